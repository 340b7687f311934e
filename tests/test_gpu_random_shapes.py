"""Differential test over seeded random shapes (`-m gpu`): the default kernel selection (k_fused_all with whichever
exchange the member count picks, start values formed in the kernel, activations exported by the last launch)
against the same call restricted to the round-1 kernels without inter-workgroup exchange, on dictionaries and
batches of arbitrary size, both layouts, with and without the synthesis, with error traces (several launches).
The restricted path is itself held to the oracle by tests/test_gpu_parity.py and tests/test_gpu_configs.py; here the
point is that no shape falls between the kernels' cases.  rtol 1e-9 (both paths are float64; they differ in
summation order only)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    M = int(rng.integers(1, 33))
    N = int(rng.choice([int(rng.integers(500, 1100)), int(rng.integers(1100, 4200)), int(rng.integers(4200, 12000)),
                        512 * int(rng.integers(1, 20))]))
    n_utt = int(rng.integers(1, 6))
    lens = [int(rng.integers(1, 260)) for _ in range(n_utt)]
    if seed >= 24:                       # long batches: every group of the persistent grid walks several frame tiles
        lens = [int(rng.integers(900, 2500)) for _ in range(n_utt + 2)]
        N = min(N, 6000)
    return M, N, lens, rng


@pytest.mark.parametrize("seed", range(30))
def test_default_selection_equals_the_restricted_path(seed):
    import exemplars_vc_amd as evc
    from oracle import evc_oracle as o
    M, N, lens, rng = _case(seed)
    T = sum(lens)
    p = o.synth_problem(M, N, T, seed=1000 + seed)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    K = int(rng.integers(1, 25))
    layout = "frame_major" if seed % 2 else "bin_major"
    tr = (lambda a: np.ascontiguousarray(a.T)) if layout == "frame_major" else (lambda a: a)
    kw = dict(layout=layout, iters=K, eps_mode=["zero_replace", "add", "clamp"][seed % 3],
              init=["sklearn", "const"][seed % 2], utt_offsets=offs)
    if kw["init"] == "const":
        kw["init_value"] = 0.37
    if seed % 4 == 3:
        kw.update(check_every=5, info=True)              # several launches, error trace
    if seed % 5 == 2:                                    # the KL update (scikit-learn's guard only, no L1)
        kw.update(loss="kl", eps_mode="zero_replace")
    got = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), **kw)
    want = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), all_resident=False, cooperative=False, **kw)
    for g, w, name in zip(got[:2], want[:2], ("H", "Y")):
        r, z = rel_err(g, w)
        assert r <= 1e-9 and z == 0.0, f"seed {seed} M={M} N={N} lens={lens} K={K} {layout}: {name} rel err {r:.2e}"
    if seed % 4 == 3:
        # (a residual at the rounding floor - M = 1 is solved exactly by one update - is noise: absolute term)
        np.testing.assert_allclose(got[2]["err"], want[2]["err"], rtol=1e-9, atol=1e-10 * np.linalg.norm(p["X"]),
                                   equal_nan=True)
    # the solve alone (no synthesis) exports H the same way
    h_only = evc.solve_activations(tr(p["A"]), tr(p["X"]), **{k: v for k, v in kw.items() if k != "info"})
    r, z = rel_err(h_only, got[0])
    assert r == 0.0 and z == 0.0, f"seed {seed}: solve and convert disagree on H ({r:.2e})"
