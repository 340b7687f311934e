"""Prepared dictionaries (evc_dict_prepare, include/evc.h): the reference builds A and B once per run
(04_align_n_nmf.py:230-246,350-361); a solve that is handed the prepared image must give bitwise the result of the
call that imports the caller's matrices itself - on every kernel route - and skip the per-call import."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle():
    from oracle import evc_oracle as o
    return o


CASES = [  # M, N, T, dtype, loss, extra solver keywords, expected kernel
    (25, 4096, 688, np.float64, "frobenius", {}, "k_fused_all"),
    (25, 4096, 688, np.float64, "kl", {}, None),
    (25, 1000, 300, np.float64, "frobenius", {}, None),
    (25, 512, 100, np.float32, "frobenius", {}, None),                      # float32 rides the float64 kernels
    (25, 700, 200, np.float64, "frobenius", {"fused": False}, "k_gemm_nt"),
    (25, 256, 64, np.float64, "frobenius", {"algo": "gram"}, "k_gemm_nt"),
    (513, 512, 70, np.float64, "frobenius", {}, "k_gemm_nt"),
    (201, 512, 90, np.float32, "frobenius", {}, "k_gemm2"),
    (201, 512, 90, np.float32, "kl", {"fused_w": 4}, "k_fused_wide"),
    (201, 640, 5000, np.float32, "frobenius", {}, "k_fused_wide"),
]


@pytest.mark.parametrize("M,N,T,dt,loss,extra,kernel", CASES)
@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
def test_prepared_dictionary_is_bitwise_the_unprepared_call(M, N, T, dt, loss, extra, kernel, layout):
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, N, T, seed=M + N)
    tr = (lambda z: np.ascontiguousarray(z.astype(dt))) if layout == "bin_major" else \
        (lambda z: np.ascontiguousarray(z.T.astype(dt)))
    A, X, B = tr(p["A"]), tr(p["X"]), tr(p["B"])
    kw = dict(layout=layout, iters=12, eps_mode="zero_replace", init="sklearn", loss=loss, **extra)
    pd = evc.prepare_dictionary(A, B, layout=layout, loss=loss)
    assert (pd.M, pd.Mb, pd.N) == (M, M, N)
    H0, Y0, i0 = evc.convert(A, X, B, info=True, **kw)
    H1, Y1, i1 = evc.convert(pd, X, info=True, **kw)
    assert i0["prepared"] == 0 and i1["prepared"] == 1 and i0["kernel"] == i1["kernel"]
    if kernel:
        assert i1["kernel"] == kernel, i1
    assert np.array_equal(H0, H1) and np.array_equal(Y0, Y1)
    H2, i2 = evc.solve_activations(pd, X, info=True, **kw)            # the same handle without the synthesis
    assert np.array_equal(H2, H0) and i2["prepared"] == 1
    # a dictionary prepared without B takes the caller's B per call
    pd_a = evc.prepare_dictionary(A, layout=layout, loss=loss)
    H3, Y3 = evc.convert(pd_a, X, B, **kw)
    assert np.array_equal(H3, H0) and np.array_equal(Y3, Y0)


def test_prepared_dictionary_mismatches_are_refused():
    import exemplars_vc_amd as evc
    from exemplars_vc_amd._lib import EvcError
    o = oracle()
    p = o.synth_problem(25, 512, 40, seed=1)
    pd = evc.prepare_dictionary(p["A"], p["B"])
    with pytest.raises(ValueError):
        evc.solve_activations(pd, p["X"].astype(np.float32), iters=3, dtype="f32")        # another dtype
    with pytest.raises(ValueError):
        evc.solve_activations(pd, p["X"], iters=3, loss="kl", eps_mode="zero_replace")    # another loss
    with pytest.raises(ValueError):
        evc.solve_activations(pd, p["X"][:20], iters=3)                                    # another bin count
    pd32 = evc.prepare_dictionary(p["A"].astype(np.float32))
    with pytest.raises(EvcError):       # a float32 dictionary with M <= 32 serves the (float64) fused route only
        evc.solve_activations(pd32, p["X"].astype(np.float32), iters=3, fused=False)
    with pytest.raises(ValueError):
        evc.convert(pd32, p["X"].astype(np.float32), iters=3)                              # holds no B


def test_cached_dictionary_by_identity():
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 512, 40, seed=2)
    a = evc.cached_dictionary(p["A"], p["B"])
    b = evc.cached_dictionary(p["A"], p["B"])
    assert a is b
    A2 = p["A"].copy()
    assert evc.cached_dictionary(A2, p["B"]) is not a
