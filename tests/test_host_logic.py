"""Host-side behaviour that needs no GPU: argument validation of the drop-in surfaces, loud
failure without a device, the utterance partitioner."""
import numpy as np
import pytest

import exemplars_vc_amd as evc
from exemplars_vc_amd.shard import partition_utterances


def has_gpu():
    import torch
    return torch.cuda.is_available()


def test_no_cpu_fallback():
    if has_gpu():
        pytest.skip("device present")
    A = np.ones((4, 8)); X = np.ones((4, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        evc.solve_activations(A, X, iters=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        evc.synthesize(A, np.ones((8, 3)))


def test_product_package_does_not_import_the_oracle():
    import sys
    import importlib
    for m in [k for k in sys.modules if k.startswith("oracle")]:
        del sys.modules[m]
    importlib.reload(evc)
    assert not any(k.startswith("oracle") for k in sys.modules)
    import os, glob
    pkg = os.path.dirname(evc.__file__)
    for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        assert "oracle" not in open(f).read().replace("the oracle", ""), f


def test_factorize_validation_mirrors_sklearn():
    from exemplars_vc_amd.compat.factorize import _factorize
    X = np.ones((5, 4))
    with pytest.raises(ValueError, match="Negative values"):
        _factorize(X, -np.ones((3, 4)))
    with pytest.raises(ValueError, match="full of zeros"):
        _factorize(X, np.zeros((3, 4)))
    with pytest.raises(ValueError, match="wrong second dimension"):
        _factorize(X, np.ones((3, 5)))
    with pytest.raises(TypeError, match="same dtype"):
        _factorize(X, np.ones((3, 4), dtype=np.float32))
    with pytest.raises(ValueError, match="2D"):
        _factorize(np.ones(4), np.ones((3, 4)))


def test_pymf_and_nmf_tool_surface_contracts():
    from exemplars_vc_amd.compat.pymf import NMF as PNMF
    from exemplars_vc_amd.compat.nmf_tool import NMF as TNMF
    m = PNMF(np.ones((4, 6)), num_bases=3)
    assert (m._data_dimension, m._num_samples, m._num_bases) == (4, 6, 3)
    # pymf's default compute_w=True no longer raises: the dictionary update runs on the host, every
    # activation update on the GPU - so without a device the call ends where the solver asks for one, with a warning first
    with pytest.raises(AttributeError):
        m.factorize(niter=2, compute_w=False)      # W not set
    with pytest.warns(RuntimeWarning, match="outside the accelerated path"):
        with pytest.raises(RuntimeError):
            PNMF(np.ones((4, 6)), num_bases=3).factorize(niter=2)
    np.random.seed(3)
    m._init_h()
    np.random.seed(3)
    assert np.array_equal(m.H, np.random.random((3, 6)) + 1e-4)   # pymf base.py:174-177
    t = TNMF(max_iter=5, optimizer="pg")
    with pytest.raises(NotImplementedError):
        t.fit_transform(np.ones((4, 6)), 3, True, np.ones((4, 3)))
    with pytest.raises(NotImplementedError):
        TNMF(max_iter=5).fit_transform(np.ones((4, 6)), 3, False, 0)


def test_solver_argument_errors_without_touching_the_device(monkeypatch):
    import exemplars_vc_amd.solver as S
    import torch

    class Stop(Exception):
        pass
    # let validation run, stop right before any device work
    monkeypatch.setattr(S, "require_device", lambda d=None: torch.device("cpu"))
    with pytest.raises(ValueError, match="number of bins"):
        S.solve_activations(np.ones((4, 8)), np.ones((5, 3)), iters=1)
    with pytest.raises(ValueError, match="expected"):
        S.solve_activations(np.ones((4, 8)), np.ones((4, 3)), np.ones((7, 3)), iters=1)
    with pytest.raises(KeyError):
        S.solve_activations(np.ones((4, 8)), np.ones((4, 3)), iters=1, eps_mode="bogus")


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_partition_is_a_balanced_partition(n):
    rng = np.random.default_rng(n)
    lens = list(rng.integers(100, 1500, size=37))
    parts = partition_utterances(lens, n)
    assert sorted(i for p in parts for i in p) == list(range(37))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lens)
    assert partition_utterances(lens, n) == parts          # deterministic
    assert partition_utterances([], n) == [[] for _ in range(n)]


def test_bench_names_the_kernel_instance_a_pmc_summary_must_be_of():
    """bench.py ties a committed rocprofv3 PMC summary to a run by the kernel's template instance (roofline.traffic_run_match):
    the instance names follow the library's routing (evc_fused_all.hip pick_c, wide_layout / wide64_layout)."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    assert bench.kernel_instance("k_fused_all", 25, 8, "frobenius") == "k_fused_all<7, 0, false>"
    assert bench.kernel_instance("k_fused_all", 25, 1, "kl") == "k_fused_all<7, 1, true>"
    # k_fused_wide: 8 wavefronts; tagged hand-offs = static schedule with reduce slices (1 and 2 utterances of 688 frames)
    assert bench.kernel_instance("k_fused_wide", 201, 42, "frobenius", 688) == "k_fused_wide<13, 8, true>"
    assert bench.kernel_instance("k_fused_wide", 201, 23, "frobenius", 1376) == "k_fused_wide<13, 8, true>"
    assert bench.kernel_instance("k_fused_wide", 201, 8, "frobenius", 4128) == "k_fused_wide<13, 8, false>"
    assert bench.kernel_instance("k_fused_wide", 64, 3, "frobenius", 11008) == "k_fused_wide<4, 8, false>"
    assert bench.kernel_instance("k_fused_wide64", 513, 1, "frobenius", 11008) == "k_fused_wide64<8>"
    assert bench.kernel_instance("k_fused_wide64", 201, 1, "frobenius", 11008) == "k_fused_wide64<3>"
    assert bench.kernel_instance("k_gemm2", 201, 1, "frobenius") is None
