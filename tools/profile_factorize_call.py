#!/usr/bin/env python3
"""cProfile of the literal drop-in call _factorize(X, W) on numpy arrays (one C2 utterance)."""
import cProfile, os, pstats, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exemplars_vc_amd.compat.factorize import _factorize
rng = np.random.default_rng(0)
W = rng.random((4096, 25)) + 1e-3
X = (rng.random((688, 4096)) * (rng.random((688, 4096)) < 8 / 4096)) @ W + 1e-6
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(3):
        _factorize(X, W, tol=tol)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        _factorize(X, W, tol=tol)
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
