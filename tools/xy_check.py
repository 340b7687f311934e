#!/usr/bin/env python3
"""GPU box: k_fused_xy against the float64 oracle and against k_fused_all on the same call (round 4 bring-up).

    python tools/xy_check.py [--big]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import exemplars_vc_amd as evc  # noqa: E402
from oracle import evc_oracle as o  # noqa: E402


def rel(got, want):
    nz = want != 0
    r = float(np.max(np.abs(got[nz] - want[nz]) / np.abs(want[nz]))) if nz.any() else 0.0
    z = float(np.max(np.abs(got[~nz]))) if (~nz).any() else 0.0
    return r, z


def case(M, N, T, K, l1=0.0, eps_mode="zero_replace", eps=None, seed=0, oracle=True, offs=None):
    p = o.synth_problem(M, N, T, seed=seed + M + N + T)
    kw = dict(iters=K, eps_mode=eps_mode, init="sklearn", l1=l1, info=True)
    if eps is not None:
        kw["eps"] = eps
    if offs is not None:
        kw["utt_offsets"] = np.asarray(offs, dtype=np.int32)
    t0 = time.time()
    H, info = evc.solve_activations(p["A"], p["X"], pair_tiles=True, **kw)
    H2, info2 = evc.solve_activations(p["A"], p["X"], **kw)
    msg = f"M={M} N={N} T={T} K={K} l1={l1} {eps_mode}: {info['kernel']} x{info['members']} redo={info['redo']} | {info2['kernel']} x{info2['members']}"
    r2, z2 = rel(H, H2)
    msg += f" | vs k_fused_all {r2:.2e}"
    bad = info["kernel"] != "k_fused_xy" or info["redo"] != 0 or r2 > 1e-9 or z2 != 0 or not np.isfinite(H).all()
    if oracle:
        mode = {"zero_replace": o.EPS_ZERO_REPLACE, "add": o.EPS_ADD, "clamp": o.EPS_CLAMP}[eps_mode]
        e = {"zero_replace": o.SK_EPSILON, "add": 1e-9, "clamp": 1e-15}[eps_mode] if eps is None else eps
        if offs is None:
            h0 = np.full((N, T), np.sqrt(p["X"].mean() / N))
        else:
            h0 = np.empty((N, T))
            for a, b in zip(offs[:-1], offs[1:]):
                h0[:, a:b] = np.sqrt(p["X"][:, a:b].mean() / N)
        want = o.mu_solve(p["A"], p["X"], h0, K, eps_mode=mode, eps=e, l1=l1, algo="factored")
        r, z = rel(H, want)
        msg += f" | vs oracle {r:.2e} (zeros {z:.1e})"
        bad = bad or r > 1e-8 or z != 0
    print(("FAIL " if bad else "ok   ") + msg + f"  [{time.time() - t0:.1f}s]", flush=True)
    return not bad


ok = True
ok &= case(25, 1024, 100, 12)
ok &= case(25, 512, 64, 10)                     # 2 members
ok &= case(25, 768, 33, 10)                     # 3 members, lone tile at the end (3 frame tiles)
ok &= case(25, 4096, 200, 20)
ok &= case(25, 4096, 16, 9)                     # one lone tile
ok &= case(13, 1024, 70, 8, l1=0.05)            # MSTEPS 4 (M <= 16), spare bin
ok &= case(32, 1024, 64, 8, l1=0.1)             # M % 4 == 0: start value in registers
ok &= case(28, 2048, 96, 8, l1=0.02)
ok &= case(25, 4096, 150, 15, l1=0.25)          # C5's penalty through the spare bin
ok &= case(25, 1024, 80, 10, eps_mode="add")
ok &= case(25, 1024, 80, 10, eps_mode="clamp")
ok &= case(7, 2048, 50, 6)
ok &= case(25, 3000, 100, 10)                   # padded exemplars (N not a multiple of 256)
ok &= case(25, 4096, 90, 12, offs=[0, 37, 90])  # two utterances, different start values, boundary inside a tile
if "--big" in sys.argv:
    ok &= case(25, 16384, 688, 100, l1=0.25, oracle=False)
    ok &= case(25, 4096, 688 * 4, 100, oracle=False)
    ok &= case(25, 4096, 688, 100)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
