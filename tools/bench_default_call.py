#!/usr/bin/env python3
"""The script's DEFAULT call semantics (`_factorize`: tol=1e-4, stop test every 10 iterations, <=150) on a batch
of utterances - per-utterance stop rules evaluated on the device between 10-iteration launches."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

# usage: bench_default_call.py [utterances=96] [M=25] [N=4096] [dtype=f64]
U = int(sys.argv[1]) if len(sys.argv) > 1 else 96
Tu = 688
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
DT = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
A /= A.norm(dim=1, keepdim=True)
Hs = torch.rand(U * Tu, N, generator=g, device=dev, dtype=torch.float64)
Hs *= (torch.rand(U * Tu, N, generator=g, device=dev, dtype=torch.float64) < 8.0 / N)
X = (Hs @ A + 1e-6).to(DT).contiguous(); del Hs
A = A.to(DT)
offs = np.arange(U + 1, dtype=np.int32) * Tu
H = torch.empty(U * Tu, N, dtype=DT, device=dev)
for tol, iters in [(0.0, 150), (1e-4, 150), (1e-3, 150)]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, info = evc.solve_activations(A, X, layout="frame_major", iters=iters, eps_mode="zero_replace", init="sklearn",
                                        utt_offsets=offs, out=H, check_every=10 if tol > 0 else 0,
                                        stop_rule="sklearn" if tol > 0 else "none", tol=tol, info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ni = info["n_iter"]
    print(json.dumps({"M": M, "N": N, "dtype": str(DT).split(".")[1], "kernel": info["kernel"], "launches": info["launches"],
                      "utterances": U, "tol": tol, "max_iter": iters, "seconds": dt, "frames_per_s": U * Tu / dt,
                      "n_iter_min": int(ni.min()), "n_iter_max": int(ni.max()), "n_iter_mean": float(ni.mean()),
                      "frame_iterations_per_s": float(ni.mean()) * U * Tu / dt}))

if M != 25:
    sys.exit(0)
# the literal drop-in: _factorize(X, W) on numpy arrays, one utterance per call (upload, solve, download)
import warnings
from exemplars_vc_amd.compat.factorize import _factorize
Xn, Wn = X[:Tu].cpu().numpy(), A.cpu().numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for tol in (1e-4, 1e-3):
        for rep in range(3):
            t0 = time.perf_counter(); Hn = _factorize(Xn, Wn, tol=tol); dt = time.perf_counter() - t0
        print(json.dumps({"call": "_factorize(X[688x25], W[4096x25]) numpy in/out", "tol": tol, "seconds": dt,
                          "frames_per_s": Tu / dt}))
