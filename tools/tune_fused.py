#!/usr/bin/env python3
"""Time the tuning variants of the fused kernel on the bench workload (GPU box only).
usage: python tools/tune_fused.py [utterances] [c_req ...]   0 = k_fused_res, 1 or 2 = general kernel"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

U = int(sys.argv[1]) if len(sys.argv) > 1 else 96
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 2]
M, N, K, Tu = 25, 4096, 100, 688
T = U * Tu
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
A /= A.norm(dim=1, keepdim=True)
Hs = torch.rand(T, N, generator=g, device=dev, dtype=torch.float64)
Hs *= (torch.rand(T, N, generator=g, device=dev, dtype=torch.float64) < 8.0 / N)
X = (Hs @ A + 1e-6).contiguous(); del Hs
H = torch.empty(T, N, dtype=torch.float64, device=dev)
offs = np.arange(U + 1, dtype=np.int32) * Tu
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
ref = None
for rep in range(2):
    for v in variants:
        evc.solve_activations(A, X, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                              utt_offsets=offs, out=H, fused_c=v, loop_events=(e0, e1))
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        chk = float(H[:64].sum())
        if ref is None: ref = chk
        print(f"rep{rep} fused_c={v}: loop {ms:8.2f} ms  {T * 1e3 / ms / 1e3:9.1f} kframes/s  "
              f"checksum rel diff {abs(chk - ref) / abs(ref):.1e}", flush=True)
