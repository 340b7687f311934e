"""Loop time of k_fused_wide over (ranges per group, wavefronts per workgroup) at the STFT flow's shape.
   python tools/tune_wide.py [utterances] [M N K]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402


def main():
    U = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (201, 4096, 150)
    combos = [tuple(int(v) for v in a.split(":")) for a in sys.argv[5:]] or [(0, 0)]
    T = 688 * U
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    A = torch.rand(N, M, generator=g, device=dev) + 1e-3
    A /= A.norm(dim=1, keepdim=True)
    Hs = torch.rand(T, N, generator=g, device=dev) * (torch.rand(T, N, generator=g, device=dev) < 8.0 / N)
    X = (Hs @ A + 1e-6).contiguous()
    del Hs
    offs = np.arange(U + 1, dtype=np.int32) * 688
    H = torch.empty(T, N, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    fl = K * (4 * M * N + 3 * N) * T
    for (c, w) in combos:
        info = {}
        ms = []
        for rep in range(6):
            evc.solve_activations(A, X, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                  utt_offsets=offs, out=H, loop_events=(e0, e1), fused_c=max(c, 0), fused_w=w,
                                  fused=c >= 0, solve_info=info)
            e1.synchronize()
            if rep >= 2:
                ms.append(e0.elapsed_time(e1))
        m = float(np.median(ms))
        print(json.dumps({"utt": U, "c": c, "w": w, "members": info.get("members"), "kernel": info.get("kernel"),
                          "loop_ms": m, "us_per_iter": 1e3 * m / K, "tflops": fl / m / 1e9,
                          "frac": fl / m / 1e9 / 157.3}), flush=True)


if __name__ == "__main__":
    main()
