#!/usr/bin/env python3
"""Griffin-Lim timing on the GPU box: the size synthesize2() runs (688 frames, fft 400, hop 80,
300 iterations) on the GPU and through the oracle's numpy restatement of zz_audio_utilities on the host."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc
from oracle import evc_oracle as o

T, F, hop, K = 688, 400, 80, 300
rng = np.random.default_rng(1)
mag = rng.random((T, F // 2 + 1)) ** 3
x0 = rng.standard_normal(T * hop + F)
dm, dx = torch.from_numpy(mag).cuda(), torch.from_numpy(x0).cuda()
evc.griffin_lim(dm, F, hop, 3, dx)
torch.cuda.synchronize(); t0 = time.perf_counter()
xg = evc.griffin_lim(dm, F, hop, K, dx)
torch.cuda.synchronize(); tg = time.perf_counter() - t0
t0 = time.perf_counter(); xc, _ = o.griffin_lim(mag, F, hop, 30, x0); tc = (time.perf_counter() - t0) * (K / 30)
xw, _ = o.griffin_lim(mag, F, hop, K, x0) if os.environ.get("GL_FULL") else (None, None)
flops = K * 2 * (2 * T * F * (F + 2))          # two dense contractions per iteration
print(json.dumps({"case": "griffin_lim T=688 fft=400 hop=80 K=300", "gpu_ms": tg * 1e3, "cpu_s_extrapolated_from_30_iters": tc,
                  "speedup": tc / tg, "gpu_dft_tflops": flops / tg / 1e12,
                  "max_rel_diff_vs_cpu_full": None if xw is None else float(np.max(np.abs(xg.cpu().numpy() - xw)) / np.max(np.abs(xw)))}))
