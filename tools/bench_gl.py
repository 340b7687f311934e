#!/usr/bin/env python3
"""Griffin-Lim timing on the GPU box: the size synthesize2() runs (688 frames, fft 400, hop 80,
300 iterations), one utterance per call or a batch (--utterances), on the GPU and - unless --no-cpu - through the
oracle's numpy restatement of zz_audio_utilities on the host.

  python tools/bench_gl.py [--utterances 16] [--iters 300] [--no-cpu]
Under rocprofv3: `rocprofv3 --kernel-trace --stats ... -- python3 tools/bench_gl.py --no-cpu --utterances 16`."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

ap = argparse.ArgumentParser()
ap.add_argument("--utterances", type=int, default=1)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()

T, F, hop, K, U = 688, 400, 80, a.iters, a.utterances
rng = np.random.default_rng(1)
mags = [rng.random((T, F // 2 + 1)) ** 3 for _ in range(U)]
x0s = [rng.standard_normal(T * hop + F) for _ in range(U)]
dm = [torch.from_numpy(m).cuda() for m in mags]
dx = [torch.from_numpy(x).cuda() for x in x0s]


def run(k):
    if U == 1:
        return [evc.griffin_lim(dm[0], F, hop, k, dx[0])]
    return evc.griffin_lim_batch(dm, F, hop, k, dx)


run(3)
torch.cuda.synchronize(); t0 = time.perf_counter()
xg = run(K)
torch.cuda.synchronize(); tg = time.perf_counter() - t0
# the two dense contractions of an iteration, algorithmic size (2 nb = F + 2 columns, F deep)
flops = K * U * 2 * (2 * T * F * (F + 2))
out = {"case": f"griffin_lim {U} x (T=688 fft=400 hop=80) K={K}", "gpu_ms": tg * 1e3,
       "frames_per_s": U * T / tg, "gpu_dft_tflops_whole_call": flops / tg / 1e12}
if not a.no_cpu:
    from oracle import evc_oracle as o
    t0 = time.perf_counter(); o.griffin_lim(mags[0], F, hop, 30, x0s[0]); tc = (time.perf_counter() - t0) * (K / 30) * U
    out.update({"cpu_s_extrapolated_from_30_iters_of_one_utterance": tc, "speedup": tc / tg})
    if os.environ.get("GL_FULL"):
        xw, _ = o.griffin_lim(mags[0], F, hop, K, x0s[0])
        out["max_rel_diff_vs_cpu_full"] = float(np.max(np.abs(xg[0].cpu().numpy() - xw)) / np.max(np.abs(xw)))
print(json.dumps(out))
