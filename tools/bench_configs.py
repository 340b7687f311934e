#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations other than the headline (GPU box only).
Prints one JSON line per case: frames/s of solve (+synthesis), loop ms, achieved algorithmic TFLOP/s."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

def flops_loop(M, N, K, algo):
    return K * ((2 * N * N + 3 * N) if algo == "gram" else (4 * M * N + 3 * N))

def run(name, M, N, K, T, algo="factored", l1=0.0, dtype="f64", reps=2, utt=688):
    dev = torch.device("cuda")
    tdt = torch.float64 if dtype == "f64" else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(7)
    A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    A /= A.norm(dim=1, keepdim=True)
    B = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    Hs = torch.rand(T, N, generator=g, device=dev, dtype=torch.float64)
    Hs *= (torch.rand(T, N, generator=g, device=dev, dtype=torch.float64) < 8.0 / N)
    X = (Hs @ A + 1e-6).to(tdt).contiguous(); del Hs
    A, B = A.to(tdt), B.to(tdt)
    offs = np.minimum(np.arange(0, T + utt, utt), T).astype(np.int32)
    if offs[-1] != T: offs = np.append(offs, T).astype(np.int32)
    offs = np.unique(offs)
    H = torch.empty(T, N, dtype=tdt, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    best = None
    for r in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        evc.solve_activations(A, X, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                              algo=algo, l1=l1, utt_offsets=offs, out=H, loop_events=(e0, e1),
                              cooperative=not os.environ.get("EVC_NO_COOP"))
        Y = evc.synthesize(B, H, layout="frame_major")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r > 0 and (best is None or dt < best[0]): best = (dt, e0.elapsed_time(e1))
    dt, loop_ms = best
    ok = bool(torch.isfinite(H).all()) and bool((H >= 0).all())
    print(json.dumps({"case": name, "M": M, "N": N, "K": K, "T": T, "algo": algo, "dtype": dtype, "l1": l1,
                      "frames_per_s": T / dt, "step_ms": dt * 1e3, "loop_ms": loop_ms,
                      "loop_tflops_algorithmic": flops_loop(M, N, K, algo) * T / (loop_ms * 1e-3) / 1e12,
                      "finite_nonneg": ok}), flush=True)
    del A, B, X, H, Y
    evc.release_workspaces(); torch.cuda.empty_cache()

cases = {
    "C1": lambda: run("C1 M=25 N=512 K=50 (one utterance)", 25, 512, 50, 688),
    "C2_1utt": lambda: run("C2 M=25 N=4096 K=100 one utterance", 25, 4096, 100, 688),
    "C2_2utt": lambda: run("C2 M=25 N=4096 K=100 two utterances", 25, 4096, 100, 688 * 2),
    "C5_1utt": lambda: run("C5 M=25 N=16384 K=100 L1 one utterance", 25, 16384, 100, 688, l1=0.25),
    "N1024_1utt": lambda: run("M=25 N=1024 K=100 one utterance (the per-workgroup share of C2 at 4 cooperating workgroups)", 25, 1024, 100, 688),
    "C2_gram": lambda: run("C2 GRAM algebra, one utterance", 25, 4096, 100, 688, algo="gram"),
    "C2_gram16": lambda: run("C2 GRAM algebra, 16 utterances", 25, 4096, 100, 688 * 16, algo="gram"),
    "C3_1utt": lambda: run("C3 M=513 N=8192 K=200 one utterance", 513, 8192, 200, 688),
    "C3_16k": lambda: run("C3 M=513 N=8192 K=200 T=16384", 513, 8192, 200, 16384, reps=1),
    "C5": lambda: run("C5 M=25 N=16384 K=100 L1 T=16x688", 25, 16384, 100, 688 * 16, l1=0.25),
    "C5_513": lambda: run("C5 M=513 N=16384 K=100 L1 one utterance", 513, 16384, 100, 688, l1=5.13, reps=1),
    # the reference's own default flow (config use_stft=1): |Re STFT| of a complex64 transform -> float32, M=201
    "STFT_f32_1utt": lambda: run("STFT M=201 N=4096 K=150 float32 one utterance", 201, 4096, 150, 688, dtype="f32"),
    "STFT_f32_16": lambda: run("STFT M=201 N=4096 K=150 float32 16 utterances", 201, 4096, 150, 688 * 16, dtype="f32"),
    "STFT_f64_16": lambda: run("STFT M=201 N=4096 K=150 float64 16 utterances", 201, 4096, 150, 688 * 16),
    "C2_f32": lambda: run("C2 float32 (generic path) 16 utterances", 25, 4096, 100, 688 * 16, dtype="f32"),
}
for k in (sys.argv[1:] or list(cases)):
    cases[k]()
