"""Two one-utterance solves of the STFT flow (static schedule: one task per workgroup, all workgroups of a launch must be
resident) started at the same time on two streams from two host threads: do they starve each other?  Measured on one
MI355X (round 4): no - 13 ms for the pair (6.4 ms each alone), no redo, results bitwise those of the lone runs: the two
launches run one after the other.  GPU box: python tools/concurrent_small_solves.py"""
import sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import exemplars_vc_amd as evc
M, N, T = 201, 4096, 688
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(N, M, generator=g, device=dev) + 1e-3
Xs = [torch.rand(T, M, generator=g, device=dev) for _ in range(2)]
ref = [evc.solve_activations(A, X, layout="frame_major", iters=150, eps_mode="zero_replace", init="sklearn", info=True) for X in Xs]
torch.cuda.synchronize()
out = [None, None]
def work(i):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out[i] = evc.solve_activations(A, Xs[i], layout="frame_major", iters=150, eps_mode="zero_replace", init="sklearn", info=True)
    st.synchronize()
for rep in range(5):
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    same = [bool(torch.equal(out[i][0], ref[i][0])) for i in range(2)]
    print(f"rep {rep}: {dt*1e3:.1f} ms, kernels {[o[1]['kernel'] for o in out]}, redo {[int(o[1]['redo']) for o in out]}, equal to the lone runs {same}", flush=True)
