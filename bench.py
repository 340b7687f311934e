#!/usr/bin/env python3
"""bench.py - spectral frames/sec converted on the exemplar-NMF activation path.

A "step" is one pass of the hot path over one batch of synthetic utterances already
resident in HBM: activation solve (K multiplicative updates against the fixed dictionary A,
scikit-learn semantics as called by 04_align_n_nmf.py: constant init, zero->EPSILON guard,
fixed K, no early stop) followed by the synthesis Y = B H.  Default workload is BASELINE.json
configs[1] ("C2"): M=25 bins, N=4096 exemplars, K=100, float64, 256 utterances x 688 frames
(the reference's corpus is 162 utterances; 256 is the next count whose
256 x 43 sixteen-frame workgroups fill the 256 CUs in whole rounds).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N>1: one process per GPU, utterances sharded (independent shards, no data-path
collective; torch.distributed is used for the barrier and the max-over-ranks only).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md: FP64 vector/matrix 78.6 TFLOP/s is the
# public datasheet number quoted in SURVEY.md section 8d; HBM3E 8 TB/s spec)
PEAK_F64_TFLOPS = 78.6
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def algorithmic_flops_per_frame(M, N, K, Mb, algo):
    """flops per frame for the algebra actually executed."""
    if algo == "gram":
        return K * (2 * N * N + 3 * N) + 2 * M * N + 2 * Mb * N
    if algo == "literal":
        return K * (2 * N * N + 2 * M * N + 3 * N) + 2 * Mb * N
    return K * (4 * M * N + 3 * N) + 2 * M * N + 2 * Mb * N


def loop_flops_per_frame(M, N, K, algo):
    """the part of the above executed inside the iteration loop (the timed dominant kernel)"""
    if algo == "gram":
        return K * (2 * N * N + 3 * N)
    if algo == "literal":
        return K * (2 * N * N + 2 * M * N + 3 * N)
    return K * (4 * M * N + 3 * N)


def cpu_baseline(M, N, K, seed, budget_frames):
    """The reference CPU path on this box's host cores, on a bounded sample of the same
    workload: one utterance (688 frames) through the oracle's restatement of what
    04_align_n_nmf.py executes (scikit-learn MU, Gram and numerator hoisted), float64, all
    BLAS threads.  kind = "port" (the oracle; bit-exact against scikit-learn 1.7.2)."""
    from oracle import evc_oracle as o
    T = budget_frames
    p = o.synth_problem(M, N, T, seed=seed)
    X_rows = np.ascontiguousarray(p["X"].T)
    W_rows = np.ascontiguousarray(p["A"].T)
    B_rows = np.ascontiguousarray(p["B"].T)
    o.sklearn_mu_fixed_dictionary(X_rows[:32], W_rows, max_iter=2, tol=0.0)     # BLAS thread pool warm-up
    t0 = time.perf_counter()
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary(X_rows, W_rows, max_iter=K, tol=0.0)
    Y = o.s4_convert(act.T, B_rows)
    dt = time.perf_counter() - t0
    out = {"value": T / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"1 utterance of {T} frames, M={M} N={N} K={K}, float64, oracle restatement of the "
                     f"scikit-learn MU call of 04_align_n_nmf.py:212 + np.matmul(H.T,B); {dt:.2f} s wall",
           "seconds": dt}
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        out["blas"] = [{"api": i.get("internal_api"), "threads": i.get("num_threads"),
                        "lib": os.path.basename(i.get("filepath", ""))} for i in info]
        # the threads the timed numpy GEMMs actually ran on: numpy's own BLAS pool
        mine = [i for i in info if i.get("user_api") == "blas" and "numpy" in i.get("filepath", "")]
        if not mine:
            mine = [i for i in info if i.get("user_api") == "blas"]
        if mine:
            out["cores"] = int(mine[0].get("num_threads", 1))
    except Exception:
        pass
    # pymf-literal (Gram and numerator recomputed per iteration) on a quarter of the frames
    Tq = max(16, T // 4)
    H0 = np.random.default_rng(1).random((N, Tq)) + 1e-4
    t0 = time.perf_counter()
    o.pymf_factorize(p["X"][:, :Tq], p["A"], H0, niter=K, compute_err=False)
    dtl = time.perf_counter() - t0
    out["pymf_literal_frames_per_s"] = Tq / dtl
    # the installed scikit-learn itself (what 04_align_n_nmf.py:212 calls), same sample, all threads
    try:
        import warnings
        from sklearn.decomposition import non_negative_factorization
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            non_negative_factorization(X=X_rows, H=W_rows, init="custom", update_H=False, n_components=N,
                                       beta_loss="frobenius", solver="mu", tol=0, max_iter=K)
        out["sklearn_installed_frames_per_s"] = T / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        out["sklearn_installed_frames_per_s"] = None
        out["sklearn_note"] = repr(e)[:120]
    # one BLAS thread, an eighth of the utterance
    try:
        import threadpoolctl
        T8 = max(16, T // 8)
        with threadpoolctl.threadpool_limits(limits=1):
            t0 = time.perf_counter()
            o.sklearn_mu_fixed_dictionary(X_rows[:T8], W_rows, max_iter=K, tol=0.0)
            out["single_thread_frames_per_s"] = T8 / (time.perf_counter() - t0)
    except Exception:
        pass
    return out, (p, act, Y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=25)
    ap.add_argument("--exemplars", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--utterances", type=int, default=256)
    ap.add_argument("--frames", type=int, default=688, help="frames per utterance")
    ap.add_argument("--algo", default="factored", choices=["factored", "gram", "literal"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--loss", default="frobenius", choices=["frobenius", "kl"],
                    help="kl: the KL update of _factorize's signature default (not the headline metric)")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--fused-c", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-frames", type=int, default=688)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (rehearse the N>1 path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import exemplars_vc_amd as evc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    M, N, K = args.bins, args.exemplars, args.iters
    U, Tu = args.utterances, args.frames
    T = U * Tu
    tdt = torch.float64 if args.dtype == "f64" else torch.float32

    # synthetic shard of this rank (SURVEY.md 8d recipe, generated on the device in float64):
    # unit-L2 dictionary columns, ~8 active exemplars per frame, X = A H* + 1e-6
    g = torch.Generator(device=dev)
    g.manual_seed(20190131)
    A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3     # frames-as-rows
    A /= A.norm(dim=1, keepdim=True)
    B = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    B /= B.norm(dim=1, keepdim=True)
    g.manual_seed(1000 + rank)
    Hs = torch.rand(T, N, generator=g, device=dev, dtype=torch.float64)
    Hs *= (torch.rand(T, N, generator=g, device=dev, dtype=torch.float64) < (8.0 / N))
    X = Hs @ A + 1e-6
    del Hs
    A, B, X = A.to(tdt), B.to(tdt), X.to(tdt).contiguous()
    offs = np.arange(U + 1, dtype=np.int32) * Tu
    H = torch.empty(T, N, dtype=tdt, device=dev)
    Yout = torch.empty(T, M, dtype=tdt, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); ev1.record()          # force creation of the underlying hipEvent_t
    torch.cuda.synchronize()
    loop_ms = []

    def step(timed):
        # factorize() + convert(): H (T x N) and Y = H B (T x Mb) both delivered in HBM
        _, Y = evc.convert(A, X, B, layout="frame_major", iters=K, eps_mode="zero_replace",
                           init="sklearn", algo=args.algo, utt_offsets=offs, out=H, out_y=Yout,
                           fused=not args.no_fused, fused_c=args.fused_c, loop_events=(ev0, ev1),
                           loss=args.loss)
        if timed:
            ev1.synchronize()
            loop_ms.append(ev0.elapsed_time(ev1))
        return Y

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Y = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        value = world * T * args.steps / elapsed
        loop_s = float(np.mean(loop_ms)) / 1e3
        fl_loop = loop_flops_per_frame(M, N, K, args.algo) * T
        achieved = fl_loop / loop_s / 1e12
        fused = args.algo == "factored" and M <= 32 and not args.no_fused
        # float32 callers with M <= 32 are widened onto the float64 fused kernels: the arithmetic type is f64
        arith = "f64" if (args.dtype == "f64" or fused) else "f32"
        peak = PEAK_F64_TFLOPS if arith == "f64" else PEAK_F32_TFLOPS
        # HBM bytes per launch from the committed PMC passes of this very workload (rocprofv3 cannot
        # run inside the timed process); null when the profile on file is for another workload
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_bench_c2.json")))
            wl = pm["workload"]
            if fused and (wl["M"], wl["N"], wl["K"], wl["frames"], wl["dtype"]) == (M, N, K, T, args.dtype):  # noqa: E501
                traffic = pm["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        res = {
            "metric": "spectral frames/sec converted (100 NMF iters, N=4096 dict)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": arith, "data": "synthetic",
            "config": {"workload": f"C2 (BASELINE configs[1]): SF1->TF1-shaped dictionary, M={M} bins, "
                                   f"N={N} exemplars, K={K} MU iterations, {U} utterances x {Tu} frames "
                                   f"= {T} frames per GPU per step, solve + synthesis B*H",
                       "algo": args.algo, "kernel": "k_fused_res (persistent, register-resident, 1 launch per step)" if fused
                       else "k_gemm_nt(+mu epilogue), launches per iteration",
                       "frames_per_gpu": T, "parallelism": f"utterance shards x{world}", "io_dtype": args.dtype},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic,
                "kernel": "iteration loop (k_fused_res)" if fused else "iteration loop (k_gemm_nt family)",
                "launch_ms": 1e3 * loop_s,
                "algorithmic_flops_per_launch": fl_loop,
                "note": "achieved = algorithmic flops of the executed algebra (K*(4MN+3N) per frame "
                        "for FACTORED) / HIP-event time of the loop launches on the launch stream",
            },
            "algorithmic_gflop_per_frame": algorithmic_flops_per_frame(M, N, K, M, args.algo) / 1e9,
        }
        if args.loss != "frobenius":
            res["config"]["loss"] = args.loss
            res["roofline"]["note"] = "KL update: flop count of the Frobenius update is NOT applicable; see value only"
        if not args.no_cpu and args.loss == "frobenius" and world == 1:   # CPU leg: rank 0 at N=1 only
            cpu, (p, act_cpu, Y_cpu) = cpu_baseline(M, N, K, 20190131, args.cpu_frames)
            res["cpu_baseline"] = cpu
            # conservative: against the faster of the oracle port and the installed scikit-learn
            res["speedup_vs_cpu"] = value / max(cpu["value"], cpu.get("sklearn_installed_frames_per_s") or 0.0)
            # parity of the GPU path on the very sample the CPU leg timed
            Xs = np.ascontiguousarray(p["X"].T)
            Hg, Yg = evc.convert(np.ascontiguousarray(p["A"].T), Xs, np.ascontiguousarray(p["B"].T),
                                 layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                 algo=args.algo, dtype=args.dtype, fused=not args.no_fused)
            nz = act_cpu != 0
            # a lone utterance takes the cooperative launch; the timed batch runs one workgroup per frame
            # tile: check that variant of the kernel on the same sample too
            Hb = evc.solve_activations(np.ascontiguousarray(p["A"].T), Xs, layout="frame_major", iters=K,
                                       eps_mode="zero_replace", init="sklearn", algo=args.algo, dtype=args.dtype,
                                       fused=not args.no_fused, cooperative=False)
            res["parity"] = {
                "H_max_rel_err": float(np.max(np.abs(Hg[nz] - act_cpu[nz]) / act_cpu[nz])),
                "H_max_rel_err_batch_kernel": float(np.max(np.abs(Hb[nz] - act_cpu[nz]) / act_cpu[nz])),
                "Y_max_rel_err": float(np.max(np.abs(Yg - Y_cpu) / np.abs(Y_cpu))),
                "rtol_required": 1e-4,
            }
            # latency of ONE utterance (device-resident inputs, solve + synthesis): the reference's call pattern
            n1 = args.cpu_frames
            X1, H1, Y1 = X[:n1], H[:n1], Yout[:n1]
            l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(13):
                if rep == 3:
                    l0.record()
                evc.convert(A, X1, B, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                            algo=args.algo, out=H1, out_y=Y1, fused=not args.no_fused, loss=args.loss)
            l1.record()
            torch.cuda.synchronize()
            ms1 = l0.elapsed_time(l1) / 10.0
            res["one_utterance"] = {"frames": n1, "ms": ms1, "frames_per_s": n1 / ms1 * 1e3,
                                    "note": "cooperative launch: several workgroups per 16-frame tile"}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
