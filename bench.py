#!/usr/bin/env python3
"""bench.py - spectral frames/sec converted on the exemplar-NMF activation path.

A "step" is one pass of the hot path over one batch of synthetic utterances already resident in HBM:
activation solve (K multiplicative updates against the fixed dictionary A, scikit-learn semantics as called
by 04_align_n_nmf.py: constant init, zero->EPSILON guard, fixed K, no early stop) followed by the synthesis
Y = B H.  Default workload is BASELINE.json configs[1] ("C2"): M=25 bins, N=4096 exemplars, K=100, float64,
256 utterances x 688 frames (the reference's corpus is 162 utterances; 256 is the next count
whose frame tiles fill the 256 CUs in whole rounds).

  python bench.py [--gpus N --steps K --warmup W] [--config C1|C2|C3|C4|C5|C5_513|STFT]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N>1: one process per GPU (independent shards, no data-path collective; torch.distributed is used for the barrier,
the max-over-ranks and the ranks_seen count only).  `python bench.py --gpus N` without a launcher starts the N
ranks itself: the parent, before it has touched the GPU, runs `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child process, relays its output and exits with its status.  Every preset but C4 gives each rank
its own copy of the batch (weak scaling); C4 is ONE 162-utterance set split over the ranks by
shard.partition_utterances (strong scaling, BASELINE configs[3]).  Prints ONE JSON line on rank 0.

Besides the contract's fields the line carries
  roofline       algorithmic flops of the executed algebra / HIP-event time of the iteration loop (recorded by the
                 library on the launch stream), against the dense matrix peak of the arithmetic type
  cpu_baseline   the oracle's restatement of the reference's scikit-learn call on this box's cores (median of 3),
                 plus the installed scikit-learn, the pymf-literal algebra and a one-thread run (rank 0, N=1 only)
  pcie           the same step with X uploaded from and Y (and H) downloaded to page-locked host memory inside the
                 wall (SURVEY.md 8d's definition of the metric for numpy callers); never the headline `value`
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md; FP64 vector/matrix 78.6 TFLOP/s is the public datasheet
# number quoted in SURVEY.md section 8d; HBM3E 8 TB/s spec)
PEAK_F64_TFLOPS = 78.6
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

# BASELINE.json configurations: per-GPU batch = utterances x frames.  C4 is C2's sizes on the ragged
# 162-utterance set, which with --gpus N is split over the ranks (LPT shards, strong scaling).
C4_LENGTHS = [704, 216, 513, 494, 945, 640, 497, 1370, 688]     # utterance lengths of the audio bundled with the reference
PRESETS = {
    "C1": dict(bins=25, exemplars=512, iters=50, utterances=256, frames=688, dtype="f64", l1=0.0,
               label="C1 (BASELINE configs[0]): the reference's CPU-runnable case"),
    "C2": dict(bins=25, exemplars=4096, iters=100, utterances=256, frames=688, dtype="f64", l1=0.0,
               label="C2 (BASELINE configs[1])"),
    "C3": dict(bins=513, exemplars=8192, iters=200, utterances=1, frames=688, dtype="f64", l1=0.0,
               label="C3 (BASELINE configs[2]): WORLD-width spectra, one utterance per call"),
    "C4": dict(bins=25, exemplars=4096, iters=100, utterances=162, frames=688, dtype="f64", l1=0.0, ragged=True,
               label="C4 (BASELINE configs[3]): the 162-utterance set (lengths of the bundled audio, cycled), "
                     "utterance-sharded over the ranks"),
    "C5": dict(bins=25, exemplars=16384, iters=100, utterances=16, frames=688, dtype="f64", l1=0.25,
               label="C5 (BASELINE configs[4]): L1-penalised, N=16384"),
    "C5_513": dict(bins=513, exemplars=16384, iters=100, utterances=1, frames=688, dtype="f64", l1=5.13,
                   label="C5 at WORLD width (M=513)"),
    "STFT": dict(bins=201, exemplars=4096, iters=150, utterances=16, frames=688, dtype="f32", l1=0.0,
                 label="the script's own default flow: |Re STFT| of a complex64 transform, float32, M=201"),
    "STFT64": dict(bins=201, exemplars=4096, iters=150, utterances=16, frames=688, dtype="f64", l1=0.0,
                   label="the same flow with float64 spectra (04_align_n_nmf.py:398 loads audio as np.double; whether its "
                         "librosa returns complex64 or complex128 is unpinned): k_fused_wide64 with 3 whole bin tiles per "
                         "wavefront + the split one (round 4; rounds 1-3: the two contractions)"),
}


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


# ------------------------------------------------------------------------------------------------
# multi-process bookkeeping (exercised on CPU over gloo by tests/test_bench_dist.py)
# ------------------------------------------------------------------------------------------------
def dist_env(same_device=False):
    """(world, rank, local_rank) from the launcher's environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if same_device else int(os.environ.get("LOCAL_RANK", "0"))
    return world, rank, local_rank


def run_timed(step, steps, warmup, barrier):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by `barrier()` on both sides; seconds."""
    for _ in range(warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist=None, device=None):
    """the slowest rank's time (what the whole job took)"""
    if dist is None:
        return float(elapsed)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(world, units_per_rank, steps, elapsed, total_units=None):
    """whole-job aggregate: every rank processed `units_per_rank` units per step (weak scaling), or - total_units
    given - the ranks together processed that many per step (strong scaling: one set sharded over the ranks)"""
    per_step = world * units_per_rank if total_units is None else total_units
    return per_step * steps / elapsed


def shard_of_rank(lengths, world, rank):
    """utterance indices of this rank's LPT shard of one set (exemplars_vc_amd.shard.partition_utterances)"""
    from exemplars_vc_amd.shard import partition_utterances
    return partition_utterances(lengths, world)[rank]


def launcher_command(argv, gpus, port, python=None):
    """the child command `python bench.py --gpus N` runs when no launcher started it: one rank per GPU"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv, gpus, run=None):
    """Parent side of `python bench.py --gpus N` (N > 1, no launcher in the environment): start the N ranks as a
    CHILD process - never exec, and before this process has made any GPU call - relay what they print and
    return the child's exit status."""
    import subprocess
    cmd = launcher_command(argv, gpus, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    p = (run or subprocess.run)(cmd, env=env)
    return int(p.returncode)


# ------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_sample_frames(N, frames):
    """a bounded sample: the scikit-learn algebra costs 2 N^2 flop per frame-iteration (Gram), ~7 s per run at
    N=4096 on 64 cores for a 688-frame utterance; larger dictionaries get proportionally fewer frames"""
    return max(16, min(frames, int(frames * (4096.0 / N) ** 2)))


def cpu_baseline(M, N, K, l1, dtype, seed, T):
    """The reference CPU path on this box's host cores, on a bounded sample of the same workload: the oracle's
    restatement of what 04_align_n_nmf.py executes (scikit-learn MU, Gram and numerator hoisted), all BLAS threads,
    median of 3 runs after a warm-up.  kind = "port" (the oracle; bit-exact against scikit-learn 1.7.2)."""
    from oracle import evc_oracle as o
    npdt = np.float64 if dtype == "f64" else np.float32
    p = o.synth_problem(M, N, T, seed=seed)
    X_rows = np.ascontiguousarray(p["X"].T).astype(npdt)
    W_rows = np.ascontiguousarray(p["A"].T).astype(npdt)
    B_rows = np.ascontiguousarray(p["B"].T).astype(npdt)
    o.sklearn_mu_fixed_dictionary(X_rows[:32], W_rows, max_iter=2, tol=0.0, l1_reg=l1)     # BLAS pool warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        act, n_iter, _ = o.sklearn_mu_fixed_dictionary(X_rows, W_rows, max_iter=K, tol=0.0, l1_reg=l1)
        Y = o.s4_convert(act.T, B_rows)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    out = {"value": T / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{T} frames of one utterance, M={M} N={N} K={K} l1={l1}, {dtype}, oracle restatement of the "
                     f"scikit-learn MU call of 04_align_n_nmf.py:212 + np.matmul(H.T,B); median of 3 runs "
                     f"({', '.join(f'{t:.2f}' for t in times)} s)",
           "seconds": dt, "runs_s": times}
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        out["blas"] = [{"api": i.get("internal_api"), "threads": i.get("num_threads"),
                        "lib": os.path.basename(i.get("filepath", ""))} for i in info]
        mine = [i for i in info if i.get("user_api") == "blas" and "numpy" in i.get("filepath", "")]
        if not mine:
            mine = [i for i in info if i.get("user_api") == "blas"]
        if mine:
            out["cores"] = int(mine[0].get("num_threads", 1))
    except Exception:
        pass
    # pymf-literal (Gram and numerator recomputed per iteration) on a sixteenth of the frames
    Tq = max(16, T // 16)
    H0 = (np.random.default_rng(1).random((N, Tq)) + 1e-4)
    t0 = time.perf_counter()
    o.pymf_factorize(p["X"][:, :Tq], p["A"], H0, niter=K, compute_err=False)
    out["pymf_literal_frames_per_s"] = Tq / (time.perf_counter() - t0)
    out["pymf_literal_sample_frames"] = Tq
    # the installed scikit-learn itself (what 04_align_n_nmf.py:212 calls), same sample, all threads
    try:
        import warnings
        from sklearn.decomposition import non_negative_factorization
        kw = dict(alpha_W=l1 / M, l1_ratio=1.0) if l1 > 0 else {}
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            non_negative_factorization(X=X_rows, H=W_rows, init="custom", update_H=False, n_components=N,
                                       beta_loss="frobenius", solver="mu", tol=0, max_iter=K, **kw)
        out["sklearn_installed_frames_per_s"] = T / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        out["sklearn_installed_frames_per_s"] = None
        out["sklearn_note"] = repr(e)[:120]
    # one BLAS thread, an eighth of the sample
    try:
        import threadpoolctl
        T8 = max(16, T // 8)
        with threadpoolctl.threadpool_limits(limits=1):
            t0 = time.perf_counter()
            o.sklearn_mu_fixed_dictionary(X_rows[:T8], W_rows, max_iter=K, tol=0.0, l1_reg=l1)
            out["single_thread_frames_per_s"] = T8 / (time.perf_counter() - t0)
            out["single_thread_sample_frames"] = T8
    except Exception:
        pass
    return out, (p, act, Y)


def kernel_instance(kernel_tag, M, members, loss, T=None):
    """The template instance the library's routing implies for what evc_solve_info reported (None: not derivable) - the
    string rocprofv3 prints for the kernel, so that a committed PMC summary can be tied to the run."""
    if kernel_tag == "k_fused_all":
        msteps = (M + 3) // 4 if M <= 16 else 4 + (M - 16 + 3) // 4
        c = members if members in (1, 2, 4) else (0 if members in (8, 16, 32, 64) else -1)
        return f"k_fused_all<{msteps}, {c}, {'true' if loss == 'kl' else 'false'}>"
    if kernel_tag == "k_fused_wide64":      # whole bin tiles per wavefront (evc_wide64.hip, WIDE64_TPW_SET)
        return "k_fused_wide64<%d>" % next((v for v in (3, 4, 5, 7, 8) if 64 * v + 16 >= M), 8)
    if kernel_tag == "k_fused_wide":        # bin tiles, 8 wavefronts per workgroup (evc_wide.hip, WIDE_MT_SET)
        # third argument: tagged hand-offs = the static schedule with reduce slices (wide_layout: more than four ranges
        # and no more sweep tasks per iteration than the 256 CUs)
        groups = -(-(-(-T // 16)) // 8) if T else 0
        tagged = bool(T) and members is not None and members > 4 and groups * members <= 256
        return "k_fused_wide<%d, 8, %s>" % (next((v for v in (4, 6, 8, 10, 13) if 16 * v >= M), 13), "true" if tagged else "false")
    return None


def pmc_traffic(kernel_tag, M, N, K, T, dtype, members=None, loss="frobenius"):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary of this very workload
    (rocprofv3 cannot run inside the timed process); (None, reason, False) when no matching file is on record.  The third
    value says whether the summary's kernel instance (template arguments included) is the one this run reports - a
    summary of another instance or of an older build of the kernel is still shown, but flagged."""
    import glob
    want = kernel_instance(kernel_tag, M, members, loss, T)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            pm = json.load(open(path))
            wl = pm["workload"]
            if (wl["M"], wl["N"], wl["K"], wl["frames"], wl["dtype"]) == (M, N, K, T, dtype) and \
                    kernel_tag in pm.get("kernel", ""):
                match = (pm.get("kernel") == want) if want else (pm.get("members") in (None, members))
                return pm["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), bool(match)
        except Exception:
            continue
    return None, "no PMC summary under profiles/ matches this workload and kernel", False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(PRESETS), help="BASELINE configuration preset")
    ap.add_argument("--bins", type=int)
    ap.add_argument("--exemplars", type=int)
    ap.add_argument("--iters", type=int)
    ap.add_argument("--utterances", type=int)
    ap.add_argument("--frames", type=int, help="frames per utterance")
    ap.add_argument("--l1", type=float)
    ap.add_argument("--algo", default="factored", choices=["factored", "gram", "literal"])
    ap.add_argument("--dtype", choices=["f64", "f32"])
    ap.add_argument("--loss", default="frobenius", choices=["frobenius", "kl"],
                    help="kl: the KL update of _factorize's signature default (not the headline metric)")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--no-all-resident", action="store_true", help="A/B: keep k_fused_all / k_fused_xy out")
    ap.add_argument("--pair-tiles", action="store_true", help="A/B: the experimental k_fused_xy instead of k_fused_all")
    ap.add_argument("--fused-c", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-transfer-inclusive leg")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0: bounded automatically)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (rehearse the N>1 path on a 1-GPU box)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves (torch is not imported yet - this process never touches the GPU)
        sys.exit(spawn_ranks(sys.argv[1:], args.gpus))
    cfg = dict(PRESETS[args.config])
    for k in ("bins", "exemplars", "iters", "utterances", "frames", "l1", "dtype"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    custom = any(getattr(args, k) is not None for k in ("bins", "exemplars", "iters", "utterances", "frames", "l1", "dtype"))

    import torch
    import exemplars_vc_amd as evc

    world, rank, local_rank = dist_env(args.same_device)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    M, N, K, l1 = cfg["bins"], cfg["exemplars"], cfg["iters"], cfg["l1"]
    U, Tu = cfg["utterances"], cfg["frames"]
    lens = [C4_LENGTHS[i % len(C4_LENGTHS)] for i in range(U)] if cfg.get("ragged") else [Tu] * U
    # C4 is ONE set: each rank converts its LPT shard of it (strong scaling); the other presets give every rank a
    # batch of its own (weak scaling)
    sharded = bool(cfg.get("ragged")) and not custom
    T_set = int(sum(lens))
    if sharded and world > 1:
        lens = [lens[i] for i in shard_of_rank(lens, world, rank)]
        U = len(lens)
    T = int(sum(lens))
    dtype = cfg["dtype"]
    tdt = torch.float64 if dtype == "f64" else torch.float32

    # synthetic shard of this rank (SURVEY.md 8d recipe, generated on the device in float64):
    # unit-L2 dictionary columns, ~8 active exemplars per frame, X = A H* + 1e-6
    g = torch.Generator(device=dev)
    g.manual_seed(20190131)
    A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3     # frames-as-rows
    A /= A.norm(dim=1, keepdim=True)
    B = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    B /= B.norm(dim=1, keepdim=True)
    g.manual_seed(1000 + rank)
    X = torch.empty(T, M, device=dev, dtype=torch.float64)
    for a0 in range(0, T, 16384):       # in slices: the T x N helper matrices of a large batch would not fit
        a1 = min(T, a0 + 16384)
        Hs = torch.rand(a1 - a0, N, generator=g, device=dev, dtype=torch.float64)
        Hs *= (torch.rand(a1 - a0, N, generator=g, device=dev, dtype=torch.float64) < (8.0 / N))
        X[a0:a1] = Hs @ A + 1e-6
        del Hs
    A, B, X = A.to(tdt), B.to(tdt), X.to(tdt).contiguous()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    H = torch.empty(T, N, dtype=tdt, device=dev)
    Yout = torch.empty(T, M, dtype=tdt, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); ev1.record()          # force creation of the underlying hipEvent_t
    torch.cuda.synchronize()
    loop_ms = []
    sinfo = {}
    counts = {"redo": 0}
    solve_kw = dict(layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn", algo=args.algo, l1=l1,
                    fused=not args.no_fused, fused_c=args.fused_c, loss=args.loss,
                    all_resident=not args.no_all_resident, pair_tiles=args.pair_tiles)

    def step(timed):
        # factorize() + convert(): H (T x N) and Y = H B (T x Mb) both delivered in HBM
        _, Y = evc.convert(A, X, B, utt_offsets=offs, out=H, out_y=Yout, loop_events=(ev0, ev1), solve_info=sinfo,
                           **solve_kw)
        if timed:
            ev1.synchronize()
            loop_ms.append(ev0.elapsed_time(ev1))
            counts["redo"] += sinfo.get("redo", 0)
        return Y

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed_own = run_timed(step, args.steps, args.warmup, barrier)
    cdev = dev if (dist is not None and args.dist_backend == "nccl") else None
    elapsed = max_over_ranks(elapsed_own, dist, cdev)
    ranks_seen, per_rank = 1, [T * args.steps / elapsed_own]
    if dist is not None:
        one = torch.ones(1, dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)          # every rank of the group adds itself
        ranks_seen = int(round(float(one.item())))
        mine = torch.tensor([T * args.steps / elapsed_own], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        rates = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rates, mine)
        per_rank = [float(r.item()) for r in rates]

    if rank == 0:
        value = job_throughput(world, T, args.steps, elapsed, T_set if sharded else None)
        loop_s = float(np.mean(loop_ms)) / 1e3
        fl_loop = loop_flops_per_frame(M, N, K, args.algo) * T
        achieved = fl_loop / loop_s / 1e12
        # the kernel the library reports it ran (evc_solve_info), not a guess from the shape
        ktag = sinfo.get("kernel", "unknown")
        fused64 = ktag in ("k_fused_xy", "k_fused_all", "k_fused_res", "k_fused_mu")
        # float32 callers with M <= 32 are widened onto the float64 fused kernels: the arithmetic type is f64
        arith = "f64" if (dtype == "f64" or fused64) else "f32"
        peak = PEAK_F64_TFLOPS if arith == "f64" else PEAK_F32_TFLOPS
        kernel = {
            "k_fused_xy": "k_fused_xy (persistent: H and P register-resident, a member = 256 exemplars of TWO frame tiles, exchange phases of one tile inside the sweep of the other, two free-running members per CU)",
            "k_fused_all": "k_fused_all (persistent: H and P register-resident, two members per CU alternating sweep / exchange)",
            "k_fused_res": "k_fused_res (persistent, half of H register-resident, P recomputed)",
            "k_fused_mu": "k_fused_mu (persistent fused update, activations streamed)",
            "k_fused_wide": "k_fused_wide (fused FACTORED for M > 32: task queue over frame groups x exemplar ranges, "
                            "dictionary blocks shared through LDS, V resident per wavefront, H and P streamed once)",
            "k_fused_wide64": "k_fused_wide64 (fused FACTORED for float64, 144 < M <= 528: the same task queue; a workgroup's "
                              "four wavefronts split the bins of 32 frames, fragments through per-wavefront LDS rings)",
            "k_gemm2": "k_gemm2 x2 per iteration (V = H Am^T, then the update as epilogue of V At^T)",
            "k_gemm_nt": "k_gemm_nt x2 per iteration (V = H Am^T, then the update as epilogue of V At^T)",
        }.get(ktag, ktag) + f"; members per frame tile/group: {sinfo.get('members', 1)}"
        traffic, traffic_src, traffic_match = pmc_traffic(ktag, M, N, K, T, dtype, sinfo.get("members"), args.loss)
        res = {
            "metric": "spectral frames/sec converted (100 NMF iters, N=4096 dict)" if (N, K) == (4096, 100)
                      else f"spectral frames/sec converted ({K} NMF iters, N={N} dict)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": arith, "data": "synthetic",
            "config": {"workload": ("custom: " if custom else cfg["label"] + ": ") +
                                   f"SF1->TF1-shaped dictionary, M={M} bins, N={N} exemplars, K={K} MU iterations"
                                   f"{f', L1 {l1}' if l1 else ''}, {U} utterance(s) x {('216..1370 (C4 length set)' if cfg.get('ragged') else Tu)} frames = {T} frames "
                                   f"{'on rank 0 (its shard of ' + str(T_set) + ')' if sharded and world > 1 else 'per GPU'} "
                                   f"per step, solve + synthesis B*H, {dtype} in and out",
                       "preset": args.config if not custom else None,
                       "algo": args.algo, "kernel": kernel,
                       "frames_per_gpu": T, "frames_per_step_all_gpus": T_set if sharded else world * T,
                       "parallelism": (f"one {cfg['utterances']}-utterance set in {world} LPT shard(s)" if sharded
                                       else f"utterance shards x{world}"),
                       "members": sinfo.get("members", 1), "launches_per_step": sinfo.get("launches"),
                       "io_dtype": dtype},
            "ranks_seen": ranks_seen, "frames_per_s_per_rank": per_rank,
            "visible_devices": torch.cuda.device_count(), "redo_count": counts["redo"],
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_run_match": traffic_match,
                "kernel": f"iteration loop ({ktag})",
                "launch_ms": 1e3 * loop_s,
                "algorithmic_flops_per_launch": fl_loop,
                "note": "achieved = algorithmic flops of the executed algebra (K*(4MN+3N) per frame for FACTORED) / "
                        "HIP-event time of the loop launches, recorded on the launch stream",
            },
            "algorithmic_gflop_per_frame": algorithmic_flops_per_frame(M, N, K, M, args.algo) / 1e9,
        }
        if args.loss != "frobenius":
            res["config"]["loss"] = args.loss
            res["roofline"]["note"] = "KL update: flop count of the Frobenius update is NOT applicable; see value only"
        if not args.no_pcie and world == 1:
            # wall including the upload of X and the download of Y (and of H for callers that want
            # the activations), page-locked host memory.  Reported beside the headline, never as `value`.
            Xh = torch.empty(X.shape, dtype=tdt, pin_memory=True); Xh.copy_(X)
            Yh = torch.empty(Yout.shape, dtype=tdt, pin_memory=True)
            pc = {}
            for name, with_h in (("xy", False), ("xyh", True)):
                if with_h and H.numel() * H.element_size() > 8 * 2 ** 30:
                    pc[name] = None
                    continue
                Hh = torch.empty(H.shape, dtype=tdt, pin_memory=True) if with_h else None
                reps = 2
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    X.copy_(Xh, non_blocking=True)
                    step(False)
                    Yh.copy_(Yout, non_blocking=True)
                    if with_h:
                        Hh.copy_(H, non_blocking=True)
                    torch.cuda.synchronize()
                pc[name] = T * reps / (time.perf_counter() - t0)
                del Hh
            res["pcie"] = {"frames_per_s_upload_X_download_Y": pc["xy"],
                           "frames_per_s_upload_X_download_Y_and_H": pc["xyh"],
                           "note": "page-locked host buffers, transfers on the launch stream inside the wall; H is "
                                   f"{N * (8 if dtype == 'f64' else 4)} bytes per frame"}
        if not args.no_cpu and args.loss == "frobenius" and world == 1:   # CPU leg: rank 0 at N=1 only
            Tc = args.cpu_frames or cpu_sample_frames(N, Tu)
            cpu, (p, act_cpu, Y_cpu) = cpu_baseline(M, N, K, l1, dtype, 20190131, Tc)
            res["cpu_baseline"] = cpu
            # conservative: against the faster of the oracle port and the installed scikit-learn
            res["speedup_vs_cpu"] = value / max(cpu["value"], cpu.get("sklearn_installed_frames_per_s") or 0.0)
            # parity of the GPU path on the very sample the CPU leg timed
            npdt = np.float64 if dtype == "f64" else np.float32
            Xs = np.ascontiguousarray(p["X"].T).astype(npdt)
            As, Bs = np.ascontiguousarray(p["A"].T).astype(npdt), np.ascontiguousarray(p["B"].T).astype(npdt)
            kw = dict(solve_kw); kw["dtype"] = dtype
            Hg, Yg = evc.convert(As, Xs, Bs, **kw)
            nz = act_cpu != 0
            # the timed batch may run another launch mode than a lone utterance: check both on the same sample
            kw2 = dict(kw); kw2["cooperative"] = False; kw2["all_resident"] = False
            Hb = evc.solve_activations(As, Xs, **kw2)
            res["parity"] = {
                "H_max_rel_err": float(np.max(np.abs(Hg[nz] - act_cpu[nz]) / act_cpu[nz])),
                "H_max_rel_err_no_exchange_kernels": float(np.max(np.abs(Hb[nz] - act_cpu[nz]) / act_cpu[nz])),
                "Y_max_rel_err": float(np.max(np.abs(Yg - Y_cpu) / np.abs(Y_cpu))),
                "rtol_required": 1e-4, "sample_frames": Tc,
            }
            # ... and of the kernel that was TIMED, on the timed batch itself (the sample above is one utterance and may be
            # routed to another kernel than the batch: VERDICT r03 item 2 v): the first frames of the batch's first
            # utterance against the oracle started from that utterance's start value, plus Y = B H of the same frames
            from oracle import evc_oracle as o_
            step(False)
            torch.cuda.synchronize()
            S = int(min(48, lens[0]))
            X0 = X[:lens[0]].double().cpu().numpy()                  # frames as rows
            A64, B64 = A.double().cpu().numpy(), B.double().cpu().numpy()
            h0 = float(np.sqrt(X0.mean() / N))
            want = o_.mu_solve(np.ascontiguousarray(A64.T), np.ascontiguousarray(X0[:S].T), np.full((N, S), h0), K,
                               eps_mode=o_.EPS_ZERO_REPLACE, eps=o_.SK_EPSILON, l1=l1, algo="factored")
            got = H[:S].double().cpu().numpy().T
            nzs = want != 0
            floor = 1e-6 * float(np.abs(want).max())
            big = np.abs(want) > floor
            res["parity"]["timed_batch"] = {
                "kernel": ktag, "frames": S,
                "H_max_rel_err": float(np.max(np.abs(got[nzs] - want[nzs]) / np.abs(want[nzs]))),
                "H_max_rel_err_above_1e-6_of_max": float(np.max(np.abs(got[big] - want[big]) / np.abs(want[big]))),
                "Y_max_rel_err": float(np.max(np.abs(Yout[:S].double().cpu().numpy() - want.T @ B64) / np.abs(want.T @ B64))),
            }
            # latency of ONE utterance (device-resident inputs, solve + synthesis): the reference's call pattern
            n1 = min(Tu, T)
            X1, H1, Y1 = X[:n1], H[:n1], Yout[:n1]
            l0, l1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(13):
                if rep == 3:
                    l0.record()
                evc.convert(A, X1, B, out=H1, out_y=Y1, **solve_kw)
            l1e.record()
            torch.cuda.synchronize()
            ms1 = l0.elapsed_time(l1e) / 10.0
            res["one_utterance"] = {"frames": n1, "ms": ms1, "frames_per_s": n1 / ms1 * 1e3}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
