"""The N>1 path on CPU: two ranks over gloo, each converting its shard of the utterance list,
results gathered on rank 0 in utterance order.  The solver is injected (the oracle), because
this container has no GPU; on a GPU box the same code path runs the HIP solver."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from exemplars_vc_amd.shard import convert_sharded
    from oracle import evc_oracle as o
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob, Xs = _problem()
    W_rows = np.ascontiguousarray(prob["A"].T)
    B_rows = np.ascontiguousarray(prob["B"].T)

    def solver(X_cat, offs):
        acts = [o.sklearn_mu_fixed_dictionary(X_cat[offs[i]:offs[i + 1]], W_rows, 20, 0.0)[0]
                for i in range(len(offs) - 1)]
        return np.concatenate(acts, axis=0)

    Ys = convert_sharded(Xs, W_rows, B_rows, solver=solver)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), *Ys)
    else:
        assert Ys is None
    dist.barrier()
    dist.destroy_process_group()


def _problem():
    from oracle import evc_oracle as o
    prob = o.synth_problem(12, 40, 0, seed=3)
    rng = np.random.default_rng(4)
    Xs = []
    for T in [30, 7, 55, 21, 1, 40, 13]:
        Hs = rng.random((40, T)) * (rng.random((40, T)) < 0.2)
        Xs.append(np.ascontiguousarray((prob["A"] @ Hs + 1e-6).T))
    return prob, Xs


def test_two_rank_sharded_conversion_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    from oracle import evc_oracle as o
    prob, Xs = _problem()
    W_rows = np.ascontiguousarray(prob["A"].T)
    B_rows = np.ascontiguousarray(prob["B"].T)
    assert len(got.files) == len(Xs)
    for i, X in enumerate(Xs):
        act = o.sklearn_mu_fixed_dictionary(X, W_rows, 20, 0.0)[0]
        # batched vs per-utterance BLAS calls may round differently in the last bit
        np.testing.assert_allclose(got[f"arr_{i}"], act @ B_rows, rtol=1e-12, atol=0)
