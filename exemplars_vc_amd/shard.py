"""Utterance sharding across the GPUs of one node - no data-path collective.

Every frame column evolves independently given the dictionary, and the reference's
per-call semantics are per utterance, so utterances are the unit of distribution
(SURVEY.md section 8e): the dictionary is replicated on each GPU, the utterance list is
split by longest-processing-time-first, each rank (one process per GPU) solves its shard
in a single batched launch sequence, and results are put back in utterance order on the
host.  `torch.distributed` (RCCL on GPU boxes, gloo in CPU tests) is used only to hand the
finished shards to rank 0.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


def partition_utterances(lengths: Sequence[int], n_shards: int) -> List[List[int]]:
    """Deterministic LPT partition: utterance indices per shard, loads as even as possible."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * n_shards
    shards: List[List[int]] = [[] for _ in range(n_shards)]
    for i in order:
        k = min(range(n_shards), key=lambda s: (loads[s], s))
        shards[k].append(i)
        loads[k] += int(lengths[i])
    for s in shards:
        s.sort()
    return shards


def convert_utterances(X_list, A, B, *, iters=100, tol=0.0, eps_mode="zero_replace",
                       init="sklearn", algo="auto", device=None, solver: Optional[Callable] = None,
                       info=False):
    """Convert a list of utterances (each T_u x M, frames as rows) with dictionary A (N x M)
    and target dictionary B (N x Mb): one batched solve + one synthesis.
    Returns a list of (T_u x Mb) arrays; with info=True a list of (Y_u, n_iter_u) pairs (n_iter_u: the
    updates applied to utterance u - the per-call `n_iter` of the reference's scikit-learn call).
    `solver(X_cat, offsets)` -> (T x N) activations can be injected by tests; the default is the HIP path."""
    if not X_list:
        return []
    offs = np.concatenate([[0], np.cumsum([len(x) for x in X_list])]).astype(np.int32)
    X = np.concatenate([np.asarray(x) for x in X_list], axis=0)
    n_iter = None
    if solver is None:
        from .solver import convert
        out = convert(A, X, B, want_h=False, layout="frame_major", iters=iters, eps_mode=eps_mode,
                      init=init, algo=algo, utt_offsets=offs, device=device,
                      check_every=10 if tol > 0 else 0,
                      stop_rule="sklearn" if tol > 0 else "none", tol=tol, info=info)
        Y, n_iter = (out[0], out[1]["n_iter"]) if info else (out, None)
    else:
        act = solver(X, offs)
        Y = act @ np.asarray(B)
    Ys = [Y[offs[i]:offs[i + 1]] for i in range(len(X_list))]
    if info:
        return [(y, int(n_iter[i]) if n_iter is not None else -1) for i, y in enumerate(Ys)]
    return Ys


def convert_sharded(X_list, A, B, *, rank=None, world_size=None, gather=True, **kw):
    """Each rank converts its LPT shard of `X_list`; with gather=True rank 0 returns the full
    list in utterance order (other ranks return None), otherwise every rank returns
    {utterance index: Y}.  Works without torch.distributed when world_size == 1."""
    dist = None
    if world_size is None or rank is None:
        try:
            import torch.distributed as dist_mod
            if dist_mod.is_available() and dist_mod.is_initialized():
                dist = dist_mod
                rank, world_size = dist.get_rank(), dist.get_world_size()
        except Exception:  # pragma: no cover
            dist = None
    if world_size is None:
        rank, world_size = 0, 1
    elif dist is None and world_size > 1:
        import torch.distributed as dist
    lengths = [len(x) for x in X_list]
    mine = partition_utterances(lengths, world_size)[rank]
    Ys = convert_utterances([X_list[i] for i in mine], A, B, **kw)
    local = dict(zip(mine, Ys))
    if not gather:
        return local
    if world_size == 1:
        return [local[i] for i in range(len(X_list))]
    parts = [None] * world_size if rank == 0 else None
    dist.gather_object(local, parts, dst=0)
    if rank != 0:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(len(X_list))]
