#!/usr/bin/env python3
"""DTW alignment timing on the GPU box: the reference's corpus shape (162 utterance pairs of
~200-760 frames, 25 cepstral features) on the GPU, and the oracle restatement on one pair on the host."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc
from oracle import evc_oracle as o

rng = np.random.default_rng(1)
shapes = [(int(rng.integers(200, 760)), int(rng.integers(200, 760))) for _ in range(162)]
A = [np.cumsum(rng.standard_normal((ta, 25)), axis=0) for ta, _ in shapes]
B = [np.cumsum(rng.standard_normal((tb, 25)), axis=0) for _, tb in shapes]
evc.dtw_align(A[:2], B[:2])
torch.cuda.synchronize(); t0 = time.perf_counter()
paths = evc.dtw_align(A, B)
torch.cuda.synchronize(); tg = time.perf_counter() - t0
t0 = time.perf_counter(); o.dtw_align(A[0], B[0]); tc = time.perf_counter() - t0
cells = sum(a * b for a, b in shapes)
print(json.dumps({"case": "dtw 162 pairs", "gpu_s_incl_transfers": tg, "cells": cells, "gpu_gcells_per_s": cells / tg / 1e9,
                  "cpu_s_one_pair_oracle": tc, "cpu_cells_one_pair": shapes[0][0] * shapes[0][1],
                  "cpu_s_corpus_extrapolated": tc * cells / (shapes[0][0] * shapes[0][1])}))
