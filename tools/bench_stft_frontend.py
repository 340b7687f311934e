#!/usr/bin/env python3
"""STFT front end (librosa.core.stft as called at 04_align_n_nmf.py:422) timing on the GPU box: one
utterance of 688 frames (fft 400, hop 80, centred), device-resident in and out, 200 calls."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

F, hop, T, REP = 400, 80, 688, 200
y = torch.from_numpy(np.random.default_rng(2).standard_normal((T - 1) * hop)).cuda()
re, im = evc.stft(y, F, hop)
assert re.shape == (T, F // 2 + 1), re.shape
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(REP):
    evc.stft(y, F, hop)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / REP
print(json.dumps({"case": f"stft front end T={T} fft={F} hop={hop}", "call_us": dt * 1e6, "frames_per_s": T / dt,
                  "dft_tflops_whole_call": 2 * T * F * (F + 2) / dt / 1e12}))
