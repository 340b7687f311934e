"""One-utterance call time (device-resident inputs, solve + synthesis) with the dictionary imported per call and with a
prepared dictionary (evc_dict_prepare), per BASELINE shape.   python tools/bench_prepared.py > profiles/r03_prepared_dict.jsonl"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402

CASES = [("C2", 25, 4096, 100, "f64", 688), ("C1", 25, 512, 50, "f64", 688), ("C3", 513, 8192, 200, "f64", 688),
         ("STFT", 201, 4096, 150, "f32", 688), ("C5", 25, 16384, 100, "f64", 688), ("C2 x16", 25, 4096, 100, "f64", 688 * 16)]


def main():
    dev = torch.device("cuda", 0)
    for name, M, N, K, dt, T in CASES:
        tdt = torch.float64 if dt == "f64" else torch.float32
        g = torch.Generator(device=dev); g.manual_seed(3)
        A = (torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3)
        A /= A.norm(dim=1, keepdim=True)
        B = (torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3)
        Hs = torch.rand(T, N, generator=g, device=dev, dtype=torch.float64) * (torch.rand(T, N, generator=g, device=dev) < 8.0 / N)
        X = (Hs @ A + 1e-6).to(tdt).contiguous()
        del Hs
        A, B = A.to(tdt), B.to(tdt)
        H = torch.empty(T, N, dtype=tdt, device=dev)
        Y = torch.empty(T, M, dtype=tdt, device=dev)
        kw = dict(layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn", out=H, out_y=Y)
        pd = evc.prepare_dictionary(A, B, layout="frame_major")
        res = {"case": name, "M": M, "N": N, "K": K, "dtype": dt, "frames": T}
        outs = {}
        for label, a, b in (("per_call_import", A, B), ("prepared", pd, None)):
            info = {}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(13):
                if rep == 3:
                    e0.record()
                evc.convert(a, X, b, solve_info=info, **kw)
            e1.record(); torch.cuda.synchronize()
            res[label + "_ms"] = e0.elapsed_time(e1) / 10.0
            res["kernel"] = info.get("kernel")
            outs[label] = (H.clone(), Y.clone())
        res["bitwise_equal"] = bool(torch.equal(outs["per_call_import"][0], outs["prepared"][0]) and
                                    torch.equal(outs["per_call_import"][1], outs["prepared"][1]))
        res["saved_ms"] = res["per_call_import_ms"] - res["prepared_ms"]
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
