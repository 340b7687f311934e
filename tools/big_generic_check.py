import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc
N, M, T, K = 16384, 40, 131200, 2
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(5)
A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
B = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64)
X = torch.rand(T, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
kw = dict(layout="frame_major", iters=K, eps_mode="zero_replace", init="const", init_value=0.01)
H, Y = evc.convert(A, X, B, **kw)
for sl in (slice(0, 48), slice(T - 48, T), slice(65536 + 16, 65536 + 64)):
    Hs, Ys = evc.convert(A, X[sl].contiguous(), B, **kw)
    print(sl, float((H[sl] - Hs).abs().max() / Hs.abs().max()), float((Y[sl] - Ys).abs().max() / Ys.abs().max()))
print("ok", bool(torch.isfinite(H).all()))
