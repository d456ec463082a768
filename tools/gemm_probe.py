"""Does padding K = 429 -> 432 (16-byte aligned rows) speed up the MLP's first-layer GEMMs on hipBLASLt?"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 4096, 256
for K in (429, 432, 448, 416):
    x = torch.randn(B, K, device=dev); W = torch.randn(H, K, device=dev); b = torch.randn(H, device=dev)
    g = torch.randn(B, H, device=dev)
    print(f"K={K}: fwd x@W^T+b {t(lambda: torch.nn.functional.linear(x, W, b)):6.1f} us | dX g@W {t(lambda: g @ W):6.1f} us | "
          f"dW g^T@x {t(lambda: g.t() @ x):6.1f} us", flush=True)
xp = torch.randn(B, 432, device=dev)[:, :429]
W = torch.randn(H, 429, device=dev)
print("strided x (lda 432), W lda 429:", round(t(lambda: torch.nn.functional.linear(xp, W)), 1), "us")
for (M, K, N) in ((4096, 256, 128), (4096, 128, 1)):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); g = torch.randn(M, N, device=dev)
    print(f"layer {K}->{N}: fwd {t(lambda: torch.nn.functional.linear(x, W)):6.1f} dX {t(lambda: g @ W):6.1f} dW {t(lambda: g.t() @ x):6.1f} us")
