import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from torch_rechub_amd import ops
dev = torch.device("cuda:0")
VOCABS = [3, 4, 10, 27, 105, 305, 583 * 40, 40, 1460 * 40, 24, 18, 15, 633 * 40]
F, D, N = len(VOCABS), 16, 128
g = torch.Generator().manual_seed(0)
for trial in range(3):
    idx = torch.stack([torch.randint(0, v, (N,), generator=g) for v in VOCABS], 1).to(dev)
    rows = torch.randn(N, F, D, generator=g).to(dev)
    tabs = [torch.nn.Parameter(torch.zeros(v, D, device=dev)) for v in VOCABS]
    call = ops.EmbedCall(tabs, [None] * F, [idx[:, f] for f in range(F)], [])
    res = []
    for rep in range(2):
        for t in tabs:
            ops.grad_buffer(t).zero_()
        ops.scatter_rows(call, idx, rows)
        torch.cuda.synchronize()
        res.append([ops.grad_buffer(t).clone() for t in tabs])
    for f, v in enumerate(VOCABS):
        ref = torch.zeros(v, D, dtype=torch.float64)
        ref.index_add_(0, idx[:, f].cpu(), rows[:, f].cpu().double())
        e0 = (res[0][f].cpu().double() - ref).abs().max().item()
        e1 = (res[0][f] - res[1][f]).abs().max().item()
        print(f"trial {trial} V={v:6d} max|gpu-ref|={e0:.3e} max|run0-run1|={e1:.3e}")
ops.check_errors()
