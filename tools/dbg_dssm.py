import os, sys, faulthandler, time
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
os.environ["RECHUB_TABLE_ADAM"] = "dense" if variant == "dense" else "lazy"
from tools.model_bench import build
dev = torch.device("cuda:0")
trainer, x, y = build("dssm", dev, 4096, 0.01)
if variant == "noinbatch":
    trainer.in_batch_neg = False
    trainer.criterion = torch.nn.BCELoss()
if variant == "hard":
    trainer.hard_negative = True
stats = torch.zeros(3, device=dev)
if variant in ("staticrand", "randnotopk", "checkkeys"):
    import torch_rechub_amd.trainers.match_trainer as MT
    keys_static = torch.rand(4096, 4096, device=dev)
    def samp(scores, neg_ratio=None, hard_negative=False, generator=None):
        Bn = scores.size(0)
        diag = torch.eye(Bn, dtype=torch.bool, device=scores.device)
        if variant == "staticrand":
            keys = keys_static.masked_fill(diag, -1.0)
            return torch.topk(keys, k=neg_ratio, dim=1).indices
        keys = torch.rand((Bn, Bn), device=scores.device)
        if variant == "checkkeys":
            keys = keys.masked_fill(diag, -1.0)
            stats.copy_(torch.stack([keys.min(), keys.max(), (~torch.isfinite(keys)).sum().float()]))
            keys = keys.clamp(0, 1)
        idx = (keys[:, :neg_ratio] * (Bn - 1)).long()
        return idx + (idx >= torch.arange(Bn, device=scores.device).unsqueeze(1)).long()
    MT.inbatch_negative_sampling = samp
trainer.model.train(); trainer.optimizer.sync_hyper()
for i in range(4):
    trainer.train_step(x, y)
trainer.flush(); torch.cuda.synchronize(); print(variant, "eager ok", flush=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): trainer.train_step(x, y)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = trainer.train_step(x, y)
print(variant, "captured", flush=True)
for i in range(12):
    g.replay(); torch.cuda.synchronize(); print(variant, "replay", i, float(loss), stats.tolist(), flush=True)
trainer.flush(); torch.cuda.synchronize(); print(variant, "flush ok", flush=True)
