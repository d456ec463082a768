import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from tools.model_bench import build
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
print("build", scale, flush=True)
trainer, x, y = build("dssm", dev, 4096, scale)
torch.cuda.synchronize(); print("built", flush=True)
from torch_rechub_amd import _lib
orig = _lib.call
def traced(name, *a):
    rc = orig(name, *a)
    torch.cuda.synchronize()
    print("  ok", name, flush=True)
    return rc
_lib.call = traced
m = trainer.model
m.train()
trainer.optimizer.sync_hyper()
for step in range(3):
    print("step", step, flush=True)
    loss = trainer._compute_loss(x, y); torch.cuda.synchronize(); print(" fwd ok", float(loss), flush=True)
    trainer._zero_grad(); loss.backward(); torch.cuda.synchronize(); print(" bwd ok", flush=True)
    trainer.bucket.finish(assign_views=False) if trainer.optimizer._bucket is not None else None
    if step == 0 and trainer.optimizer._bucket is None:
        trainer.optimizer.attach_bucket(trainer.bucket); trainer.bucket.finish()
    trainer.optimizer.step(); torch.cuda.synchronize(); print(" opt ok", flush=True)
trainer.flush(); torch.cuda.synchronize(); print("flush ok", flush=True)
