import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from tools.model_bench import build
dev = torch.device("cuda:0")
trainer, x, y = build("dssm", dev, 4096, 0.01)
from torch_rechub_amd import _lib
orig = _lib.call
last = [""]
def traced(name, *a):
    rc = orig(name, *a)
    torch.cuda.synchronize()
    last[0] = name
    return rc
_lib.call = traced
trainer.model.train(); trainer.optimizer.sync_hyper()
import torch_rechub_amd.utils.match as M
orig_s = M.inbatch_negative_sampling
def samp(*a, **k):
    torch.cuda.synchronize(); print("   pre-sample (last kernel %s)" % last[0], flush=True)
    r = orig_s(*a, **k); torch.cuda.synchronize(); print("   sampled", int(r.min()), int(r.max()), flush=True); return r
import torch_rechub_amd.trainers.match_trainer as MT
MT.inbatch_negative_sampling = samp
for step in range(40):
    loss = trainer._compute_loss(x, y); torch.cuda.synchronize(); print("step", step, "fwd", float(loss), flush=True)
    trainer._zero_grad(); loss.backward(); torch.cuda.synchronize(); print("  bwd ok", flush=True)
    if trainer.optimizer._bucket is None:
        trainer.optimizer.attach_bucket(trainer.bucket)
    trainer.bucket.finish(); trainer.optimizer.step(); torch.cuda.synchronize(); print("  opt ok", flush=True)
trainer.flush(); torch.cuda.synchronize(); print("done", flush=True)
