#!/usr/bin/env python
"""Run a script with torch.cuda.Stream() handing out ONE stream per call site instead of the next stream of PyTorch's pool.

    python tools/probe/stream_pool_shim.py [--burn N] tools/bitwise_probe.py dp

torch.cuda.Stream() does not create a stream: it takes the next of 32 pooled hipStreams per device, round robin.  A process
that builds many trainers (each with a capture stream per capture, a side stream for the optimizer's sweep, one for the dense
all-reduce, one for the warm-up) wraps around the pool, and the SAME hipStream comes back under another role -- the origin of
one capture was a forked side stream of an earlier one, and so on.  Round 5's `tools/bitwise_probe.py dp` (ten data-parallel
trainers in one process) died in an unbounded recursion hip::Stream::EndCapture() -> EndCapture() -> ... at its last capture
(rocgdb backtrace, profiles/r06_endcapture_backtrace.txt).  With this shim -- call site = role, one stream per role for the
life of the process, which is what torch_rechub_amd.graphs.role_stream does since round 6 -- the same commit runs through.
--burn N draws N pool streams first: it shifts the pool's alignment (the crash of the unshimmed run depends on it).
"""
import runpy
import sys
import traceback

import torch

burn = 0
args = sys.argv[1:]
shim = True
while args and args[0].startswith("--"):
    if args[0] == "--burn":
        burn = int(args[1])
        args = args[2:]
    elif args[0] == "--no-shim":
        shim = False
        args = args[1:]
    else:
        raise SystemExit(f"unknown option {args[0]}")
_real = torch.cuda.Stream
_keep = [_real() for _ in range(burn)]
_by_site = {}


def _site_stream(*a, **kw):
    f = traceback.extract_stack(limit=2)[0]
    key = (f.filename, f.lineno, str(kw.get("device", a[0] if a else None)))
    s = _by_site.get(key)
    if s is None:
        s = _by_site[key] = _real(*a, **kw)
    return s


if shim:
    torch.cuda.Stream = _site_stream
sys.argv = args
runpy.run_path(args[0], run_name="__main__")
print(f"[stream_pool_shim] shim={shim} burn={burn} distinct call sites: {len(_by_site)}")
