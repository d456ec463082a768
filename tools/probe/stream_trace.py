#!/usr/bin/env python
"""Run a script and log every stream / event operation PyTorch issues (wait_stream, wait_event, Event.record, graph capture
begin / end, current-stream switches) with the raw hipStream handles -- the fork / join structure of a capture, to rebuild it
in a C++ reproducer (tools/probe/endcapture_probe.cpp).  Log: $STREAM_TRACE (default gpurun_out/stream_trace.log), flushed per
line so that it survives a segfault.

    RECHUB_STEP_FORM=deferred python tools/probe/stream_trace.py tools/bitwise_probe.py dp
"""
import os
import runpy
import sys

import torch

path = os.environ.get("STREAM_TRACE", "gpurun_out/stream_trace.log")
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
log = open(path, "w")
names = {}


def sid(s):
    h = s.cuda_stream
    if h not in names:
        names[h] = f"S{len(names)}"
    return names[h]


def emit(msg):
    cap = "C" if torch.cuda.is_current_stream_capturing() else "-"
    log.write(f"{cap} cur={sid(torch.cuda.current_stream())} {msg}\n")
    log.flush()


_ws, _we, _er = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event, torch.cuda.Event.record
evn = {}


def eid(e):
    k = id(e)
    if k not in evn:
        evn[k] = f"E{len(evn)}"
    return evn[k]


def wait_stream(self, other):
    emit(f"{sid(self)}.wait_stream({sid(other)})")
    return _ws(self, other)


def wait_event(self, ev):
    emit(f"{sid(self)}.wait_event({eid(ev)})")
    return _we(self, ev)


def record(self, stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    emit(f"{eid(self)}.record(on {sid(s)})")
    return _er(self, stream) if stream is not None else _er(self)


torch.cuda.Stream.wait_stream = wait_stream
torch.cuda.Stream.wait_event = wait_event
torch.cuda.Event.record = record
_cb, _ce = torch.cuda.CUDAGraph.capture_begin, torch.cuda.CUDAGraph.capture_end


def capture_begin(self, *a, **kw):
    emit(f"capture_begin {kw.get('capture_error_mode')}")
    return _cb(self, *a, **kw)


def capture_end(self):
    emit("capture_end ...")
    r = _ce(self)
    emit("capture_end done")
    return r


torch.cuda.CUDAGraph.capture_begin = capture_begin
torch.cuda.CUDAGraph.capture_end = capture_end
import torch.distributed as dist  # noqa: E402

for fn in ("all_reduce", "all_gather_into_tensor", "all_to_all_single", "broadcast", "all_gather"):
    real = getattr(dist, fn)

    def wrap(*a, _real=real, _fn=fn, **kw):
        emit(f"dist.{_fn} async={kw.get('async_op', False)}")
        return _real(*a, **kw)

    setattr(dist, fn, wrap)
if "--mark" in sys.argv:
    sys.argv.remove("--mark")
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
