"""hipGraph capture in segments with eager actions between them.

A hipGraph replays its kernel nodes in order on one queue (measured on ROCm 7: parallel branches of a captured graph
do not overlap), and a graph node cannot wait for work that was launched outside the graph.  The deferred lazy-Adam
sweep needs both: it runs on a side stream UNDER the next training step, and it takes the step number by value.  So
the training step is captured as consecutive graphs ("segments") sharing one memory pool, and whoever needs an eager
launch at a given point of the step calls ``cut(fn)`` there while the capture is running: the current segment is
closed, ``fn`` is remembered, and a new segment begins.  ``replay()`` launches segment, fn, segment, fn ...

Only one SegmentedGraph can capture at a time; ``active()`` returns it (or None).
"""
import gc

import torch

_active = None
_ROLE_STREAMS = {}
capture_end_hooks = []  # callables run when a SegmentedGraph.capture ends, completed OR abandoned (ops resets its capture-scoped state)


def role_stream(role, device=None):
    """THE stream of ``role`` ("capture", "sweep", "dense_allreduce", "warmup", "branch", "copy") on ``device`` -- created
    once, kept for the life of the process.

    ``torch.cuda.Stream()`` does not create a stream: it hands out the next of 32 pooled hipStreams per device, round robin.
    A process that builds many trainers (a capture stream per capture, a side stream for the optimizer's sweep, one for the
    dense all-reduce, one for the warm-up ...) wraps around the pool and gets the SAME hipStream back under another role.
    Round 5's ``tools/bitwise_probe.py dp`` died that way, deterministically, inside ``hipStreamEndCapture`` (an unbounded
    recursion ``hip::Stream::EndCapture`` -> ``EndCapture`` -> ..., rocgdb backtrace in profiles/r06_endcapture_backtrace.txt):
    the ORIGIN of the tenth data-parallel trainer's capture was the pooled stream on which the first trainer had issued its
    asynchronous RCCL all-reduces 32 ``torch.cuda.Stream()`` calls earlier (tools/probe/stream_trace.py); one extra pool
    stream drawn anywhere in between -- any change of that distance -- and the same sequence runs through
    (``PROBE_SKEW`` of the probe).  With one stream per role no hipStream ever changes its role, and a
    long-lived process uses a handful of streams however many trainers it builds."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (str(role), dev.index)
    s = _ROLE_STREAMS.get(key)
    if s is None:
        s = _ROLE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def active():
    """The SegmentedGraph whose capture is running on this thread, or None."""
    return _active


class SegmentedGraph(object):

    def __init__(self):
        self.segments = []   # [CUDAGraph, eager fn | None]: fn runs after the graph
        self.before = []     # eager fns that run before the first segment of every replay
        self.after_fns = []  # eager fns that run after the last segment of every replay
        self.seg_stream = {}  # segment index -> callable returning the stream that segment is REPLAYED on (default: current)
        self._stream = None
        self._pool = None

    # -- capture ---------------------------------------------------------------------------
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()  # one private pool: later segments read earlier ones' tensors
        # thread-local error mode: with a process group alive, RCCL's watchdog thread polls events of earlier eager
        # collectives; under the default global mode such a call from another thread invalidates a running capture
        g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
        self.segments.append([g, None])

    def capture(self, fn):
        """Run ``fn()`` under capture (it may call ``cut``); returns fn's result.  Nothing is executed on the device."""
        global _active
        if _active is not None:
            raise RuntimeError("a segmented capture is already running")
        torch.cuda.synchronize()
        gc.collect()
        self._stream = role_stream("capture")  # (one origin stream for every capture of the process: see role_stream)
        self._stream.wait_stream(torch.cuda.current_stream())
        _active = self
        try:
            with torch.cuda.stream(self._stream):
                self._begin()
                try:
                    out = fn()
                finally:
                    self.segments[-1][0].capture_end()
        finally:
            _active = None
            for hook in capture_end_hooks:
                hook()
        torch.cuda.current_stream().wait_stream(self._stream)
        return out

    def cut(self, fn):
        """Close the running segment; ``fn`` will be called (eagerly, on the replaying stream) at this point of every
        replay.  It is NOT called now: a capture does not execute anything."""
        if _active is not self:
            raise RuntimeError("cut() outside this graph's capture")
        self.segments[-1][0].capture_end()
        self.segments[-1][1] = fn
        self._begin()

    def replay_segment_on(self, index, stream_fn):
        """Segment ``index`` is replayed on ``stream_fn()`` instead of the replaying stream (the eager functions around it
        order the streams).  A captured graph is not bound to the stream it was captured on."""
        self.seg_stream[int(index)] = stream_fn

    def at_start(self, fn):
        """``fn`` runs eagerly before the first segment of every replay."""
        self.before.append(fn)

    def after(self, fn):
        """``fn`` runs eagerly after the last segment of every replay -- i.e. after the WHOLE step has been enqueued.
        (optim.TableAdam counts the replayed steps on the host this way: its deferred sweep takes the step by value)."""
        self.after_fns.append(fn)

    # -- replay ----------------------------------------------------------------------------
    def replay(self):
        for fn in self.before:
            fn()
        for i, (g, fn) in enumerate(self.segments):
            sfn = self.seg_stream.get(i)
            if sfn is not None:
                with torch.cuda.stream(sfn()):
                    g.replay()
            else:
                g.replay()
            if fn is not None:
                fn()
        for fn in self.after_fns:
            fn()

    def pool(self):
        return self._pool
