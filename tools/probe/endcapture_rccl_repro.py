#!/usr/bin/env python
"""Pure PyTorch + RCCL reproducer of the hipStreamEndCapture segfault (no torch_rechub_amd code involved).

    python tools/probe/endcapture_rccl_repro.py <variant>

Every variant captures, on a fresh origin stream O, a forked side stream F that issues asynchronous all-reduces (async_op=True:
ProcessGroupNCCL runs them on its own internal stream N, which joins the capture through an event recorded on F).  They differ
in WHERE the works are waited for:
    join-on-origin   w.wait() with O current (O waits for N's end event), then O.wait_stream(F)  -- what
                     torch_rechub_amd.distributed.DenseGradBucket.join does
    wait-in-fork     w.wait() with F current: F, a forked stream, waits for an event of N, the stream forked from F
    wait-in-fork-1   the same with ONE all-reduce
    no-gather        wait-in-fork without the synchronous all_gather in front
"""
import socket
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "wait-in-fork"
s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
t = torch.ones(1 << 16, device=dev)
g = torch.ones(1 << 12, device=dev)
out = torch.empty_like(g)
origin = torch.cuda.Stream()
fork = torch.cuda.Stream()
main = torch.cuda.current_stream()
for _ in range(2):  # eager warm-up of the communicator
    dist.all_gather_into_tensor(torch.empty_like(g), g)
    fork.wait_stream(main)
    with torch.cuda.stream(fork):
        dist.all_reduce(t, async_op=True).wait()
    main.wait_stream(fork)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
origin.wait_stream(main)
with torch.cuda.stream(origin):
    graph.capture_begin(capture_error_mode="thread_local")
    if mode != "no-gather":
        dist.all_gather_into_tensor(out, g)
    fork.wait_stream(origin)
    with torch.cuda.stream(fork):
        works = [dist.all_reduce(t, async_op=True)]
        if mode != "wait-in-fork-1":
            works.append(dist.all_reduce(g, async_op=True))
        if mode != "join-on-origin":
            for w in works:
                w.wait()
    if mode == "join-on-origin":
        for w in works:
            w.wait()
    origin.wait_stream(fork)
    print(f"[{mode}] ending capture ...", flush=True)
    graph.capture_end()
print(f"[{mode}] capture ended", flush=True)
main.wait_stream(origin)
graph.replay()
torch.cuda.synchronize()
print(f"[{mode}] replayed ok", flush=True)
dist.destroy_process_group()
