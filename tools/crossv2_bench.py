#!/usr/bin/env python
"""CrossNetV2 (3 layers, d = 429, the configs[2] width) forward + backward: one launch per layer and direction on the tile
GEMM (rh_cross_v2_fwd / rh_cross_v2_dgrad, round 5) against library GEMM + separate epilogue passes (rounds 1-4).
    python tools/crossv2_bench.py [B = 4096]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import CrossNetV2
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    d, L = 429, 3
    net = CrossNetV2(d, L).to(dev)
    x = torch.randn(B, d, device=dev, requires_grad=True)
    up = torch.randn(B, d, device=dev)
    real_ok = ops.cross_v2_layer_ok
    for name, ok in (("tile GEMM + fused epilogues", real_ok), ("library GEMM + epilogue passes", lambda *_: False)):
        ops.cross_v2_layer_ok = ok

        def step():
            net.zero_grad(set_to_none=True)
            x.grad = None
            (net(x) * up).sum().backward()

        for _ in range(5):
            step()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
            with torch.cuda.graph(g, stream=s):
                step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            g.replay()
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"CrossNetV2 d={d} L={L} B={B}, fwd + bwd, hipGraph replay: {name}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
    ops.cross_v2_layer_ok = real_ok


if __name__ == "__main__":
    main()
