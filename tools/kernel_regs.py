#!/usr/bin/env python
"""Register / LDS / scratch table of every kernel of a csrc/*.hip file from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: kernel_regs.py gemm optim ...   (compiles into /tmp/regs)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/regs", exist_ok=True)
for f in sys.argv[1:]:
    src = os.path.join(ROOT, "torch_rechub_amd", "csrc", f + ".hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-mcode-object-version=5", "-munsafe-fp-atomics", "-O3", "-std=c++17",
           "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-x", "hip", "-c", src, "-o",
           f"/tmp/regs/{f}.o", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("EXTRA", "").split()
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, out = None, {}
    for line in t.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
        for k in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "LDS Size"):
            m = re.search(r"remark: .*?" + re.escape(k) + r".*?(\d+)", line)
            if m and cur:
                out[cur][k] = int(m.group(1))
    for k, v in out.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("rechub::", "").replace("void ", "").split("(")[0][:72]
        print(f"{f:7s} {name:72s} V{v.get('VGPRs:')} A{v.get('AGPRs:')} scr{v.get('ScratchSize')} occ{v.get('Occupancy')} "
              f"lds{v.get('LDS Size')}")
