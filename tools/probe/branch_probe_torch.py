"""branch_probe.cpp inside a torch process: the library binds to the HIP runtime torch has mapped (its bundled 7.0), not to
/opt/rocm's 7.2 -- do captured graph branches run concurrently THERE?"""
import ctypes, os, sys
import torch
torch.zeros(1, device="cuda")
print("torch hip", torch.version.hip)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "branch_probe_lib.so"))
sys.stdout.flush()
lib.branch_probe_run()
