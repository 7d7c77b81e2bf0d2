import sys
sys.path.insert(0, "/root/repo")
from goldpolish_amd import core
ctx = core.Context(0)
import torch
print("avail", torch.cuda.is_available(), "count", torch.cuda.device_count())
for l in open("/proc/self/maps"):
    if "amdhip64" in l or "libhsa-runtime" in l:
        print(l.split()[-1])
