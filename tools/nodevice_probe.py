#!/usr/bin/env python3
"""Library first, torch second, in a fresh process (the order that failed with hipErrorNoDevice in round 1:
two ROCm runtimes in one process - torch bundles its own libamdhip64.so). Prints which runtime files are mapped."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mapped(sub):
    return sorted({line.split()[-1] for line in open("/proc/self/maps") if sub in line})


import physics_amd
from physics_amd import scenes

sc = scenes.c1()
w = physics_amd.World(sc.config())
sc.populate(w)
w.update_n(scenes.DT_NANOS, 20)
w.sync()
w.close()
print("after the library:", mapped("libamdhip64"), mapped("libhsa-runtime64"))
import torch

s = torch.cuda.Stream()
x = torch.ones(4, device="cuda")
with torch.cuda.stream(s):
    y = (x * 2).sum().item()
print("after torch:", mapped("libamdhip64"), mapped("libhsa-runtime64"), "sum", y)
