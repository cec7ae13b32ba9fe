"""Background load for robustness runs: back-to-back 2048^3 bf16 GEMMs (hipBLASLt, MFMA) on this GPU for the given
number of seconds, from a process of its own.

    python tools/gemm_load.py 90 &  python -m pytest tests -m gpu -q ; wait
"""
import sys
import time

import torch

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
m = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
n = 0
while time.time() - t0 < seconds:
    x = m
    for _ in range(50):
        x = x @ m
    torch.cuda.synchronize()
    n += 50
print(f"gemm_load: {n} GEMMs in {time.time() - t0:.0f} s")
