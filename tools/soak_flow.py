"""Soak of the dataflow solver's hand-off: many steps of C2 with the dataflow kernels under concurrent load against
the per-colour path on the same scene; every checkpoint must be bit-identical. (The 16-byte granule atomicity the
hand-off relies on is an observed property of gfx950, not an architectural one: this is the long-run check.)

    python tools/soak_flow.py [steps] [check_every]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import physics_amd
from physics_amd import scenes

DT = 16_666_667
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
mode = sys.argv[3] if len(sys.argv) > 3 else "flow+load"   # flow+load | flow | percolour+load
load = mode.endswith("+load")
kind = sys.argv[4] if len(sys.argv) > 4 else "both"   # copy | gemm | both
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 100
sc = scenes.c2()
flow = physics_amd.World(sc.config(flags=sc.flags | (physics_amd.FLAG_SOLVER_PER_COLOR if mode.startswith('percolour') else 0)))
ref = physics_amd.World(sc.config(flags=sc.flags | physics_amd.FLAG_SOLVER_PER_COLOR))
for w in (flow, ref):
    sc.populate(w)
side = torch.cuda.Stream()
a = torch.empty(1 << 27, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
m = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
m32 = torch.randn(2048, 2048, device="cuda", dtype=torch.float32)
big = torch.ones(1 << 26, device="cuda", dtype=torch.float32)
t0 = time.time()
done = 0
while done < steps:
    for _ in range(every // chunk):
        if load:
            with torch.cuda.stream(side):
                if kind in ("copy", "both"):
                    b.copy_(a, non_blocking=True)
                if kind in ("gemm", "both"):
                    m2 = m @ m
                if kind == "elementwise":
                    for _ in range(20):
                        big.mul_(1.0001)
                if kind == "sum":
                    for _ in range(20):
                        ssum = big.sum()
                if kind == "gemm32":
                    m2 = m32 @ m32
        flow.update_n(DT, chunk)
    flow.sync()
    ref.update_n(DT, every)
    ref.sync()
    done += every
    same = all(np.array_equal(x, y) for x, y in zip(flow.get_transforms() + flow.get_velocities(),
                                                     ref.get_transforms() + ref.get_velocities()))
    st = flow.get_stats()
    print(f"step {done}: manifolds {st.n_manifolds} colours {st.n_colors} overflow {st.overflow} "
          f"{'identical' if same else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
    if not same:
        sys.exit(1)
print("soak ok", mode)
