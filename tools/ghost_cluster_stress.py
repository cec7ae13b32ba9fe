#!/usr/bin/env python3
"""Two 33k-box towers side by side, one per world, ghosts exchanged by device copies, both on the cluster solver:
the scenario of tests/test_gpu_ghosts.py::test_cluster_solver_with_ghosts..., repeated (diagnosis of a rare hand-off
time-out).   python tools/ghost_cluster_stress.py [repetitions] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
import torch  # noqa: E402
from physics_amd import scenes  # noqa: E402

DT, REC = 16_666_667, 96
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
serial = len(sys.argv) > 3 and sys.argv[3] == "serial"  # synchronise after every world's update: nothing of two worlds overlaps
sc = scenes.c5(16, 130, 16)
width = float(sc.pos[:, 0].max() - sc.pos[:, 0].min()) + 2.0
cap = 8192
fails = 0
for rep in range(reps):
    worlds, sends = [], []
    for r in range(2):
        w = physics_amd.World(sc.config(max_ghosts=2 * cap))
        pos = sc.pos.copy()
        pos[:, 0] += np.float32(r * width)
        w.set_bodies(pos, shape_type=sc.shape_type, half_extent=sc.half_extent)
        w.set_global_ids((np.arange(sc.n) + r * sc.n).astype(np.uint32))
        lo = float(sc.pos[:, 0].min()) - 1.0 + r * width
        w.set_slab(lo if r else -1.0e6, lo + width if r == 0 else 1.0e6, 4.0)
        worlds.append(w)
        sends.append(torch.empty(cap * REC, dtype=torch.uint8, device="cuda"))
    empty = torch.full_like(sends[0], 0xFF)
    t0 = time.time()
    try:
        for step in range(steps):
            worlds[0].halo_pack_bodies_face(sends[0].data_ptr(), cap, +1)
            worlds[1].halo_pack_bodies_face(sends[1].data_ptr(), cap, -1)
            for k, w in enumerate(worlds):
                try:
                    w.sync()
                except physics_amd.PhysError as e:
                    raise physics_amd.PhysError(e.code, f"[world {k}, sync before the exchange of step {step}] {e}")
            recv = [torch.cat([empty, sends[1]]), torch.cat([sends[0], empty])]
            torch.cuda.synchronize()
            for w, blocks in zip(worlds, recv):
                w.halo_unpack_ghosts(blocks.data_ptr(), 2 * cap, 0, 0)
            for w in worlds:
                w.sync()
            for w in worlds:
                w.update(DT)
                if serial:
                    w.sync()
        for k, w in enumerate(worlds):
            try:
                w.sync()
            except physics_amd.PhysError as e:
                raise physics_amd.PhysError(e.code, f"[world {k} at final sync] {e}")
        st = [w.get_stats() for w in worlds]
        print(f"rep {rep}: ok {time.time() - t0:.2f} s manifolds {[s.n_manifolds for s in st]} colours {[s.n_colors for s in st]} ghosts {[s.n_ghosts for s in st]}", flush=True)
    except physics_amd.PhysError as e:
        fails += 1
        print(f"rep {rep}: FAILED at step {step} after {time.time() - t0:.2f} s: {e}", flush=True)
    for w in worlds:
        try:
            w.close()
        except Exception:
            pass
print("failures", fails, "of", reps)
