"""Control for DESIGN.md section 8: the collision stages on a FROZEN scene. A settled C2 stack (its poses after a
quiet roll) is loaded into a world with zero gravity, zero velocities and zero solver iterations, so no step changes
any transform; every step's manifolds must then equal those of the first step bit for bit - with a bf16 GEMM on a
second stream before every step. A difference here cannot come from a transform that was read too early or too late.

    python tools/frozen_probe.py [steps] [roll]
"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import physics_amd
from physics_amd import scenes

DT = 16_666_667
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
roll = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sc = scenes.c2()
S = physics_amd.World(sc.config())
sc.populate(S)
S.update_n(DT, roll)
S.sync()
pos, rot = S.get_transforms()
F = physics_amd.World(sc.config(gravity_force=(0.0, 0.0, 0.0), solver_iterations=0))
F.set_bodies(pos, rot=rot, shape_type=sc.shape_type, half_extent=sc.half_extent)


def snapshot(w):
    ids, cnt, nrm, pts = w.get_manifolds()
    order = np.argsort(ids[:, 0].astype(np.uint64) << np.uint64(32) | ids[:, 1])
    live = np.arange(4)[None, :] < cnt[order][:, None]
    return ids[order], cnt[order], nrm[order], np.where(live[..., None], pts[order], 0)


side = torch.cuda.Stream()
m = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
F.update(DT); F.sync()
ref = snapshot(F)
p0, r0 = F.get_transforms()
print("frozen scene:", len(ref[0]), "manifolds;", "transforms unchanged by a step:", bool(np.array_equal(p0, pos) and np.array_equal(r0, rot)))
bad = 0
for step in range(2, steps + 1):
    with torch.cuda.stream(side):
        m2 = m @ m
        m3 = m2 @ m
    F.update(DT); F.sync()
    torch.cuda.synchronize()
    cur = snapshot(F)
    same = [a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cur, ref)]
    if not all(same):
        bad += 1
        if bad <= 3:
            print(f"step {step}: ids/counts/normals/points equal: {same}; manifolds {len(cur[0])} vs {len(ref[0])}")
            if same[0] and not same[3]:
                rows = np.nonzero((cur[3] != ref[3]).reshape(len(ref[3]), -1).any(axis=1))[0]
                print("   rows", rows[:8].tolist(), "of", len(rows))
                for r in rows[:: max(1, len(rows) // 12)][:12]:
                    d = (cur[3][r].astype(np.float64) - ref[3][r].astype(np.float64))
                    print("    ", cur[0][r].tolist(), "count", int(cur[1][r]), "delta xyz,depth per point:", np.array2string(d, precision=2, suppress_small=False, max_line_width=400).replace("\n", " "))
p1, r1 = F.get_transforms()
print(f"{bad} of {steps - 1} loaded steps differ from the first step; transforms still unchanged: {bool(np.array_equal(p1, pos) and np.array_equal(r1, rot))}")
