"""Lockstep probe: world A steps with a GEMM launched on a side stream right before every step, world B quietly; both
sync every step. Reports the first step at which any observable differs, and which."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import physics_amd
from physics_amd import scenes

DT = 16_666_667
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
flags_extra = physics_amd.FLAG_SOLVER_PER_COLOR if (len(sys.argv) > 2 and sys.argv[2] == "percolour") else 0
sc = getattr(scenes, sys.argv[3] if len(sys.argv) > 3 else "c2")()
A = physics_amd.World(sc.config(flags=sc.flags | flags_extra))
B = physics_amd.World(sc.config(flags=sc.flags | flags_extra))
for w in (A, B):
    sc.populate(w)
side = torch.cuda.Stream()
m = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
for step in range(1, steps + 1):
    with torch.cuda.stream(side):
        m2 = m @ m
        m3 = m2 @ m
    A.update(DT); A.sync()
    torch.cuda.synchronize()
    B.update(DT); B.sync()
    sa, sb = A.get_stats(), B.get_stats()
    diffs = []
    for f in ("n_pairs", "n_manifolds", "n_contacts", "n_colors", "color_rounds"):
        if getattr(sa, f) != getattr(sb, f):
            diffs.append(f"{f} {getattr(sa, f)}/{getattr(sb, f)}")
    for name, x, y in zip(("pos", "rot", "lin", "ang"), A.get_transforms() + A.get_velocities(), B.get_transforms() + B.get_velocities()):
        if not np.array_equal(x, y):
            bad = np.nonzero((x != y).any(axis=1))[0]
            diffs.append(f"{name} differs at {len(bad)} bodies, first {bad[:5].tolist()}")
    if diffs:
        print(f"step {step}: " + "; ".join(diffs))
        ma, mb = A.get_manifolds(), B.get_manifolds()
        print("manifold arrays equal:", [bool(a.shape == b.shape and np.array_equal(a, b)) for a, b in zip(ma, mb)])
        ka = set(map(tuple, ma[0].tolist())); kb = set(map(tuple, mb[0].tolist()))
        miss = sorted(kb - ka); extra = sorted(ka - kb)
        print("missing in loaded world:", len(miss), miss[:12], "... ground:", sum(1 for x in miss if x[1] == 0xFFFFFFFF))
        print("extra in loaded world:", len(extra), extra[:12])
        if not miss and not extra:
            # same manifold set: compare the contact data key by key (slot order is arbitrary)
            oa = np.argsort(ma[0][:, 0].astype(np.uint64) << np.uint64(32) | ma[0][:, 1]); ob = np.argsort(mb[0][:, 0].astype(np.uint64) << np.uint64(32) | mb[0][:, 1])
            for nm, i in (("counts", 1), ("normals", 2), ("points", 3)):
                xa, xb = ma[i][oa], mb[i][ob]
                if i == 3:  # only the first `count` points of a manifold are defined
                    live = np.arange(4)[None, :] < ma[1][oa][:, None]
                    xa = np.where(live[..., None], xa, 0); xb = np.where(live[..., None], xb, 0)
                bad = np.nonzero((xa != xb).reshape(len(xa), -1).any(axis=1))[0]
                print(f"  manifold {nm}: {len(bad)} differ", [(ma[0][oa][j].tolist(), xa[j].tolist(), xb[j].tolist()) for j in bad[:2]])
        print("stats A", {f: getattr(sa, f) for f in ("n_pairs", "n_manifolds", "n_ground_manifolds", "n_contacts", "overflow")})
        print("stats B", {f: getattr(sb, f) for f in ("n_pairs", "n_manifolds", "n_ground_manifolds", "n_contacts", "overflow")})
        sys.exit(1)
print("no difference in", steps, "steps")
