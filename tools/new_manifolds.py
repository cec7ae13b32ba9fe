#!/usr/bin/env python3
"""How many manifolds are NEW per update (no manifold of the same pair in the update before: the colouring's work), and how
many rounds they take, over a stretch of a bench scene:  new_manifolds.py c5|c3|t1m|t1m_settled [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402
import bench  # noqa: E402

DT = 16_666_667
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sc = scenes.SCENES[name]()
w = physics_amd.World(sc.config(flags=sc.flags | physics_amd.FLAG_EXCLUSIVE_GPU))
sc.populate(w)
w.update_n(DT, bench.DEFAULT_PREROLL.get(name, 30))
out = []
for _ in range(steps):
    w.update(DT)
    st = w.get_stats()
    out.append((st.n_new_manifolds, st.color_rounds))
print(name, "manifolds", st.n_manifolds, "new / rounds per update:", " ".join(f"{a}/{b}" for a, b in out))
