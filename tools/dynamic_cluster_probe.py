#!/usr/bin/env python3
"""Run by tests/test_gpu_collision.py in a fresh process with PHYS_DEBUG_CLUSTER_DYNAMIC (and optionally
PHYS_DEBUG_CLUSTER_CAP) in the environment - the library reads them once. A 16 x 130 x 16 tower: the cluster solver with
DYNAMIC clusters (homes dealt out every update to the bodies that have manifolds) against the per-colour kernels
(PHYS_FLAG_SOLVER_PER_COLOR) in the same process; prints 'identical' / 'different' and what ran."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402

DT = 16_666_667
sc = scenes.c5(16, 130, 16)
states, ran, active = [], [], 0
for extra in (0, physics_amd.FLAG_SOLVER_PER_COLOR):
    w = physics_amd.World(sc.config(flags=sc.flags | (extra or physics_amd.FLAG_SOLVER_CLUSTER)))
    sc.populate(w)
    w.update_n(DT, 8)
    w.profile_enable(True)
    w.update_n(DT, 6)
    w.sync()
    prof, _ = w.profile_get()
    ran.append("solve_cluster" in prof)
    st = w.get_stats()
    assert st.overflow == 0, st.overflow
    states.append(w.get_transforms() + w.get_velocities())
    w.close()
same = all(np.array_equal(a, b) for a, b in zip(states[0], states[1]))
print("identical" if same else "different", "cluster_ran", ran[0], "per_colour_ran_cluster", ran[1])
