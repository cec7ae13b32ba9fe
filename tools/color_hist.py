#!/usr/bin/env python3
"""Colour class sizes of a scene at the state tools/ab_solver.py measures (diagnosis of the solver's step structure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 70
sc = scenes.SCENES[wl]()
w = physics_amd.World(sc.config())
sc.populate(w)
w.update_n(scenes.DT_NANOS, steps)
w.sync()
counts = w.get_color_counts()
st = w.get_stats()
print(wl, "manifolds", st.n_manifolds, "colours", st.n_colors)
print(" ".join(str(int(c)) for c in counts[:st.n_colors]))
w.close()
