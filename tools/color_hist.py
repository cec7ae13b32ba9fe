#!/usr/bin/env python3
"""Manifolds per solver colour of a workload at the start of bench.py's timed window (how sparse the trailing colours are)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402

for wl, pre in (("c5", 35), ("t1m", 105), ("c3", 155)):
    if len(sys.argv) > 1 and wl not in sys.argv[1:]:
        continue
    sc = scenes.SCENES[wl]()
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.update_n(scenes.DT_NANOS, pre)
    w.sync()
    st = w.get_stats()
    print(wl, "manifolds", st.n_manifolds, "colours", st.n_colors, [int(c) for c in w.get_color_counts()[:st.n_colors]], flush=True)
    w.close()
