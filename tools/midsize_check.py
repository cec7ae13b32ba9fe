"""Ad-hoc: mid-size scenes between the tuned sizes run clean (no overflow / hand-off timeout), twice the same bits."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import physics_amd
from physics_amd import scenes

DT = 16_666_667
for dims in ((30, 24, 30), (40, 30, 40), (64, 20, 64)):
    sc = scenes.c3(*dims)
    outs = []
    for rep in range(2):
        w = physics_amd.World(sc.config())
        sc.populate(w)
        w.update_n(DT, 150)
        w.sync()
        t0 = time.perf_counter()
        w.update_n(DT, 100)
        w.sync()
        dt = time.perf_counter() - t0
        st = w.get_stats()
        outs.append(w.get_transforms())
        w.close()
    same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    print(dims, sc.n, "manifolds", st.n_manifolds, "colors", st.n_colors, "overflow", st.overflow, f"{100 / dt:.0f} steps/s", "deterministic" if same else "DIFFERENT")
