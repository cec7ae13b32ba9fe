#!/usr/bin/env python3
"""Wall time of every single update of a scene (sync after each): shows the periodic full re-colouring and any other
step that stands out.   python tools/step_times.py c5 [first] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
count = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sc = scenes.SCENES[wl]()
w = physics_amd.World(sc.config())
sc.populate(w)
w.update_n(scenes.DT_NANOS, first)
w.sync()
out = []
for k in range(count):
    t0 = time.perf_counter()
    w.update(scenes.DT_NANOS)
    w.sync()
    out.append((time.perf_counter() - t0) * 1e3)
if len(sys.argv) > 4:  # stage times of ONE update: python tools/step_times.py c5 64 0 stages  -> update number 64
    w.profile_enable(True)
    w.update(scenes.DT_NANOS)
    w.sync()
    prof, psteps = w.profile_get()
    print("stages of update", first + count, {k: (round(ms, 3), n) for k, (ms, n) in prof.items() if n})
st = w.get_stats()
print(wl, "steps", first, "..", first + count, "manifolds", st.n_manifolds, "colours", st.n_colors, "rounds", st.color_rounds)
print(" ".join(f"{t:.2f}" for t in out))
w.close()
