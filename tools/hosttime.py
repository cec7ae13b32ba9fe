import sys, time; sys.path.insert(0, ".")
import physics_amd
from physics_amd import scenes
for name, pre in (("c1", 100), ("c2", 150)):
    sc = scenes.SCENES[name]()
    w = physics_amd.World(sc.config()); sc.populate(w)
    w.update_n(scenes.DT_NANOS, pre); w.sync()
    t0 = time.perf_counter(); w.update_n(scenes.DT_NANOS, 200); t1 = time.perf_counter(); w.sync(); t2 = time.perf_counter()
    print(name, "enqueue ms/step %.4f  total ms/step %.4f" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
    # host cost alone: three steps enqueued on an idle stream (inside the snapshot ring, so nothing waits for the GPU)
    best = 1e9
    for _ in range(20):
        w.sync()
        t0 = time.perf_counter(); w.update_n(scenes.DT_NANOS, 3); t1 = time.perf_counter()
        best = min(best, (t1 - t0) / 3)
    w.sync()
    print(name, "host-only enqueue ms/step %.4f" % (best * 1e3))
