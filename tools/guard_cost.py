import sys, time
sys.path.insert(0, '/root/repo')
import physics_amd
from physics_amd import scenes
DT = 16_666_667
for name in ("c5", "t1m"):
    for label, extra in (("exclusive", physics_amd.FLAG_EXCLUSIVE_GPU), ("guarded (default)", 0)):
        sc = scenes.SCENES[name]()
        w = physics_amd.World(sc.config(flags=sc.flags | extra))
        sc.populate(w)
        w.update_n(DT, 35 if name == "c5" else 105)
        w.sync()
        w.profile_enable(True)
        t0 = time.perf_counter()
        w.update_n(DT, 20)
        w.sync()
        dt = (time.perf_counter() - t0) / 20
        prof, _ = w.profile_get()
        print(name, label, "ms/step %.4f" % (dt * 1e3), "solve_cluster ms %.4f launches/step %.1f" % (prof["solve_cluster"][0] / 20, prof["solve_cluster"][1] / 20))
        w.close()
