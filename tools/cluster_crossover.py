#!/usr/bin/env python3
"""Where the cluster solver overtakes the dataflow kernel: towers of a few heights (and mixed piles), each stepped on the
default path with the threshold out of the way (PHYS_FLAG_SOLVER_CLUSTER) and with the cluster solver off
(PHYS_DEBUG_NO_CLUSTER is read once per process, so the two paths run in separate processes: --path).

    cluster_crossover.py --path cluster|flow  ->  one line per scene: bodies manifolds colours solver-ms-per-step"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DT = 16_666_667


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", choices=["cluster", "flow"], required=True)
    args = ap.parse_args()
    if args.path == "flow":
        os.environ["PHYS_DEBUG_NO_CLUSTER"] = "1"
    import physics_amd
    from physics_amd import scenes
    specs = [("tower 16x130x16", lambda: scenes.c5(16, 130, 16)), ("tower 16x200x16", lambda: scenes.c5(16, 200, 16)),
             ("tower 16x300x16", lambda: scenes.c5(16, 300, 16)), ("tower 16x400x16", lambda: scenes.c5(16, 400, 16)),
             ("mixed 40x30x40", lambda: scenes.c3(40, 30, 40)), ("mixed 50x30x50", lambda: scenes.c3(50, 30, 50)),
             ("mixed 50x40x50", lambda: scenes.c3(50, 40, 50))]
    for name, make in specs:
        sc = make()
        extra = physics_amd.FLAG_SOLVER_CLUSTER if args.path == "cluster" else 0
        w = physics_amd.World(sc.config(flags=sc.flags | extra | physics_amd.FLAG_EXCLUSIVE_GPU))
        sc.populate(w)
        w.update_n(DT, 120)
        w.sync()
        w.profile_enable(True)
        w.update_n(DT, 20)
        w.sync()
        prof, _ = w.profile_get()
        st = w.get_stats()
        solve = sum(v[0] for k, v in prof.items() if k.startswith("solve")) / 20.0
        rows = prof.get("rows", (0.0, 0))[0] / 20.0
        print(f"{args.path:8s} {name:18s} bodies {st.n_bodies:7d} manifolds {st.n_manifolds:8d} colours {st.n_colors:3d} "
              f"solve {solve:.3f} ms rows {rows:.3f} ms  ({'+'.join(sorted(k for k in prof if k.startswith('solve')))})")
        w.close()


if __name__ == "__main__":
    main()
