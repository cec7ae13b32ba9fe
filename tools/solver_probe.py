#!/usr/bin/env python3
"""Run by the GPU tests in a FRESH process, with PHYS_DEBUG_* switches in the environment (the library reads them once
per process): steps one scene on two solver paths - and optionally the CPU oracle - and prints whether every bit of
poses and velocities agrees, and which solver stages ran.

    solver_probe.py c5:16:130:16 --pre 8 --steps 6 --a cluster --b percolour [--oracle] [--expect-error]

paths: default | cluster (PHYS_FLAG_SOLVER_CLUSTER) | percolour (PHYS_FLAG_SOLVER_PER_COLOR)
Output (one line): identical|different a_ran=<stages> b_ran=<stages> [oracle=identical|different] manifolds=<n> colours=<n>
With --expect-error: prints 'error <code> overflow=<bits>' or 'no error'."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import physics_amd  # noqa: E402
from physics_amd import scenes  # noqa: E402

DT = 16_666_667
FLAGS = {"default": 0, "cluster": physics_amd.FLAG_SOLVER_CLUSTER, "percolour": physics_amd.FLAG_SOLVER_PER_COLOR}


def make_scene(spec):
    parts = spec.split(":")
    dims = [int(x) for x in parts[1:]]
    return {"c5": scenes.c5, "c3": scenes.c3}[parts[0]](*dims) if dims else scenes.SCENES[parts[0]]()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--pre", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--a", default="cluster")
    ap.add_argument("--b", default="percolour")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--expect-error", action="store_true")
    args = ap.parse_args()
    sc = make_scene(args.scene)
    if args.expect_error:
        w = physics_amd.World(sc.config(flags=sc.flags | FLAGS[args.a]))
        sc.populate(w)
        try:
            w.update_n(DT, args.pre + args.steps)
            w.sync()
            print("no error overflow=%d" % w.get_stats().overflow)
        except physics_amd.PhysError as e:
            print("error %d overflow=%d" % (e.code, w.get_stats().overflow))
        return
    states, ran, st = [], [], None
    for path in (args.a, args.b):
        w = physics_amd.World(sc.config(flags=sc.flags | FLAGS[path]))
        sc.populate(w)
        try:
            w.update_n(DT, args.pre)
            w.profile_enable(True)
            w.update_n(DT, args.steps)
            w.sync()
        except physics_amd.PhysError as e:  # a flagged step is an answer (never a silent divergence), not a crash
            print("error %d overflow=%d path=%s" % (e.code, w.get_stats().overflow, path))
            return
        prof, _ = w.profile_get()
        ran.append("+".join(sorted(s for s in prof if s.startswith("solve"))))
        st = w.get_stats()
        assert st.overflow == 0, st.overflow
        states.append(w.get_transforms() + w.get_velocities())
        w.close()
    same = all(np.array_equal(a, b) for a, b in zip(states[0], states[1]))
    out = f"{'identical' if same else 'different'} a_ran={ran[0]} b_ran={ran[1]}"
    if args.oracle:
        from oracle import binding as ob
        o = ob.OracleWorld(sc.config(), trig=ob.TRIG_DET)
        o.set_threads(min(len(os.sched_getaffinity(0)), 16))
        sc.populate(o)
        o.update_n(DT, args.pre + args.steps)
        ref = o.get_transforms() + o.get_velocities()
        out += " oracle=" + ("identical" if all(np.array_equal(a, b) for a, b in zip(states[0], ref)) else "different")
    print(out + f" manifolds={st.n_manifolds} colours={st.n_colors}")


if __name__ == "__main__":
    main()
