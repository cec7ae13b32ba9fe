#!/usr/bin/env python3
"""A/B of solver kernel variants on one workload: every variant runs in its own child process (the debug switches are
read once per process), steps the same scene, and prints steps/s, the solver stages and a hash of the final
transforms + velocities (all variants must print the SAME hash: the variants differ in schedule, never in bits).

    python tools/ab_solver.py c5 60 lane quad          # per-colour kernels: one lane / four lanes per manifold
    python tools/ab_solver.py c3 200 flow quad lane    # dataflow kernel against the per-colour kernels
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NC = {"PHYS_DEBUG_NO_CLUSTER": "1"}
VARIANTS = {
    "lane": dict(NC, PHYS_DEBUG_COLOR_KERNEL="lane", PHYS_DEBUG_FLOW_MAX="0"),
    "quad": dict(NC, PHYS_DEBUG_COLOR_KERNEL="quad", PHYS_DEBUG_FLOW_MAX="0"),
    "auto": dict(NC, PHYS_DEBUG_FLOW_MAX="0"),
    "flow": dict(NC, PHYS_DEBUG_FLOW_MAX="100000000"),
    "cluster": {"PHYS_DEBUG_CLUSTER_MIN": "0"},   # the cluster solver whenever the scene has clusters (>= 32768 bodies)
    "nocluster": dict(NC),
    "cluster1": {"PHYS_DEBUG_CLUSTER_MIN": "0", "PHYS_DEBUG_CLUSTERS_PER_CU": "1"},
    "cluster3": {"PHYS_DEBUG_CLUSTER_MIN": "0", "PHYS_DEBUG_CLUSTERS_PER_CU": "3"},
    "cl_nogran": {"PHYS_DEBUG_CLUSTER_MIN": "0", "PHYS_DEBUG_ABLATE": "8"},    # timing only: shared bodies treated as local
    "cl_nocompute": {"PHYS_DEBUG_CLUSTER_MIN": "0", "PHYS_DEBUG_ABLATE": "2"},  # timing only
    "cl_bare": {"PHYS_DEBUG_CLUSTER_MIN": "0", "PHYS_DEBUG_ABLATE": "10"},      # timing only: neither
    "default": {},
    "cl_norot": {"PHYS_DEBUG_ABLATE": "16"},    # same bits: rows always start on wave 0
    "cl_remass": {"PHYS_DEBUG_ABLATE": "32"},   # same bits: row masses remade in every iteration
    "cl_norot_remass": {"PHYS_DEBUG_ABLATE": "48"},
    "cluster2": {"PHYS_DEBUG_CLUSTERS_PER_CU": "2"},
    "dyn": {"PHYS_DEBUG_CLUSTER_DYNAMIC": "1"},                                     # clusters remade every update
    "dyn_cap": {"PHYS_DEBUG_CLUSTER_DYNAMIC": "1", "PHYS_DEBUG_CLUSTER_CAP": "60000"},  # ... with homes for 60k bodies only
    "flow_pipe": {"PHYS_DEBUG_FLOW_MAX": "100000000", "PHYS_DEBUG_NO_CLUSTER": "1", "PHYS_DEBUG_FLOW_PIPELINE": "1"},
    "flow_nopipe": {"PHYS_DEBUG_FLOW_MAX": "100000000", "PHYS_DEBUG_NO_CLUSTER": "1", "PHYS_DEBUG_FLOW_PIPELINE": "0"},
    "pairs4": {"PHYS_DEBUG_PAIR_LANES": "4"},
    "pairs1": {"PHYS_DEBUG_PAIR_LANES": "1"},
    "np128": {"PHYS_DEBUG_NP_THREADS": "128"},
    "np256": {"PHYS_DEBUG_NP_THREADS": "256"},
    "no_ctab": {"PHYS_DEBUG_NO_CTAB": "1"},   # timing of the rows stage without the colour-table build (colours differ)
    "cl_allcus": {"PHYS_DEBUG_CLUSTER_SPARE": "0"},
    "cl_spare16": {"PHYS_DEBUG_CLUSTER_SPARE": "16"},
    # timing diagnosis of the one-lane per-colour kernel (results are WRONG by construction: hashes differ)
    "nogather": dict(NC, PHYS_DEBUG_COLOR_KERNEL="lane", PHYS_DEBUG_FLOW_MAX="0", PHYS_DEBUG_ABLATE="1"),
    "nocompute": dict(NC, PHYS_DEBUG_COLOR_KERNEL="lane", PHYS_DEBUG_FLOW_MAX="0", PHYS_DEBUG_ABLATE="2"),
    "nostore": dict(NC, PHYS_DEBUG_COLOR_KERNEL="lane", PHYS_DEBUG_FLOW_MAX="0", PHYS_DEBUG_ABLATE="4"),
    "loadonly": dict(NC, PHYS_DEBUG_COLOR_KERNEL="lane", PHYS_DEBUG_FLOW_MAX="0", PHYS_DEBUG_ABLATE="7"),
}


def child(workload, steps, preroll):
    sys.path.insert(0, ROOT)
    import numpy as np
    import physics_amd
    from physics_amd import scenes
    sc = scenes.SCENES[workload]()
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.update_n(scenes.DT_NANOS, preroll)
    w.sync()
    t0 = time.perf_counter()
    w.update_n(scenes.DT_NANOS, steps)
    w.sync()
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for a in w.get_transforms() + w.get_velocities():
        h.update(np.ascontiguousarray(a).tobytes())
    st = w.get_stats()
    w.profile_enable(True)
    w.update_n(scenes.DT_NANOS, 10)
    w.sync()
    prof, psteps = w.profile_get()
    w.profile_enable(False)
    stages = {k: [round(ms / psteps, 4), round(n / psteps, 1)] for k, (ms, n) in prof.items()}
    print(json.dumps({"steps_per_sec": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 4),
                      "hash": h.hexdigest()[:16], "n_manifolds": int(st.n_manifolds), "n_colors": int(st.n_colors),
                      "overflow": int(st.overflow), "stages_ms_launches": stages}))
    w.close()


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    workload, steps = sys.argv[1], int(sys.argv[2])
    variants = sys.argv[3:] or ["lane", "quad"]
    preroll = {"c1": 100, "c2": 150, "c3": 150, "c5": 30, "t1m": 100}.get(workload, 30)
    hashes = set()
    for v in variants:
        env = dict(os.environ, **VARIANTS[v])
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", workload, str(steps), str(preroll)],
                             env=env, capture_output=True, text=True, timeout=900)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:]
        print(workload, v, line, flush=True)
        try:
            hashes.add(json.loads(line)["hash"])
        except Exception:
            hashes.add("FAILED " + v)
    print(workload, "BIT-IDENTICAL" if len(hashes) == 1 else f"MISMATCH {hashes}", flush=True)
    return 0  # a mismatch is reported in the output; only a crash or a timeout of a child stops a chain of runs


if __name__ == "__main__":
    sys.exit(main())
