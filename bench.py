#!/usr/bin/env python3
"""bench.py — throughput of the per-frame rigid-body step (phys_update) on MI355X.

A "step" is one PhysicsState::update over one synthetic scene resident in HBM.

N = 1 workload: C5 = BASELINE.json configs[4], the 256k stacked-box tower: the LARGEST single-GPU configuration
(BASELINE.json's metric is not quoted on one config). The other single-GPU configs (the 1M-cube north_star target,
C3, C2) follow as sub-records under "other_workloads", each with its own timing, roofline and CPU sample.

Timing: the window [preroll + W warm-up steps, then EXACTLY K timed steps] is repeated `--reps` times (default 5),
every repetition on a freshly created world stepped from the same initial scene, so every repetition times the
very same K steps of the same trajectory (the step is deterministic: bit-identical states). `value` comes from the
MEDIAN repetition; min / max are reported beside it. Each timed region is bracketed by barrier +
torch.cuda.synchronize() + phys_sync on both sides; with several ranks the time of a repetition is the MAX over ranks.

N > 1 (`--gpus N`): weak scaling; every rank owns one workload-shaped slab placed side by side along x and, once per
step, all-gathers the state of its boundary bodies over RCCL (behind the C ABI: phys_halo_exchange; one process per
GPU): the neighbours' boundary bodies take part in the rank's broad phase, narrow phase and solver as kinematic ghosts. Started without a
launcher (no WORLD_SIZE in the environment) this script starts the N ranks itself, BEFORE anything touches the GPU.

Prints ONE JSON line (rank 0). `roofline` describes the dominant kernel of the timed window, timed live with HIP
events on the library's own stream (phys_profile_*) over the SAME window on one more fresh world. `traffic` and
`avg_kernel_us_rocprofv3` come from the committed rocprofv3 passes of `bench.py --profile-window` and are emitted
only when that pass covered the same window at the same scene state (profiles/r2_<workload>_window.json says which),
otherwise null. `cpu_baseline` is the CPU oracle (a scalar C++ restatement; the Rust reference cannot be built here)
seeded with the GPU world's state at the start of the timed window and timed on the host cores for a few steps.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
PROFILE_ROUND = "r3"

KERNEL_OF_STAGE = {
    "step_full": "k_step_full", "velocity_aabb": "k_step_velocity_aabb",
    "grid": "k_cell_insert",  # small scenes; larger ones: k_cell_assign + scan + k_scatter
    "pairs": "k_find_pairs", "narrow": "k_narrowphase",
    "color": "k_color_small",  # small scenes; larger ones: k_color_round x rounds + k_color_finish
    "rows": "k_rows_build",    # + k_color_hist / k_color_offsets / k_color_place beyond 40k manifolds
    "solve": "k_solve_color",  # k_solve_color_quad (four lanes per manifold) unless PHYS_DEBUG_COLOR_KERNEL=lane
    "solve_tail": "k_solve_tail", "solve_flow": "k_solve_flow", "solve_cluster": "k_solve_cluster", "position": "k_step_position",
    "constraints": "k_constraint_solve",
}

# t1m_settled: the 1M cubes after every column has come to rest (the top cube falls 50 units: 192 updates, then the pile
# settles) - the heaviest state of the north_star scene
DEFAULT_PREROLL = {"c1": 100, "c2": 150, "c3": 150, "c5": 30, "t1m": 100, "c4": 0, "t1m_settled": 600, "ref_1m": 10, "ref_cg": 10}
WORKLOAD_NAMES = {"t1m_settled": "T_1M_cubes_settled"}
# the kernel a sub-record is ABOUT, where that is not simply the longest stage of the step (the kernels that
# restate the reference's own functions, and the pair search of the broad-phase-only configuration)
ROOFLINE_STAGE = {"ref_1m": "step_full", "ref_cg": "constraints", "c4": "pairs"}


# warm starting (every bench scene), per manifold in k_rows_build: index of the previous manifold 4, its
# geometry record 96 and impulse record 48 (a pile at rest matches nearly every manifold); + the starting impulses
# written, 12 per contact point
WARM_ROWS = 148


def stage_bytes(stage, st, iters, cluster=False):
    """Algorithmic HBM bytes of one STEP spent in `stage` (formulas: DESIGN.md 'algorithmic bytes').
    cluster: the step ran the cluster solver, whose rows are compact and built without reading the bodies."""
    n, p, m, k = st["n_bodies"], st["n_pairs"], st["n_manifolds"], st["n_contacts"]
    mb = m - st["n_ground_manifolds"]
    if stage == "step_full":
        return 120 * n
    if stage == "constraints":
        # per CG iteration and row: p, Ap, r, x read, x, r, p, Ap written, the column table (60 B)
        return 60 * st.get("cg_rows", 0) * st.get("cg_iterations_per_update", 0.0)
    if stage == "velocity_aabb":
        return 132 * n
    if stage == "position":
        return 80 * n
    if stage == "grid":
        return 24 * n + 8 * n + 28 * n  # AABB read, bucket+rank, bucket-ordered copy (ids + boxes)
    if stage == "pairs":
        return 24 * n + 8 * p
    if stage == "narrow":
        # + warm starting: index of the pair's previous manifold 4, this update's impulse record zeroed 48
        return 96 * p + 44 * st["n_ground_manifolds"] + 100 * m + 52 * m
    if stage == "color":
        return 28 * m  # ids + priority + colour + slot, once
    if stage == "rows" and cluster:
        # manifold record read (ids, count, normal 24 + 16 per point), compact row written (32 + 16 per point),
        # permutation + colour 8, per body side {used mask 8, cluster slot 4, remote-colour mask 8}
        return (24 + 32 + 8) * m + 32 * k + 20 * (m + mb) + WARM_ROWS * m + 12 * k
    if stage == "rows":
        return 100 * m + 24 * m + 76 * k + 52 * (m + mb) + WARM_ROWS * m + 12 * k
    if stage == "solve_cluster":
        # compact rows streamed once per iteration: header + normal 32 per manifold; per point {contact point, bias} 16,
        # accumulated impulses 12 read + 12 written, row masses 12 read (written once); body velocities stay in LDS.
        # Warm starting: sweep 0 reads header, normal, points and the starting impulses and writes the row masses
        # (32 m + 40 k); the last sweep leaves the 48-byte impulse record of every manifold
        return iters * (32 * m + 52 * k) + (32 * m + 40 * k) + 48 * m
    if stage in ("solve", "solve_flow"):
        # (k_solve_flow makes all iterations in one launch; same job, same compulsory bytes)
        # per body and iteration: v, w read 24 + written 24, inverse mass 4, inverse inertia diagonal 12
        # (all benchmark scenes have diagonal tensors; 36 with a full tensor)
        # Warm starting: sweep 0 is one more pass that does not write the impulses back (12 k less); + the impulse records
        return iters * (24 * m + 64 * k + 64 * (m + mb)) + (24 * m + 52 * k + 64 * (m + mb)) + 48 * m
    return 0


def _kernel_matches(name, kernel):
    """Kernel name of a profile table vs a kernel family: k_solve_flow also covers k_solve_flow_quad, k_solve_color
    also k_solve_color_quad."""
    base = name.split("(")[0].split("<")[0].replace("void ", "").strip().split("::")[-1]
    return base == kernel or base.startswith(kernel + "_")


def committed_window(workload_key):
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{workload_key}_window.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


def committed_profile_fields(workload_key, kernel, window, st):
    """(kernel name, rocprofv3 average us, PMC traffic bytes per launch, note). The committed numbers are handed out
    ONLY when they were taken over the same window (preroll, warm-up, steps) and the scene was in the same state
    (manifold and colour counts equal): nothing of another state is ever mixed into one roofline object."""
    c = committed_window(workload_key)
    if c is None:
        return None, None, None, f"no committed profile for this workload (profiles/{PROFILE_ROUND}_{workload_key}_window.json)"
    cw, cs = c.get("window", {}), c.get("scene_stats", {})
    same_window = all(cw.get(k) == window[k] for k in ("preroll", "warmup", "steps"))
    same_state = all(cs.get(k) == st[k] for k in ("n_bodies", "n_pairs", "n_manifolds", "n_colors"))
    if not (same_window and same_state):
        return None, None, None, (f"committed profile covers window {cw} at {cs.get('n_manifolds')} manifolds / "
                                  f"{cs.get('n_colors')} colours; this run: {window} at {st['n_manifolds']} / {st['n_colors']}: "
                                  f"not mixed in")
    rows = [(name, r) for name, r in c.get("kernels", {}).items() if _kernel_matches(name, kernel)]
    if not rows:
        return None, None, None, "kernel not in the committed profile"
    name, r = max(rows, key=lambda kv: kv[1].get("total_us", 0.0))
    short = name.split("(")[0].replace("void ", "").strip().split("::")[-1]
    return short, r.get("avg_us"), r.get("traffic_bytes"), f"profiles/{PROFILE_ROUND}_{workload_key}_window.json (same window, same scene state)"


def stats_dict(s):
    return {f: int(getattr(s, f)) for f in ("n_bodies", "n_pairs", "n_manifolds", "n_contacts", "n_colors",
                                            "color_rounds", "n_ground_manifolds", "cg_iterations")}


class Rig:
    """Everything one rank needs to build its world again and again (one fresh world per repetition)."""

    def __init__(self, workload, rank, world_size, local_rank, dist, rehearsal, sharded, strong=False):
        from physics_amd import scenes
        self.workload, self.rank, self.world_size, self.local_rank = workload, rank, world_size, local_rank
        self.dist, self.rehearsal, self.sharded = dist, rehearsal, sharded
        self.strong = bool(strong and sharded)
        self.halo = None
        if not sharded:
            self.scene = scenes.SCENES[workload]()
        else:
            from physics_amd import sharding
            self.scene, self.halo = sharding.make_rank_scene(workload, rank, world_size, dist, local_rank, pinned_host=rehearsal,
                                                             strong=self.strong)
        self.iters = self.scene.solver_iterations

    def make_world(self):
        import physics_amd
        # the bench has its GPU to itself (one rank per GPU, nothing else on it): PHYS_FLAG_EXCLUSIVE_GPU, stated in
        # config.flags_note (the default - a guarded start of the cluster solver's launch - costs 0.04 ms per update)
        flags = self.scene.flags | (0 if self.rehearsal else physics_amd.FLAG_EXCLUSIVE_GPU)
        w = physics_amd.World(self.scene.config(device=self.local_rank, flags=flags))
        self.scene.populate(w)
        if self.halo is not None:
            self.halo.attach(w, self.scene)
        return w

    def advance(self, world, steps):
        from physics_amd.scenes import DT_NANOS
        if steps <= 0:
            return
        if self.halo is None:
            world.update_n(DT_NANOS, steps)
        elif self.halo.before_update:  # ghost bodies: exchanged first, collided with inside the update
            for _ in range(steps):
                self.halo.exchange(world)
                world.update(DT_NANOS)
        else:                          # broad-phase-only: boundary AABBs against the grid the update built
            for _ in range(steps):
                world.update(DT_NANOS)
                self.halo.exchange(world)

    def fence(self, world):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()
        world.sync()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.rehearsal else f"cuda:{self.local_rank}")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.rehearsal else f"cuda:{self.local_rank}")
        self.dist.all_reduce(t)
        return float(t.item())


def timed_repetition(rig, preroll, warmup, steps, want_state=False):
    """One fresh world: preroll + warm-up (untimed), then EXACTLY `steps` timed steps. Returns (seconds = max over
    ranks, stats after the window, state at the START of the window if asked for)."""
    import torch
    w = rig.make_world()
    rig.advance(w, preroll)
    w.sync()
    rig.advance(w, warmup)
    state = None
    if want_state:
        w.sync()
        state = w.get_transforms() + w.get_velocities()
    rig.fence(w)
    t0 = time.perf_counter()
    rig.advance(w, steps)
    w.sync()
    torch.cuda.synchronize()
    if rig.dist is not None:
        rig.dist.barrier()
    elapsed = rig.max_over_ranks(time.perf_counter() - t0)
    st = stats_dict(w.get_stats())
    cross = int(w.get_stats().n_cross_pairs) if rig.halo is not None else 0
    w.close()
    return elapsed, st, cross, state


def timed_period(rig, preroll, warmup):
    """One more fresh world: from the start of the timed window, PHYS_COLOR_CACHE_PERIOD (64) consecutive steps - exactly
    one of them is the periodic full re-colouring update (contact_solve.h), which a 20-step window may or may not
    contain. Returns steps per second over the whole period, and the slowest single step in ms."""
    import torch
    w = rig.make_world()
    rig.advance(w, preroll)
    rig.advance(w, warmup)
    rig.fence(w)
    period = 64
    per_step = []
    t0 = time.perf_counter()
    for _ in range(period):
        t1 = time.perf_counter()
        rig.advance(w, 1)
        w.sync()
        per_step.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    if rig.dist is not None:
        rig.dist.barrier()
    elapsed = rig.max_over_ranks(time.perf_counter() - t0)
    w.close()
    # a HITCH is a step far slower than the steps around it (a growing pile makes every later step slower: no hitch)
    hitch = 0.0
    for i, t in enumerate(per_step):
        near = sorted(per_step[max(0, i - 4):i] + per_step[i + 1:i + 5])
        hitch = max(hitch, t / near[len(near) // 2])
    return period / elapsed, 1e3 * max(per_step), 1e3 * statistics.median(per_step), hitch, 1e3 * per_step[0], 1e3 * per_step[-1]


def profile_window(rig, preroll, warmup, steps, workload_key):
    """The SAME window once more on a fresh world, every launch bracketed by HIP events on the library's stream.
    Returns the roofline object of the dominant kernel, the per-stage table and the scene stats after the window."""
    w = rig.make_world()
    rig.advance(w, preroll)
    rig.advance(w, warmup)
    w.sync()
    w.profile_enable(True)
    cg_total = 0
    if rig.scene.constraints:
        for _ in range(steps):  # the CG iteration count is a per-update figure: read it after every update
            rig.advance(w, 1)
            w.sync()
            cg_total += int(w.get_stats().cg_iterations)
    else:
        rig.advance(w, steps)
    w.sync()
    prof, psteps = w.profile_get()
    w.profile_enable(False)
    st = stats_dict(w.get_stats())
    if rig.scene.constraints:
        st["cg_rows"] = 3 * len(rig.scene.constraints)
        st["cg_iterations_per_update"] = cg_total / max(steps, 1)
    counts = [int(c) for c in w.get_color_counts()[:st["n_colors"]]]
    w.close()
    iters = rig.iters
    table = {}
    for stage, (ms, launches) in prof.items():
        table[stage] = {"ms_per_step": ms / max(psteps, 1), "launches_per_step": launches / max(psteps, 1),
                        "avg_launch_us": 1e3 * ms / max(launches, 1)}
    kernel_stages = [s for s in table if s in KERNEL_OF_STAGE]
    if not kernel_stages:
        return None, table, st
    dom = max(kernel_stages, key=lambda s: table[s]["ms_per_step"])
    if ROOFLINE_STAGE.get(workload_key) in table:
        dom = ROOFLINE_STAGE[workload_key]
    cluster = "solve_cluster" in table
    b_step = stage_bytes(dom, st, iters, cluster)
    if dom == "solve" and "solve_tail" in table:
        # k_solve_color only runs the colours that got a launch of their own; the trailing small colours
        # (<= 512 manifolds each, at least two of them) are solved by k_solve_tail. Scale the stage's bytes by the
        # share of manifolds in the individually launched colours (same rule as launch_solver).
        big = len(counts)
        while big > 0 and counts[big - 1] <= 512:
            big -= 1
        if len(counts) - big < 2:
            big = len(counts)
        if sum(counts):
            b_step = b_step * sum(counts[:big]) / sum(counts)
    launches = table[dom]["launches_per_step"]
    per_launch = b_step / max(launches, 1e-9)
    dur_s = table[dom]["avg_launch_us"] * 1e-6
    achieved = per_launch / dur_s / 1e9 if dur_s > 0 else 0.0
    total_bytes = sum(stage_bytes(s, st, iters, cluster) for s in kernel_stages)
    family = KERNEL_OF_STAGE[dom]
    window = {"preroll": preroll, "warmup": warmup, "steps": steps}
    prof_name, prof_us, traffic, note = committed_profile_fields(workload_key, family, window, st)
    roof = {"bound": "hbm", "kernel": prof_name or family, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_us": round(table[dom]["avg_launch_us"], 3),
            "avg_kernel_us_rocprofv3": prof_us, "committed_profile": note,
            "launches_per_step": round(launches, 2), "stage_share_of_device_time": round(
                table[dom]["ms_per_step"] / max(sum(t["ms_per_step"] for t in table.values()), 1e-12), 3),
            "algorithmic_bytes_per_step_all_kernels": int(total_bytes),
            "all_kernels_achieved_gbs": round(total_bytes / max(sum(t["ms_per_step"] for t in table.values()), 1e-12) / 1e6, 1),
            "window": window, "scene_stats_after_window": st}
    return roof, table, st


def copy_ceiling_gbs(device):
    """Second denominator: what a plain device-to-device copy reaches on this box, read + write
    bytes per second, best of 5 copies of 1 GiB."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    b.copy_(a)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return round(best, 1)


def cpu_baseline(scene, state, window_start, sample_steps, budget_s=14.0):
    """The CPU oracle on the host cores, on a bounded sample: seeded with the GPU world's state at the start of the
    timed window (poses and velocities), one untimed step (its first colouring starts from scratch), then up to
    `sample_steps` timed steps or `budget_s` seconds, whichever comes first. One thread, then all host cores (OpenMP)."""
    from oracle import binding as ob
    from physics_amd.scenes import DT_NANOS
    pos, rot, lin, ang = state

    cg_iters = [0]

    def timed(threads):
        o = ob.OracleWorld(scene.config(), trig=ob.TRIG_DET)
        scene.populate(o, (pos, rot, lin, ang))
        o.set_threads(threads)
        o.update(DT_NANOS)
        done, t0 = 0, time.perf_counter()
        # at least `sample_steps` steps, and at least a second of them where a step is cheap (at most 400); never
        # beyond the budget
        while done < sample_steps or (time.perf_counter() - t0 < 1.0 and done < 400):
            o.update(DT_NANOS)
            done += 1
            if scene.constraints and threads == 1:
                cg_iters[0] += int(o.get_stats().cg_iterations)
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        o.close()
        return done, dt

    k1, dt = timed(1)
    # additionally the OpenMP variant of the same oracle (same bits for any thread count) on the host
    # cores this process may use, at most 16 (the CPU share of one GPU on the bench boxes)
    cores = min(len(os.sched_getaffinity(0)), 16)
    if scene.constraints:
        cores = 1  # the reference path (gravity, constraints, CG, integrate) has no OpenMP loops: one figure
    km, dt_mt = timed(cores) if cores > 1 else (k1, dt)
    extra = {}
    if scene.constraints and cg_iters[0]:
        extra = {"cg_iterations_per_update": round(cg_iters[0] / k1, 2),
                 "us_per_update": round(1e6 * dt / k1, 1),
                 "note_cg": "whole PhysicsState::update of the oracle (gathers of 6N vectors, assembly, CG, integrate) per update; "
                            "the restatement keeps the reference's per-row heap vectors (sparse_matrix.rs:30,44)"}
    return {**extra, "value": round(scene.n * k1 / dt, 1), "unit": "body-steps/s", "cores": 1, "kind": "port",
            "steps_per_sec": round(k1 / dt, 4),
            "sample": f"oracle (scalar C++ restatement of the reference step + CPU collision stages, 1 thread; the Rust "
                      f"reference is not buildable here: no cargo/rustc), workload {scene.name}, seeded with the GPU world's "
                      f"poses and velocities at step {window_start} (start of the timed window), 1 untimed step, then "
                      f"{k1} timed steps",
            "openmp": {"value": round(scene.n * km / dt_mt, 1), "unit": "body-steps/s", "cores": cores,
                       "steps_per_sec": round(km / dt_mt, 4), "steps_timed": km,
                       "note": "same oracle, same sample, OpenMP over the independent loops of the collision stages "
                               "(AABBs, grid search, narrow phase, row preparation, the manifolds of one colour); the "
                               "reference itself is single-threaded"}}


def measure_workload(rig, args, preroll, reps, with_cpu):
    """Timed repetitions + profile window + (optionally) CPU sample of one workload on this rank's rig."""
    times, st, cross, state = [], None, 0, None
    for r in range(reps):
        e, st, cross, s = timed_repetition(rig, preroll, args.warmup, args.steps, want_state=(with_cpu and r == 0))
        times.append(e)
        if s is not None:
            state = s
    med = statistics.median(times)
    n_total = int(round(rig.sum_over_ranks(float(rig.scene.n))))  # strong scaling: the slabs differ by a body or two
    pairs_total = rig.sum_over_ranks(float(st["n_pairs"] + cross))
    if rig.scene.constraints:
        st["cg_rows"] = 3 * len(rig.scene.constraints)
    rec = {
        "value": round(n_total * args.steps / med, 1), "steps_per_sec": round(args.steps / med, 2),
        "ms_per_step": round(1e3 * med / args.steps, 4),
        "repetitions": {"n": reps, "statistic": "median",
                        "steps_per_sec": [round(args.steps / t, 2) for t in times],
                        "min_steps_per_sec": round(args.steps / max(times), 2),
                        "max_steps_per_sec": round(args.steps / min(times), 2),
                        "every_repetition": "fresh world, same scene, same preroll + warm-up: the same K steps of the same trajectory"},
        "pairs_per_sec": round(pairs_total * args.steps / med, 1),
        "scene_stats": st, "n_bodies": n_total, "preroll": preroll, "cross_pairs_rank": cross,
    }
    if (rig.scene.flags & 1) and not (rig.scene.flags & 8) and st["n_manifolds"] > 0:  # FLAG_COLLISIONS, not broad-phase-only
        sps, slowest, med, hitch, first, last = timed_period(rig, preroll, args.warmup)
        rec["full_period"] = {"steps": 64, "steps_per_sec": round(sps, 2), "slowest_step_ms": round(slowest, 3),
                              "median_step_ms": round(med, 3), "first_step_ms": round(first, 3), "last_step_ms": round(last, 3),
                              "worst_step_over_median_of_its_8_neighbours": round(hitch, 3),
                              "note": "64 consecutive steps from the start of the timed window, synchronised after every step: "
                                      "contains exactly one rebuild of the persistent colour table (every 64th update; colours are "
                                      "never re-made: contact_solve.h) and, with dynamic clusters, eight deals of the homes; the scene "
                                      "keeps evolving over the period (a growing pile gets slower)"}
    # sharded: every rank steps through the profile window (the exchange is a collective); rank 0 reports its slab
    roof, table = None, {}
    if rig.rank == 0 or rig.sharded:
        roof, table, _ = profile_window(rig, preroll, args.warmup, args.steps, rig.workload)
    if rig.rank == 0:
        if roof is not None:
            rec["roofline"] = roof
            rec["stages"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in table.items()}
        if with_cpu and state is not None:
            rec["cpu_baseline"] = cpu_baseline(rig.scene, state, preroll + args.warmup, args.cpu_steps)
    return rec


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) as children of this one,
    before anything here has touched torch or the GPU, and pass their output and exit code through."""
    port = os.environ.get("MASTER_PORT", "29517")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps K per repetition")
    ap.add_argument("--warmup", type=int, default=5, help="untimed warm-up steps W in front of the K timed ones")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed window (median reported)")
    ap.add_argument("--workload", default="c5", choices=["c1", "c2", "c3", "c5", "t1m", "c4", "t1m_settled", "ref_1m", "ref_cg"])
    ap.add_argument("--preroll", type=int, default=-1, help="untimed set-up steps in front of the warm-up (default per workload)")
    ap.add_argument("--cpu-steps", type=int, default=4, help="timed oracle steps of the CPU sample (bounded by 14 s per variant)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other single-GPU workloads appended at N=1")
    ap.add_argument("--weak", action="store_true",
                    help="--workload c4 --gpus N: N workload-shaped slabs side by side (weak scaling) instead of the ONE 1M-body "
                         "scene cut into N equal-count slabs (strong scaling, BASELINE config 4: the default for c4)")
    ap.add_argument("--profile-window", action="store_true",
                    help="ONE world, preroll + warm-up + K steps and nothing else: the command the committed rocprofv3 "
                         "kernel-trace / PMC passes run (tools/profile_window.py cuts the last K steps out of the trace)")
    args = ap.parse_args()

    env_ws = os.environ.get("WORLD_SIZE")
    if env_ws is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world_size = int(env_ws or "1")
    if world_size != args.gpus and not (args.gpus == 1 and env_ws is not None):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}: launch with matching values "
              f"(or without a launcher: this script starts its own ranks)", file=sys.stderr)
        sys.exit(2)

    import torch
    from physics_amd.scenes import DT_NANOS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # PHYS_BENCH_REHEARSAL=1: rehearse the N > 1 flow on a ONE-GPU box - every rank uses GPU 0, the
    # collective runs over gloo on pinned host buffers (which the halo kernels read / write directly).
    rehearsal = os.environ.get("PHYS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        # several PROCESSES on one GPU: the cluster solver needs all its workgroups resident at once and cannot know of
        # another process's launch (inside one process its launches are chained); the rehearsal is about the exchange
        if world_size > 1:
            os.environ.setdefault("PHYS_DEBUG_NO_CLUSTER", "1")
    # PHYS_BENCH_FORCE_DIST=1: take the sharded path (RCCL collective included) even with one rank, so the
    # N > 1 code is exercised end to end on a one-GPU box
    sharded = world_size > 1 or os.environ.get("PHYS_BENCH_FORCE_DIST") == "1"
    if sharded:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist_mod.init_process_group("gloo")
        else:
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    else:
        torch.cuda.set_device(local_rank)
    n_gpus = world_size

    preroll = args.preroll if args.preroll >= 0 else DEFAULT_PREROLL[args.workload]
    strong = sharded and args.workload == "c4" and not args.weak
    rig = Rig(args.workload, rank, world_size, local_rank, dist, rehearsal, sharded, strong=strong)

    if args.profile_window:
        w = rig.make_world()
        rig.advance(w, preroll)
        rig.advance(w, args.warmup)
        w.sync()
        rig.advance(w, args.steps)
        w.sync()
        st = stats_dict(w.get_stats())
        w.close()
        if rank == 0:
            print(json.dumps({"profile_window": {"workload": args.workload, "preroll": preroll, "warmup": args.warmup,
                                                 "steps": args.steps}, "scene_stats": st}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    with_cpu = rank == 0 and not sharded and not args.no_cpu_baseline
    rec = measure_workload(rig, args, preroll, args.reps, with_cpu)

    out = None
    if rank == 0:
        out = {
            "metric": "rigid_body_steps_per_sec", "value": rec["value"],
            "unit": "body-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong" if rig.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": rig.scene.name, "n_bodies": rec["n_bodies"], "bodies_per_gpu": rig.scene.n,
                       "solver_iterations": rig.iters, "dt_nanos": DT_NANOS, "preroll_steps": preroll,
                       "flags_note": "PHYS_FLAG_EXCLUSIVE_GPU: nothing else runs on the benchmark's GPU, so the cluster solver "
                                     "skips the all-or-nothing count of its workgroups (the default, for hosts that share the "
                                     "GPU with a renderer: +0.04 ms per update on C5 and on the 1M cubes: tools/guard_cost.py), and the dataflow solver of mid-size scenes (C3) may use three workgroups per CU",
                       "timed_window": f"steps {preroll + args.warmup}..{preroll + args.warmup + args.steps} of the trajectory",
                       "sharding": "none" if not sharded else (
                           f"x-slabs x{n_gpus}, one process per GPU; " +
                           ("ONE scene cut into equal-count slabs (phys_slab_*); " if rig.strong else "one workload-shaped slab per rank; ") +
                           ("per step: all-gather of fixed-size record blocks over gloo on pinned host buffers (rehearsal)" if rehearsal else
                            "per step: one grouped ncclSend / ncclRecv pair per slab face with the neighbouring rank (phys_comm_set_neighbours; "
                            "fixed-size blocks, behind the C ABI: phys_halo_exchange) - exercised with ONE rank only on the builder's "
                            "one-GPU boxes: unverified with more than 1 rank until the driver's multi-GPU run")),
                       "rccl_ranks": n_gpus if sharded and not rehearsal else (0 if not sharded else f"{n_gpus} (gloo rehearsal on one GPU)")},
            "steps_per_sec": rec["steps_per_sec"], "repetitions": rec["repetitions"],
            "pairs_per_sec": rec["pairs_per_sec"], "scene_stats": rec["scene_stats"],
        }
        if sharded:
            out["config"]["cross_slab_contacts"] = (
                "broad phase only: candidate pairs across slab faces are found and counted (AABB halo exchange)"
                if not rig.halo.before_update else
                "solved with a shared impulse: every rank sees its neighbours' boundary bodies as DYNAMIC ghost bodies (96-byte "
                "records with mass and inverse inertia, exchanged before the update) and solves the two-body contact on its "
                "side from the same state as the neighbour does on his, each keeping its own body's half; exact for an "
                "isolated pair (momentum to 1e-6, tests/test_gpu_ghosts.py), an approximation of the single-world run in a "
                "pile (a rank sees the neighbour's bodies only as far as its ghosts reach), never its bits")
        if "roofline" in rec:
            roof = rec["roofline"]
            if not sharded:
                roof["copy_ceiling"] = copy_ceiling_gbs(f"cuda:{local_rank}")  # GB/s a device-to-device copy reaches here
                roof["frac_of_copy_ceiling"] = round(roof["achieved"] / roof["copy_ceiling"], 5)
            out["roofline"] = roof
            out["stages"] = rec["stages"]
        if "cpu_baseline" in rec:
            out["cpu_baseline"] = rec["cpu_baseline"]
        if "full_period" in rec:
            out["full_period"] = rec["full_period"]

    if rig.strong:
        # the union over ranks of (local pairs, cross pairs) must equal the pair count of the ONE world holding the whole
        # scene at the same step (broad-phase-only bodies fall alike on every rank: identical positions, identical AABBs)
        total_pairs = int(round(rig.sum_over_ranks(float(rec["scene_stats"]["n_pairs"] + rec.get("cross_pairs_rank", 0)))))
        if rank == 0:
            single = Rig(args.workload, 0, 1, local_rank, None, False, False)
            w1 = single.make_world()
            single.advance(w1, preroll + args.warmup + args.steps)
            w1.sync()
            one = int(w1.get_stats().n_pairs)
            w1.close()
            out["config"]["pairs_union_check"] = {"sum_over_ranks_local_plus_cross": total_pairs, "single_world": one,
                                                  "equal": total_pairs == one}
    if rank == 0 and not sharded and not args.no_extra and args.workload == "c5":
        # the other single-GPU configurations, measured the same way (fewer repetitions: they are sub-records)
        others = {}
        for wl in ("t1m", "c3", "c2", "c4", "t1m_settled", "ref_1m", "ref_cg"):
            r2 = Rig(wl, 0, 1, local_rank, None, False, False)
            rr = measure_workload(r2, args, DEFAULT_PREROLL[wl], 3, not args.no_cpu_baseline)
            rr["workload"] = WORKLOAD_NAMES.get(wl, r2.scene.name)
            if wl in ("t1m", "t1m_settled"):
                rr["north_star_target_steps_per_sec"] = 60.0
            if wl == "ref_cg" and "roofline" in rr and "cpu_baseline" in rr:
                # k_constraint_solve beside the one-thread oracle, per CG iteration - said plainly
                it_gpu = rr["roofline"]["scene_stats_after_window"].get("cg_iterations_per_update", 0.0)
                us_gpu = rr["stages"]["constraints"]["avg_launch_us"] / max(it_gpu, 1e-9)
                cb = rr["cpu_baseline"]
                us_cpu = cb.get("us_per_update", 0.0) / max(cb.get("cg_iterations_per_update", 1e-9), 1e-9)
                rr["cg"] = {"rows": rr["scene_stats"].get("cg_rows"), "iterations_per_update_gpu_window": round(it_gpu, 2),
                            "gpu_us_per_cg_iteration": round(us_gpu, 2),
                            "cpu_oracle_us_per_cg_iteration": round(us_cpu, 2),
                            "gpu_kernel_slower_than_one_cpu_thread": bool(us_gpu > us_cpu),
                            "note": "k_constraint_solve is ONE workgroup that keeps nalgebra's summation order (8 strided "
                                    "accumulator chains per dot, three dots per iteration) so that lambda and the iteration "
                                    "count equal the reference's bit for bit; gpu figure = kernel time (set-up included) / "
                                    "iterations, cpu figure = whole oracle update / iterations"}
            others[wl] = rr
        out["other_workloads"] = others
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
