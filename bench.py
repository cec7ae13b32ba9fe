#!/usr/bin/env python3
"""bench.py — throughput of the per-frame rigid-body step (phys_update) on MI355X.

A "step" is one PhysicsState::update over one synthetic scene resident in HBM.
N = 1 workload: BASELINE.json configs[1] = C2, 10 000 falling cubes + ground contacts, f32. The scene is
pre-rolled (untimed, part of set-up) until the pile is in contact, so the timed steps carry contacts; by default
1000 steps are timed, during which the pile collapses and the contact count doubles.
N > 1: weak scaling; every rank owns one C2-shaped slab placed side by side along x, the broad phase
exchanges boundary AABBs with one RCCL all-gather per step, narrow phase + solver stay on owned bodies.

Prints ONE JSON line (rank 0). `value` = bodies * steps / seconds over all ranks (whole-job aggregate);
`steps_per_sec` is the plain reference-style figure. `roofline` describes the dominant kernel of the
timed workload, timed live with HIP events on the library's own stream (phys_profile_*) in a second,
separate pass of K steps; `cpu_baseline` is the CPU oracle (a single-threaded C++ restatement; the Rust
reference cannot be built here) timed on the host cores on a bounded sample of the same trajectory.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

KERNEL_OF_STAGE = {
    "step_full": "k_step_full", "velocity_aabb": "k_step_velocity_aabb",
    "grid": "k_cell_insert",  # small scenes; larger ones: k_cell_assign + scan + k_scatter
    "pairs": "k_find_pairs", "narrow": "k_narrowphase",
    "color": "k_color_small",  # small scenes; larger ones: k_color_round x rounds + k_color_finish
    "rows": "k_rows_build",    # + k_color_hist / k_color_offsets / k_color_place beyond 24k manifolds
    "solve": "k_solve_color",
    "solve_tail": "k_solve_tail", "solve_flow": "k_solve_flow", "position": "k_step_position",
}


def stage_bytes(stage, st, iters):
    """Algorithmic HBM bytes of one STEP spent in `stage` (formulas: DESIGN.md 'algorithmic bytes')."""
    n, p, m, k = st["n_bodies"], st["n_pairs"], st["n_manifolds"], st["n_contacts"]
    mb = m - st["n_ground_manifolds"]
    if stage == "step_full":
        return 120 * n
    if stage == "velocity_aabb":
        return 132 * n
    if stage == "position":
        return 80 * n
    if stage == "grid":
        return 24 * n + 8 * n + 28 * n  # AABB read, bucket+rank, bucket-ordered copy (ids + boxes)
    if stage == "pairs":
        return 24 * n + 8 * p
    if stage == "narrow":
        return 96 * p + 44 * st["n_ground_manifolds"] + 100 * m
    if stage == "color":
        return 28 * m  # ids + priority + colour + slot, once
    if stage == "rows":
        return 100 * m + 24 * m + 76 * k + 52 * (m + mb)
    if stage in ("solve", "solve_flow"):
        # (k_solve_flow makes all iterations in one launch; same job, same compulsory bytes)
        # per body and iteration: v, w read 24 + written 24, inverse mass 4, inverse inertia diagonal 12
        # (all benchmark scenes have diagonal tensors; 36 with a full tensor)
        return iters * (24 * m + 64 * k + 64 * (m + mb))
    return 0


def _kernel_matches(name, kernel):
    """Row name of a rocprofv3 table vs a kernel family: k_solve_flow also covers its k_solve_flow_quad variant."""
    base = name.split("(")[0].split("<")[0].replace("void ", "").strip().split("::")[-1]
    return base == kernel or base.startswith(kernel + "_")


def rocprof_avg_us(workload_key, kernel):
    """(kernel name, average duration in us) from the committed rocprofv3 --kernel-trace run of this same script:
    the per-kernel statistics of the launches of its profile pass (profiles/r1_<workload>_profile_pass_stats.csv,
    cut out of the trace by tools/trace_tail.py; the whole-run --stats summary is r1_<workload>_kernel_stats.csv).
    HIP-event brackets (avg_launch_us) additionally contain the dispatch of the launch (about 2-4 us per launch)."""
    import csv
    for name in (f"r1_{workload_key}_profile_pass_stats.csv", f"r1_{workload_key}_kernel_stats.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        rows = [r for r in csv.DictReader(open(path)) if _kernel_matches(r["Name"], kernel)]
        if rows:
            r = max(rows, key=lambda r: int(r["Calls"]))
            return r["Name"].split("(")[0].replace("void ", "").strip().split("::")[-1], round(float(r["AverageNs"]) / 1e3, 3)
    return None, None


def pmc_traffic(workload_key, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, produced by
    profiles/collect_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same script)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    table = json.load(open(path)).get(workload_key, {})
    rows = [row for name, row in table.items() if isinstance(row, dict) and _kernel_matches(name, kernel)]
    if rows:
        return max(rows, key=lambda r: r["launches_sampled"])["traffic_bytes"]
    return None


def stats_dict(s):
    return {f: int(getattr(s, f)) for f in ("n_bodies", "n_pairs", "n_manifolds", "n_contacts", "n_colors",
                                            "color_rounds", "n_ground_manifolds")}


def run_timed(world, steps, dist=None, halo=None):
    from physics_amd.scenes import DT_NANOS
    import torch
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    world.sync()
    t0 = time.perf_counter()
    if halo is None:
        world.update_n(DT_NANOS, steps)
    else:
        for _ in range(steps):
            world.update(DT_NANOS)
            halo.exchange(world)
    world.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def profile_pass(world, steps, iters, workload_key="c2"):
    """K more steps with per-launch HIP events; returns the roofline object of the dominant kernel and
    the per-stage table."""
    from physics_amd.scenes import DT_NANOS
    world.sync()
    world.profile_enable(True)
    world.update_n(DT_NANOS, steps)
    world.sync()
    prof, psteps = world.profile_get()
    world.profile_enable(False)
    st = stats_dict(world.get_stats())
    table = {}
    for stage, (ms, launches) in prof.items():
        table[stage] = {"ms_per_step": ms / max(psteps, 1), "launches_per_step": launches / max(psteps, 1),
                        "avg_launch_us": 1e3 * ms / max(launches, 1)}
    kernel_stages = [s for s in table if s in KERNEL_OF_STAGE]
    dom = max(kernel_stages, key=lambda s: table[s]["ms_per_step"])
    b_step = stage_bytes(dom, st, iters)
    if dom == "solve" and "solve_tail" in table:
        # k_solve_color only runs the colours that got a launch of their own; the trailing small colours
        # (<= 512 manifolds each, at least two of them) are solved by k_solve_tail. Scale the stage's bytes by the
        # share of manifolds in the individually launched colours (same rule as launch_solver).
        counts = [int(c) for c in world.get_color_counts()[:st["n_colors"]]]
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
    total_bytes = sum(stage_bytes(s, st, iters) for s in kernel_stages)
    family = KERNEL_OF_STAGE[dom].split("+")[0]
    prof_name, prof_us = rocprof_avg_us(workload_key, family)
    roof = {"bound": "hbm", "kernel": prof_name or KERNEL_OF_STAGE[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic(workload_key, family),
            "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_us": round(table[dom]["avg_launch_us"], 3),
            "avg_kernel_us_rocprofv3": prof_us,
            "launches_per_step": round(launches, 2), "stage_share_of_device_time": round(
                table[dom]["ms_per_step"] / max(sum(t["ms_per_step"] for t in table.values()), 1e-12), 3),
            "algorithmic_bytes_per_step_all_kernels": int(total_bytes)}
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


def cpu_baseline(scene, preroll, sample_steps):
    from oracle import binding as ob
    from physics_amd.scenes import DT_NANOS

    def timed(threads):
        o = ob.OracleWorld(scene.config(), trig=ob.TRIG_DET)
        scene.populate(o)
        o.set_threads(threads)
        o.update_n(DT_NANOS, preroll)
        t0 = time.perf_counter()
        o.update_n(DT_NANOS, sample_steps)
        dt = time.perf_counter() - t0
        o.close()
        return dt

    dt = timed(1)
    # additionally the OpenMP variant of the same oracle (same bits for any thread count) on the host
    # cores this process may use, at most 16 (the CPU share of one GPU on the bench boxes)
    cores = min(len(os.sched_getaffinity(0)), 16)
    dt_mt = timed(cores) if cores > 1 else dt
    return {"value": round(scene.n * sample_steps / dt, 1), "unit": "body-steps/s", "cores": 1, "kind": "port",
            "steps_per_sec": round(sample_steps / dt, 3),
            "sample": f"oracle (scalar C++ restatement + CPU collision stages, 1 thread; the Rust reference is not "
                      f"buildable here), same scene, steps {preroll}..{preroll + sample_steps} of the same trajectory (a "
                      f"bounded sample: the first steps of the timed window; later steps carry more contacts)",
            "openmp": {"value": round(scene.n * sample_steps / dt_mt, 1), "unit": "body-steps/s", "cores": cores,
                       "steps_per_sec": round(sample_steps / dt_mt, 3),
                       "note": "same oracle, same sample, OpenMP over the independent loops of the collision stages "
                               "(AABBs, grid search, narrow phase, row preparation, the manifolds of one colour); the "
                               "reference itself is single-threaded"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=-1, help="timed steps (default: 1000 for c1 / c2, 200 otherwise: SURVEY.md §8 D)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5", "t1m", "c4"])
    ap.add_argument("--preroll", type=int, default=-1, help="untimed set-up steps (default per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 1M-body target run appended at N=1")
    args = ap.parse_args()
    if args.steps < 0:
        args.steps = 1000 if args.workload in ("c1", "c2") else 200

    import torch
    import physics_amd
    from physics_amd import scenes
    from physics_amd.scenes import DT_NANOS

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # PHYS_BENCH_REHEARSAL=1: rehearse the N > 1 flow on a ONE-GPU box - every rank uses GPU 0, the
    # collective runs over gloo on pinned host buffers (which the halo kernels read / write directly).
    rehearsal = os.environ.get("PHYS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
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

    default_preroll = {"c1": 100, "c2": 150, "c3": 150, "c5": 30, "t1m": 100, "c4": 0}
    preroll = args.preroll if args.preroll >= 0 else default_preroll[args.workload]
    halo = None
    if not sharded:
        scene = scenes.SCENES[args.workload]()
    else:
        from physics_amd import sharding
        scene, halo = sharding.make_rank_scene(args.workload, rank, world_size, dist, local_rank, pinned_host=rehearsal)
    world = physics_amd.World(scene.config(device=local_rank))
    scene.populate(world)
    if halo is not None:
        halo.attach(world, scene)
    iters = scene.solver_iterations

    # set-up (untimed): reach the contact-rich state, then W warm-up steps
    world.update_n(DT_NANOS, preroll)
    world.sync()
    if halo is None:
        world.update_n(DT_NANOS, args.warmup)
    else:
        for _ in range(args.warmup):
            world.update(DT_NANOS)
            halo.exchange(world)
    world.sync()

    elapsed = run_timed(world, args.steps, dist, halo)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = stats_dict(world.get_stats())
    n_total = scene.n * n_gpus
    pairs_local = st["n_pairs"] + (int(world.get_stats().n_cross_pairs) if halo is not None else 0)
    if dist is not None:
        t = torch.tensor([pairs_local], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{local_rank}")
        dist.all_reduce(t)
        pairs_total = float(t.item())
    else:
        pairs_total = float(pairs_local)

    out = None
    if rank == 0:
        # rank 0 profiles its own slab (no collective inside: the other ranks wait at the final barrier)
        roof, table, st2 = profile_pass(world, min(args.steps, 100), iters, args.workload)
        out = {
            "metric": "rigid_body_steps_per_sec", "value": round(n_total * args.steps / elapsed, 1),
            "unit": "body-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": scene.name, "n_bodies": n_total, "bodies_per_gpu": scene.n,
                       "solver_iterations": iters, "dt_nanos": DT_NANOS, "preroll_steps": preroll,
                       "sharding": "none" if not sharded else f"x-slabs x{n_gpus}, halo all-gather per step"},
            "steps_per_sec": round(args.steps / elapsed, 2),
            "pairs_per_sec": round(pairs_total * args.steps / elapsed, 1),
            "scene_stats": st,
        }
        if roof is not None:
            if not sharded:
                roof["copy_ceiling"] = copy_ceiling_gbs(f"cuda:{local_rank}")  # GB/s a device-to-device copy reaches here
                roof["frac_of_copy_ceiling"] = round(roof["achieved"] / roof["copy_ceiling"], 5)
            out["roofline"] = roof
            out["stages"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in table.items()}
    elif halo is None:
        pass
    world.close()

    if rank == 0 and not sharded:
        if not args.no_cpu_baseline:
            sample = 200 if args.workload in ("c1", "c2") else 10
            out["cpu_baseline"] = cpu_baseline(scene, preroll, sample)
        if not args.no_extra and args.workload == "c2":
            # north_star target: >= 1M bodies at >= 60 steps/s on one MI355X
            sc = scenes.target_1m()
            w = physics_amd.World(sc.config(device=local_rank))
            sc.populate(w)
            w.update_n(DT_NANOS, 100)
            w.sync()
            k = 60
            e = run_timed(w, k)
            roof1, table1, st1 = profile_pass(w, 20, sc.solver_iterations, "t1m")
            out["target_1m"] = {"workload": sc.name, "n_bodies": sc.n, "preroll_steps": 100, "steps": k,
                                "steps_per_sec": round(k / e, 2), "target_steps_per_sec": 60.0,
                                "body_steps_per_sec": round(sc.n * k / e, 1), "scene_stats": st1, "roofline": roof1,
                                "stages": {a: {kk: round(vv, 3) for kk, vv in v.items()} for a, v in table1.items()}}
            w.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
