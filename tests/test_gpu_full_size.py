"""GPU: BASELINE.json's full-size configurations, checked through size-independent properties (the oracle
needs minutes at these sizes), plus edge cases of the ABI (empty / degenerate inputs, capacity limits,
re-upload). Bit-exact oracle parity at the sizes the oracle finishes in seconds is in test_gpu_collision.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 16_666_667


def _world(sc, **kw):
    import physics_amd
    w = physics_amd.World(sc.config(**kw))
    sc.populate(w)
    return w


def _check_pairs(pairs, aabb):
    assert (pairs[:, 0] < pairs[:, 1]).all()
    key = pairs[:, 0].astype(np.uint64) << np.uint64(32) | pairs[:, 1].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all(), "pairs not strictly sorted / duplicated"
    a, b = aabb[pairs[:, 0]], aabb[pairs[:, 1]]
    assert ((a[:, :3] <= b[:, 3:]) & (b[:, :3] <= a[:, 3:])).all(), "a reported pair does not overlap"


def _check_coloring(w):
    ids, counts, normals, points = w.get_manifolds()
    assert np.isfinite(normals).all() and np.isfinite(points).all()
    assert ((counts >= 1) & (counts <= 4)).all()
    nn = np.linalg.norm(normals.astype(np.float64), axis=1)
    assert np.abs(nn - 1.0).max() < 1e-3
    return ids


def test_c4_1m_bodies_broadphase_only():
    """C4: 1 000 000 bodies, dense AABB overlaps, broad phase only."""
    from physics_amd import scenes
    sc = scenes.c4()
    w = _world(sc)
    pairs = w.broadphase()
    aabb = w.get_aabbs()
    assert len(pairs) > 1_000_000
    _check_pairs(pairs, aabb)
    # completeness on a random sample of bodies: brute force against everybody
    rng = np.random.default_rng(0)
    for i in rng.choice(sc.n, 40, replace=False):
        ov = np.nonzero(((aabb[i, :3] <= aabb[:, 3:]) & (aabb[:, :3] <= aabb[i, 3:])).all(1))[0]
        ov = ov[ov != i]
        mine = np.concatenate([pairs[pairs[:, 0] == i, 1], pairs[pairs[:, 1] == i, 0]])
        assert sorted(ov.tolist()) == sorted(mine.tolist())
    # the update path reports the same number of candidate pairs
    w.update(DT)
    w.sync()
    assert w.get_stats().n_pairs == len(pairs)
    # a second world built from the same scene finds the same pair SET (whatever the emission order was)
    assert np.array_equal(w.broadphase(), _world(sc).broadphase())


def test_c3_100k_mixed_full_step_properties():
    """C3: 100 000 mixed spheres / cubes, 8 solver iterations: two runs are bit-identical, nothing is NaN,
    manifolds are well formed, nobody falls through the ground."""
    from physics_amd import scenes
    sc = scenes.c3()
    out = []
    for _ in range(2):
        w = _world(sc)
        w.update_n(DT, 120)
        w.sync()
        out.append(w.get_transforms() + w.get_velocities())
        st = w.get_stats()
        if _ == 0:
            ids = _check_coloring(w)
            assert st.n_manifolds == len(ids) and st.n_colors >= 2 and st.overflow == 0
            pos = out[0][0]
            assert np.isfinite(pos).all() and pos[:, 1].min() > 0.5  # radius / half extent 1, slop + softness
        w.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_c5_256k_tower_properties():
    """C5: 256 000 boxes in resting contact (16 x 1000 x 16, spacing exactly 2.0): dense contact graph. 40 steps, run
    twice: the two runs agree in every bit (poses, velocities, counters), nothing is NaN, no box sinks into the
    plane, and the colouring stays proper (no body carries two manifolds of one colour)."""
    from physics_amd import scenes
    sc = scenes.c5()
    runs = []
    for r in range(2):
        w = _world(sc)
        w.update_n(DT, 36)
        w.profile_enable(True)
        w.update_n(DT, 4)
        w.sync()
        assert "solve_cluster" in w.profile_get()[0], "C5 is the cluster solver's scene"
        w.profile_enable(False)
        st = w.get_stats()
        runs.append(w.get_transforms() + w.get_velocities() + (np.array([st.n_pairs, st.n_manifolds, st.n_contacts, st.n_colors]),))
        if r == 0:
            assert st.overflow == 0
            assert st.n_manifolds > 2 * sc.n  # every box touches its neighbours
            assert st.n_colors <= 64
            pos, rot = runs[0][0], runs[0][1]
            assert np.isfinite(pos).all() and np.isfinite(rot).all()
            # half extent 1: the lowest layer rests on the plane. 8 iterations cannot carry a 1000-layer tower (neither
            # could a CPU sequential-impulse solver: SURVEY section 7), so the bottom layer is pressed into the plane by
            # the push-out cap (max_bias): 0.88 after 40 steps on this build. The bound is there to catch a collapse
            # (a missing ground contact lets the layer fall by ~0.5 in 40 steps), not to certify the stack
            assert pos[:, 1].min() > 0.8
            ids = _check_coloring(w)
            assert len(ids) == st.n_manifolds
        w.close()
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def test_target_1m_cubes_steps():
    """north_star target scene: 1M cubes; a few steps run, contacts form at the bottom layer first."""
    from physics_amd import scenes
    sc = scenes.target_1m()
    w = _world(sc)
    w.update_n(DT, 40)
    w.sync()
    st = w.get_stats()
    assert st.overflow == 0 and st.n_ground_manifolds >= 9000  # 100 x 100 bottom layer has landed
    pos, _ = w.get_transforms()
    assert np.isfinite(pos).all()


# ---- edge cases -----------------------------------------------------------------------------------------
def test_single_body_and_no_contacts():
    import physics_amd
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS | physics_amd.FLAG_GROUND_PLANE, gravity_offset=(0, 0, 0))
    w = physics_amd.World(cfg)
    w.set_bodies(np.array([[0, 50, 0]], np.float32), shape_type=np.array([physics_amd.SHAPE_BOX], np.uint32),
                 half_extent=np.ones((1, 3), np.float32))
    w.update_n(DT, 5)
    w.sync()
    st = w.get_stats()
    assert (st.n_pairs, st.n_manifolds, st.n_colors) == (0, 0, 0)
    ids, counts, normals, points = w.get_manifolds()
    assert len(ids) == 0


def test_reupload_with_different_size_and_far_coordinates():
    """set_bodies twice (grow, shrink); bodies 1e6 away exercise the wrap-around of the bucket table."""
    import physics_amd
    from oracle import binding as ob
    cfg = lambda: physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS)
    w = physics_amd.World(cfg())
    rng = np.random.default_rng(4)
    for n, off in ((50, 0.0), (3000, 1.0e6), (7, -3.0e5)):
        pos = (rng.uniform(-8, 8, size=(n, 3)) + off).astype(np.float32)
        st = np.full(n, physics_amd.SHAPE_SPHERE, np.uint32)
        he = np.full((n, 3), 0.9, np.float32)
        w.set_bodies(pos, shape_type=st, half_extent=he)
        o = ob.OracleWorld(cfg(), trig=ob.TRIG_DET)
        o.set_bodies(pos, shape_type=st, half_extent=he)
        assert np.array_equal(w.broadphase(), o.broadphase())


def test_manifold_capacity_overflow_is_reported_not_silently_solved():
    import physics_amd
    from physics_amd import scenes
    sc = scenes.c5(6, 6, 6)
    w = physics_amd.World(sc.config(max_manifolds=100))
    sc.populate(w)
    w.update(DT)
    with pytest.raises(physics_amd.PhysError) as e:
        w.sync()
    assert e.value.code == -5
    assert w.get_stats().overflow & 2
    # velocities were only integrated, never touched by a solver running on a truncated contact set
    lin, _ = w.get_velocities()
    assert np.allclose(lin[:, 1], -9.81 * 0.016666668, atol=1e-6) and not lin[:, [0, 2]].any()


def test_stats_and_stage_profile_are_consistent():
    import physics_amd
    from physics_amd import scenes
    sc = scenes.c1()
    w = _world(sc)
    w.update_n(DT, 80)
    w.profile_enable(True)
    w.update_n(DT, 10)
    prof, steps = w.profile_get()
    assert steps == 10 and ("solve_flow" in prof or "solve" in prof or "solve_tail" in prof)
    assert all(ms >= 0 and n > 0 for ms, n in prof.values())
    counts = w.get_color_counts()
    st = w.get_stats()
    assert counts.sum() == st.n_manifolds and (counts[:st.n_colors] > 0).all()


def test_dataflow_solver_gives_up_instead_of_hanging():
    """Every spin of the dataflow kernels is bounded. With PHYS_DEBUG_FLOW_STALL one row is handed a ticket that is
    never published (its body's chain can never advance): the launch must still END, flag the step, and phys_sync
    must report PHYS_ERR_HIP - never a hung GPU. Runs in a child process (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, '.')\n"
        "import physics_amd\n"
        "from physics_amd import scenes\n"
        "sc = scenes.c1()\n"
        "w = physics_amd.World(sc.config()); sc.populate(w)\n"
        "try:\n"
        "    w.update_n(16666667, 120); w.sync()\n"
        "    print('NO ERROR', w.get_stats().overflow)\n"
        "except physics_amd.PhysError as e:\n"
        "    print('ERROR', e.code, w.get_stats().overflow & 16)\n"
    )
    env = dict(os.environ, PHYS_DEBUG_FLOW_STALL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=180)
    assert "ERROR -3 16" in out.stdout, out.stdout + out.stderr


def test_zero_length_step_is_harmless():
    """dt = 0 with collisions on: the contact bias divides by dt, so such a step is taken as a plain RigidBody::step
    (which moves nothing): no NaN, poses unchanged."""
    import physics_amd
    from physics_amd import scenes
    sc = scenes.c1()
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.update_n(DT, 90)
    before = w.get_transforms()
    w.update(0)
    w.sync()
    after = w.get_transforms()
    assert np.isfinite(after[0]).all() and np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])


def test_overflow_in_an_early_step_of_a_batch_is_still_reported():
    """Device-side error flags are sticky until phys_sync: the per-step word is zeroed by the next step, so a capacity
    miss in an EARLY step of a phys_update_n batch used to be gone by the time the host looked. Spheres
    start in contact and fly apart: the first steps overflow max_manifolds, the last steps have no contact at all."""
    import physics_amd
    n = 64
    pos = np.zeros((n, 3), np.float32)
    pos[:, 0] = 1.9 * (np.arange(n) % 8)
    pos[:, 2] = 1.9 * (np.arange(n) // 8)
    pos[:, 1] = 50.0
    vel = np.zeros((n, 3), np.float32)
    vel[:, 0] = 40.0 * (np.arange(n) % 8)   # neighbours separate at 40 units/s: apart after a few steps
    vel[:, 2] = 40.0 * (np.arange(n) // 8)
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_offset=(0, 0, 0), gravity_force=(0, 0, 0),
                                     max_manifolds=16)
    w = physics_amd.World(cfg)
    w.set_bodies(pos, lin_vel=vel, shape_type=np.full(n, physics_amd.SHAPE_SPHERE, np.uint32), half_extent=np.ones((n, 3), np.float32))
    w.update_n(DT, 30)
    st = w.get_stats()
    assert st.n_manifolds == 0, "the scene was meant to be contact-free at the end of the batch"
    assert st.overflow & 2, "phys_get_stats must still show the early overflow"
    with pytest.raises(physics_amd.PhysError) as e:
        w.sync()
    assert e.value.code == -5
    w.sync()  # reported once, then cleared: the following steps were clean
    assert w.get_stats().overflow == 0


def test_more_than_64_manifolds_at_one_body_is_an_error_on_every_step():
    """PHYS_MAX_COLORS = 64 manifolds per body. A slab carrying 81 spheres exceeds it: the step is flagged (bit 2) and
    its contact solve skipped; the NEXT step must not inherit the saturated colouring silently (round 1 did: kept
    colours set no flag, two rows of one colour raced on the slab) - it is flagged again."""
    import physics_amd
    k = 9
    n = 1 + k * k
    pos = np.zeros((n, 3), np.float32)
    he = np.ones((n, 3), np.float32) * 0.5
    st = np.full(n, physics_amd.SHAPE_SPHERE, np.uint32)
    st[0] = physics_amd.SHAPE_BOX
    he[0] = (10.0, 0.5, 10.0)
    pos[0] = (0, 5.0, 0)
    g = (np.arange(k) - (k - 1) / 2.0) * 1.5
    pos[1:, 0] = np.repeat(g, k)
    pos[1:, 2] = np.tile(g, k)
    pos[1:, 1] = 5.0 + 0.5 + 0.5 - 0.005
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_offset=(0, 0, 0), gravity_force=(0, 0, 0))
    w = physics_amd.World(cfg)
    w.set_bodies(pos, shape_type=st, half_extent=he)
    for step in range(3):
        w.update(DT)
        s = w.get_stats()
        assert s.n_manifolds >= k * k and (s.overflow & 4), f"step {step}: overflow {s.overflow}"
        with pytest.raises(physics_amd.PhysError) as e:
            w.sync()
        assert e.value.code == -5 and "64" in str(e.value)
    lin, _ = w.get_velocities()
    assert not lin.any()  # no gravity, and no solve ever ran on the improperly coloured set
