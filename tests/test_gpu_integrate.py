"""GPU parity of rows A2/A8 (gravity + RigidBody::step) through the C ABI, against the CPU oracle
and the golden vectors. Bar: bit-exact vs oracle(det trig); <= 1e-6 relative vs oracle(libm)
(1 ulp of sinf/cosf per call is the only permitted difference)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = 16_666_667
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def _mk(cfg=None):
    import physics_amd
    return physics_amd.World(cfg)


def _oracle(cfg=None, trig=1):
    from oracle import binding as ob
    import physics_amd
    return ob.OracleWorld(cfg if cfg is not None else physics_amd.default_config(), trig=trig)


def _random_state(n, seed):
    rng = np.random.default_rng(seed)
    pos = rng.normal(scale=5.0, size=(n, 3)).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    lin = rng.normal(size=(n, 3)).astype(np.float32)
    ang = rng.normal(scale=2.0, size=(n, 3)).astype(np.float32)
    ang[::7] = 0.0  # exercise the `angular_velocity != 0` branch
    mass = rng.uniform(0.5, 4.0, size=n).astype(np.float32)
    return pos, q, lin, ang, mass


def test_free_fall_golden_G2():
    import physics_amd
    g = GOLD["G2"]
    w = _mk(physics_amd.default_config(gravity_offset=(0, 0, 0)))
    w.set_bodies(np.array([[0, g["y0"], 0]], np.float32))
    w.update_n(g["dt_nanos"], g["steps"])
    pos, rot = w.get_transforms()
    lin, _ = w.get_velocities()
    assert pos[0, 1] == np.float32(g["y"])
    assert lin[0, 1] == np.float32(g["vy"])
    assert np.array_equal(rot[0], np.array([0, 0, 0, 1], np.float32))


@pytest.mark.parametrize("n", [1, 63, 1000, 100_003])
def test_update_bit_exact_vs_oracle(n):
    import physics_amd
    pos, q, lin, ang, mass = _random_state(n, 1234 + n)
    steps = 1000 if n <= 1000 else 50
    w, o = _mk(), _oracle()
    for x in (w, o):
        x.set_bodies(pos, rot=q, lin_vel=lin, ang_vel=ang, mass=mass)
        x.update_n(DT, steps)
    for name, a, b in zip(("pos", "rot"), w.get_transforms(), o.get_transforms()):
        assert np.array_equal(a, b), f"{name} differs from oracle (max abs {np.abs(a - b).max()})"
    for name, a, b in zip(("lin", "ang"), w.get_velocities(), o.get_velocities()):
        assert np.array_equal(a, b), name
    f, t = w.get_forces()
    assert not f.any() and not t.any()  # rigid_body.rs:38-39


def test_update_vs_libm_oracle_tolerance():
    """Against the oracle linked to the host libm (what the reference itself calls): tolerance 1e-4
    relative on transforms over 1000 steps (north_star), expected ~1e-6."""
    pos, q, lin, ang, mass = _random_state(512, 99)
    w, o = _mk(), _oracle(trig=0)
    for x in (w, o):
        x.set_bodies(pos, rot=q, lin_vel=lin, ang_vel=ang, mass=mass)
        x.update_n(DT, 1000)
    for a, b in zip(w.get_transforms(), o.get_transforms()):
        rel = np.abs(a - b).max() / np.abs(b).max()
        assert rel <= 1e-4, rel


def test_full_inertia_and_exact_rotation():
    import physics_amd
    n = 300
    pos, q, lin, ang, mass = _random_state(n, 5)
    rng = np.random.default_rng(6)
    inertia = np.tile(np.eye(3, dtype=np.float32) * 2.0, (n, 1, 1)) + rng.normal(scale=0.2, size=(n, 3, 3)).astype(np.float32)
    for flags in (0, physics_amd.FLAG_EXACT_ROTATION):
        cfg = physics_amd.default_config(flags=flags)
        w, o = _mk(cfg), _oracle(physics_amd.default_config(flags=flags))
        for x in (w, o):
            x.set_bodies(pos, rot=q, lin_vel=lin, ang_vel=ang, mass=mass, inertia=inertia.reshape(n, 9))
            x.update_n(DT, 200)
        for a, b in zip(w.get_transforms(), o.get_transforms()):
            assert np.array_equal(a, b)


def test_forces_gravity_step_calls():
    """apply_force_* (rigid_body.rs:43-62), apply_gravity (physics.rs:87-94) and step (physics.rs:95-99)
    as separate calls."""
    pos, q, lin, ang, mass = _random_state(40, 8)
    w, o = _mk(), _oracle()
    for x in (w, o):
        x.set_bodies(pos, rot=q, lin_vel=lin, ang_vel=ang, mass=mass)
        x.apply_force_centre_of_gravity(3, [1.0, 2.0, 3.0])
        x.apply_force_at_position(5, [0.5, -1.0, 0.25], [1.0, 1.0, 1.0])
        x.apply_force_at_offset(7, [0.0, 4.0, 0.0], [0.3, 0.0, -0.2])
        x.apply_gravity()
    for a, b in zip(w.get_forces(), o.get_forces()):
        assert np.array_equal(a, b)
    for x in (w, o):
        x.step(DT)
        x.update(DT)
        x.apply_force_at_offset(0, [1.0, 0.0, 0.0], [0.0, 1.0, 0.0])
        x.update(DT)
    for a, b in zip(w.get_transforms(), o.get_transforms()):
        assert np.array_equal(a, b)
    for a, b in zip(w.get_velocities(), o.get_velocities()):
        assert np.array_equal(a, b)


def test_error_paths():
    import physics_amd
    w = _mk()
    with pytest.raises(physics_amd.PhysError) as e:
        w.update(DT)  # no bodies: the reference panics
    assert e.value.code == -8
    w.set_bodies(np.zeros((2, 3), np.float32), inertia=np.zeros((2, 9), np.float32))
    with pytest.raises(physics_amd.PhysError) as e:
        w.update(DT)  # singular inertia: try_inverse().unwrap() panic
    assert e.value.code == -4
    w.set_bodies(np.zeros((2, 3), np.float32))
    with pytest.raises(physics_amd.PhysError) as e:
        w.apply_force_centre_of_gravity(2, [0, 0, 0])
    assert e.value.code == -6


def test_instance_matrices():
    pos, q, lin, ang, mass = _random_state(100, 11)
    w, o = _mk(), _oracle()
    for x in (w, o):
        x.set_bodies(pos, rot=q)
    assert np.array_equal(w.get_instance_matrices(), o.get_instance_matrices())
