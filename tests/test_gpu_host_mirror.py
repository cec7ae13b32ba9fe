"""GPU: the C++ host mirror (include/physics_state.hpp: PhysicsState / Entity / RigidBody / ConstraintSolver
with the reference's names) drives the reference's demo scene (src/lib.rs:20-42) through the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))


def _run(frames):
    exe = os.path.join(ROOT, "tests", "cpp", "demo_scene")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "demo_scene.cpp"), "-o", exe,
                               "-L", os.path.join(ROOT, "physics_amd", "csrc"), "-lphysics_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "physics_amd", "csrc")])
    out = subprocess.run([exe, str(frames)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_demo_scene_first_frame_is_golden_g1():
    g = GOLD["G1"]
    r = _run(1)
    assert np.array_equal(np.float32(r["pos"]), np.float32(g["pos"]))
    assert np.array_equal(np.float32(r["rot_ijkw"]), np.float32(g["rot_ijkw"]))
    assert np.array_equal(np.float32(r["lambda"]), np.float32(g["lambda"]))
    assert np.array_equal(np.float32(r["model_col3"]), np.float32(g["pos"] + [1.0]))  # translation column of T*R


def test_pub_field_edits_between_frames_reach_the_device():
    """lib.rs mutates bodies through pub fields; the mirror must notice. Same sequence on the oracle."""
    from oracle import binding as ob
    import physics_amd
    r = _run(3)
    o = ob.OracleWorld(physics_amd.default_config(), trig=ob.TRIG_DET)
    q = ob.quat_from_euler(1.0, 0.0, 0.0, ob.TRIG_LIBM)  # the C++ mirror builds the quaternion with libm
    o.set_bodies(np.array([[1, 0, 0]], np.float32), rot=q.reshape(1, 4))
    o.add_constraint_fix_point(0, [0, 0, 0])
    o.add_constraint_fix_orientation(0, [0, 0, 0])
    o.update_n(16_666_667, 3)
    pos, rot = o.get_transforms()
    lin, ang = o.get_velocities()
    assert np.array_equal(np.float32(r["pos"]), pos[0])
    # replay the edit: position overwritten, one force applied, one more frame (warm start kept)
    lam = o.get_lambda()
    o2 = ob.OracleWorld(physics_amd.default_config(), trig=ob.TRIG_DET)
    o2.set_bodies(np.array([[2, 0, 0]], np.float32), rot=rot, lin_vel=lin, ang_vel=ang)
    o2.add_constraint_fix_point(0, [0, 0, 0])
    o2.add_constraint_fix_orientation(0, [0, 0, 0])
    o2.apply_force_at_offset(0, [0, 1, 0], [1, 0, 0])
    o2.update(16_666_667)
    # set_bodies resets previous_solution (the mirror re-uploads on an edit, as documented), so this matches
    assert np.allclose(np.float32(r["after_edit_pos"]), o2.get_transforms()[0][0], rtol=0, atol=0)
    assert len(lam) == 6


def test_python_mirror_reads_like_the_reference_demo():
    """src/lib.rs:20-42 + the frame loop of :55-59 written against physics_amd.state."""
    import datetime
    from physics_amd.state import (ConstraintSolver, Entity, FixedOrientationConstraint, FixToPointConstraint,
                                   PhysicsState, RigidBody)
    from oracle import binding as ob
    g = GOLD["G1"]
    rigid_body = RigidBody.new(0)
    rigid_body.position = np.array([1.0, 0.0, 0.0], np.float32)
    rigid_body.rotation = ob.quat_from_euler(1.0, 0.0, 0.0, ob.TRIG_DET)  # UnitQuaternion::from_euler_angles
    physics_state = PhysicsState(
        entities=[Entity(rigid_body, 0)],
        constraint_solver=ConstraintSolver([FixToPointConstraint(rigid_body.index, [0, 0, 0]),
                                            FixedOrientationConstraint(rigid_body.index, [0, 0, 0])]))
    physics_state.update(datetime.timedelta(microseconds=16666) + datetime.timedelta(microseconds=0))  # ~1/60 s
    physics_state = PhysicsState(entities=[Entity(rigid_body, 0)], constraint_solver=physics_state.constraint_solver)
    rigid_body.position = np.array([1.0, 0.0, 0.0], np.float32)
    rigid_body.rotation = ob.quat_from_euler(1.0, 0.0, 0.0, ob.TRIG_DET)
    rigid_body.lin_velocity[:] = 0
    rigid_body.angular_velocity[:] = 0
    physics_state.update(g["dt_nanos"])
    b = physics_state.entities[0].body
    assert np.array_equal(b.position, np.float32(g["pos"])) and np.array_equal(b.rotation, np.float32(g["rot_ijkw"]))
    assert np.array_equal(physics_state.previous_solution, np.float32(g["lambda"]))
    assert not b.force.any() and not b.torque.any()
    # pub-field edit + force between frames, then apply_gravity / step as separate calls
    b.apply_force_at_offset([0, 1, 0], [1, 0, 0])
    physics_state.apply_gravity()
    assert np.allclose(physics_state.entities[0].body.force, [0, -8.81, 0], atol=1e-6)
    physics_state.step(g["dt_nanos"])
    assert not physics_state.entities[0].body.force.any()
    m = physics_state.instance_matrices()
    assert m.shape == (1, 16) and np.array_equal(m[0, 12:15], physics_state.entities[0].body.position)
