"""Shared by the CPU (oracle) and GPU (HIP) golden tests: load a G4-G6 scene of tests/golden/reference_vectors.json
into any world with the ABI-shaped interface and compare a frame snapshot bit for bit."""
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))


def load_scene(world, scene):
    world.set_bodies(np.array(scene["pos0"], np.float32), rot=np.array(scene["rot0_ijkw"], np.float32),
                     lin_vel=np.array(scene["lin0"], np.float32), ang_vel=np.array(scene["ang0"], np.float32),
                     mass=np.array(scene["mass"], np.float32))
    for kind, body, target in scene["constraints"]:
        if kind == "point":
            world.add_constraint_fix_point(body, target)
        else:
            world.add_constraint_fix_orientation(body, target)


def assert_frame(world, frame, what):
    pos, rot = world.get_transforms()
    lin, ang = world.get_velocities()
    for name, got, want in (("pos", pos, frame["pos"]), ("rot", rot, frame["rot_ijkw"]), ("lin_vel", lin, frame["lin_vel"]),
                            ("ang_vel", ang, frame["ang_vel"])):
        want = np.array(want, np.float32)
        assert np.array_equal(got, want), f"{what}: {name} differs from the numpy-f32 derivation:\n{got}\n{want}"
    lam = world.get_lambda()
    assert np.array_equal(lam, np.array(frame["lambda"], np.float32)), f"{what}: lambda {lam} vs {frame['lambda']}"
    st = world.get_stats()
    assert st.cg_converged == 1 and st.cg_iterations == frame["cg_iterations"], f"{what}: CG iterations {st.cg_iterations}"
