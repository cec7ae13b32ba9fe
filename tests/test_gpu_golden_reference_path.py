"""GPU: the HIP path against the numpy-float32 derivation of the reference's update() (tests/golden/make_golden.py,
class State) - no oracle in between. G4: two bodies of unequal mass, 12 constraint rows (8-accumulator dot product and
its tail, multi-iteration CG, warm start across frames, quirk Q4); G5: the demo scene of lib.rs:20-42 for 300 frames
(quirk Q1 rotation + euler_angles every frame); G6: quirk Q3 with three bodies. Bar: bit for bit, CG iteration counts
included. (The reference itself cannot be run: no Rust toolchain - rows A2-A8 stay "parity unpinned"; this is the
independent restatement the HIP path and the oracle are both held to.)"""
import pytest

from golden_util import GOLD, assert_frame, load_scene

pytestmark = pytest.mark.gpu


def _world():
    import physics_amd
    return physics_amd.World(physics_amd.default_config())


def test_g4_two_bodies_twelve_rows_multi_iteration_cg_and_warm_start():
    g = GOLD["G4"]
    w = _world()
    load_scene(w, g)
    for k, frame in enumerate(g["frames"]):
        w.update(g["dt_nanos"])
        assert_frame(w, frame, f"G4 frame {k + 1}")


def test_g5_demo_scene_300_frames():
    g = GOLD["G5"]
    w = _world()
    load_scene(w, g)
    for k in range(1, 301):
        w.update(g["dt_nanos"])
        if str(k) in g["frames"]:
            assert_frame(w, g["frames"][str(k)], f"G5 frame {k}")


def test_g6_quirk_q3_three_bodies():
    for c, case in enumerate(GOLD["G6"]["cases"]):
        w = _world()
        load_scene(w, case)
        for k, frame in enumerate(case["frames"]):
            w.update(GOLD["G6"]["dt_nanos"])
            assert_frame(w, frame, f"G6 case {c} frame {k + 1}")
