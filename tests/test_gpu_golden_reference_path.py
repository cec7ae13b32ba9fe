"""GPU: the HIP path against the numpy-float32 derivation of the reference's update() (tests/golden/make_golden.py,
class State) - no oracle in between. G4: two bodies of unequal mass, 12 constraint rows (8-accumulator dot product and
its tail, multi-iteration CG, warm start across frames, quirk Q4); G5: the demo scene of lib.rs:20-42 for 300 frames
(quirk Q1 rotation + euler_angles every frame); G6: quirk Q3 with three bodies; G7: both gimbal branches of euler_angles;
G3: the reference's own block-SpMV unit tests through the general device product. Bar: bit for bit, CG iteration counts
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


def test_g7_gimbal_branches_of_euler_angles():
    for c, case in enumerate(GOLD["G7"]["cases"]):
        w = _world()
        load_scene(w, case)
        for k, frame in enumerate(case["frames"]):
            w.update(GOLD["G7"]["dt_nanos"])
            assert_frame(w, frame, f"G7 case {c} ({case['branch']}) frame {k + 1}")


@pytest.mark.parametrize("case", GOLD["G3"], ids=lambda c: c["name"])
def test_g3_the_references_own_spmv_unit_tests_on_the_device(case):
    """sparse_matrix.rs:65-119 - the only tests the reference holds - through phys_block_spmv (the general
    add_block / multiply_vector / tr_multiply_vector on the device). Integer-valued: exact."""
    import numpy as np
    import physics_amd
    blocks = [(b["i"], b["j"], np.array(b["data"], np.float32)) for b in case["blocks"]]
    out = physics_amd.block_spmv(case["nrows"], case["ncols"], blocks, case["vector"], transpose=case["transpose"])
    assert np.array_equal(out, np.array(case["expected"], np.float32))


def test_block_spmv_random_overlapping_blocks_equal_the_oracle_bit_for_bit():
    """Beyond the reference's three cases: 400 random dense blocks of random shapes (1..7 x 1..7, the reference's own
    3 x 6 constraint blocks among them) scattered over a 300 x 500 matrix so that many overlap - the accumulation order
    over blocks and the left-to-right inner sums are then visible in the last bit - both products, against the oracle's
    restatement of sparse_matrix.rs:25-50. And the reference's assert_eq on the vector length is an error code."""
    import numpy as np
    import physics_amd
    from oracle import binding as ob
    rng = np.random.default_rng(5)
    nrows, ncols = 300, 500
    blocks = []
    for k in range(400):
        il, jl = (3, 6) if k % 4 == 0 else (int(rng.integers(1, 8)), int(rng.integers(1, 8)))
        blocks.append((int(rng.integers(0, nrows - il + 1)), int(rng.integers(0, ncols - jl + 1)),
                       rng.normal(size=(il, jl)).astype(np.float32)))
    for transpose in (False, True):
        v = rng.normal(size=nrows if transpose else ncols).astype(np.float32)
        got = physics_amd.block_spmv(nrows, ncols, blocks, v, transpose=transpose)
        want = ob.spmv(nrows, ncols, blocks, v, transpose=transpose)
        assert np.array_equal(got, want), f"transpose={transpose}: max diff {np.abs(got - want).max()}"
    with pytest.raises(physics_amd.PhysError) as e:
        physics_amd.block_spmv(nrows, ncols, blocks, np.zeros(7, np.float32))
    assert e.value.code == -1
