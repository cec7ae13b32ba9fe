"""CPU: the synthetic scenes of SURVEY.md §8 row D (sizes, generator, determinism)."""
import numpy as np

from physics_amd import scenes


def test_splitmix64_known_answers():
    # reference outputs of splitmix64 seeded with 1234567 (Vigna's test vector)
    got = scenes.splitmix64(1234567, 5)
    assert [int(x) for x in got] == [6457827717110365317, 3203168211198807973, 9817491932198370423,
                                     4593380528125082431, 16408922859458223821]


def test_scene_sizes_and_layout():
    assert scenes.c1().n == 64 and scenes.c2().n == 10_000
    c3 = scenes.c3()
    assert c3.n == 100_000 and set(np.unique(c3.shape_type)) == {1, 2}
    assert scenes.c5(16, 10, 16).n == 2560
    p = scenes.c2().pos
    assert abs(p[:, 1].min() - 2.0) <= 0.05 and abs(p[:, 1].max() - (2.0 + 15 * 2.5)) <= 0.05
    t = scenes.c5(4, 10, 4).pos
    assert t[:, 1].min() == 1.0 and np.all(np.diff(np.unique(t[:, 1])) == 2.0)
    assert np.array_equal(scenes.c2().pos, scenes.c2().pos)  # deterministic


def test_configs_carry_the_documented_overrides():
    cfg = scenes.c2().config()
    assert list(cfg.gravity_offset) == [0.0, 0.0, 0.0] and cfg.solver_iterations == 8
    assert cfg.flags == 3
    assert scenes.c4(10, 10, 10).config().flags == 1 | 8
