"""CPU: analytic known-answer tests for the collision-stage specification (include/spec/collide.h,
contact_solve.h) evaluated through the oracle. These stages have no reference counterpart (SURVEY §8
A10-A12: "parity unpinned" by the reference), so closed-form answers are what pins them."""
import numpy as np
import pytest

from oracle import binding as ob
from physics_amd import (FLAG_COLLISIONS, FLAG_GROUND_PLANE, GROUND_ID, SHAPE_BOX, SHAPE_NONE, SHAPE_SPHERE,
                         default_config, scenes)

DT = scenes.DT_NANOS
MARGIN = 0.02


def world(flags=FLAG_COLLISIONS, **kw):
    return ob.OracleWorld(default_config(flags=flags, gravity_offset=(0, 0, 0), **kw), trig=ob.TRIG_DET)


def boxes(n, he=1.0):
    return np.full(n, SHAPE_BOX, np.uint32), np.full((n, 3), he, np.float32)


# ---- A10 broad phase ------------------------------------------------------------------------------
@pytest.mark.parametrize("dx,expect", [(1.5, 1), (2.0, 1), (2.0 + 2 * MARGIN, 1), (2.0 + 2 * MARGIN + 1e-3, 0), (3.0, 0)])
def test_two_unit_aabbs_overlap_touch_disjoint(dx, expect):
    w = world()
    st, he = boxes(2)
    w.set_bodies(np.array([[0, 0, 0], [dx, 0, 0]], np.float32), shape_type=st, half_extent=he)
    assert len(w.broadphase()) == expect
    assert len(w.broadphase_grid()) == expect


def test_aabb_of_rotated_box_and_sphere():
    w = world()
    q45 = np.array([0, 0, np.sin(np.pi / 8), np.cos(np.pi / 8)], np.float32)  # 45 deg about z
    w.set_bodies(np.array([[0, 0, 0], [10, 5, 0], [20, 0, 0]], np.float32),
                 rot=np.stack([q45, [0, 0, 0, 1], [0, 0, 0, 1]]).astype(np.float32),
                 shape_type=np.array([SHAPE_BOX, SHAPE_SPHERE, SHAPE_NONE], np.uint32),
                 half_extent=np.array([[1, 1, 1], [0.5, 9, 9], [1, 1, 1]], np.float32))
    a = w.get_aabbs()
    r = np.sqrt(2.0) + MARGIN
    assert np.allclose(a[0], [-r, -r, -1 - MARGIN, r, r, 1 + MARGIN], atol=1e-6)
    assert np.allclose(a[1], [9.5 - MARGIN, 4.5 - MARGIN, -0.5 - MARGIN, 10.5 + MARGIN, 5.5 + MARGIN, 0.5 + MARGIN], atol=1e-6)
    assert (a[2, :3] > a[2, 3:]).all()  # NONE: inverted box, never overlaps


def test_lattice_has_exactly_the_face_neighbour_pairs():
    nx, ny, nz = 6, 5, 4
    pos = scenes.lattice(nx, ny, nz, 1.9, 5.0, 0.0)  # spacing < 2: faces overlap, diagonals do too
    st, he = boxes(len(pos))
    w = world()
    w.set_bodies(pos, shape_type=st, half_extent=he)
    full = (3 * nx - 2) * (3 * ny - 2) * (3 * nz - 2)  # ordered 26-neighbour incidences + self
    expect_26 = (full - nx * ny * nz) // 2
    assert len(w.broadphase()) == expect_26
    pos2 = scenes.lattice(nx, ny, nz, 2.03, 5.0, 0.0)  # gap 0.03 < 2*margin: only nothing but faces... none diag
    w.set_bodies(pos2, shape_type=st, half_extent=he)
    # every axis gap is 0.03 <= 2*margin, so faces, edges and corners all still overlap
    assert len(w.broadphase()) == expect_26
    pos3 = scenes.lattice(nx, ny, nz, 2.5, 5.0, 0.0)
    w.set_bodies(pos3, shape_type=st, half_extent=he)
    assert len(w.broadphase()) == 0


def test_sweep_and_grid_drivers_agree_on_a_random_soup():
    rng = np.random.default_rng(5)
    n = 3000
    pos = rng.uniform(-15, 15, size=(n, 3)).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    st = rng.integers(0, 3, n).astype(np.uint32)
    he = rng.uniform(0.2, 1.5, size=(n, 3)).astype(np.float32)
    w = world()
    w.set_bodies(pos, rot=q, shape_type=st, half_extent=he)
    a, b = w.broadphase(), w.broadphase_grid()
    assert len(a) > 1000 and np.array_equal(a, b)
    # brute force on a subset
    bb = w.get_aabbs()[:400]
    brute = [(i, j) for i in range(400) for j in range(i + 1, 400)
             if (bb[i, :3] <= bb[j, 3:]).all() and (bb[j, :3] <= bb[i, 3:]).all()]
    sub = [tuple(p) for p in a if p[0] < 400 and p[1] < 400]
    assert sub == brute


# ---- A11 narrow phase -----------------------------------------------------------------------------
def test_sphere_sphere_closed_form():
    w = world()
    d = np.array([1.2, 0.9, -0.4], np.float32)
    w.set_bodies(np.stack([np.zeros(3, np.float32), d]), shape_type=np.full(2, SHAPE_SPHERE, np.uint32),
                 half_extent=np.array([[1.0, 0, 0], [0.7, 0, 0]], np.float32))
    w.collide_now()
    ids, counts, normals, points = w.get_manifolds()
    dist = np.linalg.norm(d.astype(np.float64))
    assert ids.tolist() == [[0, 1]] and counts[0] == 1
    assert np.allclose(normals[0], d / dist, atol=1e-6)
    assert abs(points[0, 0, 3] - (1.7 - dist)) < 1e-6
    assert np.allclose(points[0, 0, :3], (d / dist) * (1.0 - (1.7 - dist) / 2), atol=1e-6)


def test_cube_resting_on_plane_gives_four_bottom_corners():
    w = world(FLAG_COLLISIONS | FLAG_GROUND_PLANE)
    st, he = boxes(1)
    y = 0.99
    w.set_bodies(np.array([[3, y, -2]], np.float32), shape_type=st, half_extent=he)
    w.collide_now()
    ids, counts, normals, points = w.get_manifolds()
    assert ids.tolist() == [[0, GROUND_ID]] and counts[0] == 4
    assert np.array_equal(normals[0], np.array([0, -1, 0], np.float32))  # A (body) -> B (ground)
    corners = sorted(tuple(np.round(p[:3], 5)) for p in points[0])
    assert corners == sorted([(2.0, y - 1, -3.0), (4.0, y - 1, -3.0), (2.0, y - 1, -1.0), (4.0, y - 1, -1.0)])
    assert np.allclose(points[0, :, 3], 1.0 - y, atol=1e-6)  # depth = -(y - 1)


def test_box_box_face_contact_and_sphere_box():
    w = world()
    st, he = boxes(2)
    w.set_bodies(np.array([[0, 0, 0], [0.5, 1.9, 0.25]], np.float32), shape_type=st, half_extent=he)
    w.collide_now()
    ids, counts, normals, points = w.get_manifolds()
    assert counts[0] == 4 and np.allclose(normals[0], [0, 1, 0])
    assert np.allclose(points[0, :, 3], 0.1, atol=1e-6)
    xs = sorted(set(np.round(points[0, :, 0], 5)))
    zs = sorted(set(np.round(points[0, :, 2], 5)))
    assert xs == [-0.5, 1.0] and zs == [-0.75, 1.0]  # overlap rectangle of the two faces
    # sphere above a box face
    w.set_bodies(np.array([[0, 0, 0], [0.3, 1.4, -0.2]], np.float32),
                 shape_type=np.array([SHAPE_BOX, SHAPE_SPHERE], np.uint32),
                 half_extent=np.array([[1, 1, 1], [0.5, 0, 0]], np.float32))
    w.collide_now()
    ids, counts, normals, points = w.get_manifolds()
    assert counts[0] == 1 and np.allclose(normals[0], [0, 1, 0]) and abs(points[0, 0, 3] - 0.1) < 1e-6
    assert np.allclose(points[0, 0, :3], [0.3, 1.0, -0.2], atol=1e-6)


@pytest.mark.parametrize("offset,expect", [
    ((2.0, 0.0, 0.0), 4),    # face neighbour, touching: full 2 x 2 patch at depth 0 - a contact
    ((2.0, 2.0, 0.0), 0),    # edge-diagonal neighbour: the patch is a line - a sliver, not a contact
    ((2.0, 2.0, 2.0), 0),    # corner-diagonal neighbour: the patch is a point
    ((2.0, 1.95, 0.0), 0),   # patch 0.05 wide (< 0.1 x half extent 1), depth 0: still a sliver
    ((2.0, 1.8, 0.0), 4),    # patch 0.2 wide: a contact, whatever its depth
    ((1.99, 1.95, 0.0), 4),  # a sliver again, but it penetrates by 0.01 > margin / 4: kept
    ((1.996, 1.95, 0.0), 0),  # penetrates by 0.004 <= margin / 4: dropped
])
def test_sliver_rule_of_the_box_face_contact(offset, expect):
    """collide.h box_face_contact: a face contact whose patch is narrower than 0.1 x the smallest half extent of the
    two faces and whose deepest point penetrates by no more than a quarter of the margin is not a contact."""
    w = world()
    st, he = boxes(2)
    w.set_bodies(np.array([[0, 0, 0], offset], np.float32), shape_type=st, half_extent=he)
    w.collide_now()
    ids, cnt, nrm, pts = w.get_manifolds()
    if expect == 0:
        assert len(ids) == 0
        return
    assert len(ids) == 1 and cnt[0] == expect
    assert np.allclose(np.abs(nrm[0]), (1, 0, 0), atol=1e-6)
    assert np.allclose(pts[0, :, 3], 2.0 - offset[0], atol=1e-5)


def test_tilted_edge_resting_on_a_face_is_not_a_sliver():
    """A box tilted by 20 degrees about z, its lower edge 0.002 deep in the top face of another: a LINE contact (two
    points), but its patch - the incident face inside the reference face's side planes, before the depth filter - is as
    wide as the face, so the sliver rule leaves it alone."""
    w = world()
    st, he = boxes(2)
    ang = np.deg2rad(20.0)
    q = np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2)], np.float32)
    low = np.cos(ang) + np.sin(ang)  # lowest corner below the centre of the tilted unit cube
    pos = np.array([[0, 0, 0], [0.3, 1.0 + low - 0.002, 0.0]], np.float32)
    w.set_bodies(pos, rot=np.stack([[0, 0, 0, 1], q]).astype(np.float32), shape_type=st, half_extent=he)
    w.collide_now()
    ids, cnt, nrm, pts = w.get_manifolds()
    assert len(ids) == 1 and cnt[0] == 2
    assert np.allclose(pts[0, :2, 3], 0.002, atol=2e-5)


def test_box_box_edge_edge_contact():
    w = world()
    st, he = boxes(2)
    qx = np.array([np.sin(np.pi / 8), 0, 0, np.cos(np.pi / 8)], np.float32)  # 45 deg about x
    qz = np.array([0, 0, np.sin(np.pi / 8), np.cos(np.pi / 8)], np.float32)  # 45 deg about z
    gap = 2 * np.sqrt(2.0) - 0.05
    w.set_bodies(np.array([[0, 0, 0], [0, gap, 0]], np.float32), rot=np.stack([qz, qx]), shape_type=st, half_extent=he)
    w.collide_now()
    ids, counts, normals, points = w.get_manifolds()
    assert counts[0] == 1 and np.allclose(np.abs(normals[0]), [0, 1, 0], atol=1e-5)
    assert abs(points[0, 0, 3] - 0.05) < 1e-4


# ---- A12 solver -----------------------------------------------------------------------------------
def test_single_cube_comes_to_rest_on_the_plane():
    w = world(FLAG_COLLISIONS | FLAG_GROUND_PLANE)
    st, he = boxes(1)
    w.set_bodies(np.array([[0, 3.0, 0]], np.float32), shape_type=st, half_extent=he)
    w.update_n(DT, 400)
    pos, rot = w.get_transforms()
    lin, ang = w.get_velocities()
    assert abs(pos[0, 1] - 1.0) < 0.02  # rest at y = 1 +- slop
    assert np.abs(lin).max() < 0.2 and np.abs(ang).max() < 1e-3


def test_symmetric_head_on_hit_conserves_linear_momentum():
    w = world(gravity_force=(0, 0, 0), friction=0.0)
    w.set_bodies(np.array([[-1.5, 0, 0], [1.5, 0, 0]], np.float32), lin_vel=np.array([[2, 0, 0], [-2, 0, 0]], np.float32),
                 shape_type=np.full(2, SHAPE_SPHERE, np.uint32), half_extent=np.ones((2, 3), np.float32))
    for _ in range(60):
        w.update(DT)
        lin, _ = w.get_velocities()
        assert abs(float(lin[0, 0]) + float(lin[1, 0])) < 1e-5  # total momentum stays zero
    pos, _ = w.get_transforms()
    assert pos[1, 0] - pos[0, 0] >= 2.0 - 0.05  # they did not pass through each other
    lin, _ = w.get_velocities()
    assert abs(lin[0, 0]) < 0.3  # inelastic: approach velocity removed


def test_warm_started_stack_stands_where_a_cold_one_creeps():
    """contact_solve.h warm starting: a column of 12 unit cubes, eight iterations. Started every update from the impulses
    the previous update ended with, the column is at rest after 4 s (bottom cube at y = 1 to 2e-4, the top one 0.022 below its
    lattice height - eleven contacts inside their slop - speeds below 0.1); from zero every update (PHYS_FLAG_NO_WARM_START, rounds 1-2) eight iterations
    never carry the column: it has sunk by more than 0.1 and keeps creeping."""
    from physics_amd import FLAG_NO_WARM_START
    out = {}
    for label, extra in (("warm", 0), ("cold", FLAG_NO_WARM_START)):
        w = world(FLAG_COLLISIONS | FLAG_GROUND_PLANE | extra)
        pos = scenes.lattice(1, 12, 1, 2.0, 1.0, 0.0)
        st, he = boxes(len(pos))
        w.set_bodies(pos, shape_type=st, half_extent=he)
        w.update_n(DT, 240)
        out[label] = (w.get_transforms()[0], w.get_velocities()[0])
    y_warm, v_warm = out["warm"][0][:, 1], out["warm"][1]
    assert abs(y_warm[0] - 1.0) < 2e-4 and np.abs(y_warm - (1.0 + 2.0 * np.arange(12))).max() < 0.03, y_warm
    assert np.abs(v_warm).max() < 0.1
    y_cold = out["cold"][0][:, 1]
    assert (1.0 + 2.0 * 11) - y_cold[-1] > 0.1, y_cold


def test_colouring_is_proper_and_order_independent():
    sc = scenes.c3(6, 5, 6)
    w = ob.OracleWorld(sc.config(), trig=ob.TRIG_DET)
    sc.populate(w)
    w.update_n(DT, 120)
    ids, counts, _, _ = w.get_manifolds()
    colors = w.get_colors()
    assert len(ids) > 100
    seen = set()
    for (a, b), c in zip(ids.tolist(), colors.tolist()):
        assert (a, c) not in seen
        seen.add((a, c))
        if b != GROUND_ID:
            assert (b, c) not in seen
            seen.add((b, c))
    assert colors.max() + 1 == w.get_stats().n_colors


def test_both_solver_drivers_give_the_same_bits():
    """contact_solve.h drives the same row arithmetic three ways: Jacobians of all rows made beforehand (the
    dataflow kernels, which make them while they wait), row by row on the way (the per-colour kernels and the oracle),
    or with lever arms, Jacobians and - in the first sweep - the row masses remade from the contact points and the body
    positions (`solve_manifold_geo`, the cluster solver's compact rows). Random manifolds, full inertia tensors, with and
    without a second body, eight sweeps: not one bit may differ between the three."""
    from oracle import binding as ob
    rng = np.random.default_rng(5)
    for trial in range(300):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        count = int(rng.integers(1, 5))
        has_b = int(rng.integers(0, 2))
        pts = np.concatenate([np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), [rng.uniform(-0.02, 0.05)]])
                              for _ in range(4)])
        a = rng.normal(scale=0.3, size=(2, 3, 3))
        inertia = [np.eye(3) * rng.uniform(0.5, 2.0) + m @ m.T for m in a]
        flat = np.concatenate([n, pts, rng.uniform(-2, 2, 3), rng.uniform(0.3, 2.0, 2), inertia[0].reshape(-1),
                               inertia[1].reshape(-1), rng.normal(scale=2.0, size=12), [rng.uniform(0.0, 1.0)]])
        assert ob.solve_drivers_mismatch(flat.astype(np.float32), count, has_b) == 0, trial
