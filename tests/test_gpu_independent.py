"""GPU: the narrow phase (A11) and the contact solver (A12) checked against an INDEPENDENT float64 numpy reference.

The HIP kernels and the CPU oracle compile the same arithmetic headers (include/spec/collide.h, contact_solve.h), so
GPU-vs-oracle parity cannot see a wrong SAT axis, clip plane or effective mass: both sides would be wrong alike. This
file imports neither `oracle` nor anything built from include/spec (tests/test_abi.py checks that): every expectation
below is recomputed here, in float64, from the definitions - brute-force 15-axis SAT, closed-form sphere contacts, box
vertices against the plane, and a sequential-impulse solve of isolated manifolds written from the textbook formulas
(lambda = m_eff * (bias - v_n), m_eff = 1 / (1/mA + 1/mB + (rA x n).IA^-1 (rA x n) + ...)).
Tolerances are written at each assert (float32 kernels against float64 references: 1e-4 absolute on unit-sized
shapes unless stated)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 16_666_667
DT_S = float(np.float32(np.float32(16_666_667) / np.float32(1e9)))
MARGIN = 0.02
GROUND = 0xFFFFFFFF


# ---------------------------------------------------------------- float64 geometry, written for this file only
def quat_to_matrix(q):
    """Rotation matrix of quaternion [i, j, k, w] (columns = body axes in world coordinates)."""
    i, j, k, w = [float(x) for x in q]
    return np.array([[w * w + i * i - j * j - k * k, 2 * (i * j - w * k), 2 * (w * j + i * k)],
                     [2 * (w * k + i * j), w * w - i * i + j * j - k * k, 2 * (j * k - w * i)],
                     [2 * (i * k - w * j), 2 * (w * i + j * k), w * w - i * i - j * j + k * k]])


def random_quats(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def renormalised(q32):
    """The float32 quaternion the device sees, as float64 (its norm differs from 1 by ~1e-7: irrelevant at 1e-4)."""
    return q32.astype(np.float64)


def box_radius(R, h, L):
    return float(np.abs(L @ R) @ h)


def sat_axes(RA, RB, min_cross):
    """The 15 candidate separating axes of two boxes as unit vectors with a label and |a_i x b_j| (1 for face axes);
    cross products shorter than min_cross (near-parallel edges, covered by the face axes) are left out."""
    out = []
    for i in range(3):
        out.append((RA[:, i], ("A", i), 1.0))
    for j in range(3):
        out.append((RB[:, j], ("B", j), 1.0))
    for i in range(3):
        for j in range(3):
            c = np.cross(RA[:, i], RB[:, j])
            n = np.linalg.norm(c)
            if n > min_cross:
                out.append((c / n, ("E", i, j), n))
    return out


def sat_separation(cA, RA, hA, cB, RB, hB, min_cross=0.02):
    """max over the candidate axes of the signed separation (negative = penetration along that axis), and the list.
    The kernel treats edge pairs with |a_i x b_j| < 0.01 as parallel; 0.02 here keeps this axis set inside the kernel's."""
    d = cB - cA
    seps = [(abs(float(d @ L)) - box_radius(RA, hA, L) - box_radius(RB, hB, L), L, lab, cn) for L, lab, cn in sat_axes(RA, RB, min_cross)]
    return max(s for s, _, _, _ in seps), seps


def outside_distance(p, c, R, h):
    """How far p lies outside box (c, R, h) along its worst axis (<= 0: inside)."""
    return float(np.max(np.abs((p - c) @ R) - h))


def face_patch(cR, RR, hR, axis, n_ref, cI, RI, hI):
    """The contact patch of a face contact, in float64 and written for this file: the face of the incident box most
    anti-parallel to n_ref, clipped (Sutherland-Hodgman) to the four side planes of the reference face. Returns
    (patch width = the smaller extent along the reference face's two in-plane axes, the smallest half extent of the
    two faces, deepest point below the reference face) - the three numbers of collide.h's sliver rule - or None when
    nothing is left of the incident face."""
    dots = n_ref @ RI
    j = int(np.argmax(np.abs(dots)))
    jsgn = -1.0 if dots[j] > 0 else 1.0
    j1, j2 = (j + 1) % 3, (j + 2) % 3
    fc = cI + RI[:, j] * jsgn * hI[j]
    e1, e2 = RI[:, j1] * hI[j1], RI[:, j2] * hI[j2]
    poly = [fc + e1 + e2, fc - e1 + e2, fc - e1 - e2, fc + e1 - e2]
    r1, r2 = (axis + 1) % 3, (axis + 2) % 3
    for u, lim in ((RR[:, r1], hR[r1]), (-RR[:, r1], hR[r1]), (RR[:, r2], hR[r2]), (-RR[:, r2], hR[r2])):
        out = []
        for k in range(len(poly)):
            a, b = poly[k], poly[(k + 1) % len(poly)]
            da, db = (a - cR) @ u - lim, (b - cR) @ u - lim
            if da <= 0:
                out.append(a)
                if db > 0:
                    out.append(a + (b - a) * (da / (da - db)))
            elif db <= 0:
                out.append(a + (b - a) * (da / (da - db)))
        poly = out
        if not poly:
            return None
    P = np.array(poly)
    c1, c2 = (P - cR) @ RR[:, r1], (P - cR) @ RR[:, r2]
    width = min(c1.max() - c1.min(), c2.max() - c2.min())
    face = min(hR[r1], hR[r2], hI[j1], hI[j2])
    deepest = float(np.max(hR[axis] - (P - cR) @ n_ref))
    return float(width), float(face), deepest


SLIVER_REL, SLIVER_DEPTH = 0.1, 0.25 * MARGIN  # collide.h: narrower than 0.1 x face AND no deeper than margin / 4 -> no contact


def sliver_readings(cA, RA, hA, cB, RB, hB, seps, s_star):
    """Every face axis the kernel may have chosen as reference (within its preference band of the best axis), with the
    patch numbers of that reading."""
    d = cB - cA
    out = []
    for sv, L, lab, _ in seps:
        if lab[0] == "E" or sv < s_star - (0.03 + 0.11 * abs(s_star)) - 1e-4:
            continue
        if lab[0] == "A":
            n_ref = L if d @ L >= 0 else -L
            fp = face_patch(cA, RA, hA, lab[1], n_ref, cB, RB, hB)
        else:
            n_ref = -L if d @ L >= 0 else L
            fp = face_patch(cB, RB, hB, lab[1], n_ref, cA, RA, hA)
        out.append((lab, fp))
    return out


def pair_base(k, n_pairs, spacing=12.0):
    """Centre of pair k on a compact 3-D lattice around the origin: coordinates stay below ~100, where float32 resolves
    8e-6 (a row of thousands of pairs along x would put the 1e-4 tolerances of this file below float32's resolution)."""
    side = int(np.ceil(n_pairs ** (1.0 / 3.0)))
    ix, iy, iz = k % side, (k // side) % side, k // (side * side)
    return (np.array([ix, iy, iz], np.float64) - 0.5 * (side - 1)) * spacing


def manifold_dict(world):
    ids, counts, normals, points = world.get_manifolds()
    return {(int(a), int(b)): (int(c), n.astype(np.float64), p.astype(np.float64)) for (a, b), c, n, p in zip(ids, counts, normals, points)}


def pair_world(pos, rot, shape, he, ground=False, **cfg):
    import physics_amd
    flags = physics_amd.FLAG_COLLISIONS | (physics_amd.FLAG_GROUND_PLANE if ground else 0)
    kw = dict(flags=flags, gravity_force=(0, 0, 0), gravity_offset=(0, 0, 0))
    kw.update(cfg)
    w = physics_amd.World(physics_amd.default_config(**kw))
    w.set_bodies(pos, rot=rot, shape_type=shape, half_extent=he)
    return w


# ---------------------------------------------------------------- narrow phase: box - box
def _box_pairs(rng, n_pairs):
    """Pair k = bodies (2k, 2k+1), 12 units from every other pair (shapes reach at most 2.6 from their centre); relative poses from deep overlap to clear separation,
    random orientations plus a share of axis-aligned / face-parallel cases (the clipper's home ground)."""
    import physics_amd
    n = 2 * n_pairs
    pos = np.zeros((n, 3), np.float64)
    he = rng.uniform(0.5, 1.5, size=(n, 3))
    rot = random_quats(rng, n)
    aligned = rng.random(n_pairs) < 0.25
    for k in range(n_pairs):
        base = pair_base(k, n_pairs)
        if aligned[k]:
            rot[2 * k] = (0, 0, 0, 1)
            rot[2 * k + 1] = (0, 0, 0, 1) if rng.random() < 0.5 else rot[2 * k + 1]
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        if aligned[k] and rng.random() < 0.7:
            direction = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0]) + rng.normal(size=3) * 0.15
            direction /= np.linalg.norm(direction)
        reach = he[2 * k] @ np.abs(direction) + he[2 * k + 1] @ np.abs(direction)
        pos[2 * k] = base
        pos[2 * k + 1] = base + direction * reach * rng.uniform(0.55, 1.25)
        if k % 10 == 9:
            # diagonal neighbours of a stack: two axis-aligned boxes that (nearly) touch along an edge or at a corner -
            # the home ground of the sliver rule (patch of no width at depth ~0), scattered +-0.03 around exact touch
            rot[2 * k] = (0, 0, 0, 1)
            rot[2 * k + 1] = (0, 0, 0, 1)
            touch = rng.choice([-1.0, 1.0], size=3) * (he[2 * k] + he[2 * k + 1])
            if rng.random() < 0.7:
                touch[rng.integers(3)] *= rng.uniform(0.0, 0.8)  # edge neighbour: real overlap on one axis
            pos[2 * k + 1] = base + touch + rng.uniform(-0.03, 0.03, size=3)
    shape = np.full(n, physics_amd.SHAPE_BOX, np.uint32)
    return pos.astype(np.float32), rot, shape, he.astype(np.float32)


def test_box_box_manifolds_against_brute_force_sat():
    rng = np.random.default_rng(2024)
    n_pairs = 3000
    pos, rot, shape, he = _box_pairs(rng, n_pairs)
    w = pair_world(pos, rot, shape, he)
    w.update(DT)
    w.sync()
    man = manifold_dict(w)
    assert all(b == a + 1 and a % 2 == 0 for a, b in man), "a manifold between bodies of different pairs"
    n_hit = n_miss = n_band = n_edge = n_sliver = 0
    for k in range(n_pairs):
        a, b = 2 * k, 2 * k + 1
        cA, cB = pos[a].astype(np.float64), pos[b].astype(np.float64)
        RA, RB = quat_to_matrix(renormalised(rot[a])), quat_to_matrix(renormalised(rot[b]))
        hA, hB = he[a].astype(np.float64), he[b].astype(np.float64)
        s_star, seps = sat_separation(cA, RA, hA, cB, RB, hB)
        got = man.get((a, b))
        # (i) separated <=> no manifold, outside a band of 1e-3 around the contact margin
        if s_star > MARGIN + 1e-3:
            assert got is None, f"pair {k}: separated by {s_star:.5f} along a SAT axis, yet a manifold came back"
            n_miss += 1
            continue
        if s_star > MARGIN - 1e-3:
            n_band += 1
            if got is None:
                continue
        if got is None:
            # must-have branch: only when no axis of the FULL set (near-parallel edge pairs included) separates either -
            # or under the SLIVER RULE of collide.h: some face reading the kernel may have chosen has a patch narrower
            # than 0.1 x the smallest half extent of the two faces and no point deeper than a quarter of the margin
            # (bands of 1e-3 on the width and 1e-4 on the depth)
            s_all, _ = sat_separation(cA, RA, hA, cB, RB, hB, min_cross=1e-6)
            if s_all > MARGIN - 1e-3:
                continue
            # (a reading whose patch is EMPTY - nothing of the incident face lies over the reference face: boxes that
            # face each other across a gap on two axes at once - has no points to report either)
            slivers = [lab for lab, fp in sliver_readings(cA, RA, hA, cB, RB, hB, seps, s_star)
                       if fp is None or (fp[0] < SLIVER_REL * fp[1] + 1e-3 and fp[2] <= SLIVER_DEPTH + 1e-4)]
            assert slivers, f"pair {k}: no separating axis (max separation {s_all:.5f}), no sliver reading, yet no manifold"
            n_sliver += 1
            continue
        n_hit += 1
        count, normal, pts = got
        # (iv) 1..4 points
        assert 1 <= count <= 4
        pts = pts[:count]
        depth = pts[:, 3]
        # (ii) the normal is a unit vector from A towards B along one of the candidate axes ...
        assert abs(np.linalg.norm(normal) - 1.0) < 1e-4
        assert normal @ (cB - cA) > -1e-4, "normal does not point from A to B"
        _, seps_all = sat_separation(cA, RA, hA, cB, RB, hB, min_cross=0.0099)
        along = [(sv, L, lab, cn) for sv, L, lab, cn in seps_all if abs(abs(L @ normal) - 1.0) < 2e-4]
        assert along, f"pair {k}: normal {normal} is none of the 15 SAT axes"
        s_n = max(sv for sv, _, _, _ in along)
        # ... and a near-minimum-penetration one. The kernel prefers faces for frame coherence: an edge axis must be
        # shallower than the shallowest face by 5 % + 0.01, and a face of B likewise against A's. The two preferences
        # compound, so the chosen axis may trail the best by up to ~10 % + 0.02, never more:
        assert s_n >= s_star - (0.03 + 0.11 * abs(s_star)) - 1e-4, f"pair {k}: chosen axis separation {s_n:.4f} vs best {s_star:.4f}"
        # which kind of contact it is: try every axis the normal is parallel to (an A face and a B face can share a
        # direction); at least one reading must satisfy ALL the checks of its kind
        problems = []
        for sv, L, lab, cn in along:
            if lab[0] == "E":
                # edge - edge: one point midway between the closest points of the two edges; depth = penetration along
                # that axis (the kernel adds 1e-6 to |a_i . b_j| before dividing by |a_i x b_j|: 3e-6 / |cross| of slack)
                bad = None
                if count != 1:
                    bad = "an edge contact has one point"
                elif abs(depth[0] + sv) > 1e-4 + 3e-6 / cn:
                    bad = f"edge depth {depth[0]} vs {-sv}"
                elif max(outside_distance(pts[0, :3], cA, RA, hA), outside_distance(pts[0, :3], cB, RB, hB)) > abs(sv) + MARGIN + 1e-4:
                    bad = "edge contact point away from the boxes"
                if bad is None:
                    n_edge += 1
                    problems = None
                    break
                problems.append((lab, bad))
                continue
            # face contact: reference = the box whose face normal is the manifold normal; the points are the incident
            # face clipped to the reference face: on the incident box, inside the reference side planes, at or below
            # the reference face (+ margin); depth = distance below the reference face; never deeper than the SAT
            # penetration along that axis (the incident box's deepest vertex may be clipped away, nothing is deeper)
            if lab[0] == "A":
                cR, RR, hR, n_ref, cI, RI, hI = cA, RA, hA, normal, cB, RB, hB
            else:
                cR, RR, hR, n_ref, cI, RI, hI = cB, RB, hB, -normal, cA, RA, hA
            axis = lab[1]
            bad = None
            for p, dep in zip(pts[:, :3], depth):
                local = (p - cR) @ RR
                want = hR[axis] - (p - cR) @ n_ref
                if abs(dep - want) > 1e-4:
                    bad = f"depth {dep} vs {want} below the reference face"
                elif dep < -MARGIN - 1e-4 or dep > -sv + 1e-4:
                    bad = f"depth {dep} outside [-margin, SAT penetration {-sv}]"
                elif any(abs(local[t]) > hR[t] + 1e-4 for t in range(3) if t != axis):
                    bad = "point outside the reference face's side planes"
                elif outside_distance(p, cI, RI, hI) > 1e-4:
                    bad = "point not on the incident box"
            if bad is None and count >= 2 and len({tuple(np.round(p, 4)) for p in pts[:, :3]}) != count:
                bad = "duplicate contact points"
            if bad is None:
                # ... and it is not what the sliver rule says is no contact (same bands, from the other side)
                fp = face_patch(cR, RR, hR, axis, n_ref, cI, RI, hI)
                if fp is not None and fp[0] < SLIVER_REL * fp[1] - 1e-3 and fp[2] <= SLIVER_DEPTH - 1e-4:
                    bad = f"a sliver (patch width {fp[0]:.5f} of face {fp[1]:.3f}, deepest {fp[2]:.5f}) came back as a contact"
            if bad is None:
                problems = None
                break
            problems.append((lab, bad))
        assert problems is None, f"pair {k}: manifold fits no reading of its normal: {problems}\n{pts}"
    assert n_hit > 800 and n_miss > 300 and n_edge > 20 and n_sliver > 30, (n_hit, n_miss, n_band, n_edge, n_sliver)  # the sample covers all regimes


# ---------------------------------------------------------------- narrow phase: spheres
def test_sphere_sphere_and_sphere_box_closed_forms():
    import physics_amd
    rng = np.random.default_rng(7)
    n_pairs = 2000
    n = 2 * n_pairs
    pos = np.zeros((n, 3), np.float64)
    he = rng.uniform(0.5, 1.5, size=(n, 3))
    rot = random_quats(rng, n)
    shape = np.full(n, physics_amd.SHAPE_SPHERE, np.uint32)
    kind = rng.integers(0, 3, n_pairs)  # 0 sphere-sphere, 1 sphere(A)-box(B), 2 box(A)-sphere(B)
    for k in range(n_pairs):
        a, b = 2 * k, 2 * k + 1
        base = pair_base(k, n_pairs)
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        if kind[k] == 1:
            shape[b] = physics_amd.SHAPE_BOX
        elif kind[k] == 2:
            shape[a] = physics_amd.SHAPE_BOX
        ra = he[a, 0] if shape[a] == physics_amd.SHAPE_SPHERE else float(he[a] @ np.abs(quat_to_matrix(rot[a]).T @ direction))
        rb = he[b, 0] if shape[b] == physics_amd.SHAPE_SPHERE else float(he[b] @ np.abs(quat_to_matrix(rot[b]).T @ direction))
        pos[a] = base
        pos[b] = base + direction * (ra + rb) * rng.uniform(0.6, 1.2)
    pos32, he32 = pos.astype(np.float32), he.astype(np.float32)
    w = pair_world(pos32, rot, shape, he32)
    w.update(DT)
    w.sync()
    man = manifold_dict(w)
    checked = 0
    for k in range(n_pairs):
        a, b = 2 * k, 2 * k + 1
        cA, cB = pos32[a].astype(np.float64), pos32[b].astype(np.float64)
        got = man.get((a, b))
        if kind[k] == 0:
            rA, rB = float(he32[a, 0]), float(he32[b, 0])
            dist = np.linalg.norm(cB - cA)
            n_exp = (cB - cA) / dist
            dep_exp = rA + rB - dist
            pt_exp = cA + n_exp * (rA - 0.5 * dep_exp)
        else:
            # sphere S against box X: closest point q of the box to the sphere centre (centres inside the box do not
            # occur in this sample: checked), normal S -> X = (q - s) / |q - s|, depth = r - |q - s|, point = q
            s_idx, x_idx = (a, b) if kind[k] == 1 else (b, a)
            cS, cX = pos32[s_idx].astype(np.float64), pos32[x_idx].astype(np.float64)
            RX, hX, r = quat_to_matrix(renormalised(rot[x_idx])), he32[x_idx].astype(np.float64), float(he32[s_idx, 0])
            local = (cS - cX) @ RX
            q_local = np.clip(local, -hX, hX)
            if np.array_equal(q_local, local):
                continue
            q = cX + RX @ q_local
            dist = np.linalg.norm(q - cS)
            n_sx = (q - cS) / dist
            n_exp = n_sx if kind[k] == 1 else -n_sx   # the manifold normal always points from body A to body B
            dep_exp = r - dist
            pt_exp = q
        if dep_exp < -MARGIN - 1e-4:
            assert got is None, f"pair {k} (kind {kind[k]}): gap {-dep_exp:.5f} > margin, yet a manifold"
            continue
        if dep_exp < -MARGIN + 1e-4:
            continue
        assert got is not None, f"pair {k} (kind {kind[k]}): depth {dep_exp:.5f}, no manifold"
        count, normal, pts = got
        assert count == 1
        assert np.abs(normal - n_exp).max() < 1e-4, f"pair {k}: normal {normal} vs {n_exp}"
        assert abs(pts[0, 3] - dep_exp) < 1e-4
        assert np.abs(pts[0, :3] - pt_exp).max() < 1e-4
        checked += 1
    assert checked > 600


# ---------------------------------------------------------------- narrow phase: ground plane
def test_ground_manifolds_against_box_vertices():
    import physics_amd
    rng = np.random.default_rng(11)
    n = 3000
    he = rng.uniform(0.4, 1.6, size=(n, 3)).astype(np.float32)
    rot = random_quats(rng, n)
    rot[: n // 5] = (0, 0, 0, 1)  # flat boxes: four vertices touch at once
    shape = np.where(rng.random(n) < 0.3, physics_amd.SHAPE_SPHERE, physics_amd.SHAPE_BOX).astype(np.uint32)
    pos = np.zeros((n, 3), np.float64)
    pos[:, 0] = (np.arange(n) % 55 - 27) * 8.0
    pos[:, 2] = (np.arange(n) // 55 - 27) * 8.0
    for i in range(n):
        R = quat_to_matrix(renormalised(rot[i]))
        low = float(he[i, 0]) if shape[i] == physics_amd.SHAPE_SPHERE else float(np.abs(R[1]) @ he[i].astype(np.float64))
        pos[i, 1] = low + rng.uniform(-0.3, 0.1)  # lowest point between 0.3 below and 0.1 above the plane
    pos32 = pos.astype(np.float32)
    w = pair_world(pos32, rot, shape, he, ground=True)
    w.update(DT)
    w.sync()
    man = manifold_dict(w)
    assert all(b == GROUND for _, b in man)
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64)
    checked = 0
    for i in range(n):
        c = pos32[i].astype(np.float64)
        got = man.get((i, GROUND))
        if shape[i] == physics_amd.SHAPE_SPHERE:
            r = float(he[i, 0])
            dep_exp = 0.0 - (c[1] - r)
            if abs(dep_exp + MARGIN) < 1e-4:
                continue
            assert (got is not None) == (dep_exp > -MARGIN)
            if got:
                count, normal, pts = got
                assert count == 1 and np.abs(normal - (0, -1, 0)).max() < 1e-6
                assert abs(pts[0, 3] - dep_exp) < 1e-4
                assert np.abs(pts[0, :3] - (c[0], c[1] - r + 0.5 * dep_exp, c[2])).max() < 1e-4
                checked += 1
            continue
        R = quat_to_matrix(renormalised(rot[i]))
        verts = c + (corners * he[i].astype(np.float64)) @ R.T
        dep = -verts[:, 1]
        if (np.abs(dep + MARGIN) < 1e-4).any():
            continue
        touching = dep > -MARGIN
        assert (got is not None) == bool(touching.any())
        if got is None:
            continue
        count, normal, pts = got
        assert np.abs(normal - (0, -1, 0)).max() < 1e-6, "A = body, B = ground: the A -> B normal is -y"
        assert count == min(4, int(touching.sum())), f"body {i}: {count} points, {int(touching.sum())} vertices within the margin"
        used = set()
        for p in pts[:count]:
            d2 = np.linalg.norm(verts - p[:3], axis=1)
            j = int(np.argmin(d2))
            assert d2[j] < 1e-4 and touching[j], f"body {i}: contact point is not a touching vertex"
            assert abs(p[3] - dep[j]) < 1e-4
            used.add(j)
        assert len(used) == count
        assert int(np.argmax(dep)) in used, "the deepest vertex must be kept"
        checked += 1
    assert checked > 1500


# ---------------------------------------------------------------- contact solver
def solve_isolated_manifold(normal, pts, xA, xB, vA, wA, vB, wB, invmA, invmB, IA_inv, IB_inv, iterations, baumgarte=0.2,
                            slop=0.01, friction=0.5, max_bias=3.0):
    """Textbook sequential impulses on ONE manifold in float64 (xB None = static ground): per iteration, per point in
    index order: two friction rows (clamped to +-mu * accumulated normal impulse of that point), then the normal row
    (accumulated impulse clamped >= 0). Written from the formulas, sharing no code with include/spec."""
    n = normal
    # any orthonormal tangent pair spans the same friction disc only approximately (the clamp is a box in the chosen
    # basis), so this reference is used where friction stays zero or the basis is irrelevant (see the tests)
    t1 = np.cross(n, [1.0, 0, 0]) if abs(n[0]) < 0.9 else np.cross(n, [0, 1.0, 0])
    t1 /= np.linalg.norm(t1)
    t2 = np.cross(n, t1)
    vA, wA = vA.copy(), wA.copy()
    vB, wB = (vB.copy(), wB.copy()) if xB is not None else (np.zeros(3), np.zeros(3))
    rows = []
    for p in pts:
        rA = p[:3] - xA
        rB = p[:3] - xB if xB is not None else np.zeros(3)

        def mass(d):
            k = invmA + np.cross(rA, d) @ IA_inv @ np.cross(rA, d)
            if xB is not None:
                k += invmB + np.cross(rB, d) @ IB_inv @ np.cross(rB, d)
            return 1.0 / k
        depth = p[3]
        bias = 0.0
        if depth > slop:
            bias = min(baumgarte / DT_S * (depth - slop), max_bias)
        elif depth < 0.0:
            bias = depth / DT_S
        rows.append(dict(rA=rA, rB=rB, mn=mass(n), mt=(mass(t1), mass(t2)), bias=bias, pn=0.0, pt=[0.0, 0.0]))

    def apply(d, lam, r):
        nonlocal vA, wA, vB, wB
        vA = vA - d * lam * invmA
        wA = wA - IA_inv @ np.cross(r["rA"], d) * lam
        if xB is not None:
            vB = vB + d * lam * invmB
            wB = wB + IB_inv @ np.cross(r["rB"], d) * lam

    def rel(d, r):
        ub = d @ vB + np.cross(r["rB"], d) @ wB if xB is not None else 0.0
        return ub - (d @ vA + np.cross(r["rA"], d) @ wA)
    for _ in range(iterations):
        for r in rows:
            for t, d in enumerate((t1, t2)):
                lam = -r["mt"][t] * rel(d, r)
                lim = friction * r["pn"]
                new = max(-lim, min(r["pt"][t] + lam, lim))
                lam, r["pt"][t] = new - r["pt"][t], new
                apply(d, lam, r)
            lam = r["mn"] * (r["bias"] - rel(n, r))
            new = max(r["pn"] + lam, 0.0)
            lam, r["pn"] = new - r["pn"], new
            apply(n, lam, r)
    return vA, wA, vB, wB, [r["pn"] for r in rows]


def test_one_iteration_impulses_match_the_closed_form():
    """Isolated contacts (every body in exactly one manifold), solver_iterations = 1. In the first iteration the
    friction rows run while the accumulated normal impulse of their point is still zero, so they are clamped to zero
    and the velocities after the update are those of the normal rows alone: lambda = m_eff (bias - v_n), clamped at
    zero, point after point - recomputed here in float64 from the manifold the device reports (which
    test_*_manifolds_* above check separately) and full inertia tensors."""
    import physics_amd
    rng = np.random.default_rng(5)
    n_pairs = 1500
    pos, rot, shape, he = _box_pairs(rng, n_pairs)
    n = 2 * n_pairs
    mass = rng.uniform(0.5, 4.0, n).astype(np.float32)
    inertia = np.zeros((n, 3, 3), np.float32)
    for i in range(n):
        A = rng.normal(size=(3, 3))
        inertia[i] = (A @ A.T + 3.0 * np.eye(3)).astype(np.float32)  # symmetric positive definite, not diagonal
    lin = rng.normal(scale=1.0, size=(n, 3)).astype(np.float32)
    ang = rng.normal(scale=1.0, size=(n, 3)).astype(np.float32)
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_force=(0, 0, 0), gravity_offset=(0, 0, 0),
                                     solver_iterations=1)
    w = physics_amd.World(cfg)
    w.set_bodies(pos, rot=rot, lin_vel=lin, ang_vel=ang, mass=mass, inertia=inertia.reshape(n, 9), shape_type=shape, half_extent=he)
    w.update(DT)
    w.sync()
    man = manifold_dict(w)
    lin1, ang1 = w.get_velocities()
    checked = 0
    worst = 0.0
    for (a, b), (count, normal, pts) in man.items():
        IA_inv = np.linalg.inv(inertia[a].astype(np.float64))
        IB_inv = np.linalg.inv(inertia[b].astype(np.float64))
        vA, wA, vB, wB, pn = solve_isolated_manifold(normal, pts[:count], pos[a].astype(np.float64), pos[b].astype(np.float64),
                                                     lin[a].astype(np.float64), ang[a].astype(np.float64),
                                                     lin[b].astype(np.float64), ang[b].astype(np.float64),
                                                     1.0 / float(mass[a]), 1.0 / float(mass[b]), IA_inv, IB_inv, 1)
        for got, want in ((lin1[a], vA), (ang1[a], wA), (lin1[b], vB), (ang1[b], wB)):
            err = np.abs(got - want).max()
            worst = max(worst, err)
            # float32 rows of ~60 operations each on velocities of order 1-10: 1e-4 absolute + 1e-4 relative
            assert err < 1e-4 + 1e-4 * np.abs(want).max(), f"pair ({a},{b}): velocity {got} vs closed form {want}"
        # linear momentum of the pair is conserved by the contact impulses (Newton's third law)
        p0 = mass[a] * lin[a].astype(np.float64) + mass[b] * lin[b].astype(np.float64)
        p1 = mass[a] * lin1[a].astype(np.float64) + mass[b] * lin1[b].astype(np.float64)
        assert np.abs(p1 - p0).max() < 1e-4 * (1.0 + np.abs(p0).max())
        checked += 1
    # bodies without a manifold keep their velocities exactly
    touched = {i for ab in man for i in ab}
    free = np.array([i for i in range(n) if i not in touched])
    assert np.array_equal(lin1[free], lin[free]) and np.array_equal(ang1[free], ang[free])
    assert checked > 700, checked


def test_eight_iterations_on_isolated_sphere_contacts_match_float64_sequential_impulses():
    """Sphere - sphere and sphere - ground contacts have r x n = 0 at every point of the solve, so friction acts on
    the linear velocities only and the disc-vs-box shape of the friction clamp is the only basis-dependent part; with
    zero tangential velocity it never engages. Eight iterations of the device against eight iterations in float64."""
    import physics_amd
    rng = np.random.default_rng(9)
    n_pairs = 1000
    n = 2 * n_pairs
    pos = np.zeros((n, 3), np.float64)
    he = np.ones((n, 3)) * rng.uniform(0.5, 1.5, size=(n, 1))
    mass = rng.uniform(0.5, 4.0, n).astype(np.float32)
    lin = np.zeros((n, 3))
    for k in range(n_pairs):
        a, b = 2 * k, 2 * k + 1
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        pos[a] = pair_base(k, n_pairs, spacing=8.0)
        pos[b] = pos[a] + direction * (he[a, 0] + he[b, 0]) * rng.uniform(0.9, 1.005)
        speed = rng.uniform(0.0, 3.0)
        lin[a], lin[b] = direction * speed, -direction * speed * rng.uniform(0.0, 1.0)  # approaching along the line of centres
    pos32, he32, lin32 = pos.astype(np.float32), he.astype(np.float32), lin.astype(np.float32)
    shape = np.full(n, physics_amd.SHAPE_SPHERE, np.uint32)
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_force=(0, 0, 0), gravity_offset=(0, 0, 0),
                                     solver_iterations=8)
    w = physics_amd.World(cfg)
    w.set_bodies(pos32, lin_vel=lin32, mass=mass, shape_type=shape, half_extent=he32)
    w.update(DT)
    w.sync()
    man = manifold_dict(w)
    lin1, ang1 = w.get_velocities()
    # friction acts at the contact point (r x t != 0), but the tangential velocity here is rounding noise of the float32
    # velocities, so is the spin it can cause
    assert np.abs(ang1).max() < 1e-4, "central contacts must not spin spheres up"
    eye = np.eye(3)
    for (a, b), (count, normal, pts) in man.items():
        vA, wA, vB, wB, pn = solve_isolated_manifold(normal, pts[:count], pos32[a].astype(np.float64), pos32[b].astype(np.float64),
                                                     lin32[a].astype(np.float64), np.zeros(3), lin32[b].astype(np.float64), np.zeros(3),
                                                     1.0 / float(mass[a]), 1.0 / float(mass[b]), eye, eye, 8)
        assert np.abs(lin1[a] - vA).max() < 1e-4 and np.abs(lin1[b] - vB).max() < 1e-4
        # after the solve the bodies do not approach faster than the bias allows: v_n >= bias - eps at the one point
        v_n = (lin1[b].astype(np.float64) - lin1[a].astype(np.float64)) @ normal
        depth = pts[0, 3]
        bias = min(0.2 / DT_S * (depth - 0.01), 3.0) if depth > 0.01 else (depth / DT_S if depth < 0 else 0.0)
        assert v_n >= bias - 1e-4
    assert len(man) > 500


def test_settled_pile_has_no_approaching_contacts():
    """C1 (64 cubes on the plane) after it has come to rest: for every contact point of the last update the relative
    normal velocity computed HERE from the returned velocities, poses and manifolds is non-approaching within the
    Gauss-Seidel residual of 8 iterations (|v| of a resting pile: millimetres per second)."""
    import physics_amd
    from physics_amd import scenes
    sc = scenes.c1()
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.update_n(DT, 600)
    w.sync()
    pos1, _ = w.get_transforms()
    lin, ang = w.get_velocities()
    man = manifold_dict(w)
    x0 = pos1.astype(np.float64) - lin.astype(np.float64) * DT_S  # poses the manifolds were made from
    worst = 0.0
    n_points = 0
    for (a, b), (count, normal, pts) in man.items():
        for p in pts[:count]:
            va = lin[a].astype(np.float64) + np.cross(ang[a].astype(np.float64), p[:3] - x0[a])
            vb = np.zeros(3) if b == GROUND else lin[b].astype(np.float64) + np.cross(ang[b].astype(np.float64), p[:3] - x0[b])
            v_n = (vb - va) @ normal
            worst = min(worst, v_n)
            n_points += 1
    assert n_points > 200
    assert np.abs(lin).max() < 0.05, "the pile has not settled"
    assert worst > -0.02, f"a contact of the resting pile closes at {worst:.4f} units/s"
    assert pos1[:, 1].min() > 0.97, "a cube of half extent 1 sank into the plane"
