"""Synthetic benchmark / parity scenes of SURVEY.md §8 row D (BASELINE.json configs C1-C5).

All scenes: f32, mass 1, inertia identity (reference defaults rigid_body.rs:64-76), cube half-extent
1.0 (res/cube.obj spans [-1,1]^3), sphere radius 1.0, zero initial velocities, ground plane y = 0,
gravity force (0,-9.81,0) with gravity_offset = (0,0,0) (documented divergence from quirk Q2 for
benchmark scenes; reference-parity tests keep (0,0,1.5)). Jitter comes from splitmix64(seed=12345).
"""
import numpy as np

from ._abi import (FLAG_BROADPHASE_ONLY, FLAG_COLLISIONS, FLAG_EXCLUSIVE_GPU, FLAG_GROUND_PLANE, SHAPE_BOX, SHAPE_SPHERE,
                   default_config)

DT_NANOS = 16_666_667  # Duration::from_nanos -> as_secs_f32 = 0.016666668 (quirk Q7)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n):
    """First n outputs of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform_pm(seed, n, j):
    u = (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return ((2.0 * u - 1.0) * j).astype(np.float32)


def lattice(nx, ny, nz, spacing, y0, jitter, seed=12345, x0=0.0):
    """Bodies on an nx*ny*nz lattice, x fastest then z then y (body id order = memory order)."""
    n = nx * ny * nz
    ix = np.arange(n) % nx
    iz = (np.arange(n) // nx) % nz
    iy = np.arange(n) // (nx * nz)
    pos = np.empty((n, 3), np.float32)
    pos[:, 0] = x0 + (ix - (nx - 1) / 2.0) * spacing
    pos[:, 1] = y0 + iy * spacing
    pos[:, 2] = (iz - (nz - 1) / 2.0) * spacing
    if jitter > 0:
        pos += _uniform_pm(seed, 3 * n, jitter).reshape(n, 3)
    return pos


class Scene:
    def __init__(self, name, pos, shape_type, half_extent, flags, solver_iterations=8, rot=None, mass=None,
                 constraints=(), **cfg):
        self.name = name
        self.pos = pos
        self.shape_type = shape_type
        self.half_extent = half_extent
        self.flags = flags
        self.solver_iterations = solver_iterations
        self.rot = rot
        self.mass = mass
        self.constraints = list(constraints)  # ("point" | "orientation", body, target)
        self.cfg_overrides = cfg

    @property
    def n(self):
        return self.pos.shape[0]

    def config(self, device=0, **extra):
        kw = dict(flags=self.flags, gravity_offset=(0.0, 0.0, 0.0), solver_iterations=self.solver_iterations,
                  device=device)
        kw.update(self.cfg_overrides)
        kw.update(extra)
        return default_config(**kw)

    def populate(self, world, state=None):
        """Upload the scene - or, with `state` = (pos, rot, lin_vel, ang_vel), the same bodies at a later moment."""
        if state is None:
            world.set_bodies(self.pos, rot=self.rot, mass=self.mass, shape_type=self.shape_type, half_extent=self.half_extent)
        else:
            pos, rot, lin, ang = state
            world.set_bodies(pos, rot=rot, lin_vel=lin, ang_vel=ang, mass=self.mass, shape_type=self.shape_type,
                             half_extent=self.half_extent)
        for kind, body, target in self.constraints:
            if kind == "point":
                world.add_constraint_fix_point(body, target)
            else:
                world.add_constraint_fix_orientation(body, target)


def _cubes(n):
    return np.full(n, SHAPE_BOX, np.uint32), np.ones((n, 3), np.float32)


def falling_cubes(nx, ny, nz, name, spacing=2.5, y0=2.0, jitter=0.05, x0=0.0):
    pos = lattice(nx, ny, nz, spacing, y0, jitter, x0=x0)
    st, he = _cubes(pos.shape[0])
    return Scene(name, pos, st, he, FLAG_COLLISIONS | FLAG_GROUND_PLANE)


def c1():
    """C1: 64 cubes, 4x4x4, spacing 2.5, lowest centre y = 2.0, jitter 0.05."""
    return falling_cubes(4, 4, 4, "C1_64_cubes")


def c2():
    """C2: 10 000 cubes, 25x16x25 (x,y,z)."""
    return falling_cubes(25, 16, 25, "C2_10k_cubes")


def c3(nx=50, ny=40, nz=50):
    """C3: 100 000 mixed spheres / cubes (sphere iff splitmix64(i) & 1), 8 SI iterations."""
    pos = lattice(nx, ny, nz, 2.5, 2.0, 0.05)
    n = pos.shape[0]
    bits = splitmix64(777, n) & np.uint64(1)
    st = np.where(bits == 1, SHAPE_SPHERE, SHAPE_BOX).astype(np.uint32)
    he = np.ones((n, 3), np.float32)
    return Scene("C3_100k_mixed" if n == 100000 else f"C3_{n}_mixed", pos, st, he,
                 FLAG_COLLISIONS | FLAG_GROUND_PLANE)


def c4(nx=100, ny=100, nz=100, x0=0.0):
    """C4: 1 000 000 bodies, spacing 2.2, jitter 0.3 (dense AABB overlaps), broad-phase only."""
    pos = lattice(nx, ny, nz, 2.2, 2.0, 0.3, x0=x0)
    st, he = _cubes(pos.shape[0])
    return Scene(f"C4_{pos.shape[0]}_broadphase", pos, st, he, FLAG_COLLISIONS | FLAG_BROADPHASE_ONLY)


def c5(nx=16, ny=1000, nz=16):
    """C5: 256 000 boxes, 16x1000x16 tower, spacing exactly 2.0 (resting contact), no jitter."""
    pos = lattice(nx, ny, nz, 2.0, 1.0, 0.0)
    st, he = _cubes(pos.shape[0])
    return Scene(f"C5_{pos.shape[0]}_tower", pos, st, he, FLAG_COLLISIONS | FLAG_GROUND_PLANE)


def target_1m():
    """north_star target scene: 1M cubes dropped onto a plane (100x100x100, spacing 2.5)."""
    return falling_cubes(100, 100, 100, "T_1M_cubes")


def _quat_from_roll(roll):
    """UnitQuaternion::from_euler_angles(roll, 0, 0) as [i, j, k, w] (lib.rs:22)."""
    return np.array([np.sin(roll / 2.0), 0.0, 0.0, np.cos(roll / 2.0)], np.float32)


def reference_path(n=1_000_000):
    """The reference's OWN path at scale: PhysicsState::update with no collision stage (flags 0) and every
    reference quirk on - gravity as a 9.81 N force at the world offset (0, 0, 1.5) (physics.rs:89-92), RigidBody::step -
    for n bodies; entity 0 is the demo's body (lib.rs:20-25: position (1, 0, 0), roll 1 rad, pinned to the origin and to
    zero orientation), so the constraint solve (2 constraints, 6 rows) and the quirk-Q3 scatter run every update too."""
    nx = int(round(n ** (1.0 / 3.0)))
    pos = lattice(nx, n // (nx * nx), nx, 2.5, 2.0, 0.05)
    n = pos.shape[0]
    pos[0] = (1.0, 0.0, 0.0)
    rot = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))
    rot[0] = _quat_from_roll(1.0)
    cons = [("point", 0, (0.0, 0.0, 0.0)), ("orientation", 0, (0.0, 0.0, 0.0))]
    return Scene(f"REF_{n}_update", pos, None, None, 0, rot=rot, constraints=cons, gravity_offset=(0.0, 0.0, 1.5))


def reference_cg(n=4096):
    """The reference's constraint solver at scale: n bodies of unequal mass, each pinned to its start position and to
    zero orientation (2 constraints = 6 rows per body; A = J W J^T is diagonal with n distinct values 1 / mass, so the CG
    needs many iterations), rolled by a random angle so that both constraint kinds pull."""
    nx = int(round(n ** (1.0 / 3.0)))
    pos = lattice(nx, n // (nx * nx), nx, 2.5, 2.0, 0.05)
    n = pos.shape[0]
    u = (splitmix64(4242, 2 * n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    mass = (10.0 ** (2.0 * u[:n] - 1.0)).astype(np.float32)  # log-uniform in [0.1, 10]: condition number 100
    rot = np.stack([_quat_from_roll(r) for r in (u[n:] - 0.5)]).astype(np.float32)
    cons = []
    for b in range(n):
        cons.append(("point", b, tuple(float(x) for x in pos[b])))
        cons.append(("orientation", b, (0.0, 0.0, 0.0)))
    return Scene(f"REF_CG_{n}_bodies_{6 * n}_rows", pos, None, None, 0, rot=rot, mass=mass, constraints=cons,
                 gravity_offset=(0.0, 0.0, 1.5))


SCENES = {"c1": c1, "c2": c2, "c3": c3, "c4": c4, "c5": c5, "t1m": target_1m, "t1m_settled": target_1m,
          "ref_1m": reference_path, "ref_cg": reference_cg}
