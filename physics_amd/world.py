"""World — thin ctypes wrapper over the C ABI (include/physics_hip.h). Every method is one ABI call;
the reference-shaped object model (PhysicsState / Entity / RigidBody) lives in state.py on top."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import STAGE_NAMES, PhysDeviceView, PhysProfile, PhysStats, f32p, u32p


class PhysError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"physics_hip error {code}: {msg}")
        self.code = code


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=f32p):
    return None if a is None else a.ctypes.data_as(t)


class World:
    def __init__(self, cfg=None):
        self.lib = _abi.load_library()
        self.cfg = cfg if cfg is not None else _abi.default_config()
        self.h = C.c_void_p()
        self.n = 0
        self._ck(self.lib.phys_create(C.byref(self.cfg), C.byref(self.h)))

    def _ck(self, rc):
        if rc != 0:
            raise PhysError(rc, self.lib.phys_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.phys_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- state upload
    def set_bodies(self, pos, rot=None, lin_vel=None, ang_vel=None, mass=None, inertia=None, shape_type=None,
                   half_extent=None):
        pos = _f(pos).reshape(-1, 3)
        n = pos.shape[0]
        arrs = [pos, _f(rot), _f(lin_vel), _f(ang_vel), _f(mass), _f(inertia)]
        for a, w in zip(arrs[1:], (4, 3, 3, 1, 9)):
            if a is not None and a.size != n * w:
                raise ValueError("array size does not match the body count")
        st = None if shape_type is None else np.ascontiguousarray(shape_type, dtype=np.uint32)
        he = _f(half_extent)
        self._ck(self.lib.phys_set_bodies(self.h, n, *[_p(a) for a in arrs], _p(st, u32p), _p(he)))
        self.n = n

    def add_constraint_fix_point(self, body, target):
        t = _f(target)
        self._ck(self.lib.phys_add_constraint_fix_point(self.h, body, _p(t)))

    def add_constraint_fix_orientation(self, body, target_rpy):
        t = _f(target_rpy)
        self._ck(self.lib.phys_add_constraint_fix_orientation(self.h, body, _p(t)))

    def clear_constraints(self):
        self._ck(self.lib.phys_clear_constraints(self.h))

    def apply_force_centre_of_gravity(self, body, force):
        f = _f(force)
        self._ck(self.lib.phys_apply_force_centre_of_gravity(self.h, body, _p(f)))

    def apply_force_at_position(self, body, force, point):
        f, p = _f(force), _f(point)
        self._ck(self.lib.phys_apply_force_at_position(self.h, body, _p(f), _p(p)))

    def apply_force_at_offset(self, body, force, offset):
        f, o = _f(force), _f(offset)
        self._ck(self.lib.phys_apply_force_at_offset(self.h, body, _p(f), _p(o)))

    def set_forces(self, force=None, torque=None):
        f, t = _f(force), _f(torque)
        self._ck(self.lib.phys_set_forces(self.h, _p(f), _p(t)))

    # ---- stepping
    def apply_gravity(self):
        self._ck(self.lib.phys_apply_gravity(self.h))

    def step(self, dt_nanos):
        self._ck(self.lib.phys_step(self.h, dt_nanos))

    def update(self, dt_nanos):
        self._ck(self.lib.phys_update(self.h, dt_nanos))

    def update_n(self, dt_nanos, n):
        self._ck(self.lib.phys_update_n(self.h, dt_nanos, n))

    def sync(self):
        self._ck(self.lib.phys_sync(self.h))

    # ---- read-back
    def get_transforms(self):
        pos = np.empty((self.n, 3), np.float32)
        rot = np.empty((self.n, 4), np.float32)
        self._ck(self.lib.phys_get_transforms(self.h, _p(pos), _p(rot)))
        return pos, rot

    def get_velocities(self):
        lin = np.empty((self.n, 3), np.float32)
        ang = np.empty((self.n, 3), np.float32)
        self._ck(self.lib.phys_get_velocities(self.h, _p(lin), _p(ang)))
        return lin, ang

    def get_forces(self):
        f = np.empty((self.n, 3), np.float32)
        t = np.empty((self.n, 3), np.float32)
        self._ck(self.lib.phys_get_forces(self.h, _p(f), _p(t)))
        return f, t

    def get_instance_matrices(self):
        m = np.empty((self.n, 16), np.float32)
        self._ck(self.lib.phys_get_instance_matrices(self.h, _p(m)))
        return m

    def get_lambda(self):
        n = C.c_uint64()
        self._ck(self.lib.phys_get_lambda(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        if n.value:
            self._ck(self.lib.phys_get_lambda(self.h, _p(out), n.value, C.byref(n)))
        return out

    def get_stats(self):
        s = PhysStats()
        self._ck(self.lib.phys_get_stats(self.h, C.byref(s)))
        return s

    def broadphase(self):
        n = C.c_uint64()
        self._ck(self.lib.phys_broadphase(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.uint32)
        if n.value:
            self._ck(self.lib.phys_broadphase(self.h, _p(out, u32p), n.value, C.byref(n)))
        return out

    def get_aabbs(self):
        out = np.empty((self.n, 6), np.float32)
        self._ck(self.lib.phys_get_aabbs(self.h, _p(out)))
        return out

    def get_manifolds(self):
        n = C.c_uint64()
        self._ck(self.lib.phys_get_manifolds(self.h, None, None, None, None, 0, C.byref(n)))
        m = n.value
        ids = np.empty((m, 2), np.uint32)
        counts = np.empty(m, np.uint32)
        normals = np.empty((m, 3), np.float32)
        points = np.empty((m, 4, 4), np.float32)
        if m:
            self._ck(self.lib.phys_get_manifolds(self.h, _p(ids, u32p), _p(counts, u32p), _p(normals), _p(points), m,
                                                 C.byref(n)))
        return ids, counts, normals, points

    def get_color_counts(self):
        out = np.zeros(64, np.uint32)
        self._ck(self.lib.phys_get_color_counts(self.h, _p(out, u32p)))
        return out

    def profile_enable(self, on=True):
        self._ck(self.lib.phys_profile_enable(self.h, int(on)))

    def profile_get(self):
        """{stage: (device ms summed, launches)} and the number of profiled updates."""
        p = PhysProfile()
        self._ck(self.lib.phys_profile_get(self.h, C.byref(p)))
        return {STAGE_NAMES[k]: (p.ms[k], p.launches[k]) for k in range(len(STAGE_NAMES)) if p.launches[k]}, p.steps

    def device_view(self):
        v = PhysDeviceView()
        self._ck(self.lib.phys_get_device_view(self.h, C.byref(v)))
        return v

    # ---- sharded broad-phase
    def set_global_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self._ck(self.lib.phys_set_global_ids(self.h, _p(ids, u32p)))

    def halo_pack(self, x_lo, x_hi, reach, dev_ptr, cap, wait=True):
        """wait=False: enqueue only (no host synchronisation), returns None."""
        n = C.c_uint64()
        self._ck(self.lib.phys_halo_pack(self.h, x_lo, x_hi, reach, C.c_void_p(dev_ptr), cap, C.byref(n) if wait else None))
        return n.value if wait else None

    def halo_pairs(self, dev_ptr, n_remote, skip_first=0, skip_count=0, wait=True):
        n = C.c_uint64()
        self._ck(self.lib.phys_halo_pairs(self.h, C.c_void_p(dev_ptr), n_remote, skip_first, skip_count,
                                          C.byref(n) if wait else None))
        return n.value if wait else None

    def get_cross_pairs(self):
        n = C.c_uint64()
        self._ck(self.lib.phys_get_cross_pairs(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.uint32)
        if n.value:
            self._ck(self.lib.phys_get_cross_pairs(self.h, _p(out, u32p), n.value, C.byref(n)))
        return out

    # ---- sharded worlds with ghost bodies (SURVEY §8 rows E + N4)
    def set_slab(self, x_lo, x_hi, reach):
        self._ck(self.lib.phys_set_slab(self.h, x_lo, x_hi, reach))

    def halo_pack_bodies(self, dev_ptr, cap):
        self._ck(self.lib.phys_halo_pack_bodies(self.h, C.c_void_p(dev_ptr), cap))

    def halo_pack_bodies_face(self, dev_ptr, cap, face):
        """face < 0: bodies within reach of the low slab face only, > 0: of the high face, 0: both."""
        self._ck(self.lib.phys_halo_pack_bodies_face(self.h, C.c_void_p(dev_ptr), cap, int(face)))

    def halo_unpack_ghosts(self, dev_ptr, n_records, skip_first=0, skip_count=0):
        self._ck(self.lib.phys_halo_unpack_ghosts(self.h, C.c_void_p(dev_ptr), n_records, skip_first, skip_count))

    def get_global_ids(self):
        out = np.empty(self.n + int(self.cfg.max_ghosts), np.uint32)
        self._ck(self.lib.phys_get_global_ids(self.h, _p(out, u32p)))
        return out

    def halo_exchange(self, comm):
        """pack -> RCCL all-gather -> unpack (ghost worlds: call BEFORE update) / cross pairs (AABB mode: AFTER)."""
        self._ck(self.lib.phys_halo_exchange(self.h, comm.h))


def block_spmv(nrows, ncols, blocks, vec, transpose=False, device=0):
    """SparseMatrix::multiply_vector / tr_multiply_vector (sparse_matrix.rs:25-50) on the device: `blocks` is the
    add_block list as (row, column, 2-D array) triples; returns M v (or M^T v) as float32."""
    lib = _abi.load_library()
    desc = np.array([[i, j, np.shape(b)[0], np.shape(b)[1]] for i, j, b in blocks], np.uint64).reshape(-1)
    data = (np.concatenate([np.asarray(b, np.float32).reshape(-1) for _, _, b in blocks]).astype(np.float32)
            if blocks else np.zeros(0, np.float32))
    vec = _f(vec).reshape(-1)
    out = np.zeros(ncols if transpose else nrows, np.float32)
    rc = lib.phys_block_spmv(device, nrows, ncols, len(blocks), _p(desc, _abi.u64p), _p(data), _p(vec), vec.size,
                             int(bool(transpose)), _p(out))
    if rc != 0:
        raise PhysError(rc, lib.phys_last_error().decode())
    return out


class Comm:
    """One rank of an RCCL communicator behind the C ABI (phys_comm_*): the id travels by whatever the host has."""

    @staticmethod
    def unique_id():
        lib = _abi.load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.phys_comm_unique_id(buf)
        if rc != 0:
            raise PhysError(rc, lib.phys_last_error().decode())
        return bytes(buf)

    def __init__(self, world, unique_id, rank, n_ranks, capacity, neighbours=False):
        """neighbours: the ranks are x-slabs ordered by rank, none thinner than the reach - exchange with ranks r - 1 and
        r + 1 only (phys_comm_set_neighbours) instead of an all-gather of every rank's block."""
        self.lib = world.lib
        self.h = C.c_void_p()
        self.rank, self.n_ranks, self.capacity, self.neighbours = rank, n_ranks, capacity, bool(neighbours)
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = self.lib.phys_comm_create(world.h, buf, rank, n_ranks, capacity, C.byref(self.h))
        if rc != 0:
            raise PhysError(rc, self.lib.phys_last_error().decode())
        if neighbours:
            rc = self.lib.phys_comm_set_neighbours(self.h, 1)
            if rc != 0:
                raise PhysError(rc, self.lib.phys_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.phys_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
