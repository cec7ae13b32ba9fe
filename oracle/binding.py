"""ctypes binding of oracle/liboracle.so — the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. It
mirrors the method names of physics_amd.world.World so one harness can drive both with the same
seeded inputs."""
import ctypes as C
import os
import subprocess

import numpy as np

from physics_amd._abi import PhysConfig, PhysStats, default_config, f32p, u32p, u64p

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "liboracle.so")
TRIG_LIBM, TRIG_DET = 0, 1
_lib = None


def build():
    subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        "oracle_last_error": (C.c_char_p, []),
        "oracle_create": (C.c_int32, [C.POINTER(PhysConfig), C.c_int32, C.POINTER(vp)]),
        "oracle_destroy": (C.c_int32, [vp]),
        "oracle_set_bodies": (C.c_int32, [vp, C.c_uint64, f32p, f32p, f32p, f32p, f32p, f32p, u32p, f32p]),
        "oracle_add_constraint_fix_point": (C.c_int32, [vp, C.c_uint64, f32p]),
        "oracle_add_constraint_fix_orientation": (C.c_int32, [vp, C.c_uint64, f32p]),
        "oracle_clear_constraints": (C.c_int32, [vp]),
        "oracle_apply_force_centre_of_gravity": (C.c_int32, [vp, C.c_uint64, f32p]),
        "oracle_apply_force_at_position": (C.c_int32, [vp, C.c_uint64, f32p, f32p]),
        "oracle_apply_force_at_offset": (C.c_int32, [vp, C.c_uint64, f32p, f32p]),
        "oracle_apply_gravity": (C.c_int32, [vp]),
        "oracle_step": (C.c_int32, [vp, C.c_uint64]),
        "oracle_update": (C.c_int32, [vp, C.c_uint64]),
        "oracle_set_threads": (C.c_int32, [vp, C.c_int32]),
        "oracle_get_transforms": (C.c_int32, [vp, f32p, f32p]),
        "oracle_get_velocities": (C.c_int32, [vp, f32p, f32p]),
        "oracle_get_forces": (C.c_int32, [vp, f32p, f32p]),
        "oracle_get_instance_matrices": (C.c_int32, [vp, f32p]),
        "oracle_get_lambda": (C.c_int32, [vp, f32p, C.c_uint64, u64p]),
        "oracle_get_stats": (C.c_int32, [vp, C.POINTER(PhysStats)]),
        "oracle_broadphase": (C.c_int32, [vp, u32p, C.c_uint64, u64p]),
        "oracle_broadphase_grid": (C.c_int32, [vp, u32p, C.c_uint64, u64p]),
        "oracle_collide_now": (C.c_int32, [vp]),
        "oracle_get_aabbs": (C.c_int32, [vp, f32p]),
        "oracle_get_manifolds": (C.c_int32, [vp, u32p, u32p, f32p, f32p, C.c_uint64, u64p]),
        "oracle_get_colors": (C.c_int32, [vp, u32p, C.c_uint64]),
        "oracle_spmv": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint64, u64p, f32p, f32p, C.c_int32, f32p]),
        "oracle_dyn_dot": (C.c_float, [f32p, f32p, C.c_uint64]),
        "oracle_duration_as_secs_f32": (C.c_float, [C.c_uint64]),
        "oracle_quat_from_euler": (None, [C.c_float, C.c_float, C.c_float, C.c_int32, f32p]),
        "oracle_quat_euler_angles": (None, [f32p, C.c_int32, f32p]),
        "oracle_det_sincos": (None, [f32p, C.c_uint64, f32p, f32p]),
        "oracle_det_asin": (None, [f32p, C.c_uint64, f32p]),
        "oracle_det_atan2": (None, [f32p, f32p, C.c_uint64, f32p]),
        "oracle_solve_drivers_mismatch": (C.c_int32, [f32p, C.c_int32, C.c_int32, C.c_int32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=f32p):
    return None if a is None else a.ctypes.data_as(t)


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


class OracleWorld:
    """CPU oracle with the same method names as physics_amd.world.World."""

    def __init__(self, cfg=None, trig=TRIG_DET):
        self.lib = load()
        self.cfg = cfg if cfg is not None else default_config()
        self.h = C.c_void_p()
        self._ck(self.lib.oracle_create(C.byref(self.cfg), trig, C.byref(self.h)))
        self.n = 0

    def _ck(self, rc):
        if rc != 0:
            raise OracleError(rc, self.lib.oracle_last_error().decode())

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bodies(self, pos, rot=None, lin_vel=None, ang_vel=None, mass=None, inertia=None, shape_type=None,
                   half_extent=None):
        pos = _f(pos).reshape(-1, 3)
        n = pos.shape[0]
        arrs = [pos, _f(rot), _f(lin_vel), _f(ang_vel), _f(mass), _f(inertia)]
        st = None if shape_type is None else np.ascontiguousarray(shape_type, dtype=np.uint32)
        he = _f(half_extent)
        self._keep = arrs + [st, he]
        self._ck(self.lib.oracle_set_bodies(self.h, n, *[_p(a) for a in arrs], _p(st, u32p), _p(he)))
        self.n = n

    def add_constraint_fix_point(self, body, target):
        t = _f(target)
        self._ck(self.lib.oracle_add_constraint_fix_point(self.h, body, _p(t)))

    def add_constraint_fix_orientation(self, body, target_rpy):
        t = _f(target_rpy)
        self._ck(self.lib.oracle_add_constraint_fix_orientation(self.h, body, _p(t)))

    def clear_constraints(self):
        self._ck(self.lib.oracle_clear_constraints(self.h))

    def apply_force_centre_of_gravity(self, body, force):
        f = _f(force)
        self._ck(self.lib.oracle_apply_force_centre_of_gravity(self.h, body, _p(f)))

    def apply_force_at_position(self, body, force, point):
        f, p = _f(force), _f(point)
        self._ck(self.lib.oracle_apply_force_at_position(self.h, body, _p(f), _p(p)))

    def apply_force_at_offset(self, body, force, offset):
        f, o = _f(force), _f(offset)
        self._ck(self.lib.oracle_apply_force_at_offset(self.h, body, _p(f), _p(o)))

    def apply_gravity(self):
        self._ck(self.lib.oracle_apply_gravity(self.h))

    def step(self, dt_nanos):
        self._ck(self.lib.oracle_step(self.h, dt_nanos))

    def set_threads(self, threads):
        """OpenMP variant of the collision stages (same results for any thread count)."""
        self._ck(self.lib.oracle_set_threads(self.h, int(threads)))

    def update(self, dt_nanos):
        self._ck(self.lib.oracle_update(self.h, dt_nanos))

    def update_n(self, dt_nanos, n):
        for _ in range(n):
            self.update(dt_nanos)

    def sync(self):
        pass

    def get_transforms(self):
        pos = np.empty((self.n, 3), np.float32)
        rot = np.empty((self.n, 4), np.float32)
        self._ck(self.lib.oracle_get_transforms(self.h, _p(pos), _p(rot)))
        return pos, rot

    def get_velocities(self):
        lin = np.empty((self.n, 3), np.float32)
        ang = np.empty((self.n, 3), np.float32)
        self._ck(self.lib.oracle_get_velocities(self.h, _p(lin), _p(ang)))
        return lin, ang

    def get_forces(self):
        f = np.empty((self.n, 3), np.float32)
        t = np.empty((self.n, 3), np.float32)
        self._ck(self.lib.oracle_get_forces(self.h, _p(f), _p(t)))
        return f, t

    def get_instance_matrices(self):
        m = np.empty((self.n, 16), np.float32)
        self._ck(self.lib.oracle_get_instance_matrices(self.h, _p(m)))
        return m

    def get_lambda(self):
        n = C.c_uint64()
        self._ck(self.lib.oracle_get_lambda(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        if n.value:
            self._ck(self.lib.oracle_get_lambda(self.h, _p(out), n.value, C.byref(n)))
        return out

    def get_stats(self):
        s = PhysStats()
        self._ck(self.lib.oracle_get_stats(self.h, C.byref(s)))
        return s

    def broadphase(self):
        n = C.c_uint64()
        self._ck(self.lib.oracle_broadphase(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.uint32)
        if n.value:
            self._ck(self.lib.oracle_broadphase(self.h, _p(out, u32p), n.value, C.byref(n)))
        return out

    def broadphase_grid(self):
        n = C.c_uint64()
        self._ck(self.lib.oracle_broadphase_grid(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.uint32)
        if n.value:
            self._ck(self.lib.oracle_broadphase_grid(self.h, _p(out, u32p), n.value, C.byref(n)))
        return out

    def collide_now(self):
        self._ck(self.lib.oracle_collide_now(self.h))

    def get_aabbs(self):
        out = np.empty((self.n, 6), np.float32)
        self._ck(self.lib.oracle_get_aabbs(self.h, _p(out)))
        return out

    def get_manifolds(self):
        n = C.c_uint64()
        self._ck(self.lib.oracle_get_manifolds(self.h, None, None, None, None, 0, C.byref(n)))
        m = n.value
        ids = np.empty((m, 2), np.uint32)
        counts = np.empty(m, np.uint32)
        normals = np.empty((m, 3), np.float32)
        points = np.empty((m, 4, 4), np.float32)
        if m:
            self._ck(self.lib.oracle_get_manifolds(self.h, _p(ids, u32p), _p(counts, u32p), _p(normals), _p(points), m,
                                                   C.byref(n)))
        return ids, counts, normals, points

    def get_colors(self):
        m = self.get_stats().n_manifolds
        out = np.empty(m, np.uint32)
        if m:
            self._ck(self.lib.oracle_get_colors(self.h, _p(out, u32p), m))
        return out


# ---- free functions used by the golden-vector tests
def spmv(nrows, ncols, blocks, vec, transpose=False):
    """blocks: list of (i, j, 2-D row-major array)."""
    lib = load()
    desc = np.array([[i, j, b.shape[0], b.shape[1]] for i, j, b in blocks], np.uint64).reshape(-1)
    data = np.concatenate([np.asarray(b, np.float32).reshape(-1) for _, _, b in blocks]).astype(np.float32)
    vec = _f(vec)
    out = np.empty(ncols if transpose else nrows, np.float32)
    lib.oracle_spmv(nrows, ncols, len(blocks), _p(desc, u64p), _p(data), _p(vec), int(transpose), _p(out))
    return out


def dyn_dot(a, b):
    a, b = _f(a), _f(b)
    return float(load().oracle_dyn_dot(_p(a), _p(b), a.size))


def duration_as_secs_f32(nanos):
    return np.float32(load().oracle_duration_as_secs_f32(nanos))


def quat_from_euler(r, p, y, trig=TRIG_LIBM):
    out = np.empty(4, np.float32)
    load().oracle_quat_from_euler(r, p, y, trig, _p(out))
    return out


def quat_euler_angles(q, trig=TRIG_LIBM):
    q = _f(q)
    out = np.empty(3, np.float32)
    load().oracle_quat_euler_angles(_p(q), trig, _p(out))
    return out


def det_sincos(x):
    x = _f(x)
    s = np.empty_like(x)
    c = np.empty_like(x)
    load().oracle_det_sincos(_p(x), x.size, _p(s), _p(c))
    return s, c


def det_asin(x):
    x = _f(x)
    out = np.empty_like(x)
    load().oracle_det_asin(_p(x), x.size, _p(out))
    return out


def solve_drivers_mismatch(flat, count, has_b, iters=8):
    """Bits that differ between contact_solve.h's two drivers (Jacobians beforehand / on the way) on one manifold."""
    flat = _f(flat)
    assert flat.size == 67
    return int(load().oracle_solve_drivers_mismatch(_p(flat), int(count), int(has_b), int(iters)))


def det_atan2(y, x):
    y, x = _f(y), _f(x)
    out = np.empty_like(x)
    load().oracle_det_atan2(_p(y), _p(x), x.size, _p(out))
    return out


def libm_sincos(x):
    """glibc sinf / cosf through the oracle library (numpy's float32 sin is NOT glibc's)."""
    lib = load()
    lib.oracle_libm_sincos.restype = None
    lib.oracle_libm_sincos.argtypes = [f32p, C.c_uint64, f32p, f32p]
    x = _f(x)
    s = np.empty_like(x)
    c = np.empty_like(x)
    lib.oracle_libm_sincos(_p(x), x.size, _p(s), _p(c))
    return s, c


def libm_asin(x):
    lib = load()
    lib.oracle_libm_asin.restype = None
    lib.oracle_libm_asin.argtypes = [f32p, C.c_uint64, f32p]
    x = _f(x)
    out = np.empty_like(x)
    lib.oracle_libm_asin(_p(x), x.size, _p(out))
    return out


def libm_atan2(y, x):
    lib = load()
    lib.oracle_libm_atan2.restype = None
    lib.oracle_libm_atan2.argtypes = [f32p, f32p, C.c_uint64, f32p]
    y, x = _f(y), _f(x)
    out = np.empty_like(x)
    lib.oracle_libm_atan2(_p(y), _p(x), x.size, _p(out))
    return out
