"""ctypes view of include/physics_hip.h (structs, constants, prototypes) and the loader of
libphysics_hip.so. There is no CPU fallback: if the HIP library is missing or no gfx950 device is
usable, creation fails loudly."""
import ctypes as C
import os

PHYS_ABI_VERSION = 2

PHYS_OK = 0
PHYS_ERR_INVALID_ARG = -1
PHYS_ERR_NO_DEVICE = -2
PHYS_ERR_HIP = -3
PHYS_ERR_SINGULAR_INERTIA = -4
PHYS_ERR_CAPACITY = -5
PHYS_ERR_OUT_OF_RANGE = -6
PHYS_ERR_UNSUPPORTED = -7
PHYS_ERR_NO_BODIES = -8

SHAPE_NONE, SHAPE_SPHERE, SHAPE_BOX = 0, 1, 2
FLAG_COLLISIONS, FLAG_GROUND_PLANE, FLAG_EXACT_ROTATION, FLAG_BROADPHASE_ONLY = 1, 2, 4, 8
FLAG_SOLVER_PER_COLOR = 16
FLAG_SHARED_GPU = 32
FLAG_SOLVER_CLUSTER = 64
FLAG_EXCLUSIVE_GPU = 128
FLAG_NO_WARM_START = 256
GROUND_ID = 0xFFFFFFFF

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class PhysConfig(C.Structure):
    """struct phys_config (include/physics_hip.h)."""
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("gravity_force", C.c_float * 3),
        ("gravity_offset", C.c_float * 3),
        ("cg_max_iterations", C.c_uint32),
        ("cg_max_error", C.c_float),
        ("cg_min_error", C.c_float),
        ("solver_iterations", C.c_uint32),
        ("baumgarte", C.c_float),
        ("slop", C.c_float),
        ("friction", C.c_float),
        ("contact_margin", C.c_float),
        ("ground_height", C.c_float),
        ("max_bias", C.c_float),
        ("max_pairs", C.c_uint64),
        ("max_manifolds", C.c_uint64),
        ("max_ghosts", C.c_uint64),
    ]


class PhysStats(C.Structure):
    _fields_ = [
        ("n_bodies", C.c_uint64),
        ("n_pairs", C.c_uint64),
        ("n_manifolds", C.c_uint64),
        ("n_contacts", C.c_uint64),
        ("n_colors", C.c_uint32),
        ("color_rounds", C.c_uint32),
        ("cg_iterations", C.c_uint32),
        ("cg_converged", C.c_int32),
        ("steps", C.c_uint64),
        ("overflow", C.c_uint32),
        ("n_ground_manifolds", C.c_uint32),
        ("max_extent", C.c_float),
        ("n_halo_records", C.c_uint32),
        ("n_cross_pairs", C.c_uint64),
        ("n_ghosts", C.c_uint32),
        ("n_new_manifolds", C.c_uint32),
    ]


STAGE_COUNT = 14
STAGE_NAMES = ["step_full", "velocity_aabb", "grid", "pairs", "narrow", "color", "rows", "solve", "position",
               "constraints", "misc", "solve_tail", "solve_flow", "solve_cluster"]


class PhysProfile(C.Structure):
    _fields_ = [("ms", C.c_double * STAGE_COUNT), ("launches", C.c_uint64 * STAGE_COUNT), ("steps", C.c_uint64)]


class PhysDeviceView(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("pos", C.c_void_p),
        ("rot", C.c_void_p),
        ("lin_vel", C.c_void_p),
        ("ang_vel", C.c_void_p),
        ("aabb", C.c_void_p),
        ("stream", C.c_void_p),
        ("vel_stride", C.c_uint64),
    ]


def default_config(**overrides):
    """Reference defaults, identical to phys_config_default() (checked by tests/test_abi.py)."""
    cfg = PhysConfig()
    cfg.abi_version = PHYS_ABI_VERSION
    cfg.device = 0
    cfg.flags = 0
    cfg.gravity_force[:] = (0.0, -9.81, 0.0)   # physics.rs:90
    cfg.gravity_offset[:] = (0.0, 0.0, 1.5)    # physics.rs:91
    cfg.cg_max_iterations = 1000               # sle_solver.rs:5
    cfg.cg_max_error = 1e-2                    # sle_solver.rs:6
    cfg.cg_min_error = 1e-3                    # sle_solver.rs:7
    cfg.solver_iterations = 8
    cfg.baumgarte = 0.2
    cfg.slop = 0.01
    cfg.friction = 0.5
    cfg.contact_margin = 0.02
    cfg.ground_height = 0.0
    cfg.max_bias = 3.0
    cfg.max_pairs = 0
    cfg.max_manifolds = 0
    cfg.max_ghosts = 0
    for k, v in overrides.items():
        if k in ("gravity_force", "gravity_offset"):
            getattr(cfg, k)[:] = tuple(v)
        else:
            setattr(cfg, k, v)
    return cfg


# name -> (restype, argtypes); every symbol include/physics_hip.h declares
PROTOTYPES = {
    "phys_config_default": (None, [C.POINTER(PhysConfig)]),
    "phys_last_error": (C.c_char_p, []),
    "phys_abi_version": (C.c_uint32, []),
    "phys_create": (C.c_int32, [C.POINTER(PhysConfig), C.POINTER(C.c_void_p)]),
    "phys_destroy": (C.c_int32, [C.c_void_p]),
    "phys_set_bodies": (C.c_int32, [C.c_void_p, C.c_uint64, f32p, f32p, f32p, f32p, f32p, f32p, u32p, f32p]),
    "phys_add_constraint_fix_point": (C.c_int32, [C.c_void_p, C.c_uint64, f32p]),
    "phys_add_constraint_fix_orientation": (C.c_int32, [C.c_void_p, C.c_uint64, f32p]),
    "phys_clear_constraints": (C.c_int32, [C.c_void_p]),
    "phys_apply_force_centre_of_gravity": (C.c_int32, [C.c_void_p, C.c_uint64, f32p]),
    "phys_apply_force_at_position": (C.c_int32, [C.c_void_p, C.c_uint64, f32p, f32p]),
    "phys_apply_force_at_offset": (C.c_int32, [C.c_void_p, C.c_uint64, f32p, f32p]),
    "phys_set_forces": (C.c_int32, [C.c_void_p, f32p, f32p]),
    "phys_update": (C.c_int32, [C.c_void_p, C.c_uint64]),
    "phys_apply_gravity": (C.c_int32, [C.c_void_p]),
    "phys_step": (C.c_int32, [C.c_void_p, C.c_uint64]),
    "phys_update_n": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "phys_sync": (C.c_int32, [C.c_void_p]),
    "phys_get_transforms": (C.c_int32, [C.c_void_p, f32p, f32p]),
    "phys_get_velocities": (C.c_int32, [C.c_void_p, f32p, f32p]),
    "phys_get_forces": (C.c_int32, [C.c_void_p, f32p, f32p]),
    "phys_get_instance_matrices": (C.c_int32, [C.c_void_p, f32p]),
    "phys_get_lambda": (C.c_int32, [C.c_void_p, f32p, C.c_uint64, u64p]),
    "phys_block_spmv": (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, u64p, f32p, f32p, C.c_uint64, C.c_int32, f32p]),
    "phys_broadphase": (C.c_int32, [C.c_void_p, u32p, C.c_uint64, u64p]),
    "phys_get_aabbs": (C.c_int32, [C.c_void_p, f32p]),
    "phys_get_manifolds": (C.c_int32, [C.c_void_p, u32p, u32p, f32p, f32p, C.c_uint64, u64p]),
    "phys_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(PhysStats)]),
    "phys_get_color_counts": (C.c_int32, [C.c_void_p, u32p]),
    "phys_profile_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "phys_profile_get": (C.c_int32, [C.c_void_p, C.POINTER(PhysProfile)]),
    "phys_get_device_view": (C.c_int32, [C.c_void_p, C.POINTER(PhysDeviceView)]),
    "phys_set_global_ids": (C.c_int32, [C.c_void_p, u32p]),
    "phys_halo_pack": (C.c_int32, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_uint64, u64p]),
    "phys_halo_pairs": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
    "phys_get_cross_pairs": (C.c_int32, [C.c_void_p, u32p, C.c_uint64, u64p]),
    "phys_set_slab": (C.c_int32, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "phys_halo_pack_bodies": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "phys_halo_pack_bodies_face": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "phys_halo_unpack_ghosts": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "phys_get_global_ids": (C.c_int32, [C.c_void_p, u32p]),
    "phys_comm_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "phys_comm_create": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "phys_comm_create_local": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "phys_comm_destroy": (C.c_int32, [C.c_void_p]),
    "phys_comm_set_neighbours": (C.c_int32, [C.c_void_p, C.c_int32]),
    "phys_halo_exchange": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "phys_halo_exchange_all": (C.c_int32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32]),
    "phys_slab_histogram": (C.c_int32, [f32p, C.c_uint64, C.c_float, C.c_float, C.c_uint32, u64p]),
    "phys_slab_cuts": (C.c_int32, [u64p, C.c_uint32, C.c_float, C.c_float, C.c_int32, f32p]),
    "phys_slab_owners": (C.c_int32, [f32p, C.c_uint64, f32p, C.c_int32, C.POINTER(C.c_int32)]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libphysics_hip.so")
_lib = None


class PhysicsHipMissing(RuntimeError):
    pass


def rocm_runtime_mapped():
    """Paths of the HIP runtime(s) mapped into this process (one entry in a healthy process)."""
    try:
        return sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line})
    except OSError:
        return []


def torch_bundled_rocm_dir():
    """Directory of the ROCm runtime a PyTorch wheel ships inside itself, found WITHOUT importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def share_rocm_runtime_with_torch():
    """One ROCm runtime per process. The PyTorch wheel of this image ships its OWN libamdhip64.so / libhsa-runtime64.so
    (ROCm 7.0.2, under torch/lib, SONAME libamdhip64.so.7) next to the system's /opt/rocm (7.2.0, same SONAME).
    libphysics_hip.so asks for the SONAME, libtorch_hip.so for the file in its own directory, so the order of loading
    decided what a process got: torch first -> ONE runtime (torch's, found again by SONAME); the library first -> /opt/rocm's
    for the library and then a SECOND runtime for torch, whose initialisation fails with hipErrorNoDevice ("no ROCm-capable
    device is detected", round 1's unexplained failure: two HSA runtimes cannot both own the process's KFD queue state).
    So when a PyTorch with a bundled runtime is installed and no HIP runtime is mapped yet, that bundled runtime is
    loaded first (RTLD_GLOBAL, by path - torch itself need not be imported): library and torch then share it whichever
    comes first. Callers without PyTorch (the Rust / C++ hosts) get the system runtime, alone in its process.
    PHYS_ROCM_RUNTIME=system skips this (a process that will never import torch)."""
    if os.environ.get("PHYS_ROCM_RUNTIME") == "system" or rocm_runtime_mapped():
        return None
    d = torch_bundled_rocm_dir()
    if d is None:
        return None
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    # the RCCL of this process must be the one built against that runtime (comm.hip loads it on first use)
    rccl = os.path.join(d, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("PHYS_RCCL_PATH", rccl)
    return d


def load_library():
    """Load libphysics_hip.so (built in-tree by __graft_entry__.build()). Raises if absent: the
    product path has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PhysicsHipMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). physics_amd has no CPU fallback.")
    share_rocm_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
