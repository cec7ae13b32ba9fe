//! Raw bindings of include/physics_hip.h (UNCOMPILED SOURCE: no Rust toolchain on the build machines).
//! One `extern "C"` item per symbol the header declares, same order, same types (tests/test_abi.py checks the list
//! against the header: every phys_* symbol of physics_hip.h appears here exactly once).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const PHYS_ABI_VERSION: u32 = 2;
pub const PHYS_OK: i32 = 0;
pub const PHYS_ERR_SINGULAR_INERTIA: i32 = -4;
pub const PHYS_ERR_NO_BODIES: i32 = -8;
pub const PHYS_SHAPE_NONE: u32 = 0;
pub const PHYS_SHAPE_SPHERE: u32 = 1;
pub const PHYS_SHAPE_BOX: u32 = 2;
pub const PHYS_FLAG_COLLISIONS: u32 = 0x1;
pub const PHYS_FLAG_GROUND_PLANE: u32 = 0x2;
pub const PHYS_FLAG_EXACT_ROTATION: u32 = 0x4;
pub const PHYS_FLAG_BROADPHASE_ONLY: u32 = 0x8;
pub const PHYS_FLAG_SOLVER_PER_COLOR: u32 = 0x10;
pub const PHYS_FLAG_SHARED_GPU: u32 = 0x20;
pub const PHYS_FLAG_SOLVER_CLUSTER: u32 = 0x40;
pub const PHYS_FLAG_EXCLUSIVE_GPU: u32 = 0x80;
pub const PHYS_FLAG_NO_WARM_START: u32 = 0x100;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct phys_config {
    pub abi_version: u32,
    pub device: i32,
    pub flags: u32,
    pub gravity_force: [f32; 3],
    pub gravity_offset: [f32; 3],
    pub cg_max_iterations: u32,
    pub cg_max_error: f32,
    pub cg_min_error: f32,
    pub solver_iterations: u32,
    pub baumgarte: f32,
    pub slop: f32,
    pub friction: f32,
    pub contact_margin: f32,
    pub ground_height: f32,
    pub max_bias: f32,
    pub max_pairs: u64,
    pub max_manifolds: u64,
    pub max_ghosts: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct phys_stats {
    pub n_bodies: u64,
    pub n_pairs: u64,
    pub n_manifolds: u64,
    pub n_contacts: u64,
    pub n_colors: u32,
    pub color_rounds: u32,
    pub cg_iterations: u32,
    pub cg_converged: i32,
    pub steps: u64,
    pub overflow: u32,
    pub n_ground_manifolds: u32,
    pub max_extent: f32,
    pub n_halo_records: u32,
    pub n_cross_pairs: u64,
    pub n_ghosts: u32,
    pub n_new_manifolds: u32,
}

#[repr(C)]
pub struct phys_world {
    _private: [u8; 0],
}

#[repr(C)]
pub struct phys_comm {
    _private: [u8; 0],
}

pub const PHYS_STAGE_COUNT: usize = 14;
pub const PHYS_COMM_ID_BYTES: usize = 128;
pub const PHYS_HALO_BODY_RECORD_BYTES: usize = 96;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct phys_profile {
    pub ms: [f64; PHYS_STAGE_COUNT],
    pub launches: [u64; PHYS_STAGE_COUNT],
    pub steps: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct phys_device_view {
    pub n: u64,
    pub pos: *mut f32,
    pub rot: *mut f32,
    pub lin_vel: *mut f32,
    pub ang_vel: *mut f32,
    pub aabb: *mut f32,
    pub stream: *mut c_void,
    pub vel_stride: u64,
}

extern "C" {
    pub fn phys_config_default(cfg: *mut phys_config);
    pub fn phys_last_error() -> *const c_char;
    pub fn phys_abi_version() -> u32;
    pub fn phys_create(cfg: *const phys_config, out: *mut *mut phys_world) -> i32;
    pub fn phys_destroy(w: *mut phys_world) -> i32;
    pub fn phys_set_bodies(w: *mut phys_world, n: u64, pos: *const f32, rot_ijkw: *const f32, lin_vel: *const f32,
                           ang_vel: *const f32, mass: *const f32, inertia: *const f32, shape_type: *const u32,
                           half_extent: *const f32) -> i32;
    pub fn phys_add_constraint_fix_point(w: *mut phys_world, body: u64, target: *const f32) -> i32;
    pub fn phys_add_constraint_fix_orientation(w: *mut phys_world, body: u64, target_rpy: *const f32) -> i32;
    pub fn phys_clear_constraints(w: *mut phys_world) -> i32;
    pub fn phys_apply_force_centre_of_gravity(w: *mut phys_world, body: u64, force: *const f32) -> i32;
    pub fn phys_apply_force_at_position(w: *mut phys_world, body: u64, force: *const f32, point: *const f32) -> i32;
    pub fn phys_apply_force_at_offset(w: *mut phys_world, body: u64, force: *const f32, offset: *const f32) -> i32;
    pub fn phys_set_forces(w: *mut phys_world, force: *const f32, torque: *const f32) -> i32;
    pub fn phys_update(w: *mut phys_world, dt_nanos: u64) -> i32;
    pub fn phys_apply_gravity(w: *mut phys_world) -> i32;
    pub fn phys_step(w: *mut phys_world, dt_nanos: u64) -> i32;
    pub fn phys_update_n(w: *mut phys_world, dt_nanos: u64, n: u32) -> i32;
    pub fn phys_sync(w: *mut phys_world) -> i32;
    pub fn phys_get_transforms(w: *mut phys_world, pos_out: *mut f32, rot_ijkw_out: *mut f32) -> i32;
    pub fn phys_get_velocities(w: *mut phys_world, lin_out: *mut f32, ang_out: *mut f32) -> i32;
    pub fn phys_get_forces(w: *mut phys_world, force_out: *mut f32, torque_out: *mut f32) -> i32;
    pub fn phys_get_instance_matrices(w: *mut phys_world, out: *mut f32) -> i32;
    pub fn phys_get_lambda(w: *mut phys_world, lambda_out: *mut f32, cap: u64, n_rows: *mut u64) -> i32;
    pub fn phys_block_spmv(device: i32, nrows: u64, ncols: u64, nblocks: u64, block_desc: *const u64, data: *const f32,
                           vec: *const f32, vec_len: u64, transpose: i32, out: *mut f32) -> i32;
    pub fn phys_broadphase(w: *mut phys_world, pairs_out: *mut u32, cap: u64, n_pairs: *mut u64) -> i32;
    pub fn phys_get_aabbs(w: *mut phys_world, out: *mut f32) -> i32;
    pub fn phys_get_manifolds(w: *mut phys_world, ids_out: *mut u32, counts_out: *mut u32, normals_out: *mut f32,
                              points_out: *mut f32, cap: u64, n_manifolds: *mut u64) -> i32;
    pub fn phys_get_stats(w: *mut phys_world, out: *mut phys_stats) -> i32;
    pub fn phys_get_color_counts(w: *mut phys_world, counts_out: *mut u32) -> i32;
    pub fn phys_profile_enable(w: *mut phys_world, on: i32) -> i32;
    pub fn phys_profile_get(w: *mut phys_world, out: *mut phys_profile) -> i32;
    pub fn phys_get_device_view(w: *mut phys_world, out: *mut phys_device_view) -> i32;
    pub fn phys_set_global_ids(w: *mut phys_world, global_ids: *const u32) -> i32;
    pub fn phys_halo_pack(w: *mut phys_world, x_lo: f32, x_hi: f32, reach: f32, dev_records_out: *mut c_void, cap: u64,
                          n_records: *mut u64) -> i32;
    pub fn phys_halo_pairs(w: *mut phys_world, dev_remote_records: *const c_void, n_remote: u64, skip_first: u64,
                           skip_count: u64, n_cross_pairs: *mut u64) -> i32;
    pub fn phys_get_cross_pairs(w: *mut phys_world, pairs_out: *mut u32, cap: u64, n_pairs: *mut u64) -> i32;
    pub fn phys_set_slab(w: *mut phys_world, x_lo: f32, x_hi: f32, reach: f32) -> i32;
    pub fn phys_halo_pack_bodies(w: *mut phys_world, dev_records_out: *mut c_void, cap: u64) -> i32;
    pub fn phys_halo_pack_bodies_face(w: *mut phys_world, dev_records_out: *mut c_void, cap: u64, face: i32) -> i32;
    pub fn phys_halo_unpack_ghosts(w: *mut phys_world, dev_records: *const c_void, n_records: u64, skip_first: u64,
                                   skip_count: u64) -> i32;
    pub fn phys_get_global_ids(w: *mut phys_world, out: *mut u32) -> i32;
    pub fn phys_comm_unique_id(id_out: *mut u8) -> i32;
    pub fn phys_comm_create(w: *mut phys_world, id: *const u8, rank: i32, n_ranks: i32, capacity: u64,
                            out: *mut *mut phys_comm) -> i32;
    pub fn phys_comm_create_local(worlds: *mut *mut phys_world, n: i32, capacity: u64, comms_out: *mut *mut phys_comm) -> i32;
    pub fn phys_comm_destroy(c: *mut phys_comm) -> i32;
    pub fn phys_comm_set_neighbours(c: *mut phys_comm, enable: i32) -> i32;
    pub fn phys_halo_exchange(w: *mut phys_world, c: *mut phys_comm) -> i32;
    pub fn phys_halo_exchange_all(worlds: *mut *mut phys_world, comms: *mut *mut phys_comm, n: i32) -> i32;
    pub fn phys_slab_histogram(pos: *const f32, n: u64, x_min: f32, x_max: f32, bins: u32, hist: *mut u64) -> i32;
    pub fn phys_slab_cuts(hist: *const u64, bins: u32, x_min: f32, x_max: f32, n_ranks: i32, cuts_out: *mut f32) -> i32;
    pub fn phys_slab_owners(pos: *const f32, n: u64, cuts: *const f32, n_ranks: i32, owner_out: *mut i32) -> i32;
}
