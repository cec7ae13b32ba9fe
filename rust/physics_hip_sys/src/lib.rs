//! Raw bindings of include/physics_hip.h (UNCOMPILED SOURCE: no Rust toolchain on the build machines).
//! One `extern "C"` item per symbol the header declares, same order, same types.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const PHYS_ABI_VERSION: u32 = 1;
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
}

#[repr(C)]
pub struct phys_world {
    _private: [u8; 0],
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
    pub fn phys_broadphase(w: *mut phys_world, pairs_out: *mut u32, cap: u64, n_pairs: *mut u64) -> i32;
    pub fn phys_get_aabbs(w: *mut phys_world, out: *mut f32) -> i32;
    pub fn phys_get_manifolds(w: *mut phys_world, ids_out: *mut u32, counts_out: *mut u32, normals_out: *mut f32,
                              points_out: *mut f32, cap: u64, n_manifolds: *mut u64) -> i32;
    pub fn phys_get_stats(w: *mut phys_world, out: *mut phys_stats) -> i32;
    pub fn phys_get_color_counts(w: *mut phys_world, counts_out: *mut u32) -> i32;
    pub fn phys_set_global_ids(w: *mut phys_world, global_ids: *const u32) -> i32;
    pub fn phys_halo_pack(w: *mut phys_world, x_lo: f32, x_hi: f32, reach: f32, dev_records_out: *mut c_void, cap: u64,
                          n_records: *mut u64) -> i32;
    pub fn phys_halo_pairs(w: *mut phys_world, dev_remote_records: *const c_void, n_remote: u64, skip_first: u64,
                           skip_count: u64, n_cross_pairs: *mut u64) -> i32;
    pub fn phys_get_cross_pairs(w: *mut phys_world, pairs_out: *mut u32, cap: u64, n_pairs: *mut u64) -> i32;
}
