// physics_backend.rs — the patch a maintainer of martingoe/physics adds as `src/physics/hip_backend.rs`
// (UNCOMPILED SOURCE: no Rust toolchain on the build machines). It replaces the bodies of
// PhysicsState::update / apply_gravity / step (src/physics.rs:41-55, 87-99) with calls into
// libphysics_hip.so and leaves get_render_data (physics.rs:58-85) and all of src/rendering untouched.
//
// It has to live INSIDE the physics module tree because RigidBody is repr(Rust) and `inertia_tensor` is
// private to rigid_body.rs (:15), `mass/force/torque` are pub(crate) (:7,12,13): the gather below reads
// them through one accessor added to rigid_body.rs:
//     impl RigidBody { pub(crate) fn inertia_row_major(&self) -> [f32; 9] { /* transpose of as_slice() */ } }
use crate::physics::constraints::Constraints;
use crate::physics::PhysicsState;
use physics_hip_sys as ffi;
use std::ffi::CStr;
use std::time::Duration;

pub struct HipBackend {
    world: *mut ffi::phys_world,
    uploaded_bodies: usize,
    uploaded_constraints: usize,
}

fn check(rc: i32) {
    if rc != ffi::PHYS_OK {
        // the reference panics in the same situations (unwrap at rigid_body.rs:31, index at physics.rs:48)
        let msg = unsafe { CStr::from_ptr(ffi::phys_last_error()) }.to_string_lossy().into_owned();
        panic!("physics_hip error {rc}: {msg}");
    }
}

impl HipBackend {
    pub fn new() -> Self {
        let mut cfg = std::mem::MaybeUninit::<ffi::phys_config>::uninit();
        let mut world = std::ptr::null_mut();
        unsafe {
            ffi::phys_config_default(cfg.as_mut_ptr()); // reference constants: gravity (0,-9.81,0) at (0,0,1.5), CG 1000/1e-2/1e-3
            // the frame loop renders on the same GPU (wgpu) while the next update may already run: the default (guarded start of
            // the contact solver's one big launch) is the right one; a headless batch run that has
            // the GPU to itself may add ffi::PHYS_FLAG_EXCLUSIVE_GPU
            check(ffi::phys_create(cfg.as_ptr(), &mut world));
        }
        Self { world, uploaded_bodies: usize::MAX, uploaded_constraints: usize::MAX }
    }

    /// gather `entities` (AoS, repr(Rust)) into the flat arrays phys_set_bodies copies
    fn upload(&mut self, state: &PhysicsState) {
        let n = state.entities.len();
        let (mut pos, mut rot, mut lin, mut ang) = (Vec::with_capacity(3 * n), Vec::with_capacity(4 * n), Vec::with_capacity(3 * n), Vec::with_capacity(3 * n));
        let (mut mass, mut inertia, mut force, mut torque) = (Vec::with_capacity(n), Vec::with_capacity(9 * n), Vec::with_capacity(3 * n), Vec::with_capacity(3 * n));
        for e in &state.entities {
            let b = &e.body;
            pos.extend_from_slice(b.position.as_slice());
            rot.extend_from_slice(b.rotation.coords.as_slice()); // nalgebra order [i, j, k, w]
            lin.extend_from_slice(b.lin_velocity.as_slice());
            ang.extend_from_slice(b.angular_velocity.as_slice());
            mass.push(b.mass);
            inertia.extend_from_slice(&b.inertia_row_major());
            force.extend_from_slice(b.force.as_slice());
            torque.extend_from_slice(b.torque.as_slice());
        }
        unsafe {
            check(ffi::phys_set_bodies(self.world, n as u64, pos.as_ptr(), rot.as_ptr(), lin.as_ptr(), ang.as_ptr(),
                                       mass.as_ptr(), inertia.as_ptr(), std::ptr::null(), std::ptr::null()));
            check(ffi::phys_set_forces(self.world, force.as_ptr(), torque.as_ptr()));
            check(ffi::phys_clear_constraints(self.world));
            for c in &state.constraint_solver.constraints {
                match c {
                    Constraints::FixedPosition(c) => check(ffi::phys_add_constraint_fix_point(self.world, c.rigid_body as u64, c.position.as_slice().as_ptr())),
                    Constraints::FixedOrientation(c) => check(ffi::phys_add_constraint_fix_orientation(self.world, c.rigid_body as u64, c.position.as_slice().as_ptr())),
                }
            }
        }
        self.uploaded_bodies = n;
        self.uploaded_constraints = state.constraint_solver.constraints.len();
    }

    /// scatter the device state back so get_render_data reads body.position / body.rotation as before
    fn download(&mut self, state: &mut PhysicsState) {
        let n = state.entities.len();
        let (mut pos, mut rot, mut lin, mut ang) = (vec![0f32; 3 * n], vec![0f32; 4 * n], vec![0f32; 3 * n], vec![0f32; 3 * n]);
        unsafe {
            check(ffi::phys_get_transforms(self.world, pos.as_mut_ptr(), rot.as_mut_ptr()));
            check(ffi::phys_get_velocities(self.world, lin.as_mut_ptr(), ang.as_mut_ptr()));
        }
        for (i, e) in state.entities.iter_mut().enumerate() {
            let b = &mut e.body;
            b.position.copy_from_slice(&pos[3 * i..3 * i + 3]);
            b.rotation = nalgebra::Unit::new_unchecked(nalgebra::Quaternion::new(rot[4 * i + 3], rot[4 * i], rot[4 * i + 1], rot[4 * i + 2]));
            b.lin_velocity.copy_from_slice(&lin[3 * i..3 * i + 3]);
            b.angular_velocity.copy_from_slice(&ang[3 * i..3 * i + 3]);
            b.force.fill(0.0);  // rigid_body.rs:38-39
            b.torque.fill(0.0);
        }
    }

    /// PhysicsState::update (physics.rs:41-55). `dirty` = the caller touched entities/constraints since the
    /// last frame (lib.rs does at start-up only); when false the bodies stay resident on the GPU.
    pub fn update(&mut self, state: &mut PhysicsState, dt: &Duration, dirty: bool) {
        if dirty || self.uploaded_bodies != state.entities.len() || self.uploaded_constraints != state.constraint_solver.constraints.len() {
            self.upload(state);
        }
        unsafe { check(ffi::phys_update(self.world, dt.as_nanos() as u64)) }; // nanoseconds: as_secs_f32 is redone on the other side (quirk Q7)
        self.download(state);
    }
}

impl Drop for HipBackend {
    fn drop(&mut self) {
        unsafe { ffi::phys_destroy(self.world) };
    }
}
