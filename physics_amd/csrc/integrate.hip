// integrate.hip — gravity + semi-implicit Euler kernels for gfx950.
//
// Replaces PhysicsState::apply_gravity (reference src/physics.rs:87-94) and RigidBody::step
// (src/physics/rigid_body.rs:24-40), operation for operation (quirks Q1, Q2, Q5, Q6, Q9), on SoA
// arrays. Purely HBM-bound: one lane per body, every attribute read once and written once.
// Algorithmic bytes per body: fused gravity+step 120 B (R 68: pos 12, rot 16, v 12,
// w 12, mass 4, inv-inertia diag 12; W 52); +36 B when the full 3x3 inverse inertia is needed; +48 B
// when the force/torque accumulators are live (R 24 + W 24 zeroing).
#include "kernels.hpp"

namespace phys {

struct StepParams {
    uint32_t n;
    float dt;
    float g_force[3];   // gravity force (physics.rs:90)
    float g_torque[3];  // offset x force (rigid_body.rs:60), same for every body
    uint32_t inertia_stride;  // 0: every body shares one diagonal tensor (the reference's only case: identity) - all lanes read
                              // entry 0 and the 16 bytes per body of the array stay where they are; 1 otherwise
};


// velocity half of RigidBody::step: rigid_body.rs:27, 30-31
template <bool DIAG>
__device__ __forceinline__ void integrate_velocity(v3 F, v3 T, float mass, const float* __restrict__ inv_inertia,
                                                   uint32_t i, float dt, v3& v, v3& w, uint32_t inertia_stride = 1u) {
    v.x = v.x + F.x / mass * dt;
    v.y = v.y + F.y / mass * dt;
    v.z = v.z + F.z / mass * dt;
    const v3 L = v3_make(T.x * dt, T.y * dt, T.z * dt);
    v3 dw;
    if (DIAG) {
        // off-diagonals are exactly zero: the gemv row sum reduces to the diagonal product. `inv_inertia` is
        // the compact diagonal array here (one float4 per body)
        const float4 d = reinterpret_cast<const float4*>(inv_inertia)[i * inertia_stride];
        dw = v3_make(d.x * L.x, d.y * L.y, d.z * L.z);
    } else {
        m33 I;
#pragma unroll
        for (int k = 0; k < 9; ++k) I.m[k] = inv_inertia[9 * i + k];
        dw = m33_mul_v3(&I, L);
    }
    w = v3_add(w, dw);
}

// position half of RigidBody::step: rigid_body.rs:28, 32-37
template <bool EXACT_ROT>
__device__ __forceinline__ void integrate_position(float dt, v3 v, v3 w, v3& x, quat& q) {
    x.x = x.x + v.x * dt;
    x.y = x.y + v.y * dt;
    x.z = x.z + v.z * dt;
    if (w.x != 0.0f || w.y != 0.0f || w.z != 0.0f) {
        const float nrm = v3_norm(w);
        // Q10: |w|^2 underflowed to 0 although w != 0 (contact impulses leave w ~ 1e-27).
        // The reference would divide by zero here and poison the quaternion with NaN; skip instead.
        if (!(nrm > 0.0f)) return;
        const v3 a = v3_div(w, nrm);
        const float theta = nrm * dt;
        const float scale = EXACT_ROT ? theta : det_sinf(theta * 0.5f);  // quirk Q1
        const v3 u = v3_make((a.x * scale) / 2.0f, (a.y * scale) / 2.0f, (a.z * scale) / 2.0f);
        const float nn = v3_dot(u, u);
        const float eps = 1.1920929e-7f;  // f32::EPSILON
        quat dq;
        if (nn <= eps * eps) {
            dq.i = 0.0f; dq.j = 0.0f; dq.k = 0.0f; dq.w = 1.0f;
        } else {
            const float n = det_sqrtf(nn);
            const float f = 1.0f * det_sinf(n) / n;
            dq.i = u.x * f; dq.j = u.y * f; dq.k = u.z * f;
            dq.w = 1.0f * det_cosf(n);
        }
        q = quat_mul(dq, q);  // quirk Q6: never renormalised
    }
}

// One kernel = [apply_gravity] + RigidBody::step for every body.
// FORCES: read + zero the force/torque accumulators; otherwise they are known to be zero.
// GRAVITY: fold apply_gravity in (update path); off for a bare phys_step.
// the quirk-Q3 scatter (physics.rs:45-51): entity 0 alone receives J^T lambda, behind gravity, and only when the CG
// converged (Some(lambda)); jl / status come from k_constraint_solve of this update, null without constraints
__device__ __forceinline__ void add_constraint_force(uint32_t i, const float* __restrict__ jl, const uint32_t* __restrict__ cg_status,
                                                     v3& F, v3& T) {
    if (jl != nullptr && i == 0u && cg_status[0] != 0u) {
        F = v3_add(F, v3_make(jl[0], jl[1], jl[2]));
        T = v3_add(T, v3_make(jl[3], jl[4], jl[5]));
    }
}

template <bool FORCES, bool GRAVITY, bool DIAG, bool EXACT_ROT>
__global__ __launch_bounds__(256) void k_step_full(StepParams sp, float* __restrict__ pos, float* __restrict__ rot,
                                                   float* __restrict__ vel, float* __restrict__ force,
                                                   float* __restrict__ torque, const float* __restrict__ inv_inertia,
                                                   const float* __restrict__ jl, const uint32_t* __restrict__ cg_status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sp.n) return;
    v3 F = v3_make(0.0f, 0.0f, 0.0f), T = v3_make(0.0f, 0.0f, 0.0f);
    if (FORCES) { F = ld3(force, i); T = ld3(torque, i); }
    if (GRAVITY) {
        // apply_force_at_offset: torque += offset x F; force += F
        T = v3_add(T, v3_make(sp.g_torque[0], sp.g_torque[1], sp.g_torque[2]));
        F = v3_add(F, v3_make(sp.g_force[0], sp.g_force[1], sp.g_force[2]));
    }
    add_constraint_force(i, jl, cg_status, F, T);
    BodyVel bv = ld_vel(vel, i);
    v3 x = ld3(pos, i);
    float4 qq = reinterpret_cast<float4*>(rot)[i];
    quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
    integrate_velocity<DIAG>(F, T, bv.mass, inv_inertia, i, sp.dt, bv.v, bv.w, sp.inertia_stride);
    integrate_position<EXACT_ROT>(sp.dt, bv.v, bv.w, x, q);
    st_vel(vel, i, bv); st3(pos, i, x);
    reinterpret_cast<float4*>(rot)[i] = make_float4(q.i, q.j, q.k, q.w);
    if (FORCES) { st3(force, i, v3_make(0.0f, 0.0f, 0.0f)); st3(torque, i, v3_make(0.0f, 0.0f, 0.0f)); }
}

// collision mode, first half: [gravity] + velocity update + fattened AABB + largest-extent reduction
template <bool FORCES, bool GRAVITY, bool DIAG>
__global__ __launch_bounds__(256) void k_step_velocity_aabb(StepParams sp, const float* __restrict__ pos,
                                                            const float* __restrict__ rot, float* __restrict__ vel,
                                                            float* __restrict__ force, float* __restrict__ torque,
                                                            const float* __restrict__ inv_inertia,
                                                            const uint32_t* __restrict__ shape,
                                                            const float* __restrict__ half_extent, float margin,
                                                            float* __restrict__ aabb, StepCounters* __restrict__ ctr,
                                                            uint4* __restrict__ zero_base, uint32_t zero_count,
                                                            const float* __restrict__ jl, const uint32_t* __restrict__ cg_status,
                                                            float* __restrict__ geo /* 16 floats per body, or null */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // first kernel of the step: it also zeroes the per-step state of the stages behind it (bucket counts,
    // colouring state, counters up to max_extent_bits) instead of a memset launch in front of it
    for (uint32_t z = i; z < zero_count; z += gridDim.x * blockDim.x) zero_base[z] = make_uint4(0u, 0u, 0u, 0u);
    float ext = 0.0f;
    if (i < sp.n) {
        v3 F = v3_make(0.0f, 0.0f, 0.0f), T = v3_make(0.0f, 0.0f, 0.0f);
        if (FORCES) { F = ld3(force, i); T = ld3(torque, i); }
        if (GRAVITY) {
            T = v3_add(T, v3_make(sp.g_torque[0], sp.g_torque[1], sp.g_torque[2]));
            F = v3_add(F, v3_make(sp.g_force[0], sp.g_force[1], sp.g_force[2]));
        }
        add_constraint_force(i, jl, cg_status, F, T);
        BodyVel bv = ld_vel(vel, i);
        integrate_velocity<DIAG>(F, T, bv.mass, inv_inertia, i, sp.dt, bv.v, bv.w, sp.inertia_stride);
        st_vel(vel, i, bv);
        if (FORCES) { st3(force, i, v3_make(0.0f, 0.0f, 0.0f)); st3(torque, i, v3_make(0.0f, 0.0f, 0.0f)); }
        const v3 x = ld3(pos, i);
        const float4 qq = reinterpret_cast<const float4*>(rot)[i];
        quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
        const uint32_t type = shape[i];
        const v3 he = ld3(half_extent, i);
        const aabb_t b = body_aabb(x, q, he, type, margin);
        st3(aabb, 2 * i, b.lo);
        st3(aabb, 2 * i + 1, b.hi);
        if (geo) {  // the narrow phase's view of this body: one line (world.hpp)
            float4* g = reinterpret_cast<float4*>(geo) + 4 * (size_t)i;
            g[0] = make_float4(x.x, x.y, x.z, __uint_as_float(type));
            g[1] = qq;
            g[2] = make_float4(he.x, he.y, he.z, b.lo.y);
        }
        if (type != PHYS_SPEC_SHAPE_NONE)
            ext = det_maxf(b.hi.x - b.lo.x, det_maxf(b.hi.y - b.lo.y, b.hi.z - b.lo.z));
    }
    // largest extent: wave max by shuffles, one atomic per wave (max is order-independent => deterministic)
    uint32_t bits = __float_as_uint(ext > 0.0f ? ext : 0.0f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)bits, off, 64);
        bits = o > bits ? o : bits;
    }
    // one same-address atomic per wave would serialise the whole launch (~88 atomics/us on one word):
    // only waves that can still raise the maximum issue it (the plain read may be stale-low, never wrong)
    if ((threadIdx.x & 63) == 0 && bits > ctr->max_extent_bits) atomicMax(&ctr->max_extent_bits, bits);
}

// AABBs only (phys_broadphase / phys_get_aabbs on the current poses)
__global__ __launch_bounds__(256) void k_aabb_only(uint32_t n, const float* __restrict__ pos,
                                                   const float* __restrict__ rot, const uint32_t* __restrict__ shape,
                                                   const float* __restrict__ half_extent, float margin,
                                                   float* __restrict__ aabb, StepCounters* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float ext = 0.0f;
    if (i < n) {
        const v3 x = ld3(pos, i);
        const float4 qq = reinterpret_cast<const float4*>(rot)[i];
        quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
        const uint32_t type = shape[i];
        const aabb_t b = body_aabb(x, q, ld3(half_extent, i), type, margin);
        st3(aabb, 2 * i, b.lo);
        st3(aabb, 2 * i + 1, b.hi);
        if (type != PHYS_SPEC_SHAPE_NONE)
            ext = det_maxf(b.hi.x - b.lo.x, det_maxf(b.hi.y - b.lo.y, b.hi.z - b.lo.z));
    }
    uint32_t bits = __float_as_uint(ext > 0.0f ? ext : 0.0f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)bits, off, 64);
        bits = o > bits ? o : bits;
    }
    // one same-address atomic per wave would serialise the whole launch (~88 atomics/us on one word):
    // only waves that can still raise the maximum issue it (the plain read may be stale-low, never wrong)
    if ((threadIdx.x & 63) == 0 && bits > ctr->max_extent_bits) atomicMax(&ctr->max_extent_bits, bits);
}

// collision mode, second half: position + rotation update
template <bool EXACT_ROT>
__global__ __launch_bounds__(256) void k_step_position(uint32_t n, float dt, float* __restrict__ pos,
                                                       float* __restrict__ rot, const float* __restrict__ vel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 x = ld3(pos, i);
    const BodyVel bv = ld_vel(vel, i);
    const v3 v = bv.v, w = bv.w;
    float4 qq = reinterpret_cast<float4*>(rot)[i];
    quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
    integrate_position<EXACT_ROT>(dt, v, w, x, q);
    st3(pos, i, x);
    reinterpret_cast<float4*>(rot)[i] = make_float4(q.i, q.j, q.k, q.w);
}

// PhysicsState::apply_gravity as a stand-alone call (physics.rs:87-94): accumulators += gravity
__global__ __launch_bounds__(256) void k_apply_gravity(StepParams sp, float* __restrict__ force,
                                                       float* __restrict__ torque) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sp.n) return;
    v3 F = ld3(force, i), T = ld3(torque, i);
    T = v3_add(T, v3_make(sp.g_torque[0], sp.g_torque[1], sp.g_torque[2]));
    F = v3_add(F, v3_make(sp.g_force[0], sp.g_force[1], sp.g_force[2]));
    st3(force, i, F); st3(torque, i, T);
}

// RigidBody::apply_force_* on one body (rigid_body.rs:43-62); mode 0 centre, 1 at position, 2 at offset
__global__ void k_apply_force_one(uint32_t body, int mode, v3 f, v3 arg, const float* __restrict__ pos,
                                  float* __restrict__ force, float* __restrict__ torque) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    v3 F = ld3(force, body), T = ld3(torque, body);
    if (mode == 1) T = v3_add(T, v3_cross(v3_sub(arg, ld3(pos, body)), f));
    if (mode == 2) T = v3_add(T, v3_cross(arg, f));
    F = v3_add(F, f);
    st3(force, body, F); st3(torque, body, T);
}

// Instance::to_raw (graphics.rs:13-21): column-major T(p) * R(q), 16 floats per body (row N1)
__global__ __launch_bounds__(256) void k_instance_matrices(uint32_t n, const float* __restrict__ pos,
                                                           const float* __restrict__ rot, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 x = ld3(pos, i);
    const float4 qq = reinterpret_cast<const float4*>(rot)[i];
    quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
    m33 R;
    quat_to_m33(q, &R);
    float4* o = reinterpret_cast<float4*>(out) + 4 * (size_t)i;
    o[0] = make_float4(R.m[0], R.m[3], R.m[6], 0.0f);
    o[1] = make_float4(R.m[1], R.m[4], R.m[7], 0.0f);
    o[2] = make_float4(R.m[2], R.m[5], R.m[8], 0.0f);
    o[3] = make_float4(x.x, x.y, x.z, 1.0f);
}

static inline StepParams make_params(const phys_world* w, float dt) {
    StepParams sp;
    sp.n = (uint32_t)w->n;
    sp.dt = dt;
    const float* F = w->cfg.gravity_force;
    const float* o = w->cfg.gravity_offset;
    for (int k = 0; k < 3; ++k) sp.g_force[k] = F[k];
    // offset.cross(&force), rigid_body.rs:60
    sp.g_torque[0] = o[1] * F[2] - o[2] * F[1];
    sp.g_torque[1] = o[2] * F[0] - o[0] * F[2];
    sp.g_torque[2] = o[0] * F[1] - o[1] * F[0];
    sp.inertia_stride = (w->all_diag_inertia && w->uniform_inertia) ? 0u : 1u;
    return sp;
}

static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

template <bool FORCES, bool GRAVITY>
static void launch_full(phys_world* w, const StepParams& sp, bool with_constraints) {
    const bool diag = w->all_diag_inertia;
    const bool exact = (w->cfg.flags & PHYS_FLAG_EXACT_ROTATION) != 0;
    const dim3 g = grid_for(w->n), b(256);
#define LAUNCH(D, E)                                                                                              \
    hipLaunchKernelGGL((k_step_full<FORCES, GRAVITY, D, E>), g, b, 0, w->stream, sp, w->pos.p, w->rot.p, w->vel.p, \
                       w->force.p, w->torque.p, D ? w->inv_inertia_diag.p : w->inv_inertia.p, jl, cg_status)
    // constraints solved in this update: their force on entity 0 rides along (launch_step_full(.., constraints = true))
    const float* jl = with_constraints ? w->cg_jl.p : nullptr;
    const uint32_t* cg_status = with_constraints ? w->cg_status.p : nullptr;
    PHYS_PROF(w, PHYS_STAGE_STEP_FULL);
    if (diag && exact) LAUNCH(true, true);
    else if (diag) LAUNCH(true, false);
    else if (exact) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
}

// gravity (optional) + RigidBody::step for every body, one launch
void launch_step_full(phys_world* w, float dt, bool gravity, bool constraints) {
    if (w->n == 0) return;
    const StepParams sp = make_params(w, dt);
    if (w->forces_dirty) { if (gravity) launch_full<true, true>(w, sp, constraints); else launch_full<true, false>(w, sp, constraints); }
    else                 { if (gravity) launch_full<false, true>(w, sp, constraints); else launch_full<false, false>(w, sp, constraints); }
    w->forces_dirty = false;
}

void launch_step_velocity_aabb(phys_world* w, float dt, bool gravity, bool zero_step, bool constraints) {
    if (w->n == 0) return;
    uint4* zero_base = zero_step ? reinterpret_cast<uint4*>(w->step_zero.p) : nullptr;
    const uint32_t zero_count = zero_step ? (uint32_t)(w->step_zero_reset_bytes / 16) : 0u;
    const StepParams sp = make_params(w, dt);
    const dim3 g = grid_for(w->n), b(256);
    const bool diag = w->all_diag_inertia;
    const float margin = w->cfg.contact_margin;
#define LAUNCH(F, G, D)                                                                                            \
    hipLaunchKernelGGL((k_step_velocity_aabb<F, G, D>), g, b, 0, w->stream, sp, w->pos.p, w->rot.p, w->vel.p,      \
                       w->force.p, w->torque.p, D ? w->inv_inertia_diag.p : w->inv_inertia.p, w->shape.p,                \
                       w->half_extent.p, margin, w->aabb.p, w->counters.p, zero_base, zero_count, jl, cg_status, w->geo.p)
    const float* jl = constraints ? w->cg_jl.p : nullptr;
    const uint32_t* cg_status = constraints ? w->cg_status.p : nullptr;
    const int sel = (w->forces_dirty ? 4 : 0) | (gravity ? 2 : 0) | (diag ? 1 : 0);
    PHYS_PROF(w, PHYS_STAGE_VELOCITY_AABB);
    switch (sel) {
        case 0: LAUNCH(false, false, false); break;
        case 1: LAUNCH(false, false, true); break;
        case 2: LAUNCH(false, true, false); break;
        case 3: LAUNCH(false, true, true); break;
        case 4: LAUNCH(true, false, false); break;
        case 5: LAUNCH(true, false, true); break;
        case 6: LAUNCH(true, true, false); break;
        default: LAUNCH(true, true, true); break;
    }
#undef LAUNCH
    w->forces_dirty = false;
    w->aabbs_valid = true;
}

void launch_aabb_only(phys_world* w) {
    if (w->n == 0) return;
    PHYS_PROF(w, PHYS_STAGE_VELOCITY_AABB);
    hipLaunchKernelGGL(k_aabb_only, grid_for(w->n), dim3(256), 0, w->stream, (uint32_t)w->n, w->pos.p, w->rot.p,
                       w->shape.p, w->half_extent.p, w->cfg.contact_margin, w->aabb.p, w->counters.p);
    w->aabbs_valid = true;
}

void launch_step_position(phys_world* w, float dt) {
    if (w->n == 0) return;
    const dim3 g = grid_for(w->n), b(256);
    PHYS_PROF(w, PHYS_STAGE_POSITION);
    if (w->cfg.flags & PHYS_FLAG_EXACT_ROTATION)
        hipLaunchKernelGGL((k_step_position<true>), g, b, 0, w->stream, (uint32_t)w->n, dt, w->pos.p, w->rot.p, w->vel.p);
    else
        hipLaunchKernelGGL((k_step_position<false>), g, b, 0, w->stream, (uint32_t)w->n, dt, w->pos.p, w->rot.p, w->vel.p);
    w->aabbs_valid = false;
}

void launch_apply_gravity(phys_world* w) {
    if (w->n == 0) return;
    const StepParams sp = make_params(w, 0.0f);
    hipLaunchKernelGGL(k_apply_gravity, grid_for(w->n), dim3(256), 0, w->stream, sp, w->force.p, w->torque.p);
    w->forces_dirty = true;
}

void launch_apply_force_one(phys_world* w, uint32_t body, int mode, const float f[3], const float arg[3]) {
    const v3 fv = v3_make(f[0], f[1], f[2]);
    const v3 av = arg ? v3_make(arg[0], arg[1], arg[2]) : v3_make(0.0f, 0.0f, 0.0f);
    hipLaunchKernelGGL(k_apply_force_one, dim3(1), dim3(64), 0, w->stream, body, mode, fv, av, w->pos.p, w->force.p, w->torque.p);
    w->forces_dirty = true;
}

void launch_instance_matrices(phys_world* w, float* d_out) {
    if (w->n == 0) return;
    hipLaunchKernelGGL(k_instance_matrices, grid_for(w->n), dim3(256), 0, w->stream, (uint32_t)w->n, w->pos.p, w->rot.p, d_out);
}

}  // namespace phys
