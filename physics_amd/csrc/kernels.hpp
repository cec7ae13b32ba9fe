// kernels.hpp — shared includes and the host-side launch interface between the translation units
// of libphysics_hip.so. Every launch_* enqueues on w->stream and returns without synchronising.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/spec/collide.h"
#include "../../include/spec/contact_solve.h"
#include "../../include/spec/det_math.h"
#include "../../include/spec/vec.h"
#include "world.hpp"

static_assert(PHYS_MAX_COLORS == phys::kMaxColors, "colour limit mismatch");

#define PHYS_PROF_CAT2(a, b) a##b
#define PHYS_PROF_CAT(a, b) PHYS_PROF_CAT2(a, b)
#define PHYS_PROF(w, st) phys::ProfScope PHYS_PROF_CAT(_prof_scope_, __LINE__)((w)->prof, (w)->stream, (st))

namespace phys {

// 12-byte packed attribute access as ONE dwordx3 memory instruction per lane (a wave then covers one
// contiguous 768-B span). Written as three scalar accesses hipcc emits three dword instructions, each
// touching every line of the span again.
struct alignas(4) packed3 { float x, y, z; };
__device__ __forceinline__ v3 ld3(const float* __restrict__ p, uint32_t i) {
    const packed3 t = reinterpret_cast<const packed3*>(p)[i];
    return v3_make(t.x, t.y, t.z);
}
__device__ __forceinline__ void st3(float* __restrict__ p, uint32_t i, v3 v) {
    packed3 t; t.x = v.x; t.y = v.y; t.z = v.z;
    reinterpret_cast<packed3*>(p)[i] = t;
}

// Velocity record of one body: 32 bytes {v.xyz, inv_mass, w.xyz, mass} = two dwordx4 accesses and ONE
// 32-byte sector per gather (the solver gathers it by body id 8 x colours times per step), instead of pieces
// of three separate arrays. lin_velocity / angular_velocity of the reference (rigid_body.rs:9-10) are the
// first three floats of each half.
struct BodyVel { v3 v; float inv_mass; v3 w; float mass; };
__device__ __forceinline__ BodyVel ld_vel(const float* __restrict__ vel, uint32_t i) {
    const float4 a = reinterpret_cast<const float4*>(vel)[2 * (size_t)i];
    const float4 b = reinterpret_cast<const float4*>(vel)[2 * (size_t)i + 1];
    BodyVel r;
    r.v = v3_make(a.x, a.y, a.z); r.inv_mass = a.w;
    r.w = v3_make(b.x, b.y, b.z); r.mass = b.w;
    return r;
}
__device__ __forceinline__ void st_vel(float* __restrict__ vel, uint32_t i, const BodyVel& r) {
    reinterpret_cast<float4*>(vel)[2 * (size_t)i] = make_float4(r.v.x, r.v.y, r.v.z, r.inv_mass);
    reinterpret_cast<float4*>(vel)[2 * (size_t)i + 1] = make_float4(r.w.x, r.w.y, r.w.z, r.mass);
}

// inverse inertia of one body. DIAG: every body's tensor is diagonal (the reference's only case: identity,
// rigid_body.rs:71), stored as one float4 per body = 16 B and one sector per gather instead of 36 B / two.
// The zero off-diagonals are put back, so the arithmetic is the general path's (only signed zeros can differ).
template <bool DIAG>
__device__ __forceinline__ m33 ld_inertia_c(const float* __restrict__ p, uint32_t i) {
    m33 M;
    if (DIAG) {
        const float4 d = reinterpret_cast<const float4*>(p)[i];
#pragma unroll
        for (int k = 0; k < 9; ++k) M.m[k] = 0.0f;
        M.m[0] = d.x; M.m[4] = d.y; M.m[8] = d.z;
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) M.m[k] = p[9 * (size_t)i + k];
    }
    return M;
}

// Persistent colouring: hash table (a << 32 | b) -> colour of this update's manifolds, looked up by the next
// update's narrow phase. Open addressing, linear probing; the table has at least 1.5 slots per manifold SLOT of the
// world, so an insert always finds room. The layout depends on arrival order, the answers (exact key matches) do
// not. The job rides along in k_rows_build (launch_coloring fills it in, launch_solver hands it over).
struct ColorTableJob {
    unsigned long long* keys;        // null: nothing to do
    uint32_t* cols;
    uint32_t mask;
    uint32_t* slots;                 // [0] = count, then the slot of every manifold (sparse clear two updates later)
    unsigned long long* other_keys;  // the table the narrow phase of THIS update read: emptied here (null: not yet used)
    const uint32_t* other_slots;
    const uint32_t* man_a; const uint32_t* man_b; const uint32_t* man_color; const uint64_t* man_prio;
};

__device__ __forceinline__ void color_table_update(const ColorTableJob& job, uint32_t M) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    if (job.other_keys) {
        // revisit exactly the slots the last build filled: no memset of a capacity-sized table, no extra launch
        const uint32_t count = job.other_slots[0];
        for (uint32_t i = tid; i < count; i += nthreads) job.other_keys[job.other_slots[1 + i]] = ~0ull;
    }
    if (tid == 0) job.slots[0] = M;
    for (uint32_t m = tid; m < M; m += nthreads) {
        const unsigned long long key = ((unsigned long long)job.man_a[m] << 32) | job.man_b[m];
        uint32_t h = (uint32_t)(job.man_prio[m] >> 20) & job.mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&job.keys[h], ~0ull, key);
            if (prev == ~0ull || prev == key) { job.cols[h] = job.man_color[m]; job.slots[1 + m] = h; break; }
            h = (h + 1) & job.mask;
        }
    }
}

// integrate.hip
void launch_step_full(phys_world* w, float dt, bool gravity);
void launch_step_velocity_aabb(phys_world* w, float dt, bool gravity, bool zero_step);  // zero_step: also zero the per-step state
void launch_aabb_only(phys_world* w);
void launch_step_position(phys_world* w, float dt);
void launch_apply_gravity(phys_world* w);
void launch_apply_force_one(phys_world* w, uint32_t body, int mode, const float f[3], const float arg[3]);
void launch_instance_matrices(phys_world* w, float* d_out);

// broadphase.hip
int32_t collision_alloc(phys_world* w);
void zero_step_state(phys_world* w, bool including_extent);  // ONE memset: counters + bucket counts + colouring state
void launch_broadphase(phys_world* w);
void build_sorted_grid(phys_world* w);  // bucket_start / sorted_ids / sorted_box from the current AABBs (bucket counts zeroed)
int32_t sorted_pairs_to_host(phys_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs);

// narrowphase.hip / solver.hip
void launch_narrowphase(phys_world* w);
void launch_coloring(phys_world* w);
void snapshot_counters_async(phys_world* w);  // abi.hip
StepCounters* snapshot_acquire(phys_world* w);  // abi.hip: pinned slot a kernel may fill itself ...
void snapshot_commit(phys_world* w);             // ... then mark it in flight
void poll_snapshots(phys_world* w);           // abi.hip
void launch_solver(phys_world* w, float dt);

// constraints.hip
int32_t constraints_alloc(phys_world* w);
void launch_constraint_phase(phys_world* w);

// halo.hip
int32_t halo_pack(phys_world* w, float x_lo, float x_hi, float reach, void* dev_out, uint64_t cap, uint64_t* n_records);
int32_t halo_pairs(phys_world* w, const void* dev_remote, uint64_t n_remote, uint64_t skip_first, uint64_t skip_count,
                   uint64_t* n_cross);

// cluster.hip
int32_t cluster_assign(phys_world* w, const float* host_pos);
void launch_cluster_sort(phys_world* w, unsigned blocks);
void launch_exclusive_scan(phys_world* w, const uint32_t* in, uint32_t count, uint32_t* out);  // broadphase.hip; count % 4 == 0
void launch_solve_cluster(phys_world* w, void* row_all, uint64_t cap, float friction, const float* inertia, uint32_t stride,
                          bool diag, long long timeout_ticks);

int32_t halo_pack_bodies(phys_world* w, void* dev_out, uint64_t cap);
int32_t halo_unpack_ghosts(phys_world* w, const void* dev_records, uint64_t n_records, uint64_t skip_first, uint64_t skip_count);

}  // namespace phys
