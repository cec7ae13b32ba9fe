// kernels.hpp — shared includes and the host-side launch interface between the translation units
// of libphysics_hip.so. Every launch_* enqueues on w->stream and returns without synchronising.
#pragma once
#include <algorithm>
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

// Persistent colouring: ONE hash table (a << 32 | b) -> {colour, update stamp} that lives across updates. Open addressing,
// linear probing, 16-byte entries {key, stamp << 32 | manifold index << 6 | colour} (the index of the pair's manifold in
// the update that stamped the entry: what the next update warm-starts from); at least 1.5 slots per manifold SLOT.
//   * the narrow phase of update E looks every manifold up: an exact key match whose stamp is E - 1 ("was there in the
//     previous update") keeps its colour and is re-stamped E on the spot - one 8-byte store to the line the probe has
//     just read. Entries that are not re-stamped are dead from then on.
//   * k_rows_build inserts only the manifolds that were NEW in this update (a few per cent of a steady pile), into the
//     first slot of their chain that is not live: claimed by a 64-bit CAS on the VALUE word (stamp becomes E), key stored
//     after. A slot stamped E is nobody else's business, whatever key it shows, so a half-written claim is never
//     taken for a match; pairs are unique within an update, so nobody looks for a key stamped E.
//   * a key's live entry is always the first entry of that key in its chain (a dead one before it would have been
//     reused by the insert), chains never shrink (slots go back to empty only in the reset), so a probe may stop at the
//     first empty slot or the first entry with its key.
//   * every PHYS_COLOR_CACHE_PERIOD-th update the table is REBUILT: emptied by one memset behind the narrow phase (which has
//     taken the kept colours from it by then) and refilled by k_rows_build with every manifold of that update. No colour
//     changes; the dead entries of the period are gone.
// Round 1-2a rebuilt a table per update inside k_rows_build (an atomic and two scattered stores per manifold, plus the
// sparse clear of the other table): 0.26 of k_rows_build's 0.50 ms on C5. The layout depends on arrival order, the
// answers (exact key + stamp matches) do not.
constexpr uint32_t kColorTableMaxWalk = 4096;  // slots a look-up may walk before it gives up (a full table must not hang a wave)
struct ColorTableJob {
    ulonglong2* tab;  // null: nothing to do
    uint32_t mask;
    uint32_t stamp;   // of the update whose manifolds are being inserted
    uint32_t all;     // != 0: table rebuild - the manifolds that kept their colour are inserted too
    const uint32_t* man_color; const uint64_t* man_prio;
};

__device__ __forceinline__ void color_table_insert(const ColorTableJob& job, uint32_t a, uint32_t b, uint32_t m, StepCounters* ctr) {
    const unsigned long long key = ((unsigned long long)a << 32) | b;
    const unsigned long long val = ((unsigned long long)job.stamp << 32) | ((unsigned long long)(m & 0x3FFFFFFu) << 6) | (job.man_color[m] & 63u);
    uint32_t h = (uint32_t)(job.man_prio[m] >> 20) & job.mask;
    // bounded (see the walk in k_narrowphase); a manifold that finds no slot would be coloured afresh next time where the
    // oracle keeps its colour: the update is flagged (bit 6), never a silent divergence
    for (uint32_t walked = 0; ; ++walked) {
        if (walked == 4u * kColorTableMaxWalk) { flag_overflow(ctr, 64u); return; }
        unsigned long long* vp = &job.tab[h].y;
        const unsigned long long seen = __hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(seen >> 32) != job.stamp && atomicCAS(vp, seen, val) == seen) {
            job.tab[h].x = key;
            return;
        }
        if ((uint32_t)(__hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == job.stamp) h = (h + 1) & job.mask;
        // else: somebody else's CAS on a dead slot failed too, or the value changed under us - look at the slot again
    }
}

// ---- uniform grid of the broad phase: cell -> bucket -----------------------------------------------------------------
// The table has 2^bits buckets, bits = bx + by + bz split over the axes in proportion to the scene's extent (a tower 16
// cells wide and 980 high gets x 4, y 10, z 5 instead of 7 + 7 + 7: with equal bits its 128-cell axis wrapped 7.6 times
// and every bucket held the bodies of eight different cells). Cell coordinates are taken modulo the axis size (far
// bodies alias; aliased candidates fail the overlap test, and neighbouring cells never alias: every axis has >= 4 cells).
// Buckets are numbered BRICK-major: a brick is 4 x 4 x 4 cells = 64 consecutive buckets (the low two bits of each
// coordinate interleaved), bricks x-fastest. So the bodies of a brick are one contiguous run of the bucket-sorted
// arrays, which is what lets one workgroup stage a brick and its half-shell halo in LDS (k_find_pairs_brick).
// (struct GridShape: world.hpp)
__host__ __device__ __forceinline__ uint32_t grid_bucket_masked(uint32_t x, uint32_t y, uint32_t z, const GridShape& g) {
    const uint32_t brick = ((((z >> 2) << g.sy) | (y >> 2)) << g.sx) | (x >> 2);
    const uint32_t local = (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2) | ((x & 2u) << 2) | ((y & 2u) << 3) | ((z & 2u) << 4);
    return (brick << 6) | local;
}
__host__ __device__ __forceinline__ uint32_t grid_bucket(int cx, int cy, int cz, const GridShape& g) {
    return grid_bucket_masked((uint32_t)cx & g.mx, (uint32_t)cy & g.my, (uint32_t)cz & g.mz, g);
}
__device__ __forceinline__ int grid_cell_coord(float c, float inv_cell) {
    float t = floorf(c * inv_cell);
    t = t < -1.0e9f ? -1.0e9f : (t > 1.0e9f ? 1.0e9f : t);
    return (int)t;
}

// Cluster solver: which cluster's workgroup owns a row. A body has a HOME cluster (cluster_slot / slots) or none (ghost
// bodies of a sharded world; bodies beyond the capacity of a dynamic clustering). A row is owned by the home of its
// body A, else by the home of its body B, else - both homeless - by a cluster picked from A's id. A side whose body's
// home is the owner is served from that workgroup's LDS; every other side is "another cluster's body".
constexpr uint32_t kNoHome = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t cluster_home(const uint32_t* __restrict__ cluster_slot, uint32_t body, uint32_t slots) {
    const uint32_t s = cluster_slot[body];
    return s == kNoHome ? kNoHome : s / slots;
}
__device__ __forceinline__ uint32_t cluster_row_owner(uint32_t a, uint32_t home_a, uint32_t home_b, uint32_t clusters) {
    if (home_a != kNoHome) return home_a;
    if (home_b != kNoHome) return home_b;
    return ((a * 2654435761u) >> 7) % clusters;
}

// integrate.hip
void launch_step_full(phys_world* w, float dt, bool gravity, bool constraints = false);  // constraints: entity 0 += J^T lambda of this update
void launch_step_velocity_aabb(phys_world* w, float dt, bool gravity, bool zero_step, bool constraints = false);  // zero_step: also zero the per-step state
void launch_aabb_only(phys_world* w);
void launch_step_position(phys_world* w, float dt);
void launch_apply_gravity(phys_world* w);
void launch_apply_force_one(phys_world* w, uint32_t body, int mode, const float f[3], const float arg[3]);
void launch_instance_matrices(phys_world* w, float* d_out);

// broadphase.hip
int32_t collision_alloc(phys_world* w);
void grid_plan(phys_world* w, const float* host_pos, const float* host_half_extent);  // table size and its split over the axes
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
void launch_constraint_phase(phys_world* w, bool gravity_pending);  // Q = accumulators (+ gravity when still pending)

// halo.hip
int32_t halo_pack(phys_world* w, float x_lo, float x_hi, float reach, void* dev_out, uint64_t cap, uint64_t* n_records);
int32_t halo_pairs(phys_world* w, const void* dev_remote, uint64_t n_remote, uint64_t skip_first, uint64_t skip_count,
                   uint64_t* n_cross);

// cluster.hip
int32_t cluster_assign(phys_world* w, const float* host_pos);
// Which single-launch solver for a dense scene whose GPU is the world's alone (PHYS_FLAG_EXCLUSIVE_GPU: the dataflow kernel may
// then fill the chip with three workgroups per CU, like the cluster kernel). Fitted to measurements of this build, ms per
// sweep: the four-lane dataflow kernel at 672 workgroups 8.4e-8 M + 1.86e-8 K + 0.0009 C (C3 0.325 ms per solve, a 182k
// tower 0.380, the 1M cubes in mid-fall 0.405, C5 1.07); the cluster kernel max(0.0041 C, 6.9e-8 M) + 10 % (C3 0.513, the
// tower 0.636, the 1M cubes 0.235, C5 0.63): rows cost the dataflow kernel throughput, colours cost the cluster kernel
// its chain. (Both give the same bits: the choice may change from update to update.)
inline bool flow_quad_beats_cluster(uint32_t manifolds, uint32_t contacts, uint32_t colors) {
    if (!colors || !contacts) return false;
    const double flow = 8.4e-8 * manifolds + 1.86e-8 * contacts + 0.0009 * colors;
    const double cluster = 1.1 * std::max(0.0041 * colors, 6.9e-8 * manifolds);
    return flow < cluster;
}
void launch_cluster_sort(phys_world* w, unsigned blocks, StepCounters* snap_out /* host-mapped slot for the counters, or null */);
#ifdef __HIPCC__
// the step counters copied out to a host-mapped slot by the first workgroup of a kernel that runs after their last writer
__device__ __forceinline__ void counters_snapshot(const StepCounters* ctr, StepCounters* snap_out) {
    if (snap_out && blockIdx.x == 0) {
        constexpr uint32_t words = (uint32_t)(sizeof(StepCounters) / 4);
        for (uint32_t k = threadIdx.x; k < words; k += blockDim.x)
            reinterpret_cast<uint32_t*>(snap_out)[k] =
                __hip_atomic_load(reinterpret_cast<const uint32_t*>(ctr) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif
void launch_exclusive_scan(phys_world* w, uint32_t* in, uint32_t count, uint32_t* out, bool zero_in);  // broadphase.hip; count % 4 == 0
bool scan_is_one_launch(uint32_t count);  // ... in which case zero_in leaves the counters zeroed behind the scan
bool cluster_plan_dynamic(phys_world* w);  // cluster.hip: clusters / slots of this update from the hint (dynamic clusters)
void launch_solve_cluster(phys_world* w, void* row_all, uint64_t cap, float friction, const float* inertia, uint32_t stride,
                          bool diag, long long timeout_ticks);

int32_t halo_pack_bodies(phys_world* w, void* dev_out, uint64_t cap);
int32_t halo_pack_bodies_faces(phys_world* w, void* dev_out, uint64_t cap, float x_lo, float x_hi);
int32_t halo_unpack_ghosts(phys_world* w, const void* dev_records, uint64_t n_records, uint64_t skip_first, uint64_t skip_count);

}  // namespace phys
