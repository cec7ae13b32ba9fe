// cluster.hip — the contact solver for scenes whose colour classes fill the chip: body velocities RESIDENT IN LDS,
// one persistent workgroup per spatial cluster of bodies, all iterations and colours in ONE launch.
//
// Why (measured on C5, 2.9M manifolds, 33 colours): one launch per colour costs 19.4 us for 38 MB.
// Of that, 7.4 us is the scattered 32-byte gather + write-back of the two body velocity records of every row (whole
// 128-byte lines move for 32 useful bytes, through memory, 264 times per step), ~5 us is what every dependent launch
// pays before and after its streaming (launch gap, two dependent round trips to memory, drain), 1.3 us is arithmetic.
// Only the streaming of the row constants (320 B per row and iteration) is compulsory.
//
// So: bodies are grouped into spatial clusters (Morton order of their positions at phys_set_bodies, `slots` bodies
// each, at most one cluster per CU); a workgroup owns one cluster for the whole solve and keeps its bodies' {v, w} in
// LDS. The rows are sorted by (cluster of body A, colour); per iteration the workgroup walks its colours in ascending
// order with a workgroup barrier between them - the spec's order of updates per body, as before - and streams its rows
// from memory exactly once per iteration. A body all of whose rows lie in one cluster never leaves LDS.
// A body touched by a row of ANOTHER cluster is `shared`: its updates travel between workgroups through the same
// data-tagged 16-byte granules as in k_solve_flow (tag = epoch | number of updates applied, sc1 stores / loads; the k-th
// update of a body may only be made by the row holding ticket k). A shared body of the own cluster is also kept in
// LDS with its tag, so a chain of updates that stays inside the workgroup never waits for memory; only an update
// that follows a REMOTE one polls the granule. The first update of a body reads `vel`, the last one writes it.
// No deadlock: every workgroup processes its rows in the global (iteration, colour) order, so the earliest unfinished
// row of the whole solve never waits; all workgroups are resident (grid <= CUs, checked on the host); every spin is
// bounded (timeout -> overflow bit 4 -> PHYS_ERR_HIP). Same arithmetic (solve_manifold_lazy), same order per body:
// bit-identical to the other solver paths.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

namespace phys {

// occupancy asked for: three workgroups per CU (3 waves per SIMD, <= 168 VGPRs; at 128 the kernel spills 2) with diagonal
// tensors, two otherwise. Measured on C5, same bits: 1 per CU 3.48 ms, 2 per CU 2.70, 3 per CU 2.20
constexpr int kClusterPerCuDiag = 3, kClusterPerCuFull = 2;
constexpr int kClusterPairPerCuDecl = 4;  // == kClusterPairPerCu of k_solve_cluster_pair below

// per-row side info packed into row_n.w (as bits): slot (16) | mode (2) per side
//   mode 0: own cluster, never shared -> LDS only          1: own cluster, shared -> LDS while its tag is current, else granule
//   mode 2: another cluster's body   -> granules only       3: no body (ground)
__host__ __device__ __forceinline__ uint32_t side_info(uint32_t slot, uint32_t mode) { return (slot & 0x3FFFu) | (mode << 14); }

// ---- host: spatial clusters from the positions at upload ------------------------------------------------------
static uint32_t spread10(uint32_t x) {
    x &= 0x3ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}

int32_t cluster_assign(phys_world* w, const float* pos /* host, 3 * n_owned */) {
    w->cluster_count = 0;
    const uint64_t n = w->n, n_owned = w->n_owned;
    static const bool off = getenv("PHYS_DEBUG_NO_CLUSTER") != nullptr;
    if (off || n < kClusterMinBodies || !w->flow_vel.p) return PHYS_OK;
    int cus = 0;
    PHYS_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w->device));
    // one workgroup per cluster, several per CU (their phases interleave: one waits for its rows while the others
    // solve); an eighth of the chip to spare: EVERY workgroup must be resident (the kernel's occupancy bound admits
    // kClusterPerCu* of them per CU; a workgroup that found no room would be waited for until the time-out)
    static const char* per_cu_env = getenv("PHYS_DEBUG_CLUSTERS_PER_CU");
    // two lanes per row (k_solve_cluster_pair, four workgroups per CU) with diagonal tensors; PHYS_DEBUG_CLUSTER_KERNEL=lane
    // keeps the one-lane kernel (A/B measurements; same bits)
    static const char* kernel_env = getenv("PHYS_DEBUG_CLUSTER_KERNEL");
    w->cluster_pair = w->all_diag_inertia && !(kernel_env && kernel_env[0] == 'l');
    const int per_cu_max = w->cluster_pair ? kClusterPairPerCuDecl : (w->all_diag_inertia ? kClusterPerCuDiag : kClusterPerCuFull);
    const int per_cu = per_cu_env ? std::min(per_cu_max, std::max(1, atoi(per_cu_env))) : per_cu_max;
    const uint32_t max_clusters = (uint32_t)std::max(8, per_cu * (cus - cus / 8));
    uint32_t slots = (uint32_t)((n + max_clusters - 1) / max_clusters);
    slots = (slots + 63u) / 64u * 64u;
    if (slots > kClusterMaxSlots) return PHYS_OK;  // velocities would not fit the CU's LDS: per-colour launches
    const uint32_t clusters = (uint32_t)((n + slots - 1) / slots);
    // isotropic Morton key over the bounding box of the owned bodies (ghost slots: behind everybody)
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (uint64_t i = 0; i < n_owned; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos[3 * i + a]); hi[a] = std::max(hi[a], pos[3 * i + a]); }
    const float span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), std::max(hi[2] - lo[2], 1e-6f));
    const float scale = 1023.0f / span;
    std::vector<uint64_t> keyed(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t key = 0x40000000ull;  // ghosts
        if (i < n_owned) {
            uint32_t q[3];
            for (int a = 0; a < 3; ++a) {
                const float t = (pos[3 * i + a] - lo[a]) * scale;
                q[a] = t <= 0.0f ? 0u : (t >= 1023.0f ? 1023u : (uint32_t)t);
            }
            key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        }
        keyed[i] = (key << 32) | i;
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> cslot(n), body_of((size_t)clusters * slots, 0xFFFFFFFFu);
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t i = (uint32_t)keyed[r];
        cslot[i] = (uint32_t)r;  // = cluster * slots + slot
        body_of[r] = i;
    }
    PHYS_HIP_TRY(w->cluster_slot.resize(n));
    PHYS_HIP_TRY(w->cluster_body.resize(body_of.size()));
    PHYS_HIP_TRY(w->body_shared.resize(2 * n));  // 64-bit mask of remote colours per body
    PHYS_HIP_TRY(w->seg_count.resize((size_t)clusters * PHYS_MAX_COLORS + 4));
    PHYS_HIP_TRY(w->seg_start.resize((size_t)clusters * PHYS_MAX_COLORS + 4));
    PHYS_HIP_TRY(w->man_rank.resize(w->max_manifolds));
    PHYS_HIP_TRY(hipMemcpyAsync(w->cluster_slot.p, cslot.data(), 4 * n, hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipMemcpyAsync(w->cluster_body.p, body_of.data(), 4 * body_of.size(), hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    w->cluster_count = clusters;
    w->cluster_slots = slots;
    return PHYS_OK;
}

// ---- per step: rows sorted by (cluster of body A, colour); bodies touched by a row of another cluster ----------
__global__ __launch_bounds__(256) void k_cluster_keys(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                      const uint32_t* __restrict__ man_b, const uint32_t* __restrict__ man_color,
                                                      const uint32_t* __restrict__ cluster_slot, uint32_t slots,
                                                      uint32_t* __restrict__ seg_count, uint32_t* __restrict__ man_rank,
                                                      uint32_t* __restrict__ body_shared, const StepCounters* __restrict__ ctr) {
    const uint32_t raw = ctr->n_manifolds;
    const uint32_t M = (uint64_t)raw < max_manifolds ? raw : (uint32_t)max_manifolds;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const uint32_t a = man_a[m], b = man_b[m], c = man_color[m];
        if (c >= (uint32_t)PHYS_MAX_COLORS) { man_rank[m] = 0xFFFFFFFFu; continue; }
        const uint32_t ca = cluster_slot[a] / slots;
        // arrival order inside a (cluster, colour) segment: nothing depends on it (rows of one colour share no body)
        man_rank[m] = atomicAdd(&seg_count[ca * PHYS_MAX_COLORS + c], 1u);
        // a row is owned by the cluster of its body A, so the only updates a body ever receives from ANOTHER workgroup are
        // those of cross rows in which it is body B: that body is `shared`, and the colours of those rows are its
        // remote colours (two 32-bit halves of a 64-bit mask)
        if (b != PHYS_GROUND_ID && cluster_slot[b] / slots != ca) atomicOr(&body_shared[2 * (size_t)b + (c >> 5)], 1u << (c & 31u));
    }
}

__global__ __launch_bounds__(256) void k_cluster_place(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                       const uint32_t* __restrict__ man_color,
                                                       const uint32_t* __restrict__ cluster_slot, uint32_t slots,
                                                       const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ man_rank,
                                                       uint32_t* __restrict__ row_src, const StepCounters* __restrict__ ctr) {
    const uint32_t raw = ctr->n_manifolds;
    const uint32_t M = (uint64_t)raw < max_manifolds ? raw : (uint32_t)max_manifolds;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const uint32_t r = man_rank[m];
        if (r == 0xFFFFFFFFu) continue;
        row_src[seg_start[(cluster_slot[man_a[m]] / slots) * PHYS_MAX_COLORS + man_color[m]] + r] = m;
    }
}

void launch_exclusive_scan(phys_world* w, const uint32_t* in, uint32_t count, uint32_t* out);  // broadphase.hip

// called by launch_coloring in place of the colour-major placement
void launch_cluster_sort(phys_world* w, unsigned blocks) {
    hipStream_t s = w->stream;
    const uint32_t bins = w->cluster_count * PHYS_MAX_COLORS;  // a multiple of 64
    PHYS_PROF(w, PHYS_STAGE_ROWS);
    (void)hipMemsetAsync(w->seg_count.p, 0, (size_t)bins * 4, s);
    (void)hipMemsetAsync(w->body_shared.p, 0, (size_t)w->n * 8, s);
    hipLaunchKernelGGL(k_cluster_keys, dim3(blocks), dim3(256), 0, s, w->max_manifolds, w->man_a.p, w->man_b.p, w->man_color.p,
                       w->cluster_slot.p, w->cluster_slots, w->seg_count.p, w->man_rank.p, w->body_shared.p, w->counters.p);
    launch_exclusive_scan(w, w->seg_count.p, bins, w->seg_start.p);
    hipLaunchKernelGGL(k_cluster_place, dim3(blocks), dim3(256), 0, s, w->max_manifolds, w->man_a.p, w->man_color.p,
                       w->cluster_slot.p, w->cluster_slots, w->seg_start.p, w->man_rank.p, w->row_src.p, w->counters.p);
}

// ---- the solver ----------------------------------------------------------------------------------------------
typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4c ld_gran(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, (int)0x80000010);  // sc1, volatile
}
__device__ __forceinline__ void st_gran(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, v3 v, uint32_t tag) {
    u32x4c g;
    g.x = __float_as_uint(v.x); g.y = __float_as_uint(v.y); g.z = __float_as_uint(v.z); g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, byte_off, 0, 16);  // sc1: write-through
}

struct ClusterRowArrays { uint4* hdr; float4* n; float4* pt; float4* tb; float4* acc; uint64_t cap; };

constexpr int kClusterThreads = 256;  // two workgroups (clusters) per CU: one streams rows while the other computes
template <bool DIAG>
__global__ __launch_bounds__(kClusterThreads, DIAG ? kClusterPerCuDiag : kClusterPerCuFull) void k_solve_cluster(StepCounters* ctr, uint32_t iterations, uint32_t epoch,
                                                                  ClusterRowArrays rows, float friction,
                                                                  const float* __restrict__ inv_inertia, uint32_t inertia_stride,
                                                                  float* vel, float* flow_vel, uint32_t n_bodies,
                                                                  const uint32_t* __restrict__ cluster_body,
                                                                  const uint32_t* __restrict__ body_shared,
                                                                  const uint32_t* __restrict__ seg_start, uint32_t slots,
                                                                  long long timeout_ticks, uint32_t ablate) {
    extern __shared__ __attribute__((aligned(16))) float4 s_lds[];  // [2 * slots]: {v, tag} {w, 1/m};  then the segment table
    float4* s_vel = s_lds;
    uint32_t* s_seg = reinterpret_cast<uint32_t*>(s_lds + 2 * (size_t)slots);  // PHYS_MAX_COLORS + 1 row offsets
    if (ctr->overflow) return;
    const uint32_t cluster = blockIdx.x;
    const uint32_t n_colors = ctr->n_colors;
    const uint32_t etag = epoch << 16;
    const uint32_t cap = (uint32_t)rows.cap;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(flow_vel, 0, n_bodies * 32u, 0x00020000);
    // own bodies into LDS; tag "no update applied yet"
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        float4 a = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(etag)), b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (body != 0xFFFFFFFFu) {
            const float4 v0 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body];
            const float4 v1 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body + 1];
            a = make_float4(v0.x, v0.y, v0.z, __uint_as_float(etag));
            b = make_float4(v1.x, v1.y, v1.z, v0.w);
        }
        s_vel[2 * sl] = a;
        s_vel[2 * sl + 1] = b;
    }
    if (threadIdx.x <= (uint32_t)PHYS_MAX_COLORS) s_seg[threadIdx.x] = seg_start[(size_t)cluster * PHYS_MAX_COLORS + threadIdx.x];
    __syncthreads();
    const long long t_start = wall_clock64();
    bool dead = false;
    // (Measured and dropped: a second register set prefetching the lane's row of the next colour step - 252 VGPRs, the
    // compiler's waits drained it with the current row: 2.72 vs 2.52 ms on C5; touching the next rows into the L2: 3.18.
    // What overlaps memory with arithmetic here is a SECOND workgroup on the same CU - two clusters per CU.)
    struct RowRaw { uint4 h; float4 nn, t01, t23, p0[4], p1[4], ac[4]; };
    auto fetch = [&](uint32_t d, bool with_acc, RowRaw& r) {
        // every plane at once, whatever the point count turns out to be (planes beyond it hold stale, readable data that
        // is ignored): ONE round trip to memory
        r.h = rows.hdr[d];
        r.nn = rows.n[d];
        r.t01 = rows.tb[d];
        r.t23 = rows.tb[cap + d];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.p0[k] = rows.pt[(size_t)(2 * k) * cap + d];
            r.p1[k] = rows.pt[(size_t)(2 * k + 1) * cap + d];
            r.ac[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (with_acc) r.ac[k] = rows.acc[(size_t)k * cap + d];  // written by this very lane one iteration ago
        }
    };
    const uint32_t steps = iterations * n_colors;
    uint32_t it = 0, col = 0;
    for (uint32_t step = 0; step < steps; ++step) {
        const bool last_it = it + 1 == iterations;
        {
            const uint32_t seg_lo = s_seg[col], seg_hi = s_seg[col + 1];
            for (uint32_t d = seg_lo + threadIdx.x; d < seg_hi; d += kClusterThreads) {
                RowRaw r;
                fetch(d, it != 0u, r);
                const uint4 h = r.h;
                const float4 nn = r.nn;
                const float4 t01 = r.t01, t23r = r.t23;
                const uint32_t info = __float_as_uint(nn.w);
                // slot (13 bits) | publish (1) | mode (2) per side. publish: the NEXT update of this (shared, own) body is
                // made by another workgroup, so this update must reach the granules; otherwise it stays in LDS
                uint32_t modeA = (info >> 14) & 3u, modeB = (info >> 30) & 3u;
                const uint32_t slotA = info & 0x1FFFu, slotB = (info >> 16) & 0x1FFFu;
                const bool pubA = (info >> 13) & 1u, pubB = (info >> 29) & 1u;
                if (ablate & 8u) { modeA = 0u; if (modeB == 1u) modeB = 0u; }  // PHYS_DEBUG_ABLATE (timing only, wrong results)
                solver_manifold_t sm;
                sm.count = (int)h.z;
                sm.has_b = h.y != PHYS_GROUND_ID;
                sm.n = v3_make(nn.x, nn.y, nn.z);
                tangent_basis(sm.n, &sm.t1, &sm.t2);
                float4 t23 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (sm.count > 2) t23 = t23r;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    contact_row_t& c = sm.row[k];
                    c.rA = v3_make(0.0f, 0.0f, 0.0f); c.rB = v3_make(0.0f, 0.0f, 0.0f);
                    c.normal_mass = 0.0f; c.tangent_mass[0] = 0.0f; c.tangent_mass[1] = 0.0f; c.bias = 0.0f;
                    c.pn = 0.0f; c.pt[0] = 0.0f; c.pt[1] = 0.0f;
                    if (k < sm.count) {
                        c.rA = v3_make(r.p0[k].x, r.p0[k].y, r.p0[k].z); c.normal_mass = r.p0[k].w;
                        c.rB = v3_make(r.p1[k].x, r.p1[k].y, r.p1[k].z); c.tangent_mass[0] = r.p1[k].w;
                        const float4 t = k < 2 ? t01 : t23;
                        c.tangent_mass[1] = (k & 1) ? t.z : t.x;
                        c.bias = (k & 1) ? t.w : t.y;
                        if (it != 0) { c.pn = r.ac[k].x; c.pt[0] = r.ac[k].y; c.pt[1] = r.ac[k].z; }
                    }
                }
                const uint32_t rankA = h.w & 0xFFu, degA = (h.w >> 8) & 0xFFu, rankB = (h.w >> 16) & 0xFFu, degB = h.w >> 24;
                const uint32_t tA = it * degA + rankA, tB = it * degB + rankB;
                const bool finalA = last_it && rankA + 1 == degA, finalB = last_it && rankB + 1 == degB;
                // ---- the two bodies
                v3 vA, wA, vB = v3_make(0.0f, 0.0f, 0.0f), wB = vB;
                float ima, imb = 0.0f, massA = 0.0f, massB = 0.0f;
                bool needA = false, needB = false;
                {
                    const float4 la = s_vel[2 * slotA], lb = s_vel[2 * slotA + 1];  // A is always of this cluster
                    vA = v3_make(la.x, la.y, la.z); wA = v3_make(lb.x, lb.y, lb.z); ima = lb.w;
                    if (modeA == 1u) {
                        needA = __float_as_uint(la.w) != (etag | tA);  // a remote row made the update before this one
                        if (finalA) massA = vel[8 * (size_t)h.x + 7];
                    }
                }
                if (modeB == 0u || modeB == 1u) {
                    const float4 la = s_vel[2 * slotB], lb = s_vel[2 * slotB + 1];
                    vB = v3_make(la.x, la.y, la.z); wB = v3_make(lb.x, lb.y, lb.z); imb = lb.w;
                    if (modeB == 1u) {
                        needB = __float_as_uint(la.w) != (etag | tB);
                        if (finalB) massB = vel[8 * (size_t)h.y + 7];
                    }
                } else if (modeB == 2u) {
                    const BodyVel B0 = ld_vel(vel, h.y);  // the state itself for ticket 0, the masses always
                    vB = B0.v; wB = B0.w; imb = B0.inv_mass; massB = B0.mass;
                    needB = tB != 0u && !(ablate & 8u);
                }
                uint32_t sweeps = 0;
                while (needA || needB) {
                    if (needA) {
                        const u32x4c g0 = ld_gran(rv, h.x * 32u), g1 = ld_gran(rv, h.x * 32u + 16u);
                        if (g0.w == (etag | tA) && g1.w == (etag | tA)) {
                            vA = v3_make(__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z));
                            wA = v3_make(__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z));
                            needA = false;
                        }
                    }
                    if (needB) {
                        const u32x4c g0 = ld_gran(rv, h.y * 32u), g1 = ld_gran(rv, h.y * 32u + 16u);
                        if (g0.w == (etag | tB) && g1.w == (etag | tB)) {
                            vB = v3_make(__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z));
                            wB = v3_make(__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z));
                            needB = false;
                        }
                    }
                    if (needA || needB) {
                        __builtin_amdgcn_s_sleep(8);
                        if ((++sweeps & 63u) == 0u) {
                            const bool gone = (wall_clock64() - t_start > timeout_ticks) ||
                                              (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u);
                            if (gone) { flag_overflow(ctr, 16u); dead = true; needA = false; needB = false; }
                        }
                    }
                }
                if (!dead) {
                    const m33 IA = ld_inertia_c<DIAG>(inv_inertia, h.x * inertia_stride);
                    m33 IB;
#pragma unroll
                    for (int k = 0; k < 9; ++k) IB.m[k] = 0.0f;
                    if (sm.has_b) IB = ld_inertia_c<DIAG>(inv_inertia, h.y * inertia_stride);
                    if (!(ablate & 2u)) solve_manifold_lazy(&sm, friction, ima, &IA, imb, &IB, &vA, &wA, &vB, &wB);
                    // ---- write back
                    s_vel[2 * slotA] = make_float4(vA.x, vA.y, vA.z, __uint_as_float(etag | (tA + 1u)));
                    s_vel[2 * slotA + 1] = make_float4(wA.x, wA.y, wA.z, ima);
                    if (modeA == 1u) {
                        if (finalA) { BodyVel o; o.v = vA; o.inv_mass = ima; o.w = wA; o.mass = massA; st_vel(vel, h.x, o); }
                        else if (pubA) { st_gran(rv, h.x * 32u, vA, etag | (tA + 1u)); st_gran(rv, h.x * 32u + 16u, wA, etag | (tA + 1u)); }
                    }
                    if (modeB == 0u || modeB == 1u) {
                        s_vel[2 * slotB] = make_float4(vB.x, vB.y, vB.z, __uint_as_float(etag | (tB + 1u)));
                        s_vel[2 * slotB + 1] = make_float4(wB.x, wB.y, wB.z, imb);
                    }
                    if (modeB == 1u || modeB == 2u) {
                        if (finalB) { BodyVel o; o.v = vB; o.inv_mass = imb; o.w = wB; o.mass = massB; st_vel(vel, h.y, o); }
                        else if (modeB == 2u || pubB) { st_gran(rv, h.y * 32u, vB, etag | (tB + 1u)); st_gran(rv, h.y * 32u + 16u, wB, etag | (tB + 1u)); }
                    }
                    if (!last_it) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < sm.count)
                                rows.acc[(size_t)k * cap + d] = make_float4(sm.row[k].pn, sm.row[k].pt[0], sm.row[k].pt[1], 0.0f);
                    }
                }
            }
            __syncthreads();  // LDS velocities of this colour are in place before the next colour reads them
        }
        if (++col == n_colors) { col = 0u; ++it; }
    }
    // bodies that never left this workgroup: their final state goes back to `vel` (a shared body's was written by the row
    // that made its last update, wherever that ran)
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        if (body == 0xFFFFFFFFu || (body_shared[2 * (size_t)body] | body_shared[2 * (size_t)body + 1])) continue;
        const float4 a = s_vel[2 * sl], b = s_vel[2 * sl + 1];
        float4* out = reinterpret_cast<float4*>(vel) + 2 * (size_t)body;
        const float mass = out[1].w;
        out[0] = make_float4(a.x, a.y, a.z, b.w);
        out[1] = make_float4(b.x, b.y, b.z, mass);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same solver with TWO LANES PER ROW: the even lane of a pair owns body A, the odd lane body B. At three
// workgroups per CU the one-lane kernel is bound by the latency of a colour step - row fetch, then 12 sequential rows
// of ~100 instructions each on lanes of which only half hold a row (C5: ~130 rows per step for 256 lanes). Here a
// lane fetches its own side's planes, makes its own Jacobian column (r x dir, I^-1 (r x dir)) and its own half of the
// relative velocity (dir.v + a.w); the halves meet through one DPP exchange inside the pair and are combined in the
// order of the spec, ub - ua; both lanes then make the same scalar update and apply it to their own body. Half the
// chain per lane, every lane busy, ~100 VGPRs (four workgroups per CU). Same arithmetic, same bits.
constexpr int kPairXor1 = 0xB1;  // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float pair_swap(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), kPairXor1, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t pair_swap_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kPairXor1, 0xF, 0xF, true);
}

constexpr int kClusterPairPerCu = kClusterPairPerCuDecl;
template <bool DIAG>
__global__ __launch_bounds__(kClusterThreads, kClusterPairPerCu) void k_solve_cluster_pair(
    StepCounters* ctr, uint32_t iterations, uint32_t epoch, ClusterRowArrays rows, float friction,
    const float* __restrict__ inv_inertia, uint32_t inertia_stride, float* vel, float* flow_vel, uint32_t n_bodies,
    const uint32_t* __restrict__ cluster_body, const uint32_t* __restrict__ body_shared, const uint32_t* __restrict__ seg_start,
    uint32_t slots, long long timeout_ticks) {
    extern __shared__ __attribute__((aligned(16))) float4 s_lds2[];
    float4* s_vel = s_lds2;
    uint32_t* s_seg = reinterpret_cast<uint32_t*>(s_lds2 + 2 * (size_t)slots);
    if (ctr->overflow) return;
    const uint32_t cluster = blockIdx.x;
    const uint32_t n_colors = ctr->n_colors;
    const uint32_t etag = epoch << 16;
    const uint32_t cap = (uint32_t)rows.cap;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(flow_vel, 0, n_bodies * 32u, 0x00020000);
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        float4 a = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(etag)), b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (body != 0xFFFFFFFFu) {
            const float4 v0 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body];
            const float4 v1 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body + 1];
            a = make_float4(v0.x, v0.y, v0.z, __uint_as_float(etag));
            b = make_float4(v1.x, v1.y, v1.z, v0.w);
        }
        s_vel[2 * sl] = a;
        s_vel[2 * sl + 1] = b;
    }
    if (threadIdx.x <= (uint32_t)PHYS_MAX_COLORS) s_seg[threadIdx.x] = seg_start[(size_t)cluster * PHYS_MAX_COLORS + threadIdx.x];
    __syncthreads();
    const long long t_start = wall_clock64();
    const bool side_b = (threadIdx.x & 1u) != 0u;
    const uint32_t pair = threadIdx.x >> 1;
    constexpr uint32_t kPairs = kClusterThreads / 2;
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    bool dead = false;
    const uint32_t steps = iterations * n_colors;
    uint32_t it = 0, col = 0;
    for (uint32_t step = 0; step < steps; ++step) {
        const bool last_it = it + 1 == iterations;
        const uint32_t seg_lo = s_seg[col], seg_hi = s_seg[col + 1];
        // pass count is the same for every lane of the workgroup (the DPP exchanges need both lanes of a pair, and a pair
        // whose row index falls beyond the segment simply carries zeros)
        for (uint32_t base = seg_lo; base < seg_hi; base += kPairs) {
            const uint32_t d = base + pair;
            const bool live = d < seg_hi;
            const uint32_t dd = live ? d : seg_lo;  // a readable row for idle pairs (nothing of it is used)
            // ---- this side's share of the row
            const uint4 h = rows.hdr[dd];
            const float4 nn = rows.n[dd];
            const float4 t01 = rows.tb[dd], t23r = rows.tb[cap + dd];
            float4 pr[4], ac[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pr[k] = rows.pt[(size_t)(2 * k + (side_b ? 1 : 0)) * cap + dd];  // A: {rA, normal mass}  B: {rB, tangent mass 0}
                ac[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (it != 0u) ac[k] = rows.acc[(size_t)k * cap + dd];  // written by this pair one iteration ago
            }
            const uint32_t count = live ? h.z : 0u;
            const bool has_b = h.y != PHYS_GROUND_ID;
            const uint32_t info = __float_as_uint(nn.w) >> (side_b ? 16 : 0);
            const uint32_t slot = info & 0x1FFFu, mode = live ? ((info >> 14) & 3u) : 3u;
            const bool pub = (info >> 13) & 1u;
            const uint32_t body = side_b ? h.y : h.x;
            const uint32_t tk = side_b ? (h.w >> 16) : h.w;
            const uint32_t rank = tk & 0xFFu, deg = (tk >> 8) & 0xFFu;
            const uint32_t ticket = it * deg + rank;
            const bool final_update = last_it && rank + 1 == deg;
            v3 dir[3];
            dir[2] = v3_make(nn.x, nn.y, nn.z);
            tangent_basis(dir[2], &dir[0], &dir[1]);
            // masses and bias of every point, on both lanes: this side's plane carries one of the two packed masses
            float nm[4], tm0[4], tm1[4], bias[4], pn[4], pt0[4], pt1[4];
            v3 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mine = pr[k].w, other = pair_swap(pr[k].w);
                nm[k] = side_b ? other : mine;
                tm0[k] = side_b ? mine : other;
                const float4 t = k < 2 ? t01 : t23r;
                tm1[k] = (k & 1) ? t.z : t.x;
                bias[k] = (k & 1) ? t.w : t.y;
                pn[k] = ac[k].x; pt0[k] = ac[k].y; pt1[k] = ac[k].z;
                r[k] = v3_make(pr[k].x, pr[k].y, pr[k].z);
            }
            // ---- this side's body
            v3 v = zero, w = zero;
            float inv_m = 0.0f, mass = 0.0f;
            bool need = false;
            if (mode == 0u || mode == 1u) {
                const float4 la = s_vel[2 * slot], lb = s_vel[2 * slot + 1];
                v = v3_make(la.x, la.y, la.z); w = v3_make(lb.x, lb.y, lb.z); inv_m = lb.w;
                if (mode == 1u) {
                    need = __float_as_uint(la.w) != (etag | ticket);  // a remote row made the update before this one
                    if (final_update) mass = vel[8 * (size_t)body + 7];
                }
            } else if (mode == 2u) {
                const BodyVel B0 = ld_vel(vel, body);  // the state itself for ticket 0, the masses always
                v = B0.v; w = B0.w; inv_m = B0.inv_mass; mass = B0.mass;
                need = ticket != 0u;
            }
            uint32_t sweeps = 0;
            while (need) {
                const u32x4c g0 = ld_gran(rv, body * 32u), g1 = ld_gran(rv, body * 32u + 16u);
                if (g0.w == (etag | ticket) && g1.w == (etag | ticket)) {
                    v = v3_make(__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z));
                    w = v3_make(__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z));
                    need = false;
                } else {
                    __builtin_amdgcn_s_sleep(8);
                    if ((++sweeps & 63u) == 0u) {
                        const bool gone = (wall_clock64() - t_start > timeout_ticks) ||
                                          (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u);
                        if (gone) { flag_overflow(ctr, 16u); dead = true; need = false; }
                    }
                }
            }
            // a pair acts only when both of its lanes have their body; every lane reaches the exchanges below
            const uint32_t pair_dead = (dead ? 1u : 0u) | pair_swap_u(dead ? 1u : 0u);
            const bool has_body = mode != 3u;
            m33 I;
#pragma unroll
            for (int k = 0; k < 9; ++k) I.m[k] = 0.0f;
            if (has_body) I = ld_inertia_c<DIAG>(inv_inertia, body * inertia_stride);
            v3 lin[3];  // lA / lB of the spec: dir * inverse mass (zero for a side without a body)
#pragma unroll
            for (int t = 0; t < 3; ++t) lin[t] = has_body ? v3_scale(dir[t], inv_m) : zero;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < (int)count) {  // the same for both lanes of the pair
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        // jac_row_make: a = r x dir, m = I a (zero for a side without a body)
                        const v3 a = has_body ? v3_cross(r[k], dir[t]) : zero;
                        const v3 mI = has_body ? m33_mul_v3(&I, a) : zero;
                        // row_velocity: ub - ua with ua = dir.vA + aA.wA, ub = has_b ? dir.vB + aB.wB : 0
                        const float mine = has_body ? v3_dot(dir[t], v) + v3_dot(a, w) : 0.0f;
                        const float other = pair_swap(mine);
                        const float vrel = side_b ? mine - other : other - mine;
                        float lambda;
                        if (t < 2) {  // solve_row_dir, friction
                            const float tm = t == 0 ? tm0[k] : tm1[k];
                            float& accu = t == 0 ? pt0[k] : pt1[k];
                            lambda = -tm * vrel;
                            const float maxf = friction * pn[k];
                            const float old = accu;
                            const float np = det_maxf(-maxf, det_minf(old + lambda, maxf));
                            lambda = np - old;
                            accu = np;
                        } else {      // normal
                            lambda = nm[k] * (bias[k] - vrel);
                            const float old = pn[k];
                            const float np = det_maxf(old + lambda, 0.0f);
                            lambda = np - old;
                            pn[k] = np;
                        }
                        // row_apply: A subtracts, B adds
                        if (has_body) {
                            if (side_b) { v = v3_add(v, v3_scale(lin[t], lambda)); w = v3_add(w, v3_scale(mI, lambda)); }
                            else        { v = v3_sub(v, v3_scale(lin[t], lambda)); w = v3_sub(w, v3_scale(mI, lambda)); }
                        }
                    }
                }
            }
            // ---- write back
            if (live && !pair_dead) {
                if (mode == 0u || mode == 1u) {
                    s_vel[2 * slot] = make_float4(v.x, v.y, v.z, __uint_as_float(etag | (ticket + 1u)));
                    s_vel[2 * slot + 1] = make_float4(w.x, w.y, w.z, inv_m);
                }
                if (mode == 1u || mode == 2u) {
                    if (final_update) { BodyVel o; o.v = v; o.inv_mass = inv_m; o.w = w; o.mass = mass; st_vel(vel, body, o); }
                    else if (mode == 2u || pub) { st_gran(rv, body * 32u, v, etag | (ticket + 1u)); st_gran(rv, body * 32u + 16u, w, etag | (ticket + 1u)); }
                }
                if (!last_it) {  // the A lane keeps the impulses of points 0 and 1, the B lane those of 2 and 3
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < (int)count && ((k >> 1) == (side_b ? 1 : 0)))
                            rows.acc[(size_t)k * cap + d] = make_float4(pn[k], pt0[k], pt1[k], 0.0f);
                }
            }
        }
        __syncthreads();  // LDS velocities of this colour are in place before the next colour reads them
        if (++col == n_colors) { col = 0u; ++it; }
    }
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        if (body == 0xFFFFFFFFu || (body_shared[2 * (size_t)body] | body_shared[2 * (size_t)body + 1])) continue;
        const float4 a = s_vel[2 * sl], b = s_vel[2 * sl + 1];
        float4* out = reinterpret_cast<float4*>(vel) + 2 * (size_t)body;
        const float mass = out[1].w;
        out[0] = make_float4(a.x, a.y, a.z, b.w);
        out[1] = make_float4(b.x, b.y, b.z, mass);
    }
}

void launch_solve_cluster(phys_world* w, void* hdr, void* nrm, void* pt, void* tb, void* acc, uint64_t cap, float friction,
                          const float* inertia, uint32_t stride, bool diag, long long timeout_ticks) {
    static const uint32_t ablate = getenv("PHYS_DEBUG_ABLATE") ? (uint32_t)atoi(getenv("PHYS_DEBUG_ABLATE")) : 0u;
    ClusterRowArrays rows;
    rows.hdr = (uint4*)hdr; rows.n = (float4*)nrm; rows.pt = (float4*)pt; rows.tb = (float4*)tb; rows.acc = (float4*)acc; rows.cap = cap;
    const size_t lds = (size_t)w->cluster_slots * 32 + (PHYS_MAX_COLORS + 1) * 4 + 12;
    static bool attr_set[2] = {false, false};
    if (!attr_set[diag ? 1 : 0]) {  // more than the default 64 KiB of dynamic LDS needs the attribute (once per kernel)
        if (diag) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr_set[diag ? 1 : 0] = true;
    }
    const dim3 g(w->cluster_count), b(kClusterThreads);
    if (w->cluster_pair) {
        static bool pair_attr = false;
        if (!pair_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster_pair<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
            pair_attr = true;
        }
        hipLaunchKernelGGL(k_solve_cluster_pair<true>, g, b, lds, w->stream, w->counters.p, w->cfg.solver_iterations, w->flow_epoch, rows,
                           friction, inertia, stride, w->vel.p, w->flow_vel.p, (uint32_t)w->n, w->cluster_body.p, w->body_shared.p,
                           w->seg_start.p, w->cluster_slots, timeout_ticks);
        return;
    }
    if (diag)
        hipLaunchKernelGGL(k_solve_cluster<true>, g, b, lds, w->stream, w->counters.p, w->cfg.solver_iterations, w->flow_epoch, rows, friction,
                           inertia, stride, w->vel.p, w->flow_vel.p, (uint32_t)w->n, w->cluster_body.p, w->body_shared.p, w->seg_start.p,
                           w->cluster_slots, timeout_ticks, ablate);
    else
        hipLaunchKernelGGL(k_solve_cluster<false>, g, b, lds, w->stream, w->counters.p, w->cfg.solver_iterations, w->flow_epoch, rows, friction,
                           inertia, stride, w->vel.p, w->flow_vel.p, (uint32_t)w->n, w->cluster_body.p, w->body_shared.p, w->seg_start.p,
                           w->cluster_slots, timeout_ticks, ablate);
}

}  // namespace phys
