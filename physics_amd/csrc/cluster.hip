// cluster.hip — the contact solver for scenes whose colour classes fill the chip: body velocities RESIDENT IN LDS,
// one persistent workgroup per spatial cluster of bodies, all iterations and colours in ONE launch.
//
// Why (measured on C5, 2.9M manifolds, 33 colours): one launch per colour costs 19.4 us for 38 MB.
// Of that, 7.4 us is the scattered 32-byte gather + write-back of the two body velocity records of every row (whole
// 128-byte lines move for 32 useful bytes, through memory, 264 times per step), ~5 us is what every dependent launch
// pays before and after its streaming (launch gap, two dependent round trips to memory, drain), 1.3 us is arithmetic.
// Only the streaming of the row constants is compulsory.
//
// So: bodies are grouped into spatial clusters - statically (Morton order of the owned bodies' positions at
// phys_set_bodies) where they all fit the chip's LDS, else dynamically (every few updates, the bodies that have manifolds,
// in the broad phase's bucket order) - of `slots` bodies each, up to three clusters per CU; a workgroup owns one cluster
// for the whole solve and keeps its bodies' {v, w, 1/m, x, I^-1 diagonal} in LDS. The rows are sorted by (owner cluster,
// colour) - a row belongs to the home of its body A, else of its body B; per iteration the workgroup walks its colours in
// ascending order with a workgroup barrier between them - the spec's order of updates per body, as before - and streams
// its rows from memory exactly once per iteration, COMPACT (contact points + bias, accumulated impulses, row masses made
// in iteration 0; lever arms remade from the positions in LDS: solve_manifold_geo) and fetched one or two colour steps
// ahead. A body all of whose rows lie in one cluster never leaves LDS.
// A body touched by a row of ANOTHER cluster is `shared`: its updates travel between workgroups through the same
// data-tagged 16-byte granules as in k_solve_flow (tag = epoch | number of updates applied, write-through stores, loads
// past the L2; the k-th update of a body may only be made by the row holding ticket k). A shared body of the own cluster
// is also kept in LDS with its tag, so a chain of updates that stays inside the workgroup never waits for memory; only an
// update that follows a REMOTE one polls the granule. The first update of a body reads `vel`, the last one writes it. A
// body without a home (ghosts; bodies beyond the capacity of dynamic homes) is "another cluster's body" for every row.
// No deadlock: every workgroup processes its rows in the global (iteration, colour) order, so the earliest unfinished
// row of the whole solve never waits; all workgroups are resident (grid sized on the host; an all-or-nothing start where
// other streams' work can run beside the launch: see the kernel); every spin is bounded (timeout -> overflow bit 4 ->
// PHYS_ERR_HIP). Same arithmetic, same order per body: bit-identical to the other solver paths.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "kernels.hpp"

namespace phys {

// occupancy asked for: three workgroups per CU (3 waves per SIMD, <= 168 VGPRs; at 128 the kernel spills 2) with diagonal
// tensors, two otherwise. Measured on C5, same bits: 1 per CU 3.48 ms, 2 per CU 2.70, 3 per CU 2.20
constexpr int kClusterPerCuDiag = 3, kClusterPerCuFull = 2;
constexpr size_t kClusterLdsPerCu = 160 * 1024;
constexpr size_t kClusterSlotBytesDecl = 64;  // == kClusterSlotBytes below
size_t cluster_lds_bytes(uint32_t slots);

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

// the (cluster, colour) counters sit behind the 2 n mask words of body_shared, on a 16-byte boundary (the scans read uint4)
static size_t seg_count_offset(uint64_t n) { return ((size_t)2 * n + 3) & ~(size_t)3; }

int32_t cluster_assign(phys_world* w, const float* pos /* host, 3 * n_owned */) {
    w->cluster_count = 0;
    w->seg_count_dirty = false;  // (both allocations below are zeroed)
    w->seg_count_bins = 0;
    const uint64_t n = w->n, n_owned = w->n_owned;
    static const bool off = getenv("PHYS_DEBUG_NO_CLUSTER") != nullptr;
    w->cluster_dynamic = false;
    w->cluster_homes_valid = false;
    if (off || n_owned < kClusterMinBodies || !w->flow_vel.p) return PHYS_OK;
    int cus = 0;
    PHYS_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w->device));
    w->cluster_cus = cus;
    // one workgroup per cluster, several per CU (their phases interleave: one waits for its rows while the others
    // solve); an eighth of the chip to spare: EVERY workgroup must be resident (the kernel's occupancy bound admits
    // kClusterPerCu* of them per CU; a workgroup that found no room would be waited for until the time-out)
    static const char* per_cu_env = getenv("PHYS_DEBUG_CLUSTERS_PER_CU");
    const int per_cu_max = w->all_diag_inertia ? kClusterPerCuDiag : kClusterPerCuFull;
    int per_cu = per_cu_env ? std::min(per_cu_max, std::max(1, atoi(per_cu_env))) : per_cu_max;
    static const int spare_div = getenv("PHYS_DEBUG_CLUSTER_SPARE") ? atoi(getenv("PHYS_DEBUG_CLUSTER_SPARE")) : 8;
    // PHYS_DEBUG_CLUSTER_DYNAMIC: dynamic clusters even where the static ones fit; PHYS_DEBUG_CLUSTER_CAP=<bodies>: fewer
    // homes than the LDS would hold, so that some bodies stay homeless (tests of exactly that; same bits)
    static const bool force_dynamic = getenv("PHYS_DEBUG_CLUSTER_DYNAMIC") != nullptr;
    static const uint64_t cap_env = getenv("PHYS_DEBUG_CLUSTER_CAP") ? strtoull(getenv("PHYS_DEBUG_CLUSTER_CAP"), nullptr, 10) : 0;
    w->cluster_cap_limit = cap_env;
    uint32_t slots = 0;
    bool fits = !force_dynamic;
    // ... and the LDS of a CU must hold all of its workgroups' bodies (64 B per slot + the segment table, in 1 KiB
    // allocation units), or the grid would not be resident: fewer, larger clusters per CU until it does
    for (; fits; --per_cu) {
        const uint32_t max_clusters = (uint32_t)std::max(8, per_cu * (cus - (spare_div ? cus / spare_div : 0)));
        slots = (uint32_t)((n_owned + max_clusters - 1) / max_clusters);
        slots = (slots + 63u) / 64u * 64u;
        const size_t per_wg = (cluster_lds_bytes(slots) + 1023) / 1024 * 1024;
        if (per_wg * (size_t)per_cu <= kClusterLdsPerCu && slots <= kClusterMaxSlots) break;
        if (per_cu == 1) fits = false;  // the owned bodies do not fit the chip's LDS
    }
    if (!fits) {
        // DYNAMIC clusters: homes are dealt out every update, to the bodies that have a manifold in it, in the broad
        // phase's bucket order (launch_cluster_sort); their number and size follow the count of such bodies
        // (cluster_plan_dynamic). Needs the sorted grid of the broad phase (n > 32768: always the case here).
        const uint32_t clusters_max = (uint32_t)std::max(8, per_cu_max * (cus - (spare_div ? cus / spare_div : 0)));
        const size_t homes_max = (size_t)cus * (kClusterLdsPerCu / kClusterSlotBytesDecl) + 64;
        PHYS_HIP_TRY(w->cluster_slot.resize(n));
        PHYS_HIP_TRY(w->cluster_body.resize(homes_max));
        // (the per-(cluster, colour) counters live behind the bodies' masks: both are zeroed every update, by ONE memset)
        PHYS_HIP_TRY(w->body_shared.resize(seg_count_offset(n) + (size_t)clusters_max * PHYS_MAX_COLORS + 4));
        PHYS_HIP_TRY(hipMemsetAsync(w->body_shared.p, 0, (seg_count_offset(n) + (size_t)clusters_max * PHYS_MAX_COLORS + 4) * 4, w->stream));
        PHYS_HIP_TRY(w->seg_start.resize((size_t)clusters_max * PHYS_MAX_COLORS + 4));
        PHYS_HIP_TRY(w->man_rank.resize(w->max_manifolds));
        PHYS_HIP_TRY(w->active_flag.resize(n + 4));
        PHYS_HIP_TRY(w->active_rank.resize(n + 4));
        w->cluster_dynamic = true;
        w->cluster_count = 0;
        w->cluster_slots = 0;
        return PHYS_OK;
    }
    const uint32_t clusters = (uint32_t)((n_owned + slots - 1) / slots);
    // isotropic Morton key over the bounding box of the owned bodies. Ghost bodies (sharded worlds) get no home in
    // any cluster: a row never has one as body A, as body B it is 'another cluster's body' for everybody (slot ~0 maps
    // to a cluster nobody runs), and no workgroup is spent on clusters that own no rows
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (uint64_t i = 0; i < n_owned; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos[3 * i + a]); hi[a] = std::max(hi[a], pos[3 * i + a]); }
    const float span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), std::max(hi[2] - lo[2], 1e-6f));
    const float scale = 1023.0f / span;
    std::vector<uint64_t> keyed(n_owned);
    for (uint64_t i = 0; i < n_owned; ++i) {
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            const float t = (pos[3 * i + a] - lo[a]) * scale;
            q[a] = t <= 0.0f ? 0u : (t >= 1023.0f ? 1023u : (uint32_t)t);
        }
        const uint64_t key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        keyed[i] = (key << 32) | i;
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> cslot(n, 0xFFFFFFFFu), body_of((size_t)clusters * slots, 0xFFFFFFFFu);
    for (uint64_t r = 0; r < n_owned; ++r) {
        const uint32_t i = (uint32_t)keyed[r];
        cslot[i] = (uint32_t)r;  // = cluster * slots + slot
        body_of[r] = i;
    }
    PHYS_HIP_TRY(w->cluster_slot.resize(n));
    PHYS_HIP_TRY(w->cluster_body.resize(body_of.size()));
    // 64-bit mask of remote colours per body; behind them the per-(cluster, colour) counters (zeroed together: one memset)
    PHYS_HIP_TRY(w->body_shared.resize(seg_count_offset(n) + (size_t)clusters * PHYS_MAX_COLORS + 4));
    PHYS_HIP_TRY(hipMemsetAsync(w->body_shared.p, 0, (seg_count_offset(n) + (size_t)clusters * PHYS_MAX_COLORS + 4) * 4, w->stream));
    PHYS_HIP_TRY(w->seg_start.resize((size_t)clusters * PHYS_MAX_COLORS + 4));
    PHYS_HIP_TRY(w->man_rank.resize(w->max_manifolds));
    PHYS_HIP_TRY(hipMemcpyAsync(w->cluster_slot.p, cslot.data(), 4 * n, hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipMemcpyAsync(w->cluster_body.p, body_of.data(), 4 * body_of.size(), hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    w->cluster_count = clusters;
    w->cluster_slots = slots;
    return PHYS_OK;
}

// ---- dynamic clusters: plan (host, from the lagged count of active bodies) and assignment (device, this update) ----
bool cluster_plan_dynamic(phys_world* w) {
    if (!w->cluster_dynamic) return false;
    if (!w->sorted_grid_valid) return false;  // the deal reads the bucket order of this update's broad phase (sorted grid only)
    // the homes stay for kClusterDynamicPeriod cluster steps (a pile changes slowly; the deal is three passes over all
    // bodies); the plan is only made when they are dealt out again
    if (w->cluster_homes_valid && w->cluster_age < kClusterDynamicPeriod) return true;
    w->cluster_homes_valid = false;
    // the count of the last deal; before the first one: a pile has about as many bodies in contact as it has manifolds
    // (between half as many and twice as many; a world seeded with a mid-fall state - 1M cubes, 360k manifolds, 250k
    // active bodies - never got a first deal with the upper bound). Too low a guess leaves some bodies without a home for
    // one period (slower, never wrong), and the deal itself then counts them.
    uint64_t active = w->hint.n_active;
    if (active == 0) active = std::min<uint64_t>(w->n_owned, (uint64_t)w->hint.n_manifolds);
    if (active == 0) return false;
    static const int spare_div = getenv("PHYS_DEBUG_CLUSTER_SPARE") ? atoi(getenv("PHYS_DEBUG_CLUSTER_SPARE")) : 8;
    static const char* per_cu_env = getenv("PHYS_DEBUG_CLUSTERS_PER_CU");
    const int per_cu_max = w->all_diag_inertia ? kClusterPerCuDiag : kClusterPerCuFull;
    int per_cu = per_cu_env ? std::min(per_cu_max, std::max(1, atoi(per_cu_env))) : per_cu_max;
    const int cus = w->cluster_cus;
    // homes for a quarter more bodies than the last known count; what does not get one is served as "another cluster's
    // body" (slower, never wrong), so this is a matter of speed only
    uint64_t want = active + active / 4;
    if (w->cluster_cap_limit && want > w->cluster_cap_limit) want = w->cluster_cap_limit;
    for (;; --per_cu) {
        const uint32_t clusters = (uint32_t)std::max(8, per_cu * (cus - (spare_div ? cus / spare_div : 0)));
        uint32_t slots = (uint32_t)((want + clusters - 1) / clusters);
        slots = std::max(64u, (slots + 63u) / 64u * 64u);
        const size_t per_wg = (cluster_lds_bytes(slots) + 1023) / 1024 * 1024;
        const bool ok = per_wg * (size_t)per_cu <= kClusterLdsPerCu && slots <= kClusterMaxSlots;
        if (ok || per_cu == 1) {
            // Only while the homes fit with the FULL number of workgroups per CU. Measured on the growing 1M-cube pile: the
            // moment the plan has to go to two or one larger workgroups per CU the per-colour launches are faster (2.10
            // against 2.33 ms at 430k active bodies, 2.66 against 3.30 at 500k; with half the bodies homeless 3.61 against
            // 4.08) - fewer workgroups hide less of each other's colour steps. (A capacity set for tests is obeyed.)
            if ((!ok || per_cu < per_cu_max) && !w->cluster_cap_limit && !per_cu_env) return false;
            // (PHYS_DEBUG_CLUSTERS_PER_CU asks for fewer, larger workgroups - never for homes that do not fit: with the
            // switch set, the growing 1M-cube pile once ran one 160 KiB workgroup per CU with half its bodies homeless and
            // ended in the hand-off time-out)
            if (!ok && !w->cluster_cap_limit) return false;
            if (!ok) slots = kClusterMaxSlots / 64u * 64u;  // one workgroup per CU, as many homes as its LDS holds
            w->cluster_count = clusters;
            w->cluster_slots = slots;
            return true;
        }
    }
}

// position of an owned body in the broad phase's bucket order (what k_scatter computed), or ~0
__device__ __forceinline__ uint32_t bucket_order(uint32_t i, const uint32_t* __restrict__ bucket_of, const uint32_t* __restrict__ rank,
                                                 const uint32_t* __restrict__ bucket_start) {
    const uint32_t bk = bucket_of[i];
    return bk == 0xFFFFFFFFu ? 0xFFFFFFFFu : bucket_start[bk] + rank[i];
}
__global__ __launch_bounds__(256) void k_active_flags(uint32_t n_owned, const unsigned long long* __restrict__ used,
                                                      const uint32_t* __restrict__ bucket_of, const uint32_t* __restrict__ rank,
                                                      const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_owned || used[i] == 0ull) return;  // flag[] was zeroed
    const uint32_t s = bucket_order(i, bucket_of, rank, bucket_start);
    if (s != 0xFFFFFFFFu) flag[s] = 1u;
}
__global__ __launch_bounds__(256) void k_cluster_homes(uint32_t n, uint32_t n_owned, const unsigned long long* __restrict__ used,
                                                       const uint32_t* __restrict__ bucket_of, const uint32_t* __restrict__ rank,
                                                       const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ active_rank,
                                                       uint32_t total_at, uint32_t homes, uint32_t* __restrict__ cluster_slot,
                                                       uint32_t* __restrict__ cluster_body, StepCounters* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) ctr->n_active = active_rank[total_at];  // the scan's grand total
    if (i >= n) return;
    uint32_t home = kNoHome;
    if (i < n_owned && used[i] != 0ull) {
        const uint32_t s = bucket_order(i, bucket_of, rank, bucket_start);
        if (s != 0xFFFFFFFFu) {
            const uint32_t r = active_rank[s];
            if (r < homes) { home = r; cluster_body[r] = i; }
        }
    }
    cluster_slot[i] = home;
}

// ---- per step: rows sorted by (owner cluster, colour); bodies touched by a row of another cluster ----------
constexpr int kKeysItems = 4;        // manifolds per lane and trip
constexpr int kKeysSlots = 2048;     // LDS table of the (cluster, colour) keys of one trip (<= 1024 of them)
__global__ __launch_bounds__(256) void k_cluster_keys(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                      const uint32_t* __restrict__ man_b, const uint32_t* __restrict__ man_color,
                                                      const uint32_t* __restrict__ cluster_slot, uint32_t slots, uint32_t clusters,
                                                      uint32_t* __restrict__ seg_count, uint32_t* __restrict__ man_rank,
                                                      uint32_t* __restrict__ body_shared, StepCounters* __restrict__ ctr) {
    // manifolds per colour and the number of colours ride along (a cluster step has no use for the colour-major sort
    // that k_color_hist / k_color_offsets serve, only for these counters): one LDS histogram per workgroup
    __shared__ uint32_t s_hist[PHYS_MAX_COLORS];
    // The rank of a manifold inside its (owner cluster, colour) segment is an arrival order nothing depends on, so it need
    // not come from one returning global atomic per manifold (C5: 755k of them, 41 us): the manifolds of a trip are
    // neighbours in space and share a few dozen keys, which are counted in an LDS table first; one global atomic per key
    // and trip then reserves the whole count.
    __shared__ uint32_t t_key[kKeysSlots], t_cnt[kKeysSlots], t_base[kKeysSlots];
    if (threadIdx.x < PHYS_MAX_COLORS) s_hist[threadIdx.x] = 0u;
    const uint32_t raw = ctr->n_manifolds;
    const uint32_t M = (uint64_t)raw < max_manifolds ? raw : (uint32_t)max_manifolds;
    constexpr uint32_t kTrip = 256u * kKeysItems;
    for (uint64_t base = (uint64_t)blockIdx.x * kTrip; base < M; base += (uint64_t)gridDim.x * kTrip) {
        for (uint32_t s = threadIdx.x; s < (uint32_t)kKeysSlots; s += 256u) { t_key[s] = 0xFFFFFFFFu; t_cnt[s] = 0u; }
        __syncthreads();
        uint32_t my_slot[kKeysItems], my_rank[kKeysItems];
#pragma unroll
        for (int j = 0; j < kKeysItems; ++j) {
            const uint64_t m = base + (uint64_t)j * 256u + threadIdx.x;
            my_slot[j] = 0xFFFFFFFFu; my_rank[j] = 0u;
            if (m >= M) continue;
            const uint32_t a = man_a[m], b = man_b[m], c = man_color[m];
            if (c >= (uint32_t)PHYS_MAX_COLORS) continue;  // (man_rank[m] = ~0 below)
            atomicAdd(&s_hist[c], 1u);
            const uint32_t ha = cluster_home(cluster_slot, a, slots);
            const uint32_t hb = b == PHYS_GROUND_ID ? kNoHome : cluster_home(cluster_slot, b, slots);
            const uint32_t owner = cluster_row_owner(a, ha, hb, clusters);
            const uint32_t key = owner * PHYS_MAX_COLORS + c;
            uint32_t s = (key * 2654435761u) >> (32 - 11);  // 2048 slots, at most 1024 keys: the walk ends
            for (;;) {
                const uint32_t seen = atomicCAS(&t_key[s], 0xFFFFFFFFu, key);
                if (seen == 0xFFFFFFFFu || seen == key) break;
                s = (s + 1u) & (uint32_t)(kKeysSlots - 1);
            }
            my_slot[j] = s;
            my_rank[j] = atomicAdd(&t_cnt[s], 1u);
            // a body with a home receives updates from ANOTHER workgroup exactly in the rows that its home does not own:
            // that body is `shared`, and the colours of those rows are its remote colours (two halves of a 64-bit mask).
            // (A's home owns the row whenever A has one.)
            if (hb != kNoHome && hb != owner) atomicOr(&body_shared[2 * (size_t)b + (c >> 5)], 1u << (c & 31u));
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < (uint32_t)kKeysSlots; s += 256u)
            if (t_cnt[s]) t_base[s] = atomicAdd(&seg_count[t_key[s]], t_cnt[s]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKeysItems; ++j) {
            const uint64_t m = base + (uint64_t)j * 256u + threadIdx.x;
            if (m < M) man_rank[m] = my_slot[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : t_base[my_slot[j]] + my_rank[j];
        }
        __syncthreads();  // the table is wiped at the top of the next trip
    }
    __syncthreads();
    // (PHYS_MAX_COLORS == 64: one wave) ONE atomicMax per workgroup - every lane with a count doing its own was up to 64
    // same-address atomics per workgroup
    if (threadIdx.x < PHYS_MAX_COLORS) {
        const uint32_t cnt = s_hist[threadIdx.x];
        if (cnt) atomicAdd(&ctr->color_count[threadIdx.x], cnt);
        const unsigned long long live = __ballot(cnt != 0u);
        if (threadIdx.x == 0 && live) atomicMax(&ctr->n_colors, 64u - (uint32_t)__clzll((long long)live));
    }
}

__global__ __launch_bounds__(256) void k_cluster_place(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                       const uint32_t* __restrict__ man_b, const uint32_t* __restrict__ man_color,
                                                       const uint32_t* __restrict__ cluster_slot, uint32_t slots, uint32_t clusters,
                                                       const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ man_rank,
                                                       uint32_t* __restrict__ row_src, const StepCounters* __restrict__ ctr,
                                                       StepCounters* snap_out /* host-mapped, may be null */) {
    counters_snapshot(ctr, snap_out);  // (k_cluster_keys made the last of them: colours and their counts)
    const uint32_t raw = ctr->n_manifolds;
    const uint32_t M = (uint64_t)raw < max_manifolds ? raw : (uint32_t)max_manifolds;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const uint32_t r = man_rank[m];
        if (r == 0xFFFFFFFFu) continue;
        const uint32_t a = man_a[m], b = man_b[m];
        const uint32_t ha = cluster_home(cluster_slot, a, slots);
        const uint32_t hb = (ha != kNoHome || b == PHYS_GROUND_ID) ? kNoHome : cluster_home(cluster_slot, b, slots);
        row_src[seg_start[cluster_row_owner(a, ha, hb, clusters) * PHYS_MAX_COLORS + man_color[m]] + r] = m;
    }
}


// called by launch_coloring in place of the colour-major placement
void launch_cluster_sort(phys_world* w, unsigned blocks, StepCounters* snap_out) {
    hipStream_t s = w->stream;
    const uint32_t bins = w->cluster_count * PHYS_MAX_COLORS;  // a multiple of 64
    PHYS_PROF(w, PHYS_STAGE_ROWS);
    if (w->cluster_dynamic && w->cluster_homes_valid) ++w->cluster_age;
    if (w->cluster_dynamic && !w->cluster_homes_valid) {
        w->cluster_homes_valid = true;
        w->cluster_age = 0;
        // homes from this update on: the bodies that have a manifold in it (used mask != 0, complete after the colouring), in the
        // bucket order of this update's broad phase - neighbours in space are neighbours in that order
        const uint32_t n = (uint32_t)w->n, homes = w->cluster_count * w->cluster_slots;
        const dim3 g((n + 255) / 256), b(256);
        (void)hipMemsetAsync(w->active_flag.p, 0, ((size_t)n + 4) * 4, s);
        (void)hipMemsetAsync(w->cluster_body.p, 0xFF, (size_t)homes * 4, s);
        hipLaunchKernelGGL(k_active_flags, g, b, 0, s, (uint32_t)w->n_owned, w->color_state.p, w->bucket_of.p, w->bucket_cursor.p,
                           w->bucket_start.p, w->active_flag.p);
        const uint32_t n4 = (n + 3u) & ~3u;  // the scans want a multiple of four; the total lands behind the last entry
        launch_exclusive_scan(w, w->active_flag.p, n4, w->active_rank.p, false);
        hipLaunchKernelGGL(k_cluster_homes, g, b, 0, s, n, (uint32_t)w->n_owned, w->color_state.p, w->bucket_of.p, w->bucket_cursor.p,
                           w->bucket_start.p, w->active_rank.p, n4, homes, w->cluster_slot.p, w->cluster_body.p, w->counters.p);
    }
    // the per-(cluster, colour) counters: all zero on entry - at first use by the allocation's memset (cluster_assign), later
    // because the one-launch scan leaves them zeroed behind it (one memset launch less per update)
    // (the three-launch scan of a larger table leaves its counters as they are: zeroed here then, and once more should a
    // later update come with a table small enough to zero itself)
    const bool self_zeroing = scan_is_one_launch(bins) && !w->seg_count_dirty;
    w->seg_count_dirty = !scan_is_one_launch(bins);
    uint32_t* seg_count = w->body_shared.p + seg_count_offset(w->n);
    (void)hipMemsetAsync(w->body_shared.p, 0, self_zeroing ? (size_t)w->n * 8 : (seg_count_offset(w->n) + (size_t)std::max<uint32_t>(bins, w->seg_count_bins)) * 4, s);
    w->seg_count_bins = bins;
    // (a trip of k_cluster_keys is 1024 manifolds: as many workgroups as the last known count needs, any number is correct)
    const uint64_t key_trips = w->hint.valid ? ((uint64_t)w->hint.n_manifolds * 5 / 4) / (256u * kKeysItems) + 1 : blocks;
    const unsigned key_blocks = (unsigned)std::min<uint64_t>(blocks, std::max<uint64_t>(1, key_trips));
    hipLaunchKernelGGL(k_cluster_keys, dim3(key_blocks), dim3(256), 0, s, w->max_manifolds, w->man_a.p, w->man_b.p, w->man_color.p,
                       w->cluster_slot.p, w->cluster_slots, w->cluster_count, seg_count, w->man_rank.p, w->body_shared.p, w->counters.p);
    launch_exclusive_scan(w, seg_count, bins, w->seg_start.p, scan_is_one_launch(bins));
    hipLaunchKernelGGL(k_cluster_place, dim3(blocks), dim3(256), 0, s, w->max_manifolds, w->man_a.p, w->man_b.p, w->man_color.p,
                       w->cluster_slot.p, w->cluster_slots, w->cluster_count, w->seg_start.p, w->man_rank.p, w->row_src.p, w->counters.p, snap_out);
}

// ---- the solver ----------------------------------------------------------------------------------------------
typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4c ld_gran(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    // sc0 | sc1 (what the volatile form emits): read past this XCD's L2. With sc1 alone ("agent scope") a poll could keep
    // hitting a line its own XCD had cached before the other XCD's write-through store landed: one run in two of two
    // worlds stepping side by side ended in the hand-off time-out (tools/ghost_cluster_stress.py), none with this form -
    // and the scope made no difference in time.
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, (int)0x80000010);
}
__device__ __forceinline__ void st_gran(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, v3 v, uint32_t tag) {
    u32x4c g;
    g.x = __float_as_uint(v.x); g.y = __float_as_uint(v.y); g.z = __float_as_uint(v.z); g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, byte_off, 0, 16);  // sc1: write-through
}

// Rows of a cluster step are COMPACT (k_rows_build writes them so when cluster_slots != 0): what the solver streams
// per iteration is 9 planes of 16 bytes instead of 16 -
//     plane 0      hdr {body a, body b, point count, tickets}
//     plane 1      n   {normal, per-side slot | publish | mode}
//     planes 2-5   geo {contact point k (world), bias k}
//     planes 6-8   acc the 12 accumulated impulses {pn, pt0, pt1} x 4, packed
//     planes 9-11  the 12 row masses {t1, t2, n} x 4, packed: written by the solver itself in iteration 0, read after
//     planes 12-13 rows whose body B belongs to ANOTHER cluster: {position of B, 1/m} {inverse inertia diagonal of B}
//                  (what never changes during a solve; gathering it by body id after the row has arrived was a second
//                  dependent round trip to memory in the chain of every colour step: 2.08 -> ~1.2 ms on C5)
// and everything else solver_prep would have stored (the lever arms) is remade from the contact point and the two body
// POSITIONS, which live in LDS beside the velocities (solve_manifold_geo: same operations, same bits;
// tests/test_collide_kat.py holds the three drivers against each other). So k_rows_build reads nothing of the bodies
// on a cluster step, and a 4-point row is 240 bytes per iteration instead of 320.
struct ClusterRowArrays { float4* all; uint64_t cap; };
constexpr int kClusterPlaneHdr = 0, kClusterPlaneN = 1, kClusterPlaneGeo = 2, kClusterPlaneAcc = 6, kClusterPlaneMass = 9, kClusterPlaneForeign = 12;
constexpr uint32_t kClusterSlotBytes = 64;  // LDS per body slot: {v, tag} {w, 1/m} {x, -} {inverse inertia diagonal, -}
size_t cluster_lds_bytes(uint32_t slots) { return (size_t)slots * kClusterSlotBytes + (PHYS_MAX_COLORS + 1) * 4 + 12; }

constexpr int kClusterThreads = 256;
#ifndef PHYS_POLL_SLEEP
#define PHYS_POLL_SLEEP 8
#endif
constexpr int kPollSleep = PHYS_POLL_SLEEP;
template <bool DIAG, bool GUARDED>
__global__ __launch_bounds__(kClusterThreads, DIAG ? kClusterPerCuDiag : kClusterPerCuFull) void k_solve_cluster(StepCounters* ctr, uint32_t iterations, uint32_t epoch,
                                                                  ClusterRowArrays rows, float friction,
                                                                  const float* __restrict__ inv_inertia, uint32_t inertia_stride,
                                                                  float* vel, const float* __restrict__ pos, float* flow_vel, uint32_t n_bodies,
                                                                  const uint32_t* __restrict__ cluster_body,
                                                                  const uint32_t* __restrict__ body_shared,
                                                                  const uint32_t* __restrict__ seg_start, uint32_t slots,
                                                                  long long timeout_ticks, uint32_t ablate, uint32_t attempt,
                                                                  uint32_t last_attempt, long long arrive_ticks,
                                                                  uint32_t warm_sweep /* sweep 0 applies the starting impulses (and
                                                                  `iterations` counts it) */, const uint32_t* __restrict__ row_src,
                                                                  float* __restrict__ man_imp /* 12 floats per manifold, or null */) {
    extern __shared__ __attribute__((aligned(16))) float4 s_lds[];  // [4 * slots]: {v, tag} {w, 1/m} {x} {I^-1 diag};  then the segment table
    float4* s_body = s_lds;
    uint32_t* s_seg = reinterpret_cast<uint32_t*>(s_lds + 4 * (size_t)slots);  // PHYS_MAX_COLORS + 1 row offsets
    if (ctr->overflow) return;
    if (GUARDED && attempt != 0u && ctr->cluster_state[attempt - 1u] == 1u) return;  // an earlier attempt went through (see below)
    const uint32_t cluster = blockIdx.x;
    const uint32_t n_colors = ctr->n_colors;
    const uint32_t etag = epoch << 16;
    const size_t cap = (size_t)rows.cap;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(flow_vel, 0, n_bodies * 32u, 0x00020000);
    // own bodies into LDS; tag "no update applied yet"
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        float4 a = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(etag)), b = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c = b, e = b;
        if (body != 0xFFFFFFFFu) {
            const float4 v0 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body];
            const float4 v1 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body + 1];
            const v3 x = ld3(pos, body);
            a = make_float4(v0.x, v0.y, v0.z, __uint_as_float(etag));
            b = make_float4(v1.x, v1.y, v1.z, v0.w);
            c = make_float4(x.x, x.y, x.z, 0.0f);
            if (DIAG) e = reinterpret_cast<const float4*>(inv_inertia)[body * inertia_stride];
        }
        s_body[4 * sl] = a;
        s_body[4 * sl + 1] = b;
        s_body[4 * sl + 2] = c;
        s_body[4 * sl + 3] = e;
    }
    if (threadIdx.x <= (uint32_t)PHYS_MAX_COLORS) s_seg[threadIdx.x] = seg_start[(size_t)cluster * PHYS_MAX_COLORS + threadIdx.x];
    // ---- ALL OR NOTHING. Every workgroup of this launch must be resident at once (they wait for each other's updates), and
    // whether they are is not a matter of counts alone: registers are handed out in contiguous ranges, so workgroups of
    // ANOTHER stream's kernels that run beside the start of this launch leave holes between this kernel's waves that never
    // close (its waves stay until the end), and a CU then takes two of its workgroups instead of three - measured: in
    // half of the runs of two worlds stepping side by side 1-2 of 520 workgroups never started, and the rest spun until the
    // time-out (tools/ghost_cluster_stress.py; never when the launch began on an idle device; a cooperative launch does not
    // help). So a launch first counts its workgroups in; nothing is written before ONE decision is made for all: go, once
    // all have begun, or - if they have not within `arrive_ticks` - called off: everybody leaves, the holes close, and the
    // next attempt (the host enqueues a few; the ones after a `go` return at once) starts on cleaner ground. Only the last
    // attempt's failure is an error (overflow bit 4).
    // (Counted in after the bodies have been read into LDS - that writes nothing outside the workgroup and overlaps the
    // arrival of the others.)
    if (!GUARDED) {
        __syncthreads();  // the bodies are in LDS
    } else {
        __shared__ uint32_t s_go;
        if (threadIdx.x == 0) {
            bool solved = false;
            for (uint32_t a = 0; a < attempt; ++a) solved = solved || __hip_atomic_load(&ctr->cluster_state[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
            uint32_t st = solved ? 2u : (arrive_ticks < 0 ? 1u : 0u);
            if (!solved && arrive_ticks >= 0) {
                atomicAdd(&ctr->cluster_arrived[attempt][blockIdx.x & 7u], 1u);
                const long long t0 = wall_clock64();
                for (;;) {
                    st = __hip_atomic_load(&ctr->cluster_state[attempt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (st != 0u) break;
                    uint32_t in = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) in += __hip_atomic_load(&ctr->cluster_arrived[attempt][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (in >= gridDim.x) {
                        atomicCAS(&ctr->cluster_state[attempt], 0u, 1u);
                    } else if (wall_clock64() - t0 > arrive_ticks) {
                        if (atomicCAS(&ctr->cluster_state[attempt], 0u, 2u) == 0u && last_attempt) flag_overflow(ctr, 16u);
                    } else {
                        __builtin_amdgcn_s_sleep(16);
                    }
                }
            }
            s_go = st == 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!s_go) return;
    }
    const long long t_start = wall_clock64();
    bool dead = false;
    // (Measured and dropped, all bit-identical: touching the next rows into the L2, 3.18 vs 2.52 ms on C5; two lanes per
    // row, the halves of the relative velocity exchanged by DPP, four workgroups per CU: 2.99 vs 2.20 - 1.5x the load
    // instructions and the scalar part of every row done twice; rotating the wave a colour segment starts on: no change;
    // FOUR lanes per row (lane q owns one of vA, wA, vB, wB as in k_solve_flow_quad) for scenes with 25-60 rows per cluster
    // and colour step: 0.74 vs 0.60 ms on C3, 1.15 vs 0.62 on the 1M cubes - a quad's second row in one step is not fetched
    // ahead, and four times the lanes issue four times the loads, polls and LDS traffic for a chain that the hand-off, not
    // the arithmetic, dominates.)
    //
    // ROWS ARE FETCHED ONE OR TWO COLOUR STEPS AHEAD. A colour step of a cluster has fewer rows than the workgroup has
    // lanes (C5: ~130 of 256), and what a step costs is a chain - row fetch (2-3 us from HBM), LDS, ~1500 dependent
    // instructions, publish, barrier - that the few workgroups of a CU cannot hide from each other (ablation at three per
    // CU: 5.0 us of the 7.6 us per step remain with neither arithmetic nor granules). So the rows of the cluster, which are
    // contiguous and in colour order, are dealt to the lanes round robin: lane l solves rows base + l, base + l + 256, ... of
    // every iteration, and asks for its next row the moment it has finished one - one or two steps before that row's colour
    // comes up. The row waits in the registers it would occupy anyway. The barrier between steps is a bare s_barrier behind
    // an LDS wait (inline asm): __syncthreads() would make every wave wait for the loads it has just issued. Only LDS
    // traffic has to be ordered by it; the planes a lane reads back (impulses, masses) it wrote itself.
    const uint32_t steps = iterations * n_colors;
    const uint32_t row_base = s_seg[0], row_end = s_seg[n_colors];
    uint32_t d = row_base + threadIdx.x;  // this lane's next row ...
    uint32_t lit = 0;                     // ... and the iteration it belongs to
    bool have = d < row_end && iterations != 0u;
    float4 hraw, nn, geo[4], acc[3], mas[3], fb, fi;
    auto fetch = [&]() {
        // every plane at once, whatever the point count turns out to be (planes beyond it hold stale, readable data that
        // is ignored): ONE round trip to memory
        const float4* row = rows.all + d;
        hraw = row[kClusterPlaneHdr * cap];
        nn = row[kClusterPlaneN * cap];
        fb = row[kClusterPlaneForeign * cap];  // stale (and ignored) unless body B is another cluster's
        if (DIAG) fi = row[(kClusterPlaneForeign + 1) * cap];
#pragma unroll
        for (int k = 0; k < 4; ++k) geo[k] = row[(kClusterPlaneGeo + k) * cap];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            acc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            mas[k] = acc[k];
            // the impulses: written by this very lane one sweep ago - or, for sweep 0 of a warm-started solve, by
            // k_rows_build (the starting impulses); the masses: by this lane in sweep 0
            if (lit != 0u || warm_sweep != 0u) acc[k] = row[(kClusterPlaneAcc + k) * cap];
            if (lit != 0u) mas[k] = row[(kClusterPlaneMass + k) * cap];
        }
    };
    hraw = make_float4(0.0f, 0.0f, 0.0f, 0.0f); nn = hraw; fb = hraw; fi = hraw;
#pragma unroll
    for (int k = 0; k < 4; ++k) geo[k] = hraw;
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[k] = hraw; mas[k] = hraw; }
    if (have) fetch();
    uint32_t it = 0, col = 0;
    for (uint32_t step = 0; step < steps; ++step) {
        const bool last_it = it + 1 == iterations;
        {
            const uint32_t seg_hi = s_seg[col + 1];
            // the lane's row is never of an earlier colour of this iteration (it would have been solved then)
            while (have && lit == it && d < seg_hi) {
                const uint32_t d_row = d;
                const uint4 h = make_uint4(__float_as_uint(hraw.x), __float_as_uint(hraw.y), __float_as_uint(hraw.z), __float_as_uint(hraw.w));
                const uint32_t info = __float_as_uint(nn.w);
                // slot (13 bits) | publish (1) | mode (2) per side. publish: the NEXT update of this (shared, own) body is
                // made by another workgroup, so this update must reach the granules; otherwise it stays in LDS
                uint32_t modeA = (info >> 14) & 3u, modeB = (info >> 30) & 3u;
                const uint32_t slotA = info & 0x1FFFu, slotB = (info >> 16) & 0x1FFFu;
                const bool pubA = (info >> 13) & 1u, pubB = (info >> 29) & 1u;
                if (ablate & 8u) { if (modeA == 1u) modeA = 0u; if (modeB == 1u) modeB = 0u; }  // PHYS_DEBUG_ABLATE (timing only, wrong results)
                geo_manifold_t gm;
                gm.count = (int)h.z;
                gm.has_b = h.y != PHYS_GROUND_ID;
                gm.n = v3_make(nn.x, nn.y, nn.z);
                tangent_basis(gm.n, &gm.t1, &gm.t2);
                {
                    const float flat[12] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w, acc[1].x, acc[1].y, acc[1].z, acc[1].w,
                                            acc[2].x, acc[2].y, acc[2].z, acc[2].w};
                    const float flam[12] = {mas[0].x, mas[0].y, mas[0].z, mas[0].w, mas[1].x, mas[1].y, mas[1].z, mas[1].w,
                                            mas[2].x, mas[2].y, mas[2].z, mas[2].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gm.pt[k] = v3_make(geo[k].x, geo[k].y, geo[k].z);
                        gm.bias[k] = geo[k].w;
                        gm.pn[k] = flat[3 * k]; gm.pt0[k] = flat[3 * k + 1]; gm.pt1[k] = flat[3 * k + 2];
                        gm.mass[k][0] = flam[3 * k]; gm.mass[k][1] = flam[3 * k + 1]; gm.mass[k][2] = flam[3 * k + 2];
                    }
                }
                const uint32_t rankA = h.w & 0xFFu, degA = (h.w >> 8) & 0xFFu, rankB = (h.w >> 16) & 0xFFu, degB = h.w >> 24;
                const uint32_t tA = it * degA + rankA, tB = it * degB + rankB;
                const bool finalA = last_it && rankA + 1 == degA, finalB = last_it && rankB + 1 == degB;
                // ---- the two bodies
                v3 vA, wA, xA, vB = v3_make(0.0f, 0.0f, 0.0f), wB = vB, xB = vB;
                float ima, imb = 0.0f;
                bool needA = false, needB = false;
                m33 IA, IB;
#pragma unroll
                for (int k = 0; k < 9; ++k) { IA.m[k] = 0.0f; IB.m[k] = 0.0f; }
                v3 vA0 = v3_make(0.0f, 0.0f, 0.0f);
                vA = vA0; wA = vA0; xA = vA0; ima = 0.0f;
                if (modeA == 0u || modeA == 1u) {  // A at home here (always, unless it has no home at all)
                    const float4* sa = s_body + 4 * slotA;
                    const float4 la = sa[0], lb = sa[1], lc = sa[2];
                    vA = v3_make(la.x, la.y, la.z); wA = v3_make(lb.x, lb.y, lb.z); ima = lb.w;
                    xA = v3_make(lc.x, lc.y, lc.z);
                    if (DIAG) { const float4 li = sa[3]; IA.m[0] = li.x; IA.m[4] = li.y; IA.m[8] = li.z; }
                    if (modeA == 1u) needA = __float_as_uint(la.w) != (etag | tA);  // a remote row made the update before this one
                } else {
                    // a body without a home (dynamic clusters beyond their capacity): like a foreign body B, its constants
                    // came with the row (planes 14, 15) - fetched here, not ahead: the case is rare by construction
                    const float4 fa = rows.all[(size_t)(kClusterPlaneForeign + 2) * cap + d_row];
                    xA = v3_make(fa.x, fa.y, fa.z); ima = fa.w;
                    if (DIAG) {
                        const float4 fia = rows.all[(size_t)(kClusterPlaneForeign + 3) * cap + d_row];
                        IA.m[0] = fia.x; IA.m[4] = fia.y; IA.m[8] = fia.z;
                    }
                    if (tA == 0u) { const BodyVel A0 = ld_vel(vel, h.x); vA = A0.v; wA = A0.w; }
                    needA = tA != 0u && !(ablate & 8u);
                }
                if (modeB == 0u || modeB == 1u) {
                    const float4* sb = s_body + 4 * slotB;
                    const float4 la = sb[0], lb = sb[1], lc = sb[2];
                    vB = v3_make(la.x, la.y, la.z); wB = v3_make(lb.x, lb.y, lb.z); imb = lb.w;
                    xB = v3_make(lc.x, lc.y, lc.z);
                    if (DIAG) { const float4 li = sb[3]; IB.m[0] = li.x; IB.m[4] = li.y; IB.m[8] = li.z; }
                    if (modeB == 1u) needB = __float_as_uint(la.w) != (etag | tB);
                } else if (modeB == 2u) {
                    xB = v3_make(fb.x, fb.y, fb.z); imb = fb.w;  // came with the row
                    if (DIAG) { IB.m[0] = fi.x; IB.m[4] = fi.y; IB.m[8] = fi.z; }
                    if (tB == 0u) {  // first update of that body in this solve (iteration 0 only): its state is still in `vel`
                        const BodyVel B0 = ld_vel(vel, h.y);
                        vB = B0.v; wB = B0.w;
                    }
                    needB = tB != 0u && !(ablate & 8u);
                }
                if (!DIAG) {  // full tensors: gathered by body id (the rare path keeps its second round trip)
                    IA = ld_inertia_c<false>(inv_inertia, h.x * inertia_stride);
                    if (gm.has_b) IB = ld_inertia_c<false>(inv_inertia, h.y * inertia_stride);
                }
                uint32_t sweeps = 0, late = 0;
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
                        __builtin_amdgcn_s_sleep(kPollSleep);
                        if ((++sweeps & 63u) == 0u) {
                            const bool gone = (wall_clock64() - t_start > timeout_ticks) ||
                                              (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u);
                            if (gone) {
                                late = (needA ? 1u : 0u) | (needB ? 2u : 0u);
                                flag_overflow(ctr, 16u); dead = true; needA = false; needB = false;
                            }
                        }
                    }
                }
                if (GUARDED && late && atomicCAS(&ctr->debug[0], 0u, 0xC1u) == 0u) {
                    // the first lane to give up says what it waited for (outside the poll loop, and without reading the granules again:
                    // cold code in the row path costs the whole kernel 1-2 % at its register limit - so only in the guarded variant,
                    // where a time-out is a defect; in the other one it means PHYS_FLAG_EXCLUSIVE_GPU was set on a GPU that is shared)
                    ctr->debug[1] = cluster | (gridDim.x << 16);
                    ctr->debug[2] = d_row; ctr->debug[3] = h.x; ctr->debug[4] = h.y;
                    ctr->debug[5] = (tA & 0xFFFFu) | (tB << 16);
                    ctr->debug[6] = late | (modeA << 4) | (modeB << 8);
                    ctr->debug[7] = (col & 0xFFu) | ((it & 0xFFu) << 8);
                }
                if (!dead) {
                    if (!(ablate & 2u))
                        solve_manifold_geo(&gm, it == 0u || (ablate & 32u), (warm_sweep != 0u && it == 0u) ? 1 : 0, friction, xA, ima, &IA, xB, imb,
                                           &IB, &vA, &wA, &vB, &wB);
                    // ---- write back
                    if (modeA == 0u || modeA == 1u) {
                        s_body[4 * slotA] = make_float4(vA.x, vA.y, vA.z, __uint_as_float(etag | (tA + 1u)));
                        s_body[4 * slotA + 1] = make_float4(wA.x, wA.y, wA.z, ima);
                    }
                    if (modeA == 1u || modeA == 2u) {
                        if (finalA) { st3(vel + 8 * (size_t)h.x, 0, vA); st3(vel + 8 * (size_t)h.x + 4, 0, wA); }  // the masses stay where they are
                        else if (modeA == 2u || pubA) { st_gran(rv, h.x * 32u, vA, etag | (tA + 1u)); st_gran(rv, h.x * 32u + 16u, wA, etag | (tA + 1u)); }
                    }
                    if (modeB == 0u || modeB == 1u) {
                        s_body[4 * slotB] = make_float4(vB.x, vB.y, vB.z, __uint_as_float(etag | (tB + 1u)));
                        s_body[4 * slotB + 1] = make_float4(wB.x, wB.y, wB.z, imb);
                    }
                    if (modeB == 1u || modeB == 2u) {
                        if (finalB) { st3(vel + 8 * (size_t)h.y, 0, vB); st3(vel + 8 * (size_t)h.y + 4, 0, wB); }
                        else if (modeB == 2u || pubB) { st_gran(rv, h.y * 32u, vB, etag | (tB + 1u)); st_gran(rv, h.y * 32u + 16u, wB, etag | (tB + 1u)); }
                    }
                    if (last_it && man_imp) {  // the solve's last sweep: remembered for the next update (contact_solve.h)
                        float4* o = reinterpret_cast<float4*>(man_imp) + 3 * (size_t)row_src[d_row];
                        o[0] = make_float4(gm.pn[0], gm.pt0[0], gm.pt1[0], gm.pn[1]);
                        o[1] = make_float4(gm.pt0[1], gm.pt1[1], gm.pn[2], gm.pt0[2]);
                        o[2] = make_float4(gm.pt1[2], gm.pn[3], gm.pt0[3], gm.pt1[3]);
                    }
                    if (!last_it) {  // impulses of points beyond the count are whatever came in: never used
                        float4* out = rows.all + d_row;
                        out[(kClusterPlaneAcc + 0) * cap] = make_float4(gm.pn[0], gm.pt0[0], gm.pt1[0], gm.pn[1]);
                        if (gm.count > 1) out[(kClusterPlaneAcc + 1) * cap] = make_float4(gm.pt0[1], gm.pt1[1], gm.pn[2], gm.pt0[2]);
                        if (gm.count > 2) out[(kClusterPlaneAcc + 2) * cap] = make_float4(gm.pt1[2], gm.pn[3], gm.pt0[3], gm.pt1[3]);
                        if (it == 0u) {
                            out[(kClusterPlaneMass + 0) * cap] = make_float4(gm.mass[0][0], gm.mass[0][1], gm.mass[0][2], gm.mass[1][0]);
                            if (gm.count > 1) out[(kClusterPlaneMass + 1) * cap] = make_float4(gm.mass[1][1], gm.mass[1][2], gm.mass[2][0], gm.mass[2][1]);
                            if (gm.count > 2) out[(kClusterPlaneMass + 2) * cap] = make_float4(gm.mass[2][2], gm.mass[3][0], gm.mass[3][1], gm.mass[3][2]);
                        }
                    }
                }
                // the lane's next row: 256 further on, or its first one again in the next iteration
                d += kClusterThreads;
                if (d >= row_end) {
                    d = row_base + threadIdx.x; ++lit;
                    // a cluster with fewer rows than lanes: the next row is the one just solved, and its impulses and
                    // masses, stored a few instructions ago, are about to be read back - loads are not ordered behind
                    // stores unless waited for
                    if (d == d_row) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                have = lit < iterations && !dead;
                if (have) fetch();
            }
            // LDS velocities of this colour are in place before the next colour reads them (see above: not __syncthreads).
            // A colour in which this cluster owns no row wrote nothing: no barrier (the barrier orders this workgroup's LDS
            // traffic only - other workgroups are waited for through the granules' tags; the trailing colours of a
            // colouring are sparse, most clusters have nothing in them)
            if (seg_hi != s_seg[col]) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (++col == n_colors) { col = 0u; ++it; }
    }
    // bodies that never left this workgroup: their final state goes back to `vel` (a shared body's was written by the row
    // that made its last update, wherever that ran)
    for (uint32_t sl = threadIdx.x; sl < slots; sl += kClusterThreads) {
        const uint32_t body = cluster_body[(size_t)cluster * slots + sl];
        if (body == 0xFFFFFFFFu || (body_shared[2 * (size_t)body] | body_shared[2 * (size_t)body + 1])) continue;
        const float4 a = s_body[4 * sl], b = s_body[4 * sl + 1];
        st3(vel + 8 * (size_t)body, 0, v3_make(a.x, a.y, a.z));
        st3(vel + 8 * (size_t)body + 4, 0, v3_make(b.x, b.y, b.z));
    }
}

void launch_solve_cluster(phys_world* w, void* row_all, uint64_t cap, float friction, const float* inertia, uint32_t stride,
                          bool diag, long long timeout_ticks) {
    const uint32_t warm_sweep = w->warm ? 1u : 0u;
    const uint32_t sweeps = w->cfg.solver_iterations + warm_sweep;
    static const uint32_t ablate = getenv("PHYS_DEBUG_ABLATE") ? (uint32_t)atoi(getenv("PHYS_DEBUG_ABLATE")) : 0u;
    ClusterRowArrays rows;
    rows.all = (float4*)row_all; rows.cap = cap;
    const size_t lds = cluster_lds_bytes(w->cluster_slots);
    const dim3 g(w->cluster_count), b(kClusterThreads);
    // Every workgroup of this launch must be resident at once, which the grid size guarantees only if no OTHER launch of
    // this kernel shares the device: two worlds of one process (two streams) would each get part of the CUs and wait for
    // the rest until the time-out. Launches of this kernel are therefore chained per device, across worlds, through one
    // event. (Another PROCESS on the same GPU is beyond this: there the bounded spin ends in PHYS_ERR_HIP.)
    static std::mutex chain_lock;
    static hipEvent_t chain_event[64] = {};
    std::lock_guard<std::mutex> hold(chain_lock);
    {
        // more than the default 64 KiB of dynamic LDS needs the attribute: once per kernel AND DEVICE (function attributes
        // are per device: a world on a second device of this process - phys_comm_create_local - needs its own), under the
        // lock (two host threads), the current device being the world's (ENTER of the ABI call) - ADVICE r2
        static bool attr_set[64] = {};
        if (!attr_set[w->device & 63]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_cluster<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { (void)hipGetLastError(); set_error(std::string("hipFuncSetAttribute(k_solve_cluster): ") + hipGetErrorString(e)); }
            else attr_set[w->device & 63] = true;
        }
    }
    hipEvent_t& ev = chain_event[w->device & 63];
    if (ev) (void)hipStreamWaitEvent(w->stream, ev, 0);
    else (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    struct Record { hipEvent_t e; hipStream_t s; ~Record() { if (e) (void)hipEventRecord(e, s); } } record{ev, w->stream};
    // ... and the all-or-nothing start inside the kernel covers what runs beside a launch from other streams - where that
    // can happen: ANYWHERE, unless the caller says the GPU is this world's alone (PHYS_FLAG_EXCLUSIVE_GPU: its own kernels
    // run one after the other, every launch starts on an idle device, the count - 0.04 ms per update on C5 and on the
    // 1M cubes (tools/guard_cost.py) - is skipped). The guarded start is the default since round 3: a drop-in behind a render loop shares its GPU
    // with the renderer, and the unguarded launch's failure mode there is a 3 s spin.
    const bool guarded = worlds_on_device(w->device) > 1 || !(w->cfg.flags & PHYS_FLAG_EXCLUSIVE_GPU) || (w->cfg.flags & PHYS_FLAG_SHARED_GPU) != 0u;
    const uint32_t kAttempts = guarded ? 2u : 1u;
    for (uint32_t attempt = 0; attempt < kAttempts; ++attempt) {
        const uint32_t last = attempt + 1 == kAttempts ? 1u : 0u;
        // 100 MHz ticks: 0.5 ms, then 20 ms for the workgroups to come in (alone on the device they need ~10 us); < 0: no count
        const long long arrive_ticks = !guarded ? -1ll : (attempt == 0 ? 50000ll : 2000000ll);
#define PHYS_CLUSTER_ARGS g, b, lds, w->stream, w->counters.p, sweeps, w->flow_epoch, rows, friction, inertia, stride, \
                          w->vel.p, w->pos.p, w->flow_vel.p, (uint32_t)w->n, w->cluster_body.p, w->body_shared.p, w->seg_start.p,     \
                          w->cluster_slots, timeout_ticks, ablate, attempt, last, arrive_ticks, warm_sweep, w->row_src.p,              \
                          w->warm ? w->man_imp.p : (float*)nullptr
        if (guarded) {
            if (diag) hipLaunchKernelGGL((k_solve_cluster<true, true>), PHYS_CLUSTER_ARGS);
            else hipLaunchKernelGGL((k_solve_cluster<false, true>), PHYS_CLUSTER_ARGS);
        } else {
            if (diag) hipLaunchKernelGGL((k_solve_cluster<true, false>), PHYS_CLUSTER_ARGS);
            else hipLaunchKernelGGL((k_solve_cluster<false, false>), PHYS_CLUSTER_ARGS);
        }
#undef PHYS_CLUSTER_ARGS
    }
}

}  // namespace phys
