// broadphase.hip — AABB broad phase for gfx950. No reference counterpart; the
// result is specified as the SET of pairs (i < j) whose fattened AABBs overlap (include/spec/collide.h).
//
// Pipeline per step (all sizes live on the device; nothing returns to the host):
//   1. k_cell_assign   body -> cell of a uniform grid whose cell edge is the largest AABB extent of the
//                      step (x1.001), cell -> bucket by 3-D Morton interleave of the low cell bits
//                      (neighbouring cells = neighbouring buckets = neighbouring memory), count per bucket.
//   2. exclusive scan  bucket counts -> bucket starts (one launch up to 32768 buckets, else 3 small kernels).
//   3. k_scatter       bodies grouped by bucket: ids + a bucket-ordered COPY of the AABBs (24 B each), so
//                      the pair kernel streams candidates instead of gathering them.
//   4. k_find_pairs    one lane per body (bucket order), half shell of 14 cells; overlap test; hits are
//                      compacted with wavefront ballot + popcount into a per-wave LDS stage and flushed
//                      with one global atomic per flush (not per hit) and coalesced 8-byte stores.
// The emission order is arbitrary; nothing downstream depends on it (DESIGN.md "determinism").
// Algorithmic bytes: 24 N (each AABB read once) + 8 P (each pair written once).
#include <algorithm>
#include <vector>

#include <cstdlib>

#include "kernels.hpp"

namespace phys {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// ---- bucket of a cell: kernels.hpp (grid_bucket: brick-major, per-axis sizes) -------------------------------------
__device__ __forceinline__ uint32_t bucket_of_cell(int cx, int cy, int cz, const GridShape& g) { return grid_bucket(cx, cy, cz, g); }
__device__ __forceinline__ int cell_coord(float c, float inv_cell) { return grid_cell_coord(c, inv_cell); }
__device__ __forceinline__ float grid_inv_cell(const StepCounters* ctr) {
    const float ext = __uint_as_float(ctr->max_extent_bits);
    const float cell = ext > 0.0f ? ext * 1.001f : 1.0f;
    return 1.0f / cell;
}

__global__ __launch_bounds__(256) void k_cell_assign(uint32_t n, const float* __restrict__ aabb,
                                                     const uint32_t* __restrict__ shape,
                                                     const StepCounters* __restrict__ ctr, GridShape axis_mask,
                                                     uint32_t* __restrict__ bucket_of, uint32_t* __restrict__ rank,
                                                     uint32_t* __restrict__ bucket_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (shape[i] == PHYS_SPEC_SHAPE_NONE) { bucket_of[i] = kInvalid; return; }
    const float inv_cell = grid_inv_cell(ctr);
    const v3 lo = ld3(aabb, 2 * i), hi = ld3(aabb, 2 * i + 1);
    const int cx = cell_coord(0.5f * (lo.x + hi.x), inv_cell);
    const int cy = cell_coord(0.5f * (lo.y + hi.y), inv_cell);
    const int cz = cell_coord(0.5f * (lo.z + hi.z), inv_cell);
    const uint32_t bk = bucket_of_cell(cx, cy, cz, axis_mask);
    bucket_of[i] = bk;
    rank[i] = atomicAdd(&bucket_count[bk], 1u);  // order inside a bucket is irrelevant downstream
}

// ---- exclusive scan of the bucket counts ----------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanChunk = kScanThreads * kScanItems;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// `block_used` (bucket grid only): per block, how many of its counters are non-zero = buckets in use; k_scan_block_sums
// adds them up. Bodies per bucket in use is how CROWDED the grid is, which decides the pair kernel of later updates
// (launch_broadphase). (Counted here, where every counter is read anyway: one atomic per workgroup of k_cell_assign -
// 3906 of them at 1M bodies - cost that kernel 36 us.)
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const uint32_t* __restrict__ in, uint32_t count,
                                                              uint32_t* __restrict__ block_sums,
                                                              uint32_t* __restrict__ block_used = nullptr) {
    __shared__ uint32_t wsum[kScanThreads / 64], wused[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanItems;
    uint32_t s = 0, u = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t v = (base + k < count) ? in[base + k] : 0u;
        s += v;
        u += v != 0u ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += (uint32_t)__shfl_xor((int)s, off, 64); u += (uint32_t)__shfl_xor((int)u, off, 64); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wused[threadIdx.x >> 6] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, tu = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) { t += wsum[k]; tu += wused[k]; }
        block_sums[blockIdx.x] = t;
        if (block_used) block_used[blockIdx.x] = tu;
    }
}

// one block: exclusive scan of the block sums in place (loops with a carry for long inputs)
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ sums, uint32_t count,
                                                          const uint32_t* __restrict__ block_used = nullptr,
                                                          StepCounters* __restrict__ ctr = nullptr) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    if (block_used) {  // buckets in use, summed over the blocks of k_scan_reduce (no atomics on the way)
        uint32_t u = 0;
        for (uint32_t k = threadIdx.x; k < count; k += 1024) u += block_used[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) u += (uint32_t)__shfl_xor((int)u, off, 64);
        if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = u;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int k = 0; k < 16; ++k) t += wtot[k];
            ctr->n_used_buckets = t;
        }
    }
    __syncthreads();
    for (uint32_t base = 0; base < count; base += 1024) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < count ? sums[idx] : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (idx < count) sums[idx] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_final(const uint32_t* __restrict__ in, uint32_t count,
                                                             const uint32_t* __restrict__ block_sums,
                                                             uint32_t* __restrict__ out /*count + 1*/) {
    __shared__ uint32_t wtot[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = (base + k < count) ? in[base + k] : 0u; s += v[k]; }
    const uint32_t inc = wave_inclusive_scan(s);
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = block_sums[blockIdx.x];
    for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) off += wtot[k];
    uint32_t run = off + inc - s;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < count) out[base + k] = run;
        run += v[k];
        if (base + k + 1 == count) out[count] = run;  // grand total in the extra slot
    }
}

// small tables (<= 32768 counters): the whole exclusive scan in ONE workgroup, one launch instead of three
// (measured at 64 items per thread, for the cluster solver's 672 x 64 segment table: 25.6 us - one workgroup's lanes read
// 256-byte runs each, every line is touched by eight load instructions - against 3 x 4.8 us for the three launches)
constexpr int kScanSmallThreads = 1024;
constexpr int kScanSmallItems = 32;
// ZERO_IN: the counters are left zeroed for their next use (a per-step histogram then needs no memset launch of its own)
template <bool ZERO_IN>
__global__ __launch_bounds__(kScanSmallThreads) void k_scan_small(uint32_t* __restrict__ in, uint32_t count /* multiple of 4 */,
                                                                  uint32_t* __restrict__ out /*count + 1*/) {
    __shared__ uint32_t wtot[kScanSmallThreads / 64];
    const uint32_t base = threadIdx.x * kScanSmallItems;
    uint4 v[kScanSmallItems / 4];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScanSmallItems / 4; ++k) {
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (base + 4 * k < count) {
            v[k] = *reinterpret_cast<const uint4*>(in + base + 4 * k);
            if (ZERO_IN) *reinterpret_cast<uint4*>(in + base + 4 * k) = make_uint4(0u, 0u, 0u, 0u);
        }
        sum += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    const uint32_t inc = wave_inclusive_scan(sum);
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) run += wtot[k];
#pragma unroll
    for (int k = 0; k < kScanSmallItems / 4; ++k) {
        if (base + 4 * k < count) {
            uint4 o;
            o.x = run; o.y = o.x + v[k].x; o.z = o.y + v[k].y; o.w = o.z + v[k].z;
            *reinterpret_cast<uint4*>(out + base + 4 * k) = o;
            run = o.w + v[k].w;
        }
    }
    if (threadIdx.x == kScanSmallThreads - 1) {
        uint32_t total = 0;
        for (int k = 0; k < kScanSmallThreads / 64; ++k) total += wtot[k];
        out[count] = total;  // grand total in the extra slot
    }
}

// ---- group bodies by bucket -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter(uint32_t n, const float* __restrict__ aabb,
                                                 const uint32_t* __restrict__ bucket_of,
                                                 const uint32_t* __restrict__ rank,
                                                 const uint32_t* __restrict__ bucket_start,
                                                 uint32_t* __restrict__ sorted_ids, float* __restrict__ sorted_box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bk = bucket_of[i];
    if (bk == kInvalid) return;
    const uint32_t s = bucket_start[bk] + rank[i];
    sorted_ids[s] = i;
    st3(sorted_box, 2 * s, ld3(aabb, 2 * i));
    st3(sorted_box, 2 * s + 1, ld3(aabb, 2 * i + 1));
}

// ---- small scenes: the whole grid build in ONE workgroup ------------------------------------------
// cell assignment, scan of the bucket counts and the scatter as three phases of one launch (barriers instead of
// four kernel boundaries of ~6 us each). Bucket and rank of a body stay in registers between the phases.
constexpr int kGridSmallThreads = 1024;
constexpr int kGridSmallBodies = 2;      // per thread: n <= 2048 (10k bodies: 40 us in one workgroup vs 26 us as three launches)
constexpr int kGridSmallBuckets = 4;     // per thread: T <= 4096
__global__ __launch_bounds__(kGridSmallThreads) void k_grid_small(uint32_t n, const float* __restrict__ aabb,
                                                                  const uint32_t* __restrict__ shape,
                                                                  const StepCounters* __restrict__ ctr, GridShape axis_mask,
                                                                  uint32_t T, uint32_t* bucket_count, uint32_t* bucket_start,
                                                                  uint32_t* __restrict__ sorted_ids,
                                                                  float* __restrict__ sorted_box) {
    __shared__ uint32_t wtot[kGridSmallThreads / 64];
    const float inv_cell = grid_inv_cell(ctr);
    uint32_t bk[kGridSmallBodies], rk[kGridSmallBodies];
#pragma unroll
    for (int k = 0; k < kGridSmallBodies; ++k) {
        const uint32_t i = k * kGridSmallThreads + threadIdx.x;
        bk[k] = kInvalid; rk[k] = 0;
        if (i < n && shape[i] != PHYS_SPEC_SHAPE_NONE) {
            const v3 lo = ld3(aabb, 2 * i), hi = ld3(aabb, 2 * i + 1);
            const int cx = cell_coord(0.5f * (lo.x + hi.x), inv_cell);
            const int cy = cell_coord(0.5f * (lo.y + hi.y), inv_cell);
            const int cz = cell_coord(0.5f * (lo.z + hi.z), inv_cell);
            bk[k] = bucket_of_cell(cx, cy, cz, axis_mask);
            rk[k] = atomicAdd(&bucket_count[bk[k]], 1u);  // order inside a bucket is irrelevant downstream
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // one workgroup: performed at the L2 is all the next phase needs
    __syncthreads();
    // exclusive scan of the T counts: kGridSmallBuckets consecutive entries per thread (first touch of these
    // lines by this CU: the loads see the atomics, which were performed at the L2)
    const uint32_t base = threadIdx.x * kGridSmallBuckets;
    uint32_t v[kGridSmallBuckets];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kGridSmallBuckets; k += 4) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (base + k < T) q = *reinterpret_cast<const uint4*>(bucket_count + base + k);  // T is a multiple of 8
        v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        sum += q.x + q.y + q.z + q.w;
    }
    const uint32_t inc = wave_inclusive_scan(sum);
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) run += wtot[k];
#pragma unroll
    for (int k = 0; k < kGridSmallBuckets; k += 4) {
        if (base + k < T) {
            uint4 q;
            q.x = run; q.y = run + v[k]; q.z = q.y + v[k + 1]; q.w = q.z + v[k + 2];
            *reinterpret_cast<uint4*>(bucket_start + base + k) = q;
            run = q.w + v[k + 3];
        }
    }
    if (threadIdx.x == kGridSmallThreads - 1) {
        uint32_t total = 0;
        for (int k = 0; k < kGridSmallThreads / 64; ++k) total += wtot[k];
        bucket_start[T] = total;  // grand total in the extra slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // one workgroup: performed at the L2 is all the next phase needs
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGridSmallBodies; ++k) {
        if (bk[k] != kInvalid) {
            const uint32_t i = k * kGridSmallThreads + threadIdx.x;
            const uint32_t s = __hip_atomic_load(&bucket_start[bk[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + rk[k];
            sorted_ids[s] = i;
            st3(sorted_box, 2 * s, ld3(aabb, 2 * i));
            st3(sorted_box, 2 * s + 1, ld3(aabb, 2 * i + 1));
        }
    }
}

// ---- candidate pairs ------------------------------------------------------------------------------
constexpr int kPairThreads = 256;
constexpr int kStagePerWave = 512;  // pairs staged in LDS per wave before a flush (4 KiB)

struct PairStage {
    uint32_t* lds;    // this wave's LDS segment: kStagePerWave pairs (2 u32 each)
    uint32_t count;   // wave-uniform
};

__device__ __forceinline__ void stage_flush(PairStage& st, uint32_t* __restrict__ pairs, uint64_t max_pairs,
                                            StepCounters* __restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    if (st.count == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctr->n_pairs, st.count);
    base = (uint32_t)__shfl((int)base, 0, 64);
    if ((uint64_t)base + st.count > max_pairs) {
        if (lane == 0) flag_overflow(ctr, 1u);
    }
    for (uint32_t k = lane; k < st.count; k += 64) {
        const uint64_t dst = (uint64_t)base + k;
        if (dst < max_pairs) {
            const uint2 pr = reinterpret_cast<const uint2*>(st.lds)[k];
            reinterpret_cast<uint2*>(pairs)[dst] = pr;
        }
    }
    st.count = 0;
}

// every lane of the wave calls this together (hit may be false); compacts the hits into the LDS stage
template <int kStageCap = kStagePerWave>
__device__ __forceinline__ void stage_push(PairStage& st, bool hit, uint32_t a, uint32_t b,
                                           uint32_t* __restrict__ pairs, uint64_t max_pairs,
                                           StepCounters* __restrict__ ctr) {
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return;
    const uint32_t hits = (uint32_t)__popcll(mask);
    if (st.count + hits > (uint32_t)kStageCap) stage_flush(st, pairs, max_pairs, ctr);
    if (hit) {
        const int lane = threadIdx.x & 63;
        const uint32_t r = st.count + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        reinterpret_cast<uint2*>(st.lds)[r] = make_uint2(a, b);
    }
    st.count += hits;
}

// kLanesPerBody lanes share one body: each takes every kLanesPerBody-th cell of the half shell, so the
// dependent chain per lane (bucket range -> candidate ids/boxes) is a quarter as long and four times as
// many loads are in flight. The ranges of a lane's cells are fetched up front (independent loads).
template <int kLanesPerBody>
__global__ __launch_bounds__(kPairThreads) void k_find_pairs(const uint32_t* __restrict__ bucket_start,
                                                             uint32_t table_size, GridShape axis_mask,
                                                             const uint32_t* __restrict__ sorted_ids,
                                                             const float* __restrict__ sorted_box,
                                                             uint32_t* __restrict__ pairs, uint64_t max_pairs,
                                                             StepCounters* __restrict__ ctr) {
    constexpr int kCellsPerLane = (14 + kLanesPerBody - 1) / kLanesPerBody;
    __shared__ uint32_t stage[(kPairThreads / 64) * kStagePerWave * 2];
    PairStage st;
    st.lds = stage + (threadIdx.x >> 6) * kStagePerWave * 2;
    st.count = 0;
    const uint32_t n_active = bucket_start[table_size];
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = g / kLanesPerBody;
    const uint32_t sub = g % kLanesPerBody;
    const bool live = s < n_active;
    const float inv_cell = grid_inv_cell(ctr);
    uint32_t i = 0;
    aabb_t bi;
    bi.lo = v3_make(0, 0, 0); bi.hi = v3_make(0, 0, 0);
    int cx = 0, cy = 0, cz = 0;
    if (live) {
        i = sorted_ids[s];
        bi.lo = ld3(sorted_box, 2 * s);
        bi.hi = ld3(sorted_box, 2 * s + 1);
        cx = cell_coord(0.5f * (bi.lo.x + bi.hi.x), inv_cell);
        cy = cell_coord(0.5f * (bi.lo.y + bi.hi.y), inv_cell);
        cz = cell_coord(0.5f * (bi.lo.z + bi.hi.z), inv_cell);
    }
    // half shell: own cell (c = 0) + the 13 cells with (dz, dy, dx) > (0, 0, 0) lexicographically
    uint32_t t0[kCellsPerLane], t1[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const int c = (int)sub + kLanesPerBody * k;
        int dx, dy, dz;
        if (c == 0) { dx = 0; dy = 0; dz = 0; }
        else if (c == 1) { dx = 1; dy = 0; dz = 0; }
        else if (c < 5) { dx = c - 3; dy = 1; dz = 0; }
        else { dx = (c - 5) % 3 - 1; dy = ((c - 5) / 3) % 3 - 1; dz = 1; }
        t0[k] = 0; t1[k] = 0;
        if (live && c < 14) {
            const uint32_t bk = bucket_of_cell(cx + dx, cy + dy, cz + dz, axis_mask);
            t0[k] = bucket_start[bk];
            t1[k] = bucket_start[bk + 1];
        }
    }
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const bool own_cell = (sub == 0 && k == 0);
        uint32_t t = t0[k];
        const uint32_t t_end = t1[k];
        while (__any(t < t_end)) {
            bool hit = false;
            uint32_t j = 0;
            if (t < t_end) {
                j = sorted_ids[t];
                aabb_t bj;
                bj.lo = ld3(sorted_box, 2 * t);
                bj.hi = ld3(sorted_box, 2 * t + 1);
                // own cell: each unordered pair once by id order. Other cells: a bucket can alias a far cell
                // (wrap-around); such a candidate fails the overlap test, and the 14 buckets are distinct.
                hit = aabb_overlap(bi, bj) && (!own_cell || i < j);
                ++t;
            }
            stage_push(st, hit, i < j ? i : j, i < j ? j : i, pairs, max_pairs, ctr);
        }
    }
    stage_flush(st, pairs, max_pairs, ctr);
}

// ---- candidate pairs, one workgroup per BRICK of the grid ------------------------------------------------------------
// k_find_pairs above walks, per body, 14 cells x (bucket range -> candidates) as dependent loads through the L2: 204 us
// for 1M bodies (C4), 2 % of the HBM rate, and 6.2x the compulsory traffic, because the lanes of a workgroup sit on
// eight XCDs' worth of unrelated cells and every L2 ends up fetching most of the boxes. Here a workgroup owns one brick
// of 4 x 4 x 4 cells - 64 consecutive buckets, so its bodies are ONE run of the bucket-sorted arrays - and first stages
// what its bodies can meet: the 6 x 6 x 5 cells of the brick and its half-shell halo (x, y from -1 to 4, z from 0 to 4):
//   1. 180 lanes fetch the bucket ranges of the region's cells (one round trip), a scan turns the counts into LDS offsets;
//   2. all lanes copy the region's {box, id} records into LDS (second round trip; 28 bytes each, every record of the
//      region exactly once per brick = 2.8 records staged per body);
//   3. four lanes per own body walk the body's 14 cells in LDS; hits go through the same ballot / popcount stage.
// The chain a body waits for is two global round trips per BRICK instead of 28 per body. A region with more records
// than the stage holds (a scene far denser than its grid) is walked in global memory as before - same code path, the
// cell table then holds positions in the sorted arrays instead of in LDS. Bricks are dealt so that the workgroups of one
// XCD (blockIdx % 8) cover one contiguous eighth of the table: a box is fetched into ONE L2, and again only by the
// neighbour XCDs along the eighth's two faces. Same pair SET as k_find_pairs (each pair found by the body whose cell
// comes first in the half-shell order; own-cell pairs by id order).
constexpr int kRegX = 6, kRegY = 6, kRegZ = 5, kRegCells = kRegX * kRegY * kRegZ;  // 180
constexpr int kBrickLanesPerBody = 4;
// pairs staged per wave: 256 (2 KiB) where a body has many pairs (C5: 12 - at 128 the flushes, one same-address atomic
// each, made its search 0.244 ms instead of 0.069), 128 where it has few (C4: 1.3; a seventh workgroup per CU fits)

// WHAT THE PAIR SEARCH WAITS FOR IS THE PAIR COUNTER. Same-address atomics serialise chip-wide at ~88 per microsecond, and
// every flush of a stage is one: the one-lane-per-body kernel flushes once per wave (1M bodies: 15.6k flushes = 180 of
// its 204 us), a first version of this kernel with one brick per workgroup flushed 80k times (814 us). So the workgroups
// are PERSISTENT - a few per CU, each working through many bricks - and keep their hits in LDS across bricks; between two
// bricks, once half the stage is full, the four waves' hits go out behind ONE atomic (C4: ~2.5k atomics for 1.3M pairs).
// A wave whose own part fills up inside a brick (dense scenes) still flushes on its own.
template <int kBrickStagePerWave>
__device__ __forceinline__ void workgroup_flush(PairStage& st, uint32_t* s_cnt, uint32_t* s_base, uint32_t* __restrict__ pairs,
                                                uint64_t max_pairs, StepCounters* __restrict__ ctr, bool force) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) s_cnt[wave] = st.count;
    __syncthreads();
    const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const bool go = total != 0u && (force || total >= (uint32_t)(2 * kBrickStagePerWave));  // workgroup-uniform
    if (go && threadIdx.x == 0) {
        const uint32_t base = atomicAdd(&ctr->n_pairs, total);
        if ((uint64_t)base + total > max_pairs) flag_overflow(ctr, 1u);
        *s_base = base;
    }
    __syncthreads();
    if (go) {
        uint32_t off = *s_base;
        for (uint32_t k = 0; k < wave; ++k) off += s_cnt[k];
        for (uint32_t k = lane; k < st.count; k += 64) {
            const uint64_t dst = (uint64_t)off + k;
            if (dst < max_pairs) reinterpret_cast<uint2*>(pairs)[dst] = reinterpret_cast<const uint2*>(st.lds)[k];
        }
        st.count = 0;
    }
    __syncthreads();  // s_cnt / s_base are reused by the next flush
}

// `cap` records fit the dynamic LDS of this launch (sized by the host from the largest region of an EARLIER update:
// StepCounters::max_region; a brick with more than that walks global memory - slower, same pairs).
template <int kBrickStagePerWave>
__global__ __launch_bounds__(kPairThreads, 7) void k_find_pairs_brick(const uint32_t* __restrict__ bucket_start, uint32_t n_bricks,
                                                                   GridShape g, const uint32_t* __restrict__ sorted_ids,
                                                                   const float* __restrict__ sorted_box,
                                                                   uint32_t* __restrict__ pairs, uint64_t max_pairs,
                                                                   StepCounters* __restrict__ ctr, uint32_t cap) {
    __shared__ uint32_t stage[(kPairThreads / 64) * kBrickStagePerWave * 2];
    __shared__ uint32_t s_gstart[kRegCells];    // position of a region cell's first record in the sorted arrays
    __shared__ uint32_t s_off[kRegCells + 1];   // exclusive scan of the cells' record counts (position in s_rec)
    extern __shared__ float s_rec[];            // cap x {lo xyz, hi xyz, id}: odd stride, distinct banks for neighbouring records
    __shared__ uint32_t s_wsum[4], s_cnt[4], s_base;
    PairStage st;
    st.lds = stage + (threadIdx.x >> 6) * kBrickStagePerWave * 2;
    st.count = 0;
    const float inv_cell = grid_inv_cell(ctr);
    const uint32_t sub = threadIdx.x % kBrickLanesPerBody;
    // XCD-aware deal (a label, never relied on for correctness): workgroups with equal blockIdx % 8 share an L2 and take
    // ONE contiguous eighth of the table between them, brick by brick in turn (neighbouring bricks are in flight together
    // on one L2). (Measured and dropped: eighths of equal BODY count taken from the bucket scan - no faster on any scene.)
    uint32_t first = blockIdx.x, last = n_bricks, step = gridDim.x;
    if (gridDim.x >= 8u && n_bricks >= 8u) {
        const uint32_t label = blockIdx.x & 7u, per = n_bricks >> 3;  // n_bricks is a power of two
        step = (gridDim.x - label + 7u) >> 3;
        first = label * per + (blockIdx.x >> 3);
        last = (label + 1u) * per;
    }
    // The bucket ranges of a brick (its own run, and the 180 cells of its region, one per lane) are asked for ONE BRICK
    // AHEAD: what a brick then waits for is a single round trip - its records - instead of three in a row (ranges ->
    // records, and its own bodies behind the run's bounds), with only a few workgroups per CU to hide them.
    uint32_t nx_begin = 0, nx_end = 0, nx_c0 = 0, nx_cnt = 0;
    auto ask = [&](uint32_t brick) {
        nx_begin = bucket_start[brick * 64u];
        nx_end = bucket_start[brick * 64u + 64u];
        nx_c0 = 0; nx_cnt = 0;
        if (threadIdx.x < (uint32_t)kRegCells) {
            const uint32_t bx = brick & ((1u << g.sx) - 1u), by = (brick >> g.sx) & ((1u << g.sy) - 1u), bz = brick >> (g.sx + g.sy);
            const uint32_t r = threadIdx.x;
            const uint32_t rx = r % kRegX, ry = (r / kRegX) % kRegY, rz = r / (kRegX * kRegY);
            const uint32_t bk = grid_bucket_masked(((bx << 2) + rx - 1u) & g.mx, ((by << 2) + ry - 1u) & g.my, ((bz << 2) + rz) & g.mz, g);
            nx_c0 = bucket_start[bk];
            nx_cnt = bucket_start[bk + 1] - nx_c0;
        }
    };
    if (first < last) ask(first);
    uint32_t seen_max = 0;
    for (uint32_t brick = first; brick < last; brick += step) {
        const uint32_t own_begin = nx_begin, own_end = nx_end, my_c0 = nx_c0, my_cnt = nx_cnt;
        if (brick + step < last) ask(brick + step);
        if (own_begin == own_end) continue;  // workgroup-uniform: an empty brick
        constexpr uint32_t kBodiesPerTrip = kPairThreads / kBrickLanesPerBody;
        const uint32_t n_own = own_end - own_begin;
        // the first trip's own bodies: asked for now, needed behind the staging
        uint32_t i0 = 0;
        aabb_t b0;
        b0.lo = v3_make(0, 0, 0); b0.hi = v3_make(0, 0, 0);
        if (threadIdx.x / kBrickLanesPerBody < n_own) {
            const uint32_t sidx = own_begin + threadIdx.x / kBrickLanesPerBody;
            i0 = sorted_ids[sidx];
            b0.lo = ld3(sorted_box, 2 * sidx);
            b0.hi = ld3(sorted_box, 2 * sidx + 1);
        }
        // 1. bucket ranges of the region's cells (asked for one brick ago), scanned
        {
            const uint32_t cnt = my_cnt;
            if (threadIdx.x < (uint32_t)kRegCells) s_gstart[threadIdx.x] = my_c0;
            const uint32_t inc = wave_inclusive_scan(cnt);
            if ((threadIdx.x & 63u) == 63u) s_wsum[threadIdx.x >> 6] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) base += s_wsum[k];
            if (threadIdx.x < (uint32_t)kRegCells) s_off[threadIdx.x] = base + inc - cnt;
            if (threadIdx.x == (uint32_t)kRegCells - 1u) s_off[kRegCells] = base + inc;
            __syncthreads();
        }
        const uint32_t total = s_off[kRegCells];
        const bool staged = total <= cap;
        seen_max = total > seen_max ? total : seen_max;
        // 2. the region's records into LDS
        if (staged) {
            for (uint32_t q = threadIdx.x; q < total; q += kPairThreads) {
                uint32_t lo = 0, hi = kRegCells;  // the cell r with s_off[r] <= q < s_off[r + 1]
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_off[mid] <= q) lo = mid; else hi = mid;
                }
                const uint32_t t = s_gstart[lo] + (q - s_off[lo]);
                const v3 blo = ld3(sorted_box, 2 * t), bhi = ld3(sorted_box, 2 * t + 1);
                float* rec = s_rec + 7u * q;
                rec[0] = blo.x; rec[1] = blo.y; rec[2] = blo.z; rec[3] = bhi.x; rec[4] = bhi.y; rec[5] = bhi.z;
                rec[6] = __uint_as_float(sorted_ids[t]);
            }
            __syncthreads();
        }
        // 3. the brick's own bodies, four lanes each, against the 14 cells of their half shell.
        // (Measured and dropped, same pairs: (own body, candidate) tests enumerated per own CELL, sixteen cells per wave, both
        // sides from LDS - 2.6x slower, three dependent LDS trips per cell and nothing to overlap them; a lane walking its
        // four cells as one run, moving on the moment a cell is exhausted - fewer passes, 5-15 % slower: the cursor costs
        // more than the passes it saves.)
        for (uint32_t base = 0; base < n_own; base += kBodiesPerTrip) {
            const uint32_t jo = base + threadIdx.x / kBrickLanesPerBody;
            const bool live = jo < n_own;
            uint32_t i = i0;
            aabb_t bi = b0;
            if (base != 0u && live) {
                const uint32_t sidx = own_begin + jo;
                i = sorted_ids[sidx];
                bi.lo = ld3(sorted_box, 2 * sidx);
                bi.hi = ld3(sorted_box, 2 * sidx + 1);
            }
            // the brick's origin is a multiple of four cells on every axis: the low two bits are the place in the brick
            const uint32_t lx = (uint32_t)cell_coord(0.5f * (bi.lo.x + bi.hi.x), inv_cell) & 3u;
            const uint32_t ly = (uint32_t)cell_coord(0.5f * (bi.lo.y + bi.hi.y), inv_cell) & 3u;
            const uint32_t lz = (uint32_t)cell_coord(0.5f * (bi.lo.z + bi.hi.z), inv_cell) & 3u;
#pragma unroll
            for (int k = 0; k < (14 + kBrickLanesPerBody - 1) / kBrickLanesPerBody; ++k) {
                const int c = (int)sub + kBrickLanesPerBody * k;
                int dx, dy, dz;  // half shell: own cell (c = 0) + the 13 cells with (dz, dy, dx) > (0, 0, 0) lexicographically
                if (c == 0) { dx = 0; dy = 0; dz = 0; }
                else if (c == 1) { dx = 1; dy = 0; dz = 0; }
                else if (c < 5) { dx = c - 3; dy = 1; dz = 0; }
                else { dx = (c - 5) % 3 - 1; dy = ((c - 5) / 3) % 3 - 1; dz = 1; }
                uint32_t t = 0, t_end = 0;
                if (live && c < 14) {
                    const uint32_t rid = ((lz + (uint32_t)dz) * kRegY + (ly + (uint32_t)(dy + 1))) * kRegX + (lx + (uint32_t)(dx + 1));
                    t = staged ? s_off[rid] : s_gstart[rid];
                    t_end = t + (s_off[rid + 1] - s_off[rid]);
                }
                const bool own_cell = c == 0;
                while (__any(t < t_end)) {
                    bool hit = false;
                    uint32_t j = 0;
                    if (t < t_end) {
                        aabb_t bj;
                        if (staged) {
                            const float* rec = s_rec + 7u * t;
                            bj.lo = v3_make(rec[0], rec[1], rec[2]);
                            bj.hi = v3_make(rec[3], rec[4], rec[5]);
                            j = __float_as_uint(rec[6]);
                        } else {
                            j = sorted_ids[t];
                            bj.lo = ld3(sorted_box, 2 * t);
                            bj.hi = ld3(sorted_box, 2 * t + 1);
                        }
                        // own cell: each unordered pair once by id order. Other cells: a bucket can alias a far cell
                        // (wrap-around); such a candidate fails the overlap test, and the 14 buckets are distinct.
                        hit = aabb_overlap(bi, bj) && (!own_cell || i < j);
                        ++t;
                    }
                    stage_push<kBrickStagePerWave>(st, hit, i < j ? i : j, i < j ? j : i, pairs, max_pairs, ctr);
                }
            }
        }
        // the region tables and records are rewritten by the next brick: everybody is done with them behind the flush's barriers
        workgroup_flush<kBrickStagePerWave>(st, s_cnt, &s_base, pairs, max_pairs, ctr, /*force=*/false);
    }
    workgroup_flush<kBrickStagePerWave>(st, s_cnt, &s_base, pairs, max_pairs, ctr, /*force=*/true);
    // the largest region met: what a later update sizes its stage by (only a raise is an atomic)
    if (threadIdx.x == 0 && seen_max > ctr->max_region) atomicMax(&ctr->max_region, seen_max);
}

// ---- slot grid (small scenes) ------------------------------------------------------------------------
// With a table of >= 2 buckets per body almost every bucket holds 0-2 bodies: instead of counting, scanning and
// scattering (three dependent launches) a body simply takes one of EIGHT slots of its bucket (id + box) and the
// pair kernel reads the neighbours' slots directly. A body that finds the eight slots taken goes to a short overflow list that every body also tests against, so the pair SET is the same as with the
// sorted grid whatever the occupancy; only the speed depends on it.
constexpr uint32_t kSlotsPerBucket = 8;
constexpr uint32_t kSlotGridMaxBodies = 32768;  // beyond, streaming the sorted boxes is as fast

__global__ __launch_bounds__(256) void k_cell_insert(uint32_t n, const float* __restrict__ aabb,
                                                     const uint32_t* __restrict__ shape, StepCounters* __restrict__ ctr,
                                                     GridShape axis_mask, uint32_t* __restrict__ rank,
                                                     uint32_t* __restrict__ bucket_count, uint32_t* __restrict__ slot_ids,
                                                     float* __restrict__ slot_box, uint32_t* __restrict__ ovf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || shape[i] == PHYS_SPEC_SHAPE_NONE) return;
    const float inv_cell = grid_inv_cell(ctr);
    const v3 lo = ld3(aabb, 2 * i), hi = ld3(aabb, 2 * i + 1);
    const int cx = cell_coord(0.5f * (lo.x + hi.x), inv_cell);
    const int cy = cell_coord(0.5f * (lo.y + hi.y), inv_cell);
    const int cz = cell_coord(0.5f * (lo.z + hi.z), inv_cell);
    const uint32_t bk = bucket_of_cell(cx, cy, cz, axis_mask);
    const uint32_t r = atomicAdd(&bucket_count[bk], 1u);  // which slot is irrelevant downstream
    rank[i] = r;
    if (r < kSlotsPerBucket) {
        // id AND box go into the slot: the pair kernel then needs no gather by id behind the slot record
        slot_ids[kSlotsPerBucket * bk + r] = i;
        st3(slot_box, 2 * (kSlotsPerBucket * bk + r), lo);
        st3(slot_box, 2 * (kSlotsPerBucket * bk + r) + 1, hi);
    } else {
        ovf[atomicAdd(&ctr->n_grid_ovf, 1u)] = i;
    }
}

template <int kLanesPerBody>
__global__ __launch_bounds__(kPairThreads) void k_find_pairs_slots(uint32_t n, const float* __restrict__ aabb,
                                                                   const uint32_t* __restrict__ shape,
                                                                   const uint32_t* __restrict__ rank,
                                                                   const uint32_t* __restrict__ bucket_count,
                                                                   const uint32_t* __restrict__ slot_ids,
                                                                   const float* __restrict__ slot_box,
                                                                   const uint32_t* __restrict__ ovf, GridShape axis_mask,
                                                                   uint32_t* __restrict__ pairs, uint64_t max_pairs,
                                                                   StepCounters* __restrict__ ctr) {
    constexpr int kCellsPerLane = (14 + kLanesPerBody - 1) / kLanesPerBody;
    __shared__ uint32_t stage[(kPairThreads / 64) * kStagePerWave * 2];
    PairStage st;
    st.lds = stage + (threadIdx.x >> 6) * kStagePerWave * 2;
    st.count = 0;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = g / kLanesPerBody;
    const uint32_t sub = g % kLanesPerBody;
    const bool live = i < n && shape[i] != PHYS_SPEC_SHAPE_NONE;
    const float inv_cell = grid_inv_cell(ctr);
    aabb_t bi;
    bi.lo = v3_make(0, 0, 0); bi.hi = v3_make(0, 0, 0);
    int cx = 0, cy = 0, cz = 0;
    bool in_overflow = false;
    if (live) {
        bi.lo = ld3(aabb, 2 * i);
        bi.hi = ld3(aabb, 2 * i + 1);
        cx = cell_coord(0.5f * (bi.lo.x + bi.hi.x), inv_cell);
        cy = cell_coord(0.5f * (bi.lo.y + bi.hi.y), inv_cell);
        cz = cell_coord(0.5f * (bi.lo.z + bi.hi.z), inv_cell);
        in_overflow = rank[i] >= kSlotsPerBucket;
    }
    // half shell: own cell (c = 0) + the 13 cells with (dz, dy, dx) > (0, 0, 0) lexicographically. A body of the
    // overflow list looks at nobody's slots: its pairs come from the others' (and its own) pass over that list.
    uint32_t cnt[kCellsPerLane], first[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const int c = (int)sub + kLanesPerBody * k;
        int dx, dy, dz;
        if (c == 0) { dx = 0; dy = 0; dz = 0; }
        else if (c == 1) { dx = 1; dy = 0; dz = 0; }
        else if (c < 5) { dx = c - 3; dy = 1; dz = 0; }
        else { dx = (c - 5) % 3 - 1; dy = ((c - 5) / 3) % 3 - 1; dz = 1; }
        cnt[k] = 0; first[k] = 0;
        if (live && !in_overflow && c < 14) {
            const uint32_t bk = bucket_of_cell(cx + dx, cy + dy, cz + dz, axis_mask);
            const uint32_t have = bucket_count[bk];
            first[k] = kSlotsPerBucket * bk;
            cnt[k] = have < kSlotsPerBucket ? have : kSlotsPerBucket;
        }
    }
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const bool own_cell = (sub == 0 && k == 0);
        for (uint32_t r = 0; __any(r < cnt[k]); ++r) {
            bool hit = false;
            uint32_t j = 0;
            if (r < cnt[k]) {
                const uint32_t slot = first[k] + r;
                j = slot_ids[slot];
                aabb_t bj;
                bj.lo = ld3(slot_box, 2 * slot);
                bj.hi = ld3(slot_box, 2 * slot + 1);
                // own cell: each unordered pair once by id order. Other cells: a bucket can alias a far cell
                // (wrap-around); such a candidate fails the overlap test, and the 14 buckets are distinct.
                hit = aabb_overlap(bi, bj) && (!own_cell || i < j);
            }
            stage_push(st, hit, i < j ? i : j, i < j ? j : i, pairs, max_pairs, ctr);
        }
    }
    // everybody against the overflow list: a slotted body takes every overflow body it overlaps, two overflow
    // bodies meet once (by id order)
    const uint32_t n_ovf = ctr->n_grid_ovf;
    for (uint32_t o = 0; o < n_ovf; ++o) {
        const uint32_t j = ovf[o];
        bool hit = false;
        if (live && sub == 0 && j != i && (!in_overflow || i < j)) {
            aabb_t bj;
            bj.lo = ld3(aabb, 2 * j);
            bj.hi = ld3(aabb, 2 * j + 1);
            hit = aabb_overlap(bi, bj);
        }
        stage_push(st, hit, i < j ? i : j, i < j ? j : i, pairs, max_pairs, ctr);
    }
    stage_flush(st, pairs, max_pairs, ctr);
}

// ---- host side -------------------------------------------------------------------------------------
static uint32_t table_bits_for(uint64_t n) {
    uint32_t bits = 9;  // 512 buckets = 8 bricks at least
    while ((1ull << bits) < 2 * n && bits < 27) ++bits;
    return bits;
}

// Table size (>= 2 buckets per body, a power of two) and its split over the axes, from the scene as uploaded: every axis
// starts with 2 bits (one brick of 4 cells) and the rest go, one at a time, to the axis with the most cells per bucket
// row - cells estimated as extent of the body centres / the largest bounding diameter. Any split is correct (cells wrap
// modulo the axis size); a good one keeps far-apart cells out of the same bucket. Bodies move, the split stays: a pile
// that compresses or spreads by a factor of two costs one bit of accuracy, not correctness.
void grid_plan(phys_world* w, const float* pos, const float* half_extent) {
    const uint64_t n = w->n_owned;
    const uint32_t bits = table_bits_for(w->n);
    w->grid_table_size = 1u << bits;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f}, diam = 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos[3 * i + a]); hi[a] = std::max(hi[a], pos[3 * i + a]); }
        if (half_extent) {
            const float* h = half_extent + 3 * i;
            diam = std::max(diam, 2.0f * std::max(h[0], std::max(h[1], h[2])));
        }
    }
    const float cell = (diam > 0.0f ? diam : 1.0f) * 1.05f + 2.0f * w->cfg.contact_margin;
    double cells[3];
    for (int a = 0; a < 3; ++a) cells[a] = n ? std::max(1.0, (double)(hi[a] - lo[a]) / cell + 1.0) : 1.0;
    uint32_t ab[3] = {2, 2, 2};
    for (uint32_t left = bits - 6; left > 0; --left) {
        int best = 0;
        double worst = -1.0;
        for (int a = 0; a < 3; ++a) {
            const double load = cells[a] / (double)(1u << ab[a]);
            if (load > worst && ab[a] < 20) { worst = load; best = a; }
        }
        ab[best] += 1;
    }
    GridShape g;
    g.mx = (1u << ab[0]) - 1u; g.my = (1u << ab[1]) - 1u; g.mz = (1u << ab[2]) - 1u;
    g.sx = ab[0] - 2u; g.sy = ab[1] - 2u;
    w->grid_shape = g;
}

int32_t collision_alloc(phys_world* w) {
    const uint64_t n = w->n;
    w->max_pairs = w->cfg.max_pairs ? w->cfg.max_pairs : std::max<uint64_t>(24 * n, 4096);
    w->max_manifolds = w->cfg.max_manifolds ? w->cfg.max_manifolds : std::max<uint64_t>(17 * n, 4096);
    if (w->max_pairs > 0xFFFFFFF0ull || w->max_manifolds > 0xFFFFFFF0ull) {
        set_error("max_pairs / max_manifolds exceed u32 indexing");
        return PHYS_ERR_INVALID_ARG;
    }
    const uint32_t T = w->grid_table_size;  // grid_plan (phys_set_bodies), before this
    {
        // one allocation, zeroed by one memset per step: [bucket counts | colouring state | StepCounters]
        const size_t b_bytes = ((size_t)T * 4 + 255) / 256 * 256;
        const size_t c_bytes = (w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY) ? 0 : ((size_t)4 * n * 8 + 255) / 256 * 256;
        w->counters.free(); w->bucket_count.free(); w->color_state.free();
        w->step_zero.free();
        PHYS_HIP_TRY(w->step_zero.resize(b_bytes + c_bytes + sizeof(StepCounters)));
        w->bucket_count.point_at(reinterpret_cast<uint32_t*>(w->step_zero.p), T);
        w->color_state.point_at(reinterpret_cast<unsigned long long*>(w->step_zero.p + b_bytes), c_bytes / 8);
        w->counters.point_at(reinterpret_cast<StepCounters*>(w->step_zero.p + b_bytes + c_bytes), 1);
        w->step_zero_reset_bytes = b_bytes + c_bytes + kCountersStepResetBytes;
        w->step_zero_full_bytes = b_bytes + c_bytes + kCountersExtentResetBytes;  // not the sticky word behind it
        PHYS_HIP_TRY(hipMemsetAsync(w->step_zero.p, 0, b_bytes + c_bytes + sizeof(StepCounters), w->stream));  // sticky word too
    }
    PHYS_HIP_TRY(w->bucket_of.resize(n));
    PHYS_HIP_TRY(w->bucket_cursor.resize(n));  // rank of each body inside its bucket
    PHYS_HIP_TRY(w->bucket_start.resize((size_t)T + 1));
    PHYS_HIP_TRY(w->scan_block_sums.resize(2 * std::max<size_t>((T + kScanChunk - 1) / kScanChunk + 1, 64)));  // sums | buckets in use
    PHYS_HIP_TRY(w->sorted_ids.resize(n));
    PHYS_HIP_TRY(w->slot_ids.resize((size_t)kSlotsPerBucket * T));
    PHYS_HIP_TRY(w->slot_box.resize((size_t)6 * kSlotsPerBucket * T));
    PHYS_HIP_TRY(w->grid_ovf.resize(n));
    w->sorted_grid_valid = false;
    PHYS_HIP_TRY(w->sorted_box.resize(6 * n));
    PHYS_HIP_TRY(w->pairs.resize(2 * w->max_pairs));
    if (!(w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY)) {
        const uint64_t M = w->max_manifolds;
        PHYS_HIP_TRY(w->man_a.resize(M)); PHYS_HIP_TRY(w->man_b.resize(M));
        PHYS_HIP_TRY(w->man_color.resize(M));
        PHYS_HIP_TRY(w->man_geo.resize(32 * M));
        w->warm = !(w->cfg.flags & PHYS_FLAG_NO_WARM_START) && M < (1ull << 26);  // the colour table's value word holds 26 bits of index
        if (w->warm) {
            PHYS_HIP_TRY(w->man_geo_prev.resize(32 * M));
            PHYS_HIP_TRY(w->man_imp.resize(12 * M));
            PHYS_HIP_TRY(w->man_imp_prev.resize(12 * M));
            PHYS_HIP_TRY(w->man_prev.resize(4));  // only its address is used: "warm starting is on" (the index itself is in man_geo)
        }
        PHYS_HIP_TRY(w->man_prio.resize(M));
        PHYS_HIP_TRY(w->row_src.resize(M));
        PHYS_HIP_TRY(w->unc_list.resize(2 * M));
        {
            uint64_t cap = 4096;
            while (cap < M + M / 2) cap <<= 1;
            // PHYS_DEBUG_CTAB_SLOTS=<power of two>: a smaller table (tests of the bounded walks: crowded and overfull tables)
            static const uint64_t slots_env = getenv("PHYS_DEBUG_CTAB_SLOTS") ? strtoull(getenv("PHYS_DEBUG_CTAB_SLOTS"), nullptr, 10) : 0;
            if (slots_env >= 64 && (slots_env & (slots_env - 1)) == 0) cap = slots_env;
            PHYS_HIP_TRY(w->ctab.resize(2 * cap));
            w->ctab_mask = (uint32_t)(cap - 1);
            w->ctab_valid = false;
            w->color_epoch = 0;
        }
        PHYS_HIP_TRY(w->color_block_hist.resize((size_t)kMaxColors * 512));
        w->row_hdr.free(); w->row_n.free(); w->row_tb.free(); w->row_pt.free(); w->row_acc.free();
        w->row_all.free();
        PHYS_HIP_TRY(w->row_all.resize(64 * M));  // 16 planes x M x float4
        w->row_hdr.point_at(reinterpret_cast<uint32_t*>(w->row_all.p), 4 * M);
        w->row_n.point_at(w->row_all.p + 4 * M, 4 * M);
        w->row_tb.point_at(w->row_all.p + 8 * M, 8 * M);
        w->row_pt.point_at(w->row_all.p + 16 * M, 32 * M);
        w->row_acc.point_at(w->row_all.p + 48 * M, 16 * M);
        w->flow_vel.free();
        // the dataflow solver addresses row_acc / flow_vel through 32-bit buffer offsets
        if (!(w->cfg.flags & PHYS_FLAG_SOLVER_PER_COLOR) && 64 * M < 0xFFFFFFFFull && 32 * n < 0xFFFFFFFFull) {
            PHYS_HIP_TRY(w->flow_vel.resize(8 * n));
            // tags of an earlier scene must never look like tags of this one
            PHYS_HIP_TRY(hipMemsetAsync(w->flow_vel.p, 0, 8 * n * sizeof(float), w->stream));
            PHYS_HIP_TRY(hipMemsetAsync(w->row_acc.p, 0, 16 * M * sizeof(float), w->stream));
            // PHYS_DEBUG_FLOW_EPOCH=<n>: start the tag epoch near its wrap (test of the 16-bit epoch reset)
            static const uint32_t epoch_env = getenv("PHYS_DEBUG_FLOW_EPOCH") ? (uint32_t)strtoul(getenv("PHYS_DEBUG_FLOW_EPOCH"), nullptr, 0) : 0u;
            w->flow_epoch = epoch_env;
        }
    }
    return PHYS_OK;
}

// exclusive scan of `count` (a multiple of 4) counters into out[count + 1] on the world's stream
bool scan_is_one_launch(uint32_t count) { return count <= (uint32_t)(kScanSmallThreads * kScanSmallItems); }

// zero_in: only honoured by the one-launch scan (scan_is_one_launch(count)); the caller zeroes the counters itself otherwise
void launch_exclusive_scan(phys_world* w, uint32_t* in, uint32_t count, uint32_t* out, bool zero_in) {
    hipStream_t s = w->stream;
    if (scan_is_one_launch(count)) {
        if (zero_in) hipLaunchKernelGGL(k_scan_small<true>, dim3(1), dim3(kScanSmallThreads), 0, s, in, count, out);
        else hipLaunchKernelGGL(k_scan_small<false>, dim3(1), dim3(kScanSmallThreads), 0, s, in, count, out);
        return;
    }
    const uint32_t nblk = (count + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblk), dim3(kScanThreads), 0, s, in, count, w->scan_block_sums.p);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, w->scan_block_sums.p, nblk);
    hipLaunchKernelGGL(k_scan_final, dim3(nblk), dim3(kScanThreads), 0, s, in, count, w->scan_block_sums.p, out);
}

void zero_step_state(phys_world* w, bool including_extent) {
    PHYS_PROF(w, PHYS_STAGE_MISC);
    (void)hipMemsetAsync(w->step_zero.p, 0, including_extent ? w->step_zero_full_bytes : w->step_zero_reset_bytes, w->stream);
}

static GridShape grid_axis_mask(const phys_world* w) { return w->grid_shape; }

// bucket_start / sorted_ids / sorted_box from the current AABBs; the bucket counts must be zero
void build_sorted_grid(phys_world* w) {
    const uint32_t n = (uint32_t)w->n;
    const uint32_t T = w->grid_table_size;
    const GridShape axis_mask = grid_axis_mask(w);
    hipStream_t s = w->stream;
    const dim3 gb((n + 255) / 256), tb(256);
    w->sorted_grid_valid = true;
    if (n <= (uint32_t)(kGridSmallBodies * kGridSmallThreads) && T <= (uint32_t)(kGridSmallBuckets * kGridSmallThreads) && T % 8 == 0) {
        PHYS_PROF(w, PHYS_STAGE_GRID);
        hipLaunchKernelGGL(k_grid_small, dim3(1), dim3(kGridSmallThreads), 0, s, n, w->aabb.p, w->shape.p, w->counters.p, axis_mask, T,
                           w->bucket_count.p, w->bucket_start.p, w->sorted_ids.p, w->sorted_box.p);
        return;
    }
    { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_cell_assign, gb, tb, 0, s, n, w->aabb.p, w->shape.p, w->counters.p, axis_mask, w->bucket_of.p,
                       w->bucket_cursor.p, w->bucket_count.p); }
    if (T <= (uint32_t)(kScanSmallThreads * kScanSmallItems) && T % 4 == 0) {
        PHYS_PROF(w, PHYS_STAGE_GRID);
        hipLaunchKernelGGL(k_scan_small<false>, dim3(1), dim3(kScanSmallThreads), 0, s, w->bucket_count.p, T, w->bucket_start.p);
    } else {
        const uint32_t nblk = (T + kScanChunk - 1) / kScanChunk;
        uint32_t* used = w->scan_block_sums.p + (w->scan_block_sums.n / 2);  // second half of the buffer
        { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_scan_reduce, dim3(nblk), dim3(kScanThreads), 0, s, w->bucket_count.p, T, w->scan_block_sums.p, used); }
        { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, w->scan_block_sums.p, nblk, (const uint32_t*)used, w->counters.p); }
        { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_scan_final, dim3(nblk), dim3(kScanThreads), 0, s, w->bucket_count.p, T, w->scan_block_sums.p,
                           w->bucket_start.p); }
    }
    { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_scatter, gb, tb, 0, s, n, w->aabb.p, w->bucket_of.p, w->bucket_cursor.p, w->bucket_start.p,
                       w->sorted_ids.p, w->sorted_box.p); }
}

void launch_broadphase(phys_world* w) {
    const uint32_t n = (uint32_t)w->n;
    if (n == 0) return;
    w->grid_valid = true;
    const uint32_t T = w->grid_table_size;
    const GridShape axis_mask = grid_axis_mask(w);
    hipStream_t s = w->stream;
    if (n <= kSlotGridMaxBodies) {
        // slot grid: two launches for the whole broad phase
        w->sorted_grid_valid = false;
        { PHYS_PROF(w, PHYS_STAGE_GRID); hipLaunchKernelGGL(k_cell_insert, dim3((n + 255) / 256), dim3(256), 0, s, n, w->aabb.p, w->shape.p, w->counters.p,
                           axis_mask, w->bucket_cursor.p, w->bucket_count.p, w->slot_ids.p, w->slot_box.p, w->grid_ovf.p); }
        PHYS_PROF(w, PHYS_STAGE_PAIRS);
        hipLaunchKernelGGL((k_find_pairs_slots<4>), dim3((unsigned)(((uint64_t)n * 4 + kPairThreads - 1) / kPairThreads)), dim3(kPairThreads), 0, s,
                           n, w->aabb.p, w->shape.p, w->bucket_cursor.p, w->bucket_count.p, w->slot_ids.p, w->slot_box.p, w->grid_ovf.p, axis_mask,
                           w->pairs.p, w->max_pairs, w->counters.p);
        return;
    }
    build_sorted_grid(w);
    // small scenes are latency-bound: 4 lanes per body shorten the dependent chain; large scenes are
    // throughput-bound: one lane per body does the least total work
    static const int pair_lanes_env = getenv("PHYS_DEBUG_PAIR_LANES") ? atoi(getenv("PHYS_DEBUG_PAIR_LANES")) : 0;  // measurements
    // (measured: one lane per body is the faster one already at 100k bodies - C3: 0.051 against 0.089 ms)
    // PHYS_DEBUG_PAIR_KERNEL=body / brick forces one (measurements; same pair set)
    static const char* pair_kernel_env = getenv("PHYS_DEBUG_PAIR_KERNEL");
    // The brick kernel wins where the grid is sparsely filled - lattices, stacks of aligned boxes: many bricks of few
    // records (C4 204 -> 119 us, C5 135 -> 73) - and loses where cells are crowded (one tumbled cube sets the cell size for
    // everybody: 1M falling cubes 89 -> 102 us, C3 51 -> 82: few bricks, each a long walk for the one workgroup that has
    // it). Crowding = bodies per bucket in use, counted by k_cell_assign of an earlier update.
    const bool crowded = w->hint.valid && w->hint.n_used_buckets && (uint64_t)w->n * 10ull > (uint64_t)w->hint.n_used_buckets * 21ull;
    const bool brick = pair_kernel_env ? pair_kernel_env[0] != 'b' || pair_kernel_env[1] == 'r' : !crowded;
    if (brick && !pair_lanes_env) {
        const uint32_t n_bricks = T >> 6;
        // records staged per brick: a quarter more than the largest region of an earlier update (C4: ~400 records, a pile
        // of tumbled cubes: ~1500), 1024 while nothing is known, at most what one workgroup may have of a CU's LDS
        constexpr uint32_t kCapMax = 5000;  // 137 KiB
        uint32_t cap = w->hint.valid && w->hint.max_region ? w->hint.max_region + w->hint.max_region / 4 : 1024u;
        cap = std::min(std::max((cap + 63u) & ~63u, 256u), kCapMax);
        const size_t dyn = (size_t)cap * 28;
        // few pairs per body (of an earlier update): the small stage, which leaves room for a seventh workgroup per CU
        static const int stage_env = getenv("PHYS_DEBUG_BRICK_STAGE") ? atoi(getenv("PHYS_DEBUG_BRICK_STAGE")) : 0;  // measurements: 128 | 256
        const bool small_stage = stage_env ? stage_env == 128 : (w->hint.valid && (uint64_t)w->hint.n_pairs < 3ull * w->n);
        const size_t fixed = (kPairThreads / 64) * (small_stage ? 128 : 256) * 8 + kRegCells * 8 + 64;
        // persistent workgroups, as many as are resident at once (the LDS decides), never more than there are bricks
        // at most seven per CU (66 registers: seven waves per SIMD). Measured, us (C4 / 1M cubes in mid-fall / C5): 3 per CU
        // 173 / 145 / 74, 4: 137 / 114 / 72, 5: 117 / 100 / 69, 6: 105 / 89 / 70, 7 (small stage): 103 / 83 / -; asked for 8
        // (not all resident: the late ones start on a drained chip) 131 / 109 / 69
        static const size_t per_cu_max = getenv("PHYS_DEBUG_BRICK_PER_CU") ? (size_t)atoi(getenv("PHYS_DEBUG_BRICK_PER_CU")) : 7;  // measurements
        uint32_t per_cu = (uint32_t)std::min<size_t>(per_cu_max, (160 * 1024) / (((dyn + fixed) + 1023) / 1024 * 1024));
        uint32_t wgs = 256u * std::max(per_cu, 1u);
        while (wgs > n_bricks) wgs >>= 1;
        static bool attr_set[64] = {};  // per device (function attributes are): more than 64 KiB of dynamic LDS needs it
        if (!attr_set[w->device & 63]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_find_pairs_brick<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_find_pairs_brick<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess)
                attr_set[w->device & 63] = true;
            else (void)hipGetLastError();
        }
        PHYS_PROF(w, PHYS_STAGE_PAIRS);
        if (small_stage)
            hipLaunchKernelGGL(k_find_pairs_brick<128>, dim3(wgs), dim3(kPairThreads), dyn, s, w->bucket_start.p, n_bricks, axis_mask,
                               w->sorted_ids.p, w->sorted_box.p, w->pairs.p, w->max_pairs, w->counters.p, cap);
        else
            hipLaunchKernelGGL(k_find_pairs_brick<256>, dim3(wgs), dim3(kPairThreads), dyn, s, w->bucket_start.p, n_bricks, axis_mask,
                               w->sorted_ids.p, w->sorted_box.p, w->pairs.p, w->max_pairs, w->counters.p, cap);
        return;
    }
    if (pair_lanes_env ? pair_lanes_env == 4 : n <= 65536u) {
        PHYS_PROF(w, PHYS_STAGE_PAIRS);
        hipLaunchKernelGGL((k_find_pairs<4>), dim3((unsigned)(((uint64_t)n * 4 + kPairThreads - 1) / kPairThreads)), dim3(kPairThreads), 0, s,
                           w->bucket_start.p, T, axis_mask, w->sorted_ids.p, w->sorted_box.p, w->pairs.p, w->max_pairs, w->counters.p);
    } else {
        PHYS_PROF(w, PHYS_STAGE_PAIRS);
        hipLaunchKernelGGL((k_find_pairs<1>), dim3((n + kPairThreads - 1) / kPairThreads), dim3(kPairThreads), 0, s,
                           w->bucket_start.p, T, axis_mask, w->sorted_ids.p, w->sorted_box.p, w->pairs.p, w->max_pairs, w->counters.p);
    }
}

// phys_broadphase read-out: pairs sorted by (i, j). The sort is a host-side convenience of this
// read-out call; the per-step pipeline never sorts (nothing downstream depends on pair order).
int32_t sorted_pairs_to_host(phys_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs) {
    PHYS_HIP_TRY(hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    if (w->h_counters->overflow & 1u) {
        set_error("pair capacity exceeded: raise phys_config.max_pairs");
        return PHYS_ERR_CAPACITY;
    }
    const uint64_t m = w->h_counters->n_pairs;
    *n_pairs = m;
    if (!pairs_out || m == 0) return PHYS_OK;
    std::vector<uint64_t> keys(m);
    {
        std::vector<uint32_t> raw(2 * m);
        PHYS_HIP_TRY(hipMemcpy(raw.data(), w->pairs.p, 8 * m, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < m; ++k) keys[k] = ((uint64_t)raw[2 * k] << 32) | raw[2 * k + 1];
    }
    std::sort(keys.begin(), keys.end());
    for (uint64_t k = 0; k < m && k < cap; ++k) {
        pairs_out[2 * k] = (uint32_t)(keys[k] >> 32);
        pairs_out[2 * k + 1] = (uint32_t)keys[k];
    }
    return PHYS_OK;
}

}  // namespace phys
