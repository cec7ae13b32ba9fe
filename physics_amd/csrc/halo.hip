// halo.hip — sharded broad phase for gfx950: the per-rank halves of the one exchange
// step. One process per GPU; the all-gather itself is RCCL through torch.distributed (plumbing) on
// buffers whose device pointers are handed to these entry points.
//
//   phys_halo_pack   owned bodies whose fattened AABB comes within `reach` of a slab face (or beyond it)
//                    -> 32-byte records {lo xyz, hi xyz, global id, pad}, ballot-compacted.
//   phys_halo_pairs  every gathered remote record is tested against the owned bodies through the SAME
//                    bucket grid the local broad phase built this step; a cross pair (local index, remote
//                    global id) is emitted by the rank whose body has the smaller global id, so each
//                    cross-rank pair appears exactly once in the union over ranks.
#include <algorithm>

#include "kernels.hpp"

namespace phys {

struct HaloRecord {
    float lo[3];
    float hi[3];
    uint32_t gid;
    uint32_t pad;
};
static_assert(sizeof(HaloRecord) == 32, "halo record is 32 bytes");

__device__ __forceinline__ uint32_t part1by2h(uint32_t x) {
    x &= 0x000003ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}
__device__ __forceinline__ uint32_t bucket_h(int cx, int cy, int cz, uint32_t m) {
    return part1by2h((uint32_t)cx & m) | (part1by2h((uint32_t)cy & m) << 1) | (part1by2h((uint32_t)cz & m) << 2);
}
__device__ __forceinline__ int cell_h(float c, float inv_cell) {
    float t = floorf(c * inv_cell);
    t = t < -1.0e9f ? -1.0e9f : (t > 1.0e9f ? 1.0e9f : t);
    return (int)t;
}

__global__ __launch_bounds__(256) void k_halo_pack(uint32_t n, const float* __restrict__ aabb,
                                                   const uint32_t* __restrict__ shape,
                                                   const uint32_t* __restrict__ global_id, float x_lo, float x_hi,
                                                   float reach, HaloRecord* __restrict__ out, uint64_t cap,
                                                   StepCounters* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool take = false;
    v3 lo = v3_make(0, 0, 0), hi = v3_make(0, 0, 0);
    if (i < n && shape[i] != PHYS_SPEC_SHAPE_NONE) {
        lo = ld3(aabb, 2 * i);
        hi = ld3(aabb, 2 * i + 1);
        const float r = reach > 0.0f ? reach : __uint_as_float(ctr->max_extent_bits) * 1.001f;
        take = lo.x < x_lo + r || hi.x > x_hi - r;
    }
    const unsigned long long mask = __ballot(take);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctr->n_halo, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (take) {
        const uint64_t slot = (uint64_t)base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot < cap) {
            HaloRecord r;
            r.lo[0] = lo.x; r.lo[1] = lo.y; r.lo[2] = lo.z;
            r.hi[0] = hi.x; r.hi[1] = hi.y; r.hi[2] = hi.z;
            r.gid = global_id[i];
            r.pad = 0;
            out[slot] = r;
        } else {
            flag_overflow(ctr, 8u);
        }
    }
}

// every lane of the wave calls this together: the hits are appended with one atomic per wave
__device__ __forceinline__ void emit_cross_pairs(bool hit, uint32_t j, uint32_t rgid, uint32_t* __restrict__ cross_pairs,
                                                 uint64_t cap, StepCounters* __restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctr->n_cross_pairs, (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (hit) {
            const uint64_t slot = (uint64_t)base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (slot < cap) { cross_pairs[2 * slot] = j; cross_pairs[2 * slot + 1] = rgid; }
            else flag_overflow(ctr, 8u);
        }
    }
}

// SLOTS: the last broad phase built the slot grid of small scenes (broadphase.hip): `bucket_start` then holds the
// bucket COUNTS, ids / boxes are the slot arrays (kSlots per bucket), and bodies of the overflow list are tested
// against every record directly.
constexpr uint32_t kSlots = 8;  // == kSlotsPerBucket of broadphase.hip
template <bool SLOTS>
__global__ __launch_bounds__(256) void k_halo_pairs(uint32_t n_remote, uint32_t skip_first, uint32_t skip_count,
                                                    const HaloRecord* __restrict__ remote,
                                                    const uint32_t* __restrict__ bucket_start, uint32_t table_size,
                                                    uint32_t axis_mask, const uint32_t* __restrict__ sorted_ids,
                                                    const float* __restrict__ sorted_box,
                                                    const uint32_t* __restrict__ ovf, const float* __restrict__ aabb,
                                                    const uint32_t* __restrict__ global_id,
                                                    uint32_t* __restrict__ cross_pairs, uint64_t cap,
                                                    StepCounters* __restrict__ ctr) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const float ext = __uint_as_float(ctr->max_extent_bits);
    const float cell = ext > 0.0f ? ext * 1.001f : 1.0f;
    const float inv_cell = 1.0f / cell;
    // [skip_first, skip_first + skip_count) is this rank's own block of the gathered buffer
    const bool live = k < n_remote && (k < skip_first || k >= skip_first + skip_count) && remote[k].gid != 0xFFFFFFFFu;
    aabb_t rb;
    uint32_t rgid = 0;
    int c0[3] = {0, 0, 0}, c1[3] = {-1, -1, -1};
    if (live) {
        const HaloRecord r = remote[k];
        rb.lo = v3_make(r.lo[0], r.lo[1], r.lo[2]);
        rb.hi = v3_make(r.hi[0], r.hi[1], r.hi[2]);
        rgid = r.gid;
        // owned bodies are binned by AABB centre, and a centre lies within half a cell of its box
        for (int a = 0; a < 3; ++a) {
            c0[a] = cell_h(r.lo[a] - 0.5f * cell, inv_cell);
            c1[a] = cell_h(r.hi[a] + 0.5f * cell, inv_cell);
            if (c1[a] - c0[a] > 7) c1[a] = c0[a] + 7;  // a remote box spanning > 8 cells would alias the table
        }
    }
    // wave-uniform sweep over the largest cell range in the wave
    int span[3];
    for (int a = 0; a < 3; ++a) {
        int s = live ? c1[a] - c0[a] + 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(s, off, 64); s = o > s ? o : s; }
        span[a] = s;
    }
    for (int dz = 0; dz < span[2]; ++dz)
        for (int dy = 0; dy < span[1]; ++dy)
            for (int dx = 0; dx < span[0]; ++dx) {
                const int cx = c0[0] + dx, cy = c0[1] + dy, cz = c0[2] + dz;
                const bool in = live && cx <= c1[0] && cy <= c1[1] && cz <= c1[2];
                uint32_t t = 0, t_end = 0;
                if (in) {
                    const uint32_t bk = bucket_h(cx, cy, cz, axis_mask);
                    if (SLOTS) {
                        const uint32_t have = bucket_start[bk];
                        t = kSlots * bk;
                        t_end = t + (have < kSlots ? have : kSlots);
                    } else {
                        t = bucket_start[bk];
                        t_end = bucket_start[bk + 1];
                    }
                }
                while (__any(t < t_end)) {
                    bool hit = false;
                    uint32_t j = 0;
                    if (t < t_end) {
                        j = sorted_ids[t];
                        aabb_t bj;
                        bj.lo = ld3(sorted_box, 2 * t);
                        bj.hi = ld3(sorted_box, 2 * t + 1);
                        // the candidate must really live in the scanned cell (buckets alias distant cells)
                        const bool same_cell = cell_h(0.5f * (bj.lo.x + bj.hi.x), inv_cell) == cx &&
                                               cell_h(0.5f * (bj.lo.y + bj.hi.y), inv_cell) == cy &&
                                               cell_h(0.5f * (bj.lo.z + bj.hi.z), inv_cell) == cz;
                        hit = same_cell && aabb_overlap(rb, bj) && global_id[j] < rgid;
                        ++t;
                    }
                    emit_cross_pairs(hit, j, rgid, cross_pairs, cap, ctr);
                }
            }
    if (SLOTS) {
        // bodies that found their bucket full are not in any slot: every record meets them directly
        const uint32_t n_ovf = ctr->n_grid_ovf;
        for (uint32_t o = 0; o < n_ovf; ++o) {
            const uint32_t j = ovf[o];
            bool hit = false;
            if (live) {
                aabb_t bj;
                bj.lo = ld3(aabb, 2 * j);
                bj.hi = ld3(aabb, 2 * j + 1);
                hit = aabb_overlap(rb, bj) && global_id[j] < rgid;
            }
            emit_cross_pairs(hit, j, rgid, cross_pairs, cap, ctr);
        }
    }
}

static int32_t read_counters(phys_world* w) {
    PHYS_HIP_TRY(hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}

int32_t halo_pack(phys_world* w, float x_lo, float x_hi, float reach, void* dev_out, uint64_t cap, uint64_t* n_records) {
    if (!dev_out) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    if (!(w->cfg.flags & PHYS_FLAG_COLLISIONS)) { set_error("world created without PHYS_FLAG_COLLISIONS"); return PHYS_ERR_UNSUPPORTED; }
    if (!w->grid_valid) { set_error("phys_halo_pack needs the AABBs of an update or phys_broadphase first"); return PHYS_ERR_UNSUPPORTED; }
    const uint32_t n = (uint32_t)w->n;
    PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->n_halo, 0, 4, w->stream));
    PHYS_HIP_TRY(hipMemsetAsync(dev_out, 0xFF, cap * sizeof(HaloRecord), w->stream));  // unused slots: id 0xFFFFFFFF
    if (n) {
        PHYS_PROF(w, PHYS_STAGE_MISC);
        hipLaunchKernelGGL(k_halo_pack, dim3((n + 255) / 256), dim3(256), 0, w->stream, n, w->aabb.p, w->shape.p,
                           w->global_id.p, x_lo, x_hi, reach, (HaloRecord*)dev_out, cap, w->counters.p);
    }
    if (!n_records) return PHYS_OK;  // asynchronous form: nothing returns to the host (phys_get_stats has the count)
    const int32_t rc = read_counters(w);
    if (rc != PHYS_OK) return rc;
    if (w->h_counters->overflow & 8u) { set_error("halo buffer capacity exceeded"); return PHYS_ERR_CAPACITY; }
    *n_records = w->h_counters->n_halo;
    return PHYS_OK;
}

int32_t halo_pairs(phys_world* w, const void* dev_remote, uint64_t n_remote, uint64_t skip_first, uint64_t skip_count,
                   uint64_t* n_cross) {
    if (n_remote && !dev_remote) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    if (!w->grid_valid) { set_error("phys_halo_pairs needs the grid of an update or phys_broadphase first"); return PHYS_ERR_UNSUPPORTED; }
    if (w->max_cross_pairs == 0) {
        w->max_cross_pairs = std::max<uint64_t>(4 * w->n, 4096);
        PHYS_HIP_TRY(w->cross_pairs.resize(2 * w->max_cross_pairs));
    }
    PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->n_cross_pairs, 0, 4, w->stream));
    if (n_remote && w->n) {
        const uint32_t T = w->grid_table_size;
        uint32_t bits = 0;
        while ((1u << (3 * bits)) < T) ++bits;
        PHYS_PROF(w, PHYS_STAGE_MISC);
        if (w->sorted_grid_valid)
            hipLaunchKernelGGL(k_halo_pairs<false>, dim3((unsigned)((n_remote + 255) / 256)), dim3(256), 0, w->stream,
                               (uint32_t)n_remote, (uint32_t)skip_first, (uint32_t)skip_count, (const HaloRecord*)dev_remote, w->bucket_start.p, T,
                               (1u << bits) - 1u, w->sorted_ids.p, w->sorted_box.p, nullptr, w->aabb.p, w->global_id.p, w->cross_pairs.p,
                               w->max_cross_pairs, w->counters.p);
        else  // the slot grid of small scenes
            hipLaunchKernelGGL(k_halo_pairs<true>, dim3((unsigned)((n_remote + 255) / 256)), dim3(256), 0, w->stream,
                               (uint32_t)n_remote, (uint32_t)skip_first, (uint32_t)skip_count, (const HaloRecord*)dev_remote, w->bucket_count.p, T,
                               (1u << bits) - 1u, w->slot_ids.p, w->slot_box.p, w->grid_ovf.p, w->aabb.p, w->global_id.p, w->cross_pairs.p,
                               w->max_cross_pairs, w->counters.p);
    }
    if (!n_cross) return PHYS_OK;  // asynchronous form
    const int32_t rc = read_counters(w);
    if (rc != PHYS_OK) return rc;
    if (w->h_counters->overflow & 8u) { set_error("cross-pair capacity exceeded"); return PHYS_ERR_CAPACITY; }
    *n_cross = w->h_counters->n_cross_pairs;
    return PHYS_OK;
}

}  // namespace phys
