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

__device__ __forceinline__ uint32_t bucket_h(int cx, int cy, int cz, const GridShape& g) { return grid_bucket(cx, cy, cz, g); }
__device__ __forceinline__ int cell_h(float c, float inv_cell) { return grid_cell_coord(c, inv_cell); }

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
                                                    GridShape axis_mask, const uint32_t* __restrict__ sorted_ids,
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
            if (c1[a] - c0[a] > 7) c1[a] = c0[a] + 7;  // bound of the sweep (a remote box is at most `reach` wide; buckets met twice on
                                                       // a short axis are harmless: a candidate counts only in its TRUE cell, below)
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

// ---------------------------------------------------------------------------------------------------------
// Ghost bodies (SURVEY rows E + N4): the full state of boundary bodies crosses the cut planes, and a rank sees its
// neighbours' boundary bodies as kinematic bodies in slots [n_owned, n_owned + max_ghosts) of its own arrays, so the
// ordinary pipeline collides and solves against them. 96-byte record = six float4:
//   {pos.xyz, rot.i} {rot.jkw, lin.x} {lin.yz, ang.xy} {ang.z, half.xyz} {shape, global id, 0, 0} {0, 0, 0, 0}
// Both compactions (boundary bodies -> records, gathered records -> ghost slots) are ORDERED: a count per workgroup,
// then every workgroup sums the counts in front of it and places its items at ballot-prefix offsets. No atomics, so
// the record order and the ghost slot of every remote body are functions of the data alone and a sharded run repeats
// bit for bit.
struct BodyRecord { float4 q[6]; };
static_assert(sizeof(BodyRecord) == PHYS_HALO_BODY_RECORD_BYTES, "body record is 96 bytes");

__device__ __forceinline__ bool boundary_body(uint32_t i, uint32_t n_owned, const float* __restrict__ pos,
                                              const uint32_t* __restrict__ shape, float x_lo, float x_hi, float reach) {
    if (i >= n_owned || shape[i] == PHYS_SPEC_SHAPE_NONE) return false;
    const float x = pos[3 * (size_t)i];
    return x < x_lo + reach || x > x_hi - reach;
}
__device__ __forceinline__ bool ghost_record(uint32_t k, uint32_t n_records, uint32_t skip_first, uint32_t skip_count,
                                             const BodyRecord* __restrict__ rec, float x_lo, float x_hi, float reach) {
    if (k >= n_records || (k >= skip_first && k < skip_first + skip_count)) return false;
    const float4 q4 = rec[k].q[4];
    if (__float_as_uint(q4.y) == 0xFFFFFFFFu) return false;  // empty slot of a rank's block
    const float x = rec[k].q[0].x;
    return x >= x_lo - reach && x <= x_hi + reach;
}

// MODE 0: boundary bodies of this rank, MODE 1: gathered records that reach into this rank's slab
template <int MODE>
__global__ __launch_bounds__(256) void k_halo_count(uint32_t n_items, uint32_t n_owned, const float* __restrict__ pos,
                                                    const uint32_t* __restrict__ shape, const BodyRecord* __restrict__ rec,
                                                    uint32_t skip_first, uint32_t skip_count, float x_lo, float x_hi, float reach,
                                                    uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wc[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool take = MODE == 0 ? (i < n_items && boundary_body(i, n_owned, pos, shape, x_lo, x_hi, reach))
                                : ghost_record(i, n_items, skip_first, skip_count, rec, x_lo, x_hi, reach);
    const unsigned long long mask = __ballot(take);
    if ((threadIdx.x & 63u) == 0u) wc[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// exclusive offset of this workgroup = sum of the counts of the workgroups in front of it (a few thousand at most:
// one strided pass); returns this lane's slot, or ~0 when the lane has nothing to place
__device__ __forceinline__ uint32_t ordered_slot(bool take, const uint32_t* __restrict__ block_counts, uint32_t* total_out) {
    __shared__ uint32_t s_part[4], s_wc[4], s_base;
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256u) part += block_counts[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += (uint32_t)__shfl_xor((int)part, off, 64);
    const unsigned long long mask = __ballot(take);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) { s_part[wave] = part; s_wc[wave] = (uint32_t)__popcll(mask); }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < wave; ++k) woff += s_wc[k];
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = s_base + s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    return take ? s_base + woff + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_halo_pack_bodies(uint32_t n_owned, const float* __restrict__ pos,
                                                          const float* __restrict__ rot, const float* __restrict__ vel,
                                                          const float* __restrict__ half_extent,
                                                          const uint32_t* __restrict__ shape,
                                                          const uint32_t* __restrict__ global_id,
                                                          const float* __restrict__ inv_inertia /* 9 per body */, float x_lo, float x_hi,
                                                          float reach, const uint32_t* __restrict__ block_counts,
                                                          BodyRecord* __restrict__ out, uint32_t cap, StepCounters* ctr) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool take = boundary_body(i, n_owned, pos, shape, x_lo, x_hi, reach);
    const uint32_t slot = ordered_slot(take, block_counts, &ctr->n_halo);
    if (!take) return;
    if (slot >= cap) { flag_overflow(ctr, 8u); return; }
    const v3 x = ld3(pos, i), he = ld3(half_extent, i);
    const float4 q = reinterpret_cast<const float4*>(rot)[i];
    const BodyVel bv = ld_vel(vel, i);
    BodyRecord r;
    r.q[0] = make_float4(x.x, x.y, x.z, q.x);
    r.q[1] = make_float4(q.y, q.z, q.w, bv.v.x);
    r.q[2] = make_float4(bv.v.y, bv.v.z, bv.w.x, bv.w.y);
    r.q[3] = make_float4(bv.w.z, he.x, he.y, he.z);
    // mass properties travel too: the neighbour solves its contacts with this body as a DYNAMIC body (the impulse of a contact
    // across the plane is then the two-body impulse on both sides, not an impact on an immovable wall). A diagonal inverse
    // inertia fits the record; a full tensor is flagged and that body crosses as a kinematic one, as in round 2.
    const float* I = inv_inertia + 9 * (size_t)i;
    const bool full = I[1] != 0.0f || I[2] != 0.0f || I[3] != 0.0f || I[5] != 0.0f || I[6] != 0.0f || I[7] != 0.0f;
    r.q[4] = make_float4(__uint_as_float(shape[i]), __uint_as_float(global_id[i]), bv.inv_mass, __uint_as_float(full ? 1u : 0u));
    r.q[5] = make_float4(I[0], I[4], I[8], 0.0f);
#pragma unroll
    for (int k = 0; k < 6; ++k) out[slot].q[k] = r.q[k];
}

// every ghost slot back to "nobody" before the records of this step are placed (shape NONE takes part in nothing)
__global__ __launch_bounds__(256) void k_ghost_clear(uint32_t n_owned, uint32_t n_total, uint32_t* __restrict__ shape,
                                                     uint32_t* __restrict__ global_id, StepCounters* ctr) {
    const uint32_t i = n_owned + blockIdx.x * 256u + threadIdx.x;
    if (i < n_total) { shape[i] = PHYS_SPEC_SHAPE_NONE; global_id[i] = 0xFFFFFFFFu; }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_ghosts = 0;
}

__global__ __launch_bounds__(256) void k_halo_unpack(uint32_t n_records, uint32_t skip_first, uint32_t skip_count,
                                                     const BodyRecord* __restrict__ rec, float x_lo, float x_hi, float reach,
                                                     const uint32_t* __restrict__ block_counts, uint32_t n_owned,
                                                     uint32_t max_ghosts, float* __restrict__ pos, float* __restrict__ rot,
                                                     float* __restrict__ vel, float* __restrict__ half_extent,
                                                     uint32_t* __restrict__ shape, uint32_t* __restrict__ global_id,
                                                     float* __restrict__ inv_inertia, float* __restrict__ inv_inertia_diag,
                                                     StepCounters* ctr) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const bool take = ghost_record(k, n_records, skip_first, skip_count, rec, x_lo, x_hi, reach);
    const uint32_t slot = ordered_slot(take, block_counts, &ctr->n_ghosts);
    if (!take) return;
    if (slot >= max_ghosts) { flag_overflow(ctr, 8u); return; }
    const uint32_t i = n_owned + slot;
    float4 q[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) q[j] = rec[k].q[j];
    st3(pos, i, v3_make(q[0].x, q[0].y, q[0].z));
    reinterpret_cast<float4*>(rot)[i] = make_float4(q[0].w, q[1].x, q[1].y, q[1].z);
    // A ghost is a DYNAMIC body of this world for the length of one update: it has its owner's mass and (diagonal) inverse
    // inertia, feels this update's gravity like at home (the record holds its velocity BEFORE the owner's velocity half),
    // rests on the ground and on other ghosts, and a contact between it and an owned body is solved as the two-body
    // contact it is - on both sides of the plane alike, each side keeping its own body's half of the outcome. Its state
    // here is forgotten when the next exchange brings the owner's. A body with a full inertia tensor (flagged by the
    // sender: the record holds a diagonal) stays kinematic: inverse mass and inertia 0, F / m = 0.
    const bool kinematic = __float_as_uint(q[4].w) != 0u || !(q[4].z > 0.0f);
    BodyVel bv;
    bv.v = v3_make(q[1].w, q[2].x, q[2].y); bv.inv_mass = kinematic ? 0.0f : q[4].z;
    bv.w = v3_make(q[2].z, q[2].w, q[3].x); bv.mass = kinematic ? __uint_as_float(0x7F800000u) : 1.0f / q[4].z;
    st_vel(vel, i, bv);
    const float dx = kinematic ? 0.0f : q[5].x, dy = kinematic ? 0.0f : q[5].y, dz = kinematic ? 0.0f : q[5].z;
    float* I = inv_inertia + 9 * (size_t)i;
    I[0] = dx; I[1] = 0.0f; I[2] = 0.0f; I[3] = 0.0f; I[4] = dy; I[5] = 0.0f; I[6] = 0.0f; I[7] = 0.0f; I[8] = dz;
    reinterpret_cast<float4*>(inv_inertia_diag)[i] = make_float4(dx, dy, dz, 0.0f);
    st3(half_extent, i, v3_make(q[3].y, q[3].z, q[3].w));
    shape[i] = __float_as_uint(q[4].x);
    global_id[i] = __float_as_uint(q[4].y);
}

static int32_t halo_counts_room(phys_world* w, uint64_t items) {
    const uint64_t blocks = (items + 255) / 256 + 1;
    if (w->halo_block_counts.n < blocks) PHYS_HIP_TRY(w->halo_block_counts.resize(blocks));
    return PHYS_OK;
}

int32_t halo_pack_bodies(phys_world* w, void* dev_out, uint64_t cap) {
    return halo_pack_bodies_faces(w, dev_out, cap, w->slab_lo, w->slab_hi);
}

// the bodies within reach of the given faces only (a neighbour exchange packs one block per face: the other face is
// moved out of the way)
int32_t halo_pack_bodies_faces(phys_world* w, void* dev_out, uint64_t cap, float x_lo, float x_hi) {
    if (!dev_out) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    if (w->max_ghosts == 0) { set_error("world created without phys_config.max_ghosts"); return PHYS_ERR_UNSUPPORTED; }
    if (!(w->slab_reach > 0.0f)) { set_error("phys_set_slab first"); return PHYS_ERR_UNSUPPORTED; }
    if (cap >= 0xFFFFFFFFull) { set_error("record capacity exceeds u32"); return PHYS_ERR_INVALID_ARG; }
    const uint32_t n = (uint32_t)w->n_owned;
    const int32_t rc = halo_counts_room(w, n);
    if (rc != PHYS_OK) return rc;
    PHYS_PROF(w, PHYS_STAGE_MISC);
    PHYS_HIP_TRY(hipMemsetAsync(dev_out, 0xFF, cap * sizeof(BodyRecord), w->stream));  // unused slots: global id 0xFFFFFFFF
    if (n) {
        const dim3 g((n + 255) / 256), b(256);
        hipLaunchKernelGGL(k_halo_count<0>, g, b, 0, w->stream, n, n, w->pos.p, w->shape.p, (const BodyRecord*)nullptr, 0u, 0u,
                           x_lo, x_hi, w->slab_reach, w->halo_block_counts.p);
        hipLaunchKernelGGL(k_halo_pack_bodies, g, b, 0, w->stream, n, w->pos.p, w->rot.p, w->vel.p, w->half_extent.p, w->shape.p,
                           w->global_id.p, w->inv_inertia.p, x_lo, x_hi, w->slab_reach, w->halo_block_counts.p, (BodyRecord*)dev_out,
                           (uint32_t)cap, w->counters.p);
    }
    PHYS_HIP_TRY(hipGetLastError());
    return PHYS_OK;
}

int32_t halo_unpack_ghosts(phys_world* w, const void* dev_records, uint64_t n_records, uint64_t skip_first, uint64_t skip_count) {
    if (n_records && !dev_records) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    if (w->max_ghosts == 0) { set_error("world created without phys_config.max_ghosts"); return PHYS_ERR_UNSUPPORTED; }
    if (!(w->slab_reach > 0.0f)) { set_error("phys_set_slab first"); return PHYS_ERR_UNSUPPORTED; }
    if (n_records >= 0xFFFFFFFFull) { set_error("record count exceeds u32"); return PHYS_ERR_INVALID_ARG; }
    const int32_t rc = halo_counts_room(w, n_records);
    if (rc != PHYS_OK) return rc;
    PHYS_PROF(w, PHYS_STAGE_MISC);
    const uint32_t G = (uint32_t)w->max_ghosts, n_owned = (uint32_t)w->n_owned;
    hipLaunchKernelGGL(k_ghost_clear, dim3((G + 255) / 256), dim3(256), 0, w->stream, n_owned, (uint32_t)w->n, w->shape.p,
                       w->global_id.p, w->counters.p);
    if (n_records) {
        const dim3 g((unsigned)((n_records + 255) / 256)), b(256);
        hipLaunchKernelGGL(k_halo_count<1>, g, b, 0, w->stream, (uint32_t)n_records, n_owned, (const float*)nullptr,
                           (const uint32_t*)nullptr, (const BodyRecord*)dev_records, (uint32_t)skip_first, (uint32_t)skip_count,
                           w->slab_lo, w->slab_hi, w->slab_reach, w->halo_block_counts.p);
        hipLaunchKernelGGL(k_halo_unpack, g, b, 0, w->stream, (uint32_t)n_records, (uint32_t)skip_first, (uint32_t)skip_count,
                           (const BodyRecord*)dev_records, w->slab_lo, w->slab_hi, w->slab_reach, w->halo_block_counts.p, n_owned, G,
                           w->pos.p, w->rot.p, w->vel.p, w->half_extent.p, w->shape.p, w->global_id.p, w->inv_inertia.p,
                           w->inv_inertia_diag.p, w->counters.p);
    }
    PHYS_HIP_TRY(hipGetLastError());
    w->aabbs_valid = false;
    return PHYS_OK;
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
        PHYS_PROF(w, PHYS_STAGE_MISC);
        if (w->sorted_grid_valid)
            hipLaunchKernelGGL(k_halo_pairs<false>, dim3((unsigned)((n_remote + 255) / 256)), dim3(256), 0, w->stream,
                               (uint32_t)n_remote, (uint32_t)skip_first, (uint32_t)skip_count, (const HaloRecord*)dev_remote, w->bucket_start.p, T,
                               w->grid_shape, w->sorted_ids.p, w->sorted_box.p, nullptr, w->aabb.p, w->global_id.p, w->cross_pairs.p,
                               w->max_cross_pairs, w->counters.p);
        else  // the slot grid of small scenes
            hipLaunchKernelGGL(k_halo_pairs<true>, dim3((unsigned)((n_remote + 255) / 256)), dim3(256), 0, w->stream,
                               (uint32_t)n_remote, (uint32_t)skip_first, (uint32_t)skip_count, (const HaloRecord*)dev_remote, w->bucket_count.p, T,
                               w->grid_shape, w->slot_ids.p, w->slot_box.p, w->grid_ovf.p, w->aabb.p, w->global_id.p, w->cross_pairs.p,
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
