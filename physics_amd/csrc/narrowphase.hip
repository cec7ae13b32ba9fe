// narrowphase.hip — contact generation and the deterministic manifold colouring
// that orders the solver, for gfx950. No reference counterpart; the arithmetic is the normative scalar
// spec of include/spec/collide.h and the colouring rule of include/spec/contact_solve.h.
//
// k_narrowphase: one lane per work item (ground test of a body, or one candidate pair); manifolds are
//   compacted per workgroup (wavefront ballot + popcount prefix, wave totals through LDS, ONE global
//   atomic per workgroup) and written as 88-byte records. Emission order is arbitrary.
// colouring: synchronous Jones-Plassmann rounds on the line graph. Each round is two kernels
//   (k_color_top: u64 atomicMax of priorities per body; k_color_assign: winners take the lowest free
//   colour). max / or are order-independent, so the colours are a pure function of the manifold SET.
// Algorithmic bytes: per pair 8 + 2 x 44 (pos 12, rot 16, half 12, shape 4) = 96 B read;
//   per manifold 100 B written (ids 8, count 4, normal 12, points 64, priority 8, colour 4).
#include "kernels.hpp"

namespace phys {

constexpr uint32_t kUncolored = 0xFFFFFFFFu;

__device__ __forceinline__ v3 ld3g(const float* __restrict__ p, uint32_t i) {
    return v3_make(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}

__device__ __forceinline__ geom_t load_geom(uint32_t i, const float* __restrict__ pos, const float* __restrict__ rot,
                                            const float* __restrict__ half_extent, const uint32_t* __restrict__ shape) {
    const float4 qq = reinterpret_cast<const float4*>(rot)[i];
    quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
    return geom_make(ld3g(pos, i), q, ld3g(half_extent, i), shape[i]);
}

constexpr int kNpThreads = 256;

__global__ __launch_bounds__(kNpThreads) void k_narrowphase(
    uint32_t n_ground /* bodies tested against the plane (0 = no ground) */, const uint32_t* __restrict__ pairs,
    uint64_t max_pairs, const float* __restrict__ pos, const float* __restrict__ rot,
    const float* __restrict__ half_extent, const uint32_t* __restrict__ shape, float margin, float ground,
    uint64_t max_manifolds, uint32_t* __restrict__ man_a, uint32_t* __restrict__ man_b,
    uint32_t* __restrict__ man_count, uint32_t* __restrict__ man_color, float* __restrict__ man_normal,
    float* __restrict__ man_points, uint64_t* __restrict__ man_prio, StepCounters* __restrict__ ctr) {
    __shared__ uint32_t wcount[kNpThreads / 64];
    __shared__ uint32_t block_base;
    const uint32_t np_raw = ctr->n_pairs;
    const uint32_t n_pairs = (uint64_t)np_raw < max_pairs ? np_raw : (uint32_t)max_pairs;
    const uint32_t total = n_ground + n_pairs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * kNpThreads; base < total; base += gridDim.x * kNpThreads) {
        const uint32_t idx = base + threadIdx.x;
        manifold_t m;
        m.count = 0;
        uint32_t a = 0, b = PHYS_GROUND_ID;
        if (idx < n_ground) {
            a = idx;
            if (shape[a] != PHYS_SPEC_SHAPE_NONE) {
                const geom_t ga = load_geom(a, pos, rot, half_extent, shape);
                collide_ground(&ga, ground, margin, &m);
            }
        } else if (idx < total) {
            const uint2 pr = reinterpret_cast<const uint2*>(pairs)[idx - n_ground];
            a = pr.x; b = pr.y;
            const geom_t ga = load_geom(a, pos, rot, half_extent, shape);
            const geom_t gb = load_geom(b, pos, rot, half_extent, shape);
            collide_pair(&ga, &gb, margin, &m);
        }
        const bool has = m.count > 0;
        const unsigned long long mask = __ballot(has);
        if (lane == 0) wcount[wave] = (uint32_t)__popcll(mask);
        // contact points of this wave (for the stats counter)
        uint32_t pts = has ? (uint32_t)m.count : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pts += (uint32_t)__shfl_xor((int)pts, off, 64);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int k = 0; k < kNpThreads / 64; ++k) t += wcount[k];
            uint32_t bb = 0;
            if (t) {
                bb = atomicAdd(&ctr->n_manifolds, t);
                const uint64_t room = (uint64_t)bb < max_manifolds ? max_manifolds - bb : 0;
                const uint32_t stored = (uint64_t)t <= room ? t : (uint32_t)room;
                if (stored) atomicAdd(&ctr->n_uncolored, stored);
                if (stored != t) atomicOr(&ctr->overflow, 2u);
            }
            block_base = bb;
        }
        if (lane == 0 && pts) atomicAdd(&ctr->n_contacts, pts);
        const unsigned long long gmask = __ballot(has && b == PHYS_GROUND_ID);
        if (lane == 0 && gmask) atomicAdd(&ctr->n_ground_manifolds, (uint32_t)__popcll(gmask));
        __syncthreads();
        if (has) {
            uint32_t woff = 0;
            for (int k = 0; k < wave; ++k) woff += wcount[k];
            const uint64_t slot = (uint64_t)block_base + woff + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (slot < max_manifolds) {
                man_a[slot] = a;
                man_b[slot] = b;
                man_count[slot] = (uint32_t)m.count;
                man_color[slot] = kUncolored;
                man_prio[slot] = color_priority(a, b);
                man_normal[3 * slot + 0] = m.normal.x;
                man_normal[3 * slot + 1] = m.normal.y;
                man_normal[3 * slot + 2] = m.normal.z;
                float4* o = reinterpret_cast<float4*>(man_points) + 4 * slot;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = make_float4(m.pt[k].x, m.pt[k].y, m.pt[k].z, m.depth[k]);
            }
        }
        __syncthreads();  // wcount / block_base are reused by the next trip
    }
}

// ---- colouring --------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t stored_manifolds(const StepCounters* ctr, uint64_t max_manifolds) {
    const uint32_t m = ctr->n_manifolds;
    return (uint64_t)m < max_manifolds ? m : (uint32_t)max_manifolds;
}

__global__ __launch_bounds__(256) void k_color_top(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                   const uint32_t* __restrict__ man_b,
                                                   const uint32_t* __restrict__ man_color,
                                                   const uint64_t* __restrict__ man_prio,
                                                   unsigned long long* __restrict__ top,
                                                   StepCounters* __restrict__ ctr) {
    if (ctr->n_uncolored == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->color_rounds += 1;
    const uint32_t M = stored_manifolds(ctr, max_manifolds);
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        if (man_color[m] != kUncolored) continue;
        const unsigned long long p = man_prio[m];
        atomicMax(&top[man_a[m]], p);
        const uint32_t b = man_b[m];
        if (b != PHYS_GROUND_ID) atomicMax(&top[b], p);
    }
}

__global__ __launch_bounds__(256) void k_color_assign(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                      const uint32_t* __restrict__ man_b,
                                                      uint32_t* __restrict__ man_color, uint32_t* __restrict__ man_slot,
                                                      const uint64_t* __restrict__ man_prio,
                                                      const unsigned long long* __restrict__ top,
                                                      unsigned long long* __restrict__ top_next,
                                                      unsigned long long* __restrict__ used,
                                                      StepCounters* __restrict__ ctr) {
    if (ctr->n_uncolored == 0) return;  // uniform: n_uncolored only changes through the atomics below,
                                        // and a stale non-zero read only costs an idle pass
    const uint32_t M = stored_manifolds(ctr, max_manifolds);
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t trips = (M + stride - 1) / stride;
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t trip = 0; trip < trips; ++trip, m += stride) {
        bool winner = false;
        uint32_t c = 0;
        if (m < M && man_color[m] == kUncolored) {
            const unsigned long long p = man_prio[m];
            const uint32_t a = man_a[m], b = man_b[m];
            const bool gb = b == PHYS_GROUND_ID;
            if (top[a] == p && (gb || top[b] == p)) {
                winner = true;
                unsigned long long mask = used[a];
                if (!gb) mask |= used[b];
                while (c < (uint32_t)(PHYS_MAX_COLORS - 1) && ((mask >> c) & 1ull)) ++c;
                if (((mask >> c) & 1ull)) atomicOr(&ctr->overflow, 4u);  // more than PHYS_MAX_COLORS at one body
                // the winner is the only manifold touching a or b that colours this round
                used[a] = used[a] | (1ull << c);
                if (!gb) used[b] = used[b] | (1ull << c);
                man_color[m] = c;
            } else {
                top_next[a] = 0ull;  // losers clear the other buffer for the next round
                if (!gb) top_next[b] = 0ull;
            }
        }
        // slot of each winner inside its colour: one atomic per (wave, colour)
        unsigned long long pending = __ballot(winner);
        if (pending) {
            const uint32_t wins = (uint32_t)__popcll(pending);
            uint32_t cmax = winner ? c + 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)cmax, off, 64);
                cmax = o > cmax ? o : cmax;
            }
            if (lane == 0) {
                atomicSub(&ctr->n_uncolored, wins);
                atomicMax(&ctr->n_colors, cmax);
            }
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const uint32_t c0 = (uint32_t)__shfl((int)c, leader, 64);
                const unsigned long long same = __ballot(winner && c == c0);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&ctr->color_count[c0], (uint32_t)__popcll(same));
                base = (uint32_t)__shfl((int)base, leader, 64);
                if (winner && c == c0) man_slot[m] = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                pending &= ~same;
            }
        }
    }
}

void launch_narrowphase(phys_world* w) {
    const uint32_t n = (uint32_t)w->n;
    if (n == 0) return;
    const uint32_t n_ground = (w->cfg.flags & PHYS_FLAG_GROUND_PLANE) ? n : 0u;
    const uint64_t work = (uint64_t)n_ground + w->max_pairs;
    uint64_t blocks = (work + kNpThreads - 1) / kNpThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    { PHYS_PROF(w, PHYS_STAGE_NARROW); hipLaunchKernelGGL(k_narrowphase, dim3((unsigned)blocks), dim3(kNpThreads), 0, w->stream, n_ground, w->pairs.p,
                       w->max_pairs, w->pos.p, w->rot.p, w->half_extent.p, w->shape.p, w->cfg.contact_margin,
                       w->cfg.ground_height, w->max_manifolds, w->man_a.p, w->man_b.p, w->man_count.p, w->man_color.p,
                       w->man_normal.p, w->man_points.p, w->man_prio.p, w->counters.p); }
}

static void launch_color_round(phys_world* w, uint32_t round, unsigned blocks) {
    unsigned long long* top = w->body_top.p + (round & 1u) * w->n;
    unsigned long long* top_next = w->body_top.p + ((round + 1u) & 1u) * w->n;
    { PHYS_PROF(w, PHYS_STAGE_COLOR); hipLaunchKernelGGL(k_color_top, dim3(blocks), dim3(256), 0, w->stream, w->max_manifolds, w->man_a.p, w->man_b.p,
                       w->man_color.p, w->man_prio.p, top, w->counters.p); }
    { PHYS_PROF(w, PHYS_STAGE_COLOR); hipLaunchKernelGGL(k_color_assign, dim3(blocks), dim3(256), 0, w->stream, w->max_manifolds, w->man_a.p, w->man_b.p,
                       w->man_color.p, w->man_slot.p, w->man_prio.p, top, top_next, w->body_used.p, w->counters.p); }
}

// Runs colouring rounds until the device reports no uncoloured manifold, then leaves the final counters
// in w->h_counters (the solver launch sizes come from them). One host check per step in the steady state.
void launch_coloring(phys_world* w) {
    const uint64_t n = w->n;
    if (n == 0) return;
    hipStream_t s = w->stream;
    { PHYS_PROF(w, PHYS_STAGE_COLOR); (void)hipMemsetAsync(w->body_used.p, 0, n * 8, s); }
    { PHYS_PROF(w, PHYS_STAGE_COLOR); (void)hipMemsetAsync(w->body_top.p, 0, 2 * n * 8, s); }
    uint64_t blocks64 = (w->max_manifolds + 255) / 256;
    if (blocks64 > 256 * 8) blocks64 = 256 * 8;
    const unsigned blocks = (unsigned)blocks64;
    uint32_t round = 0;
    uint32_t batch = w->color_rounds_hint;
    for (int guard = 0; guard < 64; ++guard) {
        for (uint32_t k = 0; k < batch; ++k) launch_color_round(w, round++, blocks);
        (void)hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        if (w->prof.on) w->prof.collect(s);
        if (w->h_counters->n_uncolored == 0 || w->h_counters->overflow) break;
        batch = 2;
    }
    const uint32_t used_rounds = w->h_counters->color_rounds;
    w->color_rounds_hint = used_rounds + 1 > 2 ? used_rounds + 1 : 2;
}

}  // namespace phys
