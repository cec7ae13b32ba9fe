// narrowphase.hip — contact generation and the deterministic manifold colouring
// that orders the solver, for gfx950. No reference counterpart; the arithmetic is the normative scalar
// spec of include/spec/collide.h and the colouring rule of include/spec/contact_solve.h.
//
// k_narrowphase: one lane per work item (ground test of a body, or one candidate pair); manifolds are
//   compacted per workgroup (wavefront ballot + popcount prefix, wave totals through LDS, ONE global
//   atomic per workgroup) and written as 100-byte records. Emission order is arbitrary. A manifold that
//   existed in the previous update keeps its colour (hash-table probe); the others publish round 0 of the colouring.
// colouring: synchronous Jones-Plassmann rounds on the line graph over the NEW manifolds, one launch per round
//   (k_color_round, three rotating per-body priority buffers) + k_color_finish, or everything including the
//   colour-major sort in one workgroup for small scenes (k_color_small). max / or are order-independent, so the
//   colours are a pure function of (previous colouring, manifold SET).
// Algorithmic bytes: per pair 8 + 2 x 44 (pos 12, rot 16, half 12, shape 4) = 96 B read;
//   per manifold 100 B written (ids 8, count 4, normal 12, points 64, priority 8, colour 4).
#include <cstdlib>
#include <utility>

#include "kernels.hpp"

namespace phys {

constexpr uint32_t kUncolored = 0xFFFFFFFFu;


// one body as the narrow phase sees it: three 16-byte loads from ONE 64-byte line (world.hpp `geo`, written by
// k_step_velocity_aabb of this update from the same pose the AABBs were made of)
__device__ __forceinline__ geom_t load_geom(uint32_t i, const float* __restrict__ geo) {
    const float4* g = reinterpret_cast<const float4*>(geo) + 4 * (size_t)i;
    const float4 g0 = g[0], qq = g[1], g2 = g[2];
    quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
    return geom_make(v3_make(g0.x, g0.y, g0.z), q, v3_make(g2.x, g2.y, g2.z), __float_as_uint(g0.w));
}

// kNpThreads: 128 for small scenes (latency-bound: more workgroups in flight, the LDS slice of the clipper
// halves), 256 for everything else (launch_narrowphase)
template <int kNpThreads, int kNpItems>
__global__ __launch_bounds__(kNpThreads) void k_narrowphase(
    uint32_t n_ground /* bodies tested against the plane (0 = no ground) */, uint32_t n_owned /* pairs whose FIRST body is at
    or beyond this index are skipped (= all body slots: the ghosts of a sharded world collide like everybody else) */,
    const uint32_t* __restrict__ pairs,
    uint64_t max_pairs, const float* __restrict__ geo /* 16 floats per body: {pos, shape} {rot} {half extent, AABB lo.y} */,
    float margin, float ground, uint64_t max_manifolds, uint32_t* __restrict__ man_a, uint32_t* __restrict__ man_b,
    uint32_t* __restrict__ man_color, float* __restrict__ man_geo /* 32 floats per manifold */,
    uint64_t* __restrict__ man_prio, unsigned long long* __restrict__ used,
    unsigned long long* __restrict__ top0, ulonglong2* __restrict__ cache /* persistent colour table (kernels.hpp) */,
    uint32_t cache_mask /* 0 = keep nothing this update */, uint32_t early_probe /* ask for the table entry before the shapes
    are tested */, uint32_t stamp /* of this update */,
    uint32_t* __restrict__ unc_list /* ids of the manifolds that did not keep a colour: round 0 of the colouring */,
    const uint32_t* __restrict__ man_prev /* warm starting on (non-null; the index of the pair's previous manifold itself
    travels in the manifold record) */,
    float* __restrict__ man_imp /* ... and this update's impulse records, zeroed here (a solve that never runs leaves zeros) */,
    StepCounters* __restrict__ ctr) {
    // per-wave totals of a trip, in two sets used alternately: a wave may start the next trip (and post its totals) while
    // another still reads this trip's to place its manifolds - there is no barrier at the end of a trip any more
    __shared__ uint32_t wtot[2][4][kNpThreads / 64];
    __shared__ uint32_t block_base, unc_base;
    // polygon-clipper scratch in LDS: one 32-dword slice per lane at an odd (33) dword stride, so the lanes of
    // a wave hit distinct banks; private scratch memory would go through L1/L2 instead
    constexpr int kWsStride = sizeof(clip_ws_t) / 4 + 1;
    __shared__ float ws_lds[kNpThreads * kWsStride];
    clip_ws_t* ws = reinterpret_cast<clip_ws_t*>(ws_lds + threadIdx.x * kWsStride);
    // nothing else of a lane is indexed at run time: the shapes and the manifold stay in registers, and the kernel
    // uses no scratch memory at all (a rule of this library - tests/test_build_rules.py)
    const uint32_t np_raw = ctr->n_pairs;
    const uint32_t n_pairs = (uint64_t)np_raw < max_pairs ? np_raw : (uint32_t)max_pairs;
    const uint32_t total = n_ground + n_pairs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t trip = 0;
    uint32_t acc_pts = 0, acc_ground = 0, acc_unc = 0;  // thread 0: statistics of this workgroup's trips, added once at the end
    // A trip is a chain of dependent round trips - pair, shapes, [test], table entry, `used` masks, barrier, slot
    // reservation, barrier, stores - that the 12 waves a CU's LDS admits cannot hide from each other. A lane therefore
    // tests kNpItems work items per trip, one after the other, and everything behind the test - the ballots, the two
    // barriers, the workgroup's ONE reservation - is made once for all of them.
    struct Item {
        manifold_t m;
        uint32_t a, b, col, prev_m, kept_h;
        unsigned long long prio, seen, bit;
        bool has, uncolored;
    };
    for (uint64_t base = (uint64_t)blockIdx.x * (kNpThreads * kNpItems); base < total;
         base += (uint64_t)gridDim.x * (kNpThreads * kNpItems), ++trip) {
        uint32_t* wcount = wtot[trip & 1u][0];
        uint32_t* wpts = wtot[trip & 1u][1];
        uint32_t* wground = wtot[trip & 1u][2];
        uint32_t* wunc = wtot[trip & 1u][3];
        Item item[kNpItems];
#pragma unroll
        for (int j = 0; j < kNpItems; ++j) {
            Item& it = item[j];
            manifold_t& m = it.m;
            const uint64_t idx = base + (uint64_t)j * kNpThreads + threadIdx.x;
            m.count = 0;
            uint32_t a = 0, b = PHYS_GROUND_ID;
            ulonglong2 early = make_ulonglong2(0ull, 0ull);
            uint32_t early_h = 0;
            bool have_early = false;
            if (idx < n_ground) {
                a = (uint32_t)idx;
                // the fattened AABB of this step (k_step_velocity_aabb; lo.y = lowest corner - margin) rules most bodies out
                // without their orientation being read or a corner being made: a million-cube drop has 1 % of its bodies on
                // the plane. Conservative: a body is kept unless its AABB clears ground + margin by more than rounding.
                const float4 g2 = reinterpret_cast<const float4*>(geo)[4 * (size_t)a + 2];
                const float lo_y = g2.w;
                if (lo_y <= (ground + margin) + 1.0e-3f * (1.0f + det_absf(lo_y))) {
                    const geom_t ga = load_geom(a, geo);
                    if (ga.type != PHYS_SPEC_SHAPE_NONE) collide_ground(&ga, ground, margin, &m, ws);
                }
            } else if (idx < total) {
                const uint2 pr = reinterpret_cast<const uint2*>(pairs)[idx - n_ground];
                a = pr.x; b = pr.y;
                if (a < n_owned) {
                    // the colour-table entry this pair would keep its colour from (a random 16-byte read): asked for NOW, so
                    // that it travels while the shapes are fetched and tested instead of being one more dependent round trip
                    // behind the emission below (nearly every candidate pair of a resting pile becomes a manifold)
                    // (only where most candidate pairs DO become manifolds: a stack of aligned boxes has three sliver pairs for
                    // every contact since the sliver rule, and three wasted 64-byte line fetches out of a 100+ MB table for
                    // every useful one - launch_narrowphase decides from the counts of an earlier update)
                    if (cache_mask && early_probe) {
                        early_h = (uint32_t)(color_priority(a, b) >> 20) & cache_mask;
                        early = cache[early_h];
                        have_early = true;
                    }
                    const geom_t ga = load_geom(a, geo);
                    const geom_t gb = load_geom(b, geo);
                    collide_pair(&ga, &gb, margin, &m, ws);
                }
            }
            const bool has = m.count > 0;
            // ---- everything of a manifold that does not need its slot, BEFORE the workgroup's slot reservation: the kept
            // colour (table entry asked for above; re-stamped below), the colour's mark at the two bodies, or round 0 of the
            // colouring. Their round trips then overlap the reservation's instead of following it.
            // A manifold that turns out to be beyond the capacity has then left its marks too: that update is flagged, its
            // solve skipped and its new manifolds never reach the table, so nothing of it survives.
            unsigned long long prio = 0ull, seen_a = 0ull, seen_b = 0ull, bit = 0ull;
            uint32_t col = kUncolored, prev_m = 0xFFFFFFFFu, kept_h = 0;
            bool uncolored = false;
            if (has) {
                prio = color_priority(a, b);
                // persistent colouring (contact_solve.h): a manifold that existed in the previous update keeps its
                // colour - exact 64-bit key match in the table, stamped by the previous update; re-stamped below
                if (cache_mask) {
                    const unsigned long long key = ((unsigned long long)a << 32) | b;
                    uint32_t h = (uint32_t)(prio >> 20) & cache_mask;
                    // bounded walk: the table is rebuilt every PHYS_COLOR_CACHE_PERIOD updates and holds 1.5 slots per manifold
                    // slot, but a chain of live and dead entries without an empty slot must end the walk, not hang the GPU
                    bool ended = false;
                    for (uint32_t walked = 0; walked < kColorTableMaxWalk; ++walked) {
                        const ulonglong2 e = (have_early && h == early_h) ? early : cache[h];
                        have_early = false;
                        if (e.x == key) {
                            if ((uint32_t)(e.y >> 32) + 1u == stamp) {
                                col = (uint32_t)e.y & 63u;
                                prev_m = ((uint32_t)e.y >> 6);  // the pair's manifold of the previous update
                                kept_h = h;                     // re-stamped below, once this update's slot is known
                            }
                            ended = true;
                            break;  // a dead entry of this key: no live one follows
                        }
                        if (e.x == ~0ull) { ended = true; break; }  // empty slot: never seen
                        h = (h + 1) & cache_mask;
                    }
                    // a walk given up on might have passed over a colour the oracle's map keeps: never silently (bit 6: the
                    // update is flagged and its solve skipped, like any other capacity miss)
                    if (!ended) flag_overflow(ctr, 64u);
                }
                if (col != kUncolored) {
                    bit = 1ull << col;
                    seen_a = atomicOr(&used[a], bit);  // looked at after the barrier below
                    if (b != PHYS_GROUND_ID) seen_b = atomicOr(&used[b], bit);
                } else {
                    uncolored = true;
                    // round 0 of the colouring: per-body maximum priority (order-independent u64 max)
                    atomicMax(&top0[a], prio);
                    if (b != PHYS_GROUND_ID) atomicMax(&top0[b], prio);
                }
            }
            it.a = a; it.b = b; it.col = col; it.prev_m = prev_m; it.kept_h = kept_h;
            it.prio = prio; it.seen = seen_a | seen_b; it.bit = bit; it.has = has; it.uncolored = uncolored;
        }
        // manifolds of the wave: item-major (all of item 0, then all of item 1), lanes in order inside an item
        unsigned long long mask[kNpItems], umask[kNpItems];
        uint32_t n_has = 0, n_gnd = 0, n_unc = 0, pts = 0;
#pragma unroll
        for (int j = 0; j < kNpItems; ++j) {
            mask[j] = __ballot(item[j].has);
            umask[j] = __ballot(item[j].uncolored);
            n_has += (uint32_t)__popcll(mask[j]);
            n_unc += (uint32_t)__popcll(umask[j]);
            n_gnd += (uint32_t)__popcll(__ballot(item[j].has && item[j].b == PHYS_GROUND_ID));
            pts += item[j].has ? (uint32_t)item[j].m.count : 0u;  // contact points (for the stats counter)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pts += (uint32_t)__shfl_xor((int)pts, off, 64);
        if (lane == 0) {
            wcount[wave] = n_has;
            wpts[wave] = pts;
            wground[wave] = n_gnd;
            wunc[wave] = n_unc;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // one set of global atomics per workgroup: same-address atomics serialise chip-wide
            uint32_t t = 0, tp = 0, tg = 0, tu = 0;
            for (int k = 0; k < kNpThreads / 64; ++k) { t += wcount[k]; tp += wpts[k]; tg += wground[k]; tu += wunc[k]; }
            uint32_t bb = 0, ub = 0;
            if (t) {
                // ONE reservation for both lists: manifold slots (low word) and this workgroup's stretch of the round-0 list
                // of uncoloured manifolds (high word); see StepCounters
                const unsigned long long got = atomicAdd(reinterpret_cast<unsigned long long*>(&ctr->n_manifolds),
                                                         ((unsigned long long)tu << 32) | t);
                bb = (uint32_t)got;
                ub = (uint32_t)(got >> 32);
                const uint64_t room = (uint64_t)bb < max_manifolds ? max_manifolds - bb : 0;
                const uint32_t stored = (uint64_t)t <= room ? t : (uint32_t)room;
                if (stored != t) flag_overflow(ctr, 2u);
                acc_pts += tp; acc_ground += tg; acc_unc += tu;
            }
            block_base = bb;
            unc_base = ub;
        }
        __syncthreads();
        uint32_t woff = 0, uoff = 0;
        for (int k = 0; k < wave; ++k) { woff += wcount[k]; uoff += wunc[k]; }
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < kNpItems; ++j) {
            const Item& it = item[j];
            if (it.has) {
                const manifold_t& m = it.m;
                const uint32_t a = it.a, b = it.b, col = it.col;
                const uint64_t slot = (uint64_t)block_base + woff + (uint32_t)__popcll(mask[j] & below);
                if (slot < max_manifolds) {
                    man_a[slot] = a;
                    man_b[slot] = b;
                    man_prio[slot] = it.prio;
                    man_color[slot] = col;
                    // a kept entry is re-stamped with this update's manifold index (one 8-byte store to the line the probe read)
                    if (col != kUncolored) cache[it.kept_h].y = ((unsigned long long)stamp << 32) | ((unsigned long long)((uint32_t)slot & 0x3FFFFFFu) << 6) | col;
                    if (man_prev) {
                        float4* imp = reinterpret_cast<float4*>(man_imp) + 3 * slot;
                        imp[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); imp[1] = imp[0]; imp[2] = imp[0];
                    }
                    // (Measured and dropped: staging the records of a wave in LDS and copying them out as whole 128-byte lines -
                    // 4.5 write requests per manifold become 2 - left the kernel at 0.52 ms on C5: it waits on its chain of
                    // dependent round trips, not on the write path.)
                    float4* o = reinterpret_cast<float4*>(man_geo) + 8 * slot;  // one 128-byte line per manifold, 96 bytes used
                    // {a, b, count, kept colour + 1 (0: new in this update, coloured later: man_color)} {normal, index of the pair's
                    // manifold in the previous update or ~0}: what k_rows_build would otherwise gather, 4 bytes out of a 64-byte
                    // sector each, from two more arrays through the row permutation
                    o[0] = make_float4(__uint_as_float(a), __uint_as_float(b), __uint_as_float((uint32_t)m.count),
                                       __uint_as_float(col != kUncolored ? col + 1u : 0u));
                    o[1] = make_float4(m.normal.x, m.normal.y, m.normal.z, __uint_as_float(it.prev_m));
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[2 + k] = make_float4(m.pt[k].x, m.pt[k].y, m.pt[k].z, m.depth[k]);
                }
                if (it.uncolored) {
                    // (a manifold beyond the capacity - update flagged, never solved - names the last slot: the list must hold
                    // ids of stored manifolds only, whatever else happens to that update)
                    const uint32_t at = unc_base + uoff + (uint32_t)__popcll(umask[j] & below);
                    unc_list[at < max_manifolds ? at : (uint32_t)max_manifolds - 1u] =
                        slot < max_manifolds ? (uint32_t)slot : (uint32_t)max_manifolds - 1u;
                } else if ((it.seen & it.bit) != 0ull) {
                    // order-independent. A kept colour that was ALREADY in use at one of the bodies means the previous
                    // colouring was not proper (it saturated at PHYS_MAX_COLORS): two rows of one colour on one body
                    // would race in the solver, so the step is flagged like any other colour overflow (no solve)
                    flag_overflow(ctr, 4u);
                }
            }
            woff += (uint32_t)__popcll(mask[j]);
            uoff += (uint32_t)__popcll(umask[j]);
        }
        // (block_base / unc_base are rewritten behind the next trip's first barrier, which every wave reaches only after it
        // has placed this trip's manifolds; the per-wave totals alternate between two sets)
    }
    if (threadIdx.x == 0) {
        if (acc_pts) atomicAdd(&ctr->n_contacts, acc_pts);
        if (acc_ground) atomicAdd(&ctr->n_ground_manifolds, acc_ground);
        if (acc_unc) { atomicAdd(&ctr->n_uncolored, acc_unc); atomicAdd(&ctr->n_new_manifolds, acc_unc); }
    }
}

// ---- colouring --------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t stored_manifolds(const StepCounters* ctr, uint64_t max_manifolds) {
    const uint32_t m = ctr->n_manifolds;
    return (uint64_t)m < max_manifolds ? m : (uint32_t)max_manifolds;
}

constexpr int kColorThreads = 1024;
constexpr int kColorStage = 8192;  // losers of one workgroup and round staged in LDS (k_color_round)

// Ordering between the waves of ONE workgroup that talk through global memory (single-workgroup colouring
// kernels): every wave's stores and atomics have been performed at the L2 once its vmcnt has drained, and the
// readers load past the L1 (sc1 atomic loads), so draining + the workgroup barrier is all it takes. An
// agent-scope __threadfence() here would write back and invalidate caches of the whole XCD from all 16 waves
// (microseconds per round) for data that never leaves this CU's path to its L2.
__device__ __forceinline__ void drain_stores_for_workgroup() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One synchronous Jones-Plassmann round in ONE launch. Three per-body priority buffers rotate:
//   top      (read)   maxima over the manifolds uncoloured at the start of this round - complete;
//   top_next (atomic) losers of this round = exactly the manifolds uncoloured at the start of the next
//                     round publish their priority there (it was cleared one round ago);
//   top_clr  (store)  the buffer read one round ago, cleared at the losers' bodies for the round after next.
// Round 0's `top` is filled by k_narrowphase at emission time.
// body of one round for the manifolds m = first, first + stride, ...; returns this lane's wins
// AGG: `next_list` is a list in global memory shared by all workgroups of the launch. Same-address atomics serialise
// chip-wide (~88 per microsecond), so the losers of a workgroup are staged in LDS (`stage`, one LDS atomic per wave and
// trip) and the caller appends them with ONE global atomic; only what does not fit the stage goes out directly.
template <bool BYPASS_L1, bool AGG = false>
__device__ __forceinline__ uint32_t color_round_lanes(uint32_t first, uint32_t stride, uint32_t M,
                                                      const uint32_t* list /* null: every manifold; else M ids */,
                                                      uint32_t* next_list /* non-null: `list` holds uncoloured ids only, and
                                                                             the losers of this round are appended here */,
                                                      uint32_t* next_count,
                                                      const uint32_t* __restrict__ man_a, const uint32_t* __restrict__ man_b,
                                                      uint32_t* __restrict__ man_color, const uint64_t* __restrict__ man_prio,
                                                      const unsigned long long* top, unsigned long long* top_next,
                                                      unsigned long long* top_clr, unsigned long long* used,
                                                      StepCounters* __restrict__ ctr, uint32_t* stage = nullptr,
                                                      uint32_t* stage_n = nullptr, uint32_t stage_cap = 0) {
    uint32_t wins = 0;
    for (uint32_t i = first; i < M; i += stride) {
        // (single-launch loops: a global list was written by other waves of this workgroup one round ago)
        const uint32_t m = list ? (BYPASS_L1 ? __hip_atomic_load(&list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : list[i]) : i;
        if (!next_list && man_color[m] != kUncolored) continue;
        const unsigned long long p = man_prio[m];
        const uint32_t a = man_a[m], b = man_b[m];
        const bool gb = b == PHYS_GROUND_ID;
        bool lose = false;
        // BYPASS_L1 (single-launch finish loop): words other waves changed with atomics inside this launch
        // must come from L2, not from a line this CU cached rounds ago
        const unsigned long long ta = BYPASS_L1 ? __hip_atomic_load(&top[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : top[a];
        unsigned long long tb = p;
        if (!gb) tb = BYPASS_L1 ? __hip_atomic_load(&top[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : top[b];
        if (ta == p && tb == p) {
            unsigned long long mask = BYPASS_L1 ? __hip_atomic_load(&used[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : used[a];
            unsigned long long mb = 0ull;
            if (!gb) { mb = BYPASS_L1 ? __hip_atomic_load(&used[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : used[b]; }
            const unsigned long long ma = mask;
            mask |= mb;
            uint32_t c = 0;
            while (c < (uint32_t)(PHYS_MAX_COLORS - 1) && ((mask >> c) & 1ull)) ++c;
            if (((mask >> c) & 1ull)) flag_overflow(ctr, 4u);  // more than PHYS_MAX_COLORS at one body
            // the winner is the only manifold touching a or b that colours this round
            if (BYPASS_L1) {
                __hip_atomic_store(&used[a], ma | (1ull << c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!gb) __hip_atomic_store(&used[b], mb | (1ull << c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                used[a] = ma | (1ull << c);
                if (!gb) used[b] = mb | (1ull << c);
            }
            man_color[m] = c;
            ++wins;
        } else {
            lose = true;
            if (next_list && !AGG) next_list[atomicAdd(next_count, 1u)] = m;
            atomicMax(&top_next[a], p);
            if (BYPASS_L1) __hip_atomic_store(&top_clr[a], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else top_clr[a] = 0ull;
            if (!gb) {
                atomicMax(&top_next[b], p);
                if (BYPASS_L1) __hip_atomic_store(&top_clr[b], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else top_clr[b] = 0ull;
            }
        }
        if (AGG) {
            const unsigned long long losers = __ballot(lose);
            if (losers) {
                const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)losers) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(stage_n, (uint32_t)__popcll(losers));  // LDS
                base = (uint32_t)__shfl((int)base, leader, 64);
                if (lose) {
                    const uint32_t at = base + (uint32_t)__popcll(losers & ((1ull << lane) - 1ull));
                    if (at < stage_cap) stage[at] = m;
                    else next_list[atomicAdd(next_count, 1u)] = m;  // the stage is full (a launch far smaller than its list)
                }
            }
        }
    }
    return wins;
}

// One synchronous Jones-Plassmann round in ONE launch. Three per-body priority buffers rotate:
//   top      (read)   maxima over the manifolds uncoloured at the start of this round - complete;
//   top_next (atomic) losers of this round = exactly the manifolds uncoloured at the start of the next
//                     round publish their priority there (it was cleared one round ago);
//   top_clr  (store)  the buffer read one round ago, cleared at the losers' bodies for the round after next.
// Round 0's `top` is filled by k_narrowphase at emission time.
// The round runs over the LIST of the manifolds that were uncoloured at its start (round 0: written by the narrow
// phase; round r + 1: the losers of round r) - a steady pile has a few per cent of new manifolds per update, and
// scanning the colours of all of them in every round was 7.4 us per launch on C5, twenty times per step.
__global__ __launch_bounds__(kColorThreads) void k_color_round(uint32_t round, uint64_t max_manifolds,
                                                              const uint32_t* __restrict__ man_a,
                                                              const uint32_t* __restrict__ man_b,
                                                              uint32_t* __restrict__ man_color,
                                                              const uint64_t* __restrict__ man_prio,
                                                              const unsigned long long* __restrict__ top,
                                                              unsigned long long* __restrict__ top_next,
                                                              unsigned long long* __restrict__ top_clr,
                                                              unsigned long long* __restrict__ used,
                                                              const uint32_t* __restrict__ list, uint32_t* __restrict__ next_list,
                                                              StepCounters* __restrict__ ctr) {
    // block-uniform early exit (every thread must act on the SAME read: a barrier follows)
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_wins[kColorThreads / 64];
    if (threadIdx.x == 0) {
        const uint32_t c = ctr->unc_count[round % 3u];
        s_count = (uint64_t)c < max_manifolds ? c : (uint32_t)max_manifolds;
        // the counter read one round ago is appended to one round from now
        if (blockIdx.x == 0) ctr->unc_count[(round + 2u) % 3u] = 0u;
    }
    __syncthreads();
    const uint32_t count = s_count;
    if (blockIdx.x * blockDim.x >= count) return;
    __shared__ uint32_t s_stage[kColorStage];
    __shared__ uint32_t s_stage_n, s_stage_base;
    if (threadIdx.x == 0) s_stage_n = 0;
    __syncthreads();
    uint32_t* next_count = &ctr->unc_count[(round + 1u) % 3u];
    uint32_t wins = color_round_lanes<false, true>(blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, count, list, next_list,
                                                   next_count, man_a, man_b, man_color, man_prio, top, top_next, top_clr, used, ctr,
                                                   s_stage, &s_stage_n, (uint32_t)kColorStage);
    __syncthreads();
    const uint32_t staged = s_stage_n < (uint32_t)kColorStage ? s_stage_n : (uint32_t)kColorStage;
    if (threadIdx.x == 0 && staged) s_stage_base = atomicAdd(next_count, staged);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < staged; i += kColorThreads) next_list[s_stage_base + i] = s_stage[i];
    // ONE global atomic per workgroup (same-address atomics serialise chip-wide at ~88 per microsecond)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wins += (uint32_t)__shfl_xor((int)wins, off, 64);
    if ((threadIdx.x & 63) == 0) s_wins[threadIdx.x >> 6] = wins;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < kColorThreads / 64; ++k) t += s_wins[k];
        if (t) {
            atomicSub(&ctr->n_uncolored, t);
            // rounds used = index of the last round that coloured something + 1 (every round with an
            // uncoloured manifold colours at least the one of highest priority)
            if (round + 1 > ctr->color_rounds) atomicMax(&ctr->color_rounds, round + 1);
        }
    }
}

// Runs the rounds that are still needed after the launched ones, inside ONE workgroup (barrier between rounds), over
// the same shrinking lists, so the host never has to ask the device whether the colouring is complete. Normally
// nothing or a handful of manifolds is left.
__global__ __launch_bounds__(kColorThreads) void k_color_finish(uint32_t round, uint64_t max_manifolds,
                                                               const uint32_t* __restrict__ man_a,
                                                               const uint32_t* __restrict__ man_b,
                                                               uint32_t* __restrict__ man_color,
                                                               const uint64_t* __restrict__ man_prio,
                                                               unsigned long long* __restrict__ state /*4n*/, uint64_t n,
                                                               uint32_t* __restrict__ lists /* 2 x max_manifolds */,
                                                               StepCounters* __restrict__ ctr) {
    __shared__ uint32_t s_left;
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_wins[kColorThreads / 64];
    if (threadIdx.x == 0) {
        s_left = ctr->n_uncolored;
        const uint32_t c = ctr->unc_count[round % 3u];  // what round `round` would have read
        s_cnt[round & 1u] = (uint64_t)c < max_manifolds ? c : (uint32_t)max_manifolds;
    }
    __syncthreads();
    const uint32_t left0 = s_left;
    uint32_t left = left0;
    unsigned long long* used = state;
    while (left != 0) {
        unsigned long long* top = state + (1 + round % 3) * n;
        unsigned long long* top_next = state + (1 + (round + 1) % 3) * n;
        unsigned long long* top_clr = state + (1 + (round + 2) % 3) * n;
        const uint32_t cur = round & 1u, nxt = cur ^ 1u;
        if (threadIdx.x == 0) s_cnt[nxt] = 0;
        __syncthreads();
        uint32_t wins = color_round_lanes<true>(threadIdx.x, kColorThreads, s_cnt[cur], lists + cur * max_manifolds, lists + nxt * max_manifolds,
                                                &s_cnt[nxt], man_a, man_b, man_color, man_prio, top, top_next, top_clr, used, ctr);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wins += (uint32_t)__shfl_xor((int)wins, off, 64);
        if ((threadIdx.x & 63) == 0) s_wins[threadIdx.x >> 6] = wins;
        drain_stores_for_workgroup();  // this round's stores and atomics are performed before anyone starts the next
        __syncthreads();
        uint32_t t = 0;
        for (int k = 0; k < kColorThreads / 64; ++k) t += s_wins[k];
        left -= t;
        ++round;
        __syncthreads();
        if (t == 0) break;  // cannot happen (the highest priority always wins); never spin
    }
    if (threadIdx.x == 0 && left != left0) {
        ctr->n_uncolored = left;
        ctr->color_rounds = round;
    }
}

// ---- colour-major renumbering: counting sort of the manifolds by colour ---------------------------
// hist (per-workgroup colour histogram) -> offsets (one workgroup scans colour-major) -> place.
// No global atomics; the order inside a colour is (workgroup, arrival), which nothing depends on.
constexpr int kSortBlocksMax = 512;  // most workgroups of the hist / place kernels (the launch picks nb <= this)
constexpr int kSortChunk = 4096;   // manifolds per workgroup trip

__global__ __launch_bounds__(1024) void k_color_hist(uint64_t max_manifolds, const uint32_t* __restrict__ man_color,
                                                     uint32_t* __restrict__ block_hist /*[colour][nb]*/, uint32_t nb,
                                                     const StepCounters* __restrict__ ctr) {
    __shared__ uint32_t h[PHYS_MAX_COLORS];
    if (threadIdx.x < PHYS_MAX_COLORS) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t M = stored_manifolds(ctr, max_manifolds);
    for (uint32_t base = blockIdx.x * kSortChunk; base < M; base += gridDim.x * kSortChunk) {
#pragma unroll
        for (int k = 0; k < kSortChunk / 1024; ++k) {
            const uint32_t m = base + k * 1024 + threadIdx.x;
            if (m < M) {
                const uint32_t c = man_color[m];
                if (c < (uint32_t)PHYS_MAX_COLORS) atomicAdd(&h[c], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < PHYS_MAX_COLORS) block_hist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// one workgroup: exclusive scan of block_hist in colour-major order (in place) + per-colour totals
__global__ __launch_bounds__(1024) void k_color_offsets(uint32_t* __restrict__ block_hist, uint32_t nb, StepCounters* __restrict__ ctr) {
    // every WAVE owns a run of consecutive entries, read 64 at a time (coalesced, all loads in flight at once), scanned
    // with shuffles and a running carry; then one scan of the 16 wave totals (38 us for the 32 dependent block-wide passes
    // of the first version at 512 workgroups, 59 us with a run per THREAD - 32 dependent uncoalesced loads - 8 us so)
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const uint32_t kPerColor = nb;
    const uint32_t kTotal = PHYS_MAX_COLORS * kPerColor;   // <= 64 * 512
    const uint32_t per_wave = (kTotal + 15u) / 16u;        // nb is a power of two: a multiple of 4
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t w_begin = wave * per_wave < kTotal ? wave * per_wave : kTotal;
    const uint32_t w_end = w_begin + per_wave < kTotal ? w_begin + per_wave : kTotal;
    constexpr int kTrips = (PHYS_MAX_COLORS * kSortBlocksMax / 16 + 63) / 64;  // 32, eight at a time (registers)
    constexpr int kHalf = kTrips / 4;
    uint32_t v[kHalf];
    uint32_t run = 0;  // sum of this wave's entries in front of the current trip
    // pass 1: the wave's total; pass 2 (after the scan of the wave totals): the exclusive offsets, written in place
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t base = 0;
        if (pass == 1) {
            if (lane == 0) wtot[wave] = run;
            __syncthreads();
            for (uint32_t k = 0; k < wave; ++k) base += wtot[k];
            if (threadIdx.x == 1023) carry_s = base + run;
            run = 0;
        }
        for (int half = 0; half < 4; ++half) {
            const uint32_t h_begin = w_begin + (uint32_t)(half * kHalf) * 64u;
            if (h_begin >= w_end) break;  // wave-uniform
#pragma unroll
            for (int k = 0; k < kHalf; ++k) {
                const uint32_t i = h_begin + (uint32_t)k * 64u + lane;
                v[k] = i < w_end ? block_hist[i] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kHalf; ++k) {
                uint32_t inc = v[k];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
                    if (lane >= (uint32_t)off) inc += o;
                }
                const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
                const uint32_t i = h_begin + (uint32_t)k * 64u + lane;
                if (pass == 1 && i < w_end) {
                    const uint32_t excl = base + run + inc - v[k];
                    block_hist[i] = excl;
                    if (i % kPerColor == 0) ctr->color_start[i / kPerColor] = excl;
                }
                run += total;
            }
        }
    }
    uint32_t ncol = 0;
    __syncthreads();
    if (threadIdx.x == 0) ctr->color_start[PHYS_MAX_COLORS] = carry_s;
    __syncthreads();
    if (threadIdx.x < PHYS_MAX_COLORS) {
        const uint32_t cnt = ctr->color_start[threadIdx.x + 1] - ctr->color_start[threadIdx.x];
        ctr->color_count[threadIdx.x] = cnt;
        uint32_t cmax = cnt ? threadIdx.x + 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)cmax, off, 64);
            cmax = o > cmax ? o : cmax;
        }
        ncol = cmax;
        if (threadIdx.x == 0) ctr->n_colors = ncol;
    }
}

__global__ __launch_bounds__(1024) void k_color_place(uint64_t max_manifolds, const uint32_t* __restrict__ man_color,
                                                      const uint32_t* __restrict__ block_off /*scanned block_hist*/, uint32_t nb,
                                                      uint32_t* __restrict__ row_src, const StepCounters* __restrict__ ctr,
                                                      StepCounters* snap_out /* host-mapped, may be null */) {
    __shared__ uint32_t cursor[PHYS_MAX_COLORS];
    if (threadIdx.x < PHYS_MAX_COLORS) cursor[threadIdx.x] = block_off[threadIdx.x * nb + blockIdx.x];
    counters_snapshot(ctr, snap_out);
    __syncthreads();
    const uint32_t M = stored_manifolds(ctr, max_manifolds);
    for (uint32_t base = blockIdx.x * kSortChunk; base < M; base += gridDim.x * kSortChunk) {
#pragma unroll
        for (int k = 0; k < kSortChunk / 1024; ++k) {
            const uint32_t m = base + k * 1024 + threadIdx.x;
            if (m < M) {
                const uint32_t c = man_color[m];
                if (c < (uint32_t)PHYS_MAX_COLORS) row_src[atomicAdd(&cursor[c], 1u)] = m;
            }
        }
    }
}

// Small scenes (<= 40k manifolds): the WHOLE colouring stage in ONE launch of ONE workgroup - every
// Jones-Plassmann round (over a list of the uncoloured manifolds gathered into LDS: with persistent colouring
// only the new ones), the colour-major counting sort, and the snapshot of the counters into pinned host memory
// (the launch-size hints of later steps) - instead of ~3 round launches + finish + sort + a copy.
constexpr int kSmallList = 6144;  // ids per list; two lists: the uncoloured of this round / of the next
constexpr int kSmallTrips = 40;  // manifolds per thread kept in registers: 40 x 1024 = the `small` limit of launch_coloring
__global__ __launch_bounds__(kColorThreads) void k_color_small(uint64_t max_manifolds, const uint32_t* __restrict__ man_a,
                                                              const uint32_t* __restrict__ man_b, uint32_t* man_color,
                                                              const uint64_t* __restrict__ man_prio,
                                                              unsigned long long* __restrict__ state /*4n*/, uint64_t n,
                                                              uint32_t* __restrict__ row_src, StepCounters* ctr,
                                                              StepCounters* snap_out /* host-mapped, may be null */) {
    __shared__ uint32_t s_list[2][kSmallList];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_n, s_left;
    __shared__ uint32_t s_wins[kColorThreads / 64];
    __shared__ uint32_t h[PHYS_MAX_COLORS], cursor[PHYS_MAX_COLORS];
    const uint32_t M = stored_manifolds(ctr, max_manifolds);
    if (threadIdx.x == 0) { s_n = 0; s_left = ctr->n_uncolored; }
    if (threadIdx.x < PHYS_MAX_COLORS) h[threadIdx.x] = 0;
    // the colours of this thread's manifolds (m = k * 1024 + thread) stay in registers for all three passes
    // (gather the uncoloured, histogram, placement); a larger M than the launch expected takes the slow loops
    const bool in_regs = M <= (uint32_t)(kSmallTrips * kColorThreads);
    uint32_t col[kSmallTrips];
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kSmallTrips; ++k) {
            const uint32_t m = k * kColorThreads + threadIdx.x;
            col[k] = m < M ? man_color[m] : 0xFFFFFFFEu;  // neither a colour nor kUncolored
        }
    }
    __syncthreads();
    uint32_t left = s_left;
    if (left != 0 && left <= M) {
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kSmallTrips; ++k) {
                if (col[k] == kUncolored) {
                    const uint32_t at = atomicAdd(&s_n, 1u);
                    if (at < (uint32_t)kSmallList) s_list[0][at] = k * kColorThreads + threadIdx.x;
                }
            }
        } else {
            for (uint32_t m = threadIdx.x; m < M; m += kColorThreads) {
                if (man_color[m] == kUncolored) {
                    const uint32_t at = atomicAdd(&s_n, 1u);
                    if (at < (uint32_t)kSmallList) s_list[0][at] = m;
                }
            }
        }
        __syncthreads();
        // The rounds run over a LIST of the uncoloured manifolds that shrinks with every round (the losers of a round
        // are the list of the next one). A full re-colouring starts with more than a list holds: it scans all
        // manifolds until few enough are left.
        bool listed = s_n <= (uint32_t)kSmallList;
        uint32_t cur = 0;
        if (threadIdx.x == 0) s_cnt[0] = s_n;
        unsigned long long* used = state;
        uint32_t round = 0;
        while (left != 0) {
            unsigned long long* top = state + (1 + round % 3) * n;
            unsigned long long* top_next = state + (1 + (round + 1) % 3) * n;
            unsigned long long* top_clr = state + (1 + (round + 2) % 3) * n;
            const uint32_t nxt = cur ^ 1u;
            if (threadIdx.x == 0) s_cnt[nxt] = 0;
            __syncthreads();
            uint32_t wins = listed
                ? color_round_lanes<true>(threadIdx.x, kColorThreads, s_cnt[cur], s_list[cur], s_list[nxt], &s_cnt[nxt], man_a, man_b,
                                          man_color, man_prio, top, top_next, top_clr, used, ctr)
                : color_round_lanes<true>(threadIdx.x, kColorThreads, M, nullptr, nullptr, nullptr, man_a, man_b, man_color,
                                          man_prio, top, top_next, top_clr, used, ctr);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wins += (uint32_t)__shfl_xor((int)wins, off, 64);
            if ((threadIdx.x & 63) == 0) s_wins[threadIdx.x >> 6] = wins;
            drain_stores_for_workgroup();  // this round's stores and atomics are performed before anyone starts the next
            __syncthreads();
            uint32_t t = 0;
            for (int k = 0; k < kColorThreads / 64; ++k) t += s_wins[k];
            left -= t;
            ++round;
            if (t == 0) break;  // cannot happen (the highest priority always wins); never spin
            if (listed) {
                cur = nxt;
            } else if (left != 0 && left <= (uint32_t)kSmallList) {
                // few enough are left: list them (each thread looks at the manifolds it has been handling itself)
                if (threadIdx.x == 0) s_cnt[0] = 0;
                __syncthreads();
                for (uint32_t m = threadIdx.x; m < M; m += kColorThreads)
                    if (__hip_atomic_load(&man_color[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kUncolored)
                        s_list[0][atomicAdd(&s_cnt[0], 1u)] = m;
                listed = true;
                cur = 0;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            ctr->n_uncolored = left;
            ctr->color_rounds = round;
        }
        if (in_regs) {  // colours other waves of this workgroup wrote: read past the L1
#pragma unroll
            for (int k = 0; k < kSmallTrips; ++k)
                if (col[k] == kUncolored)
                    col[k] = __hip_atomic_load(&man_color[k * kColorThreads + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // colour-major counting sort
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kSmallTrips; ++k) if (col[k] < (uint32_t)PHYS_MAX_COLORS) atomicAdd(&h[col[k]], 1u);
    } else {
        for (uint32_t m = threadIdx.x; m < M; m += kColorThreads) {
            const uint32_t c = __hip_atomic_load(&man_color[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c < (uint32_t)PHYS_MAX_COLORS) atomicAdd(&h[c], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < PHYS_MAX_COLORS) {  // one wave: exclusive scan of the 64 counts
        const uint32_t cnt = h[threadIdx.x];
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
            if ((int)threadIdx.x >= off) inc += o;
        }
        const uint32_t start = inc - cnt;
        cursor[threadIdx.x] = start;
        ctr->color_start[threadIdx.x] = start;
        ctr->color_count[threadIdx.x] = cnt;
        if (threadIdx.x == 63) ctr->color_start[PHYS_MAX_COLORS] = inc;
        uint32_t cmax = cnt ? threadIdx.x + 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)cmax, off, 64);
            cmax = o > cmax ? o : cmax;
        }
        if (threadIdx.x == 0) ctr->n_colors = cmax;
    }
    drain_stores_for_workgroup();
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kSmallTrips; ++k)
            if (col[k] < (uint32_t)PHYS_MAX_COLORS) row_src[atomicAdd(&cursor[col[k]], 1u)] = k * kColorThreads + threadIdx.x;
    } else {
        for (uint32_t m = threadIdx.x; m < M; m += kColorThreads) {
            const uint32_t c = __hip_atomic_load(&man_color[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c < (uint32_t)PHYS_MAX_COLORS) row_src[atomicAdd(&cursor[c], 1u)] = m;
        }
    }
    if (snap_out) {  // the counters as they stand after the colouring stage
        const uint32_t words = (uint32_t)(sizeof(StepCounters) / 4);
        if (threadIdx.x < words)
            reinterpret_cast<uint32_t*>(snap_out)[threadIdx.x] =
                __hip_atomic_load(reinterpret_cast<uint32_t*>(ctr) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

void launch_narrowphase(phys_world* w) {
    const uint32_t n = (uint32_t)w->n;
    if (n == 0) return;
    // ghost bodies of a sharded world (slots behind the owned bodies) are dynamic bodies of this world for one update
    // (halo.hip k_halo_unpack): they rest on the ground and on each other like everybody else
    const uint32_t n_owned = n;
    const uint32_t n_ground = (w->cfg.flags & PHYS_FLAG_GROUND_PLANE) ? n : 0u;
    const uint64_t work = (uint64_t)n_ground + w->max_pairs;
    // colouring state of the step: used masks + three rotating priority buffers (one memset); the narrow
    // phase publishes round 0's per-body maxima as it emits manifolds
    // persistent colouring: colours of the previous update are kept (contact_solve.h)
    const uint32_t cache_mask = w->ctab_valid ? w->ctab_mask : 0u;
    const uint32_t stamp = (uint32_t)w->color_epoch + 1u;  // never 0xFFFFFFFF (the stamp of an empty slot) in a world's life
    // the colour-table entry of a pair is asked for ahead of the shape test where at least half of the pairs become manifolds
    static const char* probe_env = getenv("PHYS_DEBUG_NP_EARLY_PROBE");  // 0 / 1 forces it (measurements; same bits)
    const uint32_t early_probe = probe_env ? (uint32_t)(probe_env[0] == '1')
                                           : (uint32_t)(!w->hint.valid || 2ull * w->hint.n_manifolds >= (uint64_t)w->hint.n_pairs);
    if (w->warm) {  // last update's records become "previous": what this update's kept manifolds start from
        std::swap(w->man_geo.p, w->man_geo_prev.p);
        std::swap(w->man_imp.p, w->man_imp_prev.p);
    }
    PHYS_PROF(w, PHYS_STAGE_NARROW);
#define PHYS_NP_LAUNCH(T, kItems)                                                                                              \
    do {                                                                                                               \
        uint64_t blocks = (work + T * kItems - 1) / (T * kItems);                                                      \
        if (blocks > 256 * 16) blocks = 256 * 16;                                                                      \
        hipLaunchKernelGGL((k_narrowphase<T, kItems>), dim3((unsigned)blocks), dim3(T), np_extra_lds, w->stream, n_ground, n_owned, w->pairs.p, \
                           w->max_pairs, w->geo.p, w->cfg.contact_margin, \
                           w->cfg.ground_height, w->max_manifolds, w->man_a.p, w->man_b.p,                             \
                           w->man_color.p, w->man_geo.p, w->man_prio.p, w->color_state.p,                              \
                           w->color_state.p + n, reinterpret_cast<ulonglong2*>(w->ctab.p), cache_mask, early_probe, stamp, \
                           w->unc_list.p, w->warm ? w->man_prev.p : nullptr, w->man_imp.p, w->counters.p);             \
    } while (0)
    static const unsigned np_extra_lds = getenv("PHYS_DEBUG_NP_EXTRA_LDS") ? (unsigned)atoi(getenv("PHYS_DEBUG_NP_EXTRA_LDS")) : 0u;  // occupancy experiments
    static const int np_threads_env = getenv("PHYS_DEBUG_NP_THREADS") ? atoi(getenv("PHYS_DEBUG_NP_THREADS")) : 0;  // measurements
    // 128 threads only while the whole stage is a few workgroups (C2: 10k manifolds); measured at 230k manifolds (C3):
    // 0.175 ms with 128 threads, 0.133 with 256; at 2.9M (C5): 0.86 vs 0.55 (round 2)
    const bool few = w->hint.valid ? w->hint.n_manifolds <= 32768u : n <= 200000u;
    static const int np_items_env = getenv("PHYS_DEBUG_NP_ITEMS") ? atoi(getenv("PHYS_DEBUG_NP_ITEMS")) : 0;  // measurements
    // Workgroup shape (all variants: same manifolds, emission order is arbitrary anyway). A trip ends in one reservation
    // behind two barriers, and the waves a CU holds are what hides a trip's round trips from each other; the in-place
    // clipper's LDS slice (33 dwords per lane) admits 16 waves per CU at 118 registers. Measured, ms per update
    // (C5 / 1M cubes in mid-fall / settled 1M pile / C3):
    //   256 threads, one item per lane (4 workgroups per CU)            0.251 / 0.081 / 0.812 / 0.091
    //   256 threads, two items per lane (151 registers: 12 waves)       0.230 / 0.067 / 0.917 / 0.093
    //   512 threads, one item (2 workgroups per CU, half the atomics)   0.206 / 0.069 / 0.836 / 0.075   <- the default
    //   1024 threads, one item (a barrier over 16 waves)                0.231 / 0.084 / 0.937 / 0.080
    // PHYS_DEBUG_NP_THREADS=128|256|512|1024 and PHYS_DEBUG_NP_ITEMS=1|2 (with 256) pick one by hand.
    const int threads = np_threads_env ? np_threads_env : (few ? 128 : 512);
    if (threads == 128) PHYS_NP_LAUNCH(128, 1);
    else if (threads == 256 && np_items_env == 2) PHYS_NP_LAUNCH(256, 2);
    else if (threads == 256) PHYS_NP_LAUNCH(256, 1);
    else if (threads == 1024) PHYS_NP_LAUNCH(1024, 1);
    else PHYS_NP_LAUNCH(512, 1);
#undef PHYS_NP_LAUNCH
}

static void launch_color_round(phys_world* w, uint32_t round, unsigned blocks) {
    unsigned long long* used = w->color_state.p;
    unsigned long long* T[3] = {w->color_state.p + w->n, w->color_state.p + 2 * w->n, w->color_state.p + 3 * w->n};
    PHYS_PROF(w, PHYS_STAGE_COLOR);
    hipLaunchKernelGGL(k_color_round, dim3(blocks), dim3(kColorThreads), 0, w->stream, round, w->max_manifolds,
                       w->man_a.p, w->man_b.p, w->man_color.p, w->man_prio.p, T[round % 3], T[(round + 1) % 3],
                       T[(round + 2) % 3], used, w->unc_list.p + (round & 1u) * w->max_manifolds,
                       w->unc_list.p + ((round + 1u) & 1u) * w->max_manifolds, w->counters.p);
}

// Colouring + colour-major renumbering, entirely device-driven: `rounds` round launches (the previous
// steps' round count + 2: surplus launches exit at once), one finish launch that completes whatever is
// left, then histogram / offsets / place. No host check. A snapshot of the counters is copied to pinned
// memory asynchronously; later steps use it only as a HINT for launch sizes.
void launch_coloring(phys_world* w) {
    const uint64_t n = w->n;
    if (n == 0) return;
    hipStream_t s = w->stream;
    uint64_t blocks64 = (w->max_manifolds + kColorThreads - 1) / kColorThreads;
    if (blocks64 > 512) blocks64 = 512;
    if (w->hint.valid) {
        const uint64_t want = ((uint64_t)w->hint.n_manifolds * 5 / 4 + kColorThreads) / kColorThreads;
        if (want < blocks64) blocks64 = want ? want : 1;
    }
    const unsigned blocks = (unsigned)blocks64;
    uint32_t rounds = 0;
    // a full colouring (the first update after phys_set_bodies: nothing to keep) needs far more rounds than an incremental one
    const bool full = !w->ctab_valid;
    // every PHYS_COLOR_CACHE_PERIOD-th update the colour TABLE is rebuilt: emptied here - the narrow phase of this update
    // has taken what it keeps from it already - and refilled by k_rows_build with every manifold of this update instead of
    // the new ones only. That purges the dead entries (chains never shrink otherwise) and changes no colour.
    const bool rebuild = full || (w->color_epoch % PHYS_COLOR_CACHE_PERIOD) == 0;
    w->snap_tag_full = full;
    const bool known = w->hint.valid && (!full || w->hint.full_rounds > 0);
    const bool small = w->hint.valid && w->hint.n_manifolds <= (uint32_t)(kSmallTrips * kColorThreads);
    bool snapshot_done = false;
    // cluster solver this update? (decided here because it decides the ORDER of the rows: by (cluster, colour)
    // instead of by colour). PHYS_DEBUG_CLUSTER_MIN=<manifolds> moves the threshold (measurements; same bits either way).
    static const char* cluster_min_env = getenv("PHYS_DEBUG_CLUSTER_MIN");
    const bool cluster_forced = (w->cfg.flags & PHYS_FLAG_SOLVER_CLUSTER) != 0u;
    const uint64_t cluster_min = cluster_forced ? 0 : (cluster_min_env ? strtoull(cluster_min_env, nullptr, 10) : kClusterMinManifolds);
    // worth it where contacts are dense (C5: 11 rows per body): velocities stay in LDS for many rows each. Sparse piles
    // (the 1M-cube scene: 0.4-0.5 rows per body, contacts in the bottom layers only) leave most clusters idle and a few
    // overloaded - they keep the dataflow / per-colour kernels, which spread rows evenly over the chip
    // (dynamic clusters hold only the bodies that have manifolds: nothing idles, the row count alone decides)
    const bool dense = cluster_forced || cluster_min_env || w->cluster_dynamic || 2ull * w->hint.n_manifolds >= 3ull * w->n_owned;
    w->cluster_step = (w->cluster_count > 0 || w->cluster_dynamic) && w->hint.valid && !small && dense &&
                      w->hint.n_manifolds >= cluster_min && !(w->cfg.flags & PHYS_FLAG_SOLVER_PER_COLOR) &&
                      w->cfg.solver_iterations > 0 && w->cfg.solver_iterations < 1000 && w->hint.n_colors > 0;
    // ... unless the dataflow kernel is the faster one for this many rows and colours (kernels.hpp; only where it may take
    // the whole chip: PHYS_FLAG_EXCLUSIVE_GPU, one world on the device)
    static const bool no_flow_pref = getenv("PHYS_DEBUG_NO_FLOW_PREFERENCE") != nullptr;  // measurements; same bits
    w->flow_wide = (w->cfg.flags & PHYS_FLAG_EXCLUSIVE_GPU) && !(w->cfg.flags & PHYS_FLAG_SHARED_GPU) && worlds_on_device(w->device) == 1;
    if (w->cluster_step && !cluster_forced && !cluster_min_env && !no_flow_pref && w->flow_wide && w->hint.n_manifolds <= kFlowWideMaxManifolds &&
        flow_quad_beats_cluster(w->hint.n_manifolds, w->hint.n_contacts, w->hint.n_colors))
        w->cluster_step = false;
    if (w->cluster_step && w->cluster_dynamic) w->cluster_step = cluster_plan_dynamic(w);  // clusters and slots (every few updates)
    if (small) {
        // one workgroup does the whole stage, snapshot of the counters included
        StepCounters* slot = snapshot_acquire(w);
        StepCounters* d_slot = nullptr;
        if (slot && hipHostGetDevicePointer((void**)&d_slot, slot, 0) != hipSuccess) {
            d_slot = nullptr;
            (void)hipGetLastError();  // an answer handled here (the copy path takes over), not an error to leave behind
        }
        { PHYS_PROF(w, PHYS_STAGE_COLOR);
          hipLaunchKernelGGL(k_color_small, dim3(1), dim3(kColorThreads), 0, s, w->max_manifolds, w->man_a.p, w->man_b.p,
                             w->man_color.p, w->man_prio.p, w->color_state.p, (uint64_t)n, w->row_src.p, w->counters.p, d_slot); }
        if (d_slot) { snapshot_commit(w); snapshot_done = true; }
    } else {
    if (known) {
        const uint32_t base = full ? w->hint.full_rounds : w->hint.color_rounds;
        // as many launches as the last update of this kind needed; k_color_finish runs what is still missing over the
        // same lists (measured: handing it the second half of the rounds - one workgroup, ~10 us per round with a few
        // thousand manifolds left - is slower than the launches it saves, and far slower on a full re-colouring)
        rounds = base;
        for (uint32_t r = 0; r < rounds; ++r) launch_color_round(w, r, blocks);
    } else {
        // first step after phys_set_bodies: nothing is known about the scene yet, so this one step asks the
        // device (a single-workgroup finish / tail over millions of manifolds would take seconds)
        for (int guard = 0; guard < 4096; ++guard) {
            for (uint32_t k = 0; k < 8; ++k) launch_color_round(w, rounds++, blocks);
            (void)hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            if (w->prof.on) w->prof.collect(s);
            if (w->h_counters->n_uncolored == 0 || w->h_counters->overflow) break;
        }
    }
    { PHYS_PROF(w, PHYS_STAGE_COLOR); hipLaunchKernelGGL(k_color_finish, dim3(1), dim3(kColorThreads), 0, s, rounds, w->max_manifolds, w->man_a.p, w->man_b.p, w->man_color.p, w->man_prio.p, w->color_state.p, (uint64_t)n, w->unc_list.p, w->counters.p); }
    // workgroups of the colour sort: sized from the hint (any value is correct: the kernels stride)
    uint32_t nb = kSortBlocksMax;
    if (w->hint.valid) {
        const uint64_t want = ((uint64_t)w->hint.n_manifolds * 5 / 4) / kSortChunk + 1;
        nb = 1;
        while (nb < want && nb < (uint32_t)kSortBlocksMax) nb <<= 1;
    }
    // the snapshot of the counters (launch-size hints of later updates) is written by the last kernel of the sort itself,
    // into a host-mapped slot: the copy engine's turn between two kernels of the stream cost 4.4 us per update
    StepCounters* d_snap = nullptr;
    if (known) {
        StepCounters* slot = snapshot_acquire(w);
        if (slot && hipHostGetDevicePointer((void**)&d_snap, slot, 0) != hipSuccess) {
            d_snap = nullptr;
            (void)hipGetLastError();  // an answer handled here (the copy path takes over), not an error to leave behind
        }
    }
    if (w->cluster_step) {
        launch_cluster_sort(w, blocks * (kColorThreads / 256), d_snap);  // rows by (owner cluster, colour); counts the colours too
    } else {
    { PHYS_PROF(w, PHYS_STAGE_ROWS); hipLaunchKernelGGL(k_color_hist, dim3(nb), dim3(1024), 0, s, w->max_manifolds, w->man_color.p, w->color_block_hist.p, nb, w->counters.p); }
    { PHYS_PROF(w, PHYS_STAGE_ROWS); hipLaunchKernelGGL(k_color_offsets, dim3(1), dim3(1024), 0, s, w->color_block_hist.p, nb, w->counters.p); }
    // (n_colors and the per-colour counts are final here: k_color_place below may copy the counters out)
    { PHYS_PROF(w, PHYS_STAGE_ROWS); hipLaunchKernelGGL(k_color_place, dim3(nb), dim3(1024), 0, s, w->max_manifolds, w->man_color.p, w->color_block_hist.p, nb, w->row_src.p, w->counters.p, d_snap); }
    }
    if (d_snap) { snapshot_commit(w); snapshot_done = true; }
    }
    {
        // the new manifolds of this update go into the colour table in k_rows_build (launch_solver): one launch less
        if (rebuild) {  // start from an empty table: every manifold of this update is inserted
            PHYS_PROF(w, PHYS_STAGE_ROWS);
            (void)hipMemsetAsync(w->ctab.p, 0xFF, ((size_t)w->ctab_mask + 1) * 16, s);
        }
        w->ctab_job_pending = true;
        w->ctab_job_all = rebuild;
        w->ctab_job_stamp = (uint32_t)w->color_epoch + 1u;
        w->ctab_valid = true;
        w->color_epoch++;
    }
    if (!known && !small) {
        // ... and adopts the exact counters as the first hint (the solver launches right after use them)
        (void)hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        const StepCounters& c = *w->h_counters;
        if (!c.overflow) {
            w->hint.valid = true;
            w->hint.n_manifolds = c.n_manifolds;
            w->hint.n_pairs = c.n_pairs;
            w->hint.n_contacts = c.n_contacts;
            if (c.max_region) w->hint.max_region = c.max_region;
            w->hint.n_used_buckets = c.n_used_buckets;
            w->hint.n_colors = c.n_colors;
            if (c.n_active) w->hint.n_active = c.n_active;
            if (full) w->hint.full_rounds = c.color_rounds; else w->hint.color_rounds = c.color_rounds;
            for (int q = 0; q < kMaxColors; ++q) w->hint.color_count[q] = c.color_count[q];
        }
    } else if (!snapshot_done) {
        snapshot_counters_async(w);
    }
}

}  // namespace phys
