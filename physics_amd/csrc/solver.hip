// solver.hip — sequential-impulse contact solver for gfx950. No reference
// counterpart; the arithmetic is include/spec/contact_solve.h (solver_prep / solve_manifold).
//
// Order of work = the spec's: iterations outermost, colours ascending, manifolds of one colour in
// parallel (they share no body, so there are no float atomics and the result is independent of the
// schedule), points of a manifold in index order inside one lane.
//
// Row storage: after colouring, manifolds are renumbered colour-major (row d = colour start + slot in
// colour). Everything a solve needs per row is kept as 16-BYTE elements in plane-major arrays over d (stride =
// cap), so a lane fetches a row with ~10 dwordx4 loads (a wave: 1 KiB contiguous per load) instead of ~45
// dword loads - the row solve is a long dependent chain at one wave per SIMD, and the number of memory
// instructions in front of it is what its latency is made of:
//     row_hdr uint4  [d]              {body a, body b, point count, update tickets (k_solve_flow)}
//     row_n   float4 [d]              {normal xyz, 0}
//     row_pt  float4 [(2k)*cap + d]   {rA xyz, normal mass}          point k = 0..3
//                    [(2k+1)*cap + d] {rB xyz, tangent mass 0}
//     row_tb  float4 [d]              {tangent mass 1, bias} of points 0 and 1;  [cap + d]: points 2 and 3
//     row_acc float4 [k*cap + d]      {pn, pt0, pt1, tag}            accumulated impulses of point k
// Only the body velocities (32-byte records) and the inverse inertia are gathered by body id.
// Algorithmic bytes per manifold per iteration: ids, count, normal 24 + per point 40 (rA, rB,
// masses, bias) + 12 R + 12 W (accumulated impulses), + per body 48 (v, w R+W) + 4 (inv mass) + 12 / 36
// (inverse inertia diagonal / full).
#include <cstdlib>

#include "kernels.hpp"

namespace phys {

// inverse inertia of one body. DIAG: every body's tensor is diagonal (the reference's only case: identity,
// rigid_body.rs:71), stored as one float4 per body = 16 B and one sector per gather instead of 36 B / two.
// The zero off-diagonals are put back, so the arithmetic is the general path's (only signed zeros can differ).
template <bool DIAG>
__device__ __forceinline__ m33 ld_inertia(const float* __restrict__ p, uint32_t i);
__device__ __forceinline__ m33 ld_m33(const float* __restrict__ p, uint32_t i) {
    m33 M;
#pragma unroll
    for (int k = 0; k < 9; ++k) M.m[k] = p[9 * (size_t)i + k];
    return M;
}
template <>
__device__ __forceinline__ m33 ld_inertia<false>(const float* __restrict__ p, uint32_t i) { return ld_m33(p, i); }
template <>
__device__ __forceinline__ m33 ld_inertia<true>(const float* __restrict__ p, uint32_t i) {
    const float4 d = reinterpret_cast<const float4*>(p)[i];
    m33 M;
#pragma unroll
    for (int k = 0; k < 9; ++k) M.m[k] = 0.0f;
    M.m[0] = d.x; M.m[4] = d.y; M.m[8] = d.z;
    return M;
}

// warm starting (contact_solve.h): where the impulses a solve ends with go, and where a row's starting impulses come from
struct WarmJob {
    const uint32_t* man_prev;  // null: warm starting is off (rows start from zero, no sweep 0); otherwise only a flag
    const float* geo_prev;     // 128-byte manifold records of the previous update
    const float* imp_prev;     // 12 floats per manifold: what its solve ended with
    float* imp;                // this update's (written by the last sweep of whichever solver kernel runs)
    const uint32_t* row_src;   // row -> manifold
};
__device__ __forceinline__ void store_final_impulses(const WarmJob& wj, uint32_t d, const float pn[4], const float pt0[4], const float pt1[4]) {
    float4* o = reinterpret_cast<float4*>(wj.imp) + 3 * (size_t)wj.row_src[d];
    o[0] = make_float4(pn[0], pt0[0], pt1[0], pn[1]);
    o[1] = make_float4(pt0[1], pt1[1], pn[2], pt0[2]);
    o[2] = make_float4(pt1[2], pn[3], pt0[3], pt1[3]);
}

struct RowArrays {
    float4* all;  // the 16 planes in one piece: plane p of row d = all[p * cap + d] (0 hdr, 1 n, 2-3 tb, 4-11 pt, 12-15 acc)
    uint4* hdr;
    float4* n;
    float4* pt;
    float4* tb;
    float4* acc;
    uint64_t cap;
};

template <bool DIAG>
__global__ __launch_bounds__(256) void k_rows_build(StepCounters* __restrict__ ctr, RowArrays rows, solve_params_t sp,
                                                    const uint32_t* __restrict__ row_src,
                                                    const float* __restrict__ man_geo /* 128-byte records */,
                                                    const float* __restrict__ pos,
                                                    const float* __restrict__ vel,
                                                    const float* __restrict__ inv_inertia, uint32_t inertia_stride,
                                                    const uint32_t* __restrict__ man_color,
                                                    const unsigned long long* __restrict__ used,
                                                    int flow /* 1: k_solve_flow runs this step (tickets, no zeroed impulses) */,
                                                    ColorTableJob table, const uint32_t* __restrict__ cluster_slot,
                                                    const uint32_t* __restrict__ body_shared,
                                                    uint32_t cluster_slots /* 0: not a cluster-solver step */, uint32_t cluster_count,
                                                    WarmJob warm) {
    if (ctr->overflow) return;  // never solve a truncated set; phys_sync / phys_get_stats report it
    const uint32_t M = ctr->n_manifolds;
    const uint64_t cap = rows.cap;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < M; d += gridDim.x * blockDim.x) {
        const uint32_t m = row_src[d];
        manifold_t g;
        const float4* rec = reinterpret_cast<const float4*>(man_geo) + 8 * (size_t)m;  // ONE line through the row permutation
        const float4 r0 = rec[0], r1 = rec[1];
        const uint32_t a = __float_as_uint(r0.x), b = __float_as_uint(r0.y);
        g.count = (int)__float_as_uint(r0.z);
        g.normal = v3_make(r1.x, r1.y, r1.z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 p = rec[2 + k];
            g.pt[k] = v3_make(p.x, p.y, p.z);
            g.depth[k] = p.w;
        }
        const int has_b = b != PHYS_GROUND_ID;
        // warm starting: the impulses this row starts from = what the same pair's manifold of the previous update ended
        // with, matched point by point (contact_solve.h warm_match); zero for a new pair or with warm starting off
        float w_pn[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w_pt0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w_pt1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (warm.man_prev) {
            const uint32_t pm = __float_as_uint(r1.w);  // index of the pair's manifold in the previous update (narrow phase)
            if (pm != 0xFFFFFFFFu) {
                const float4* prec = reinterpret_cast<const float4*>(warm.geo_prev) + 8 * (size_t)pm;
                const float4* pimp = reinterpret_cast<const float4*>(warm.imp_prev) + 3 * (size_t)pm;
                const float4 q0 = prec[0], q1 = prec[1], i0 = pimp[0], i1 = pimp[1], i2 = pimp[2];
                warm_t wt;
                wt.count = (int)__float_as_uint(q0.z);
                wt.normal = v3_make(q1.x, q1.y, q1.z);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float4 p = prec[2 + k]; wt.pt[k] = v3_make(p.x, p.y, p.z); }
                wt.pn[0] = i0.x; wt.pt0[0] = i0.y; wt.pt1[0] = i0.z; wt.pn[1] = i0.w;
                wt.pt0[1] = i1.x; wt.pt1[1] = i1.y; wt.pn[2] = i1.z; wt.pt0[2] = i1.w;
                wt.pt1[2] = i2.x; wt.pn[3] = i2.y; wt.pt0[3] = i2.z; wt.pt1[3] = i2.w;
                warm_match(&g, &wt, w_pn, w_pt0, w_pt1);
            }
        }
        // persistent colouring: a manifold that is new in this update enters the colour table (kernels.hpp)
        if (table.tab && (table.all || __float_as_uint(r0.w) == 0u)) color_table_insert(table, a, b, m, ctr);
        // a kept colour came with the record; a new manifold's was made by the colouring rounds since
        const uint32_t color_m = __float_as_uint(r0.w) != 0u ? __float_as_uint(r0.w) - 1u : man_color[m];
        uint32_t ticket = 0;
        if (flow) {
            // the colours in use at a body are exactly the colours of its manifolds (all distinct), so the rank of
            // this row among the body's manifolds in solve order = number of its colours below this one
            const unsigned long long below = (1ull << color_m) - 1ull;
            const unsigned long long ua = used[a];
            ticket = (uint32_t)__popcll(ua & below) | ((uint32_t)__popcll(ua) << 8);
            if (has_b) {
                const unsigned long long ub = used[b];
                ticket |= ((uint32_t)__popcll(ub & below) << 16) | ((uint32_t)__popcll(ub) << 24);
            }
            if (flow == 2 && d == 0) ticket += 1;  // fault injection (PHYS_DEBUG_FLOW_STALL): row 0 waits for a turn that never comes
        }
        rows.hdr[d] = make_uint4(a, b, (uint32_t)g.count, ticket);
        uint32_t info = 0;
        if (cluster_slots) {
            // cluster solver (cluster.hip): where each side's velocity lives. Per side: slot (13 bits) | publish (1) |
            // mode (2): 0 own cluster, never updated by another workgroup (LDS only); 1 own cluster, shared (LDS while its
            // tag is current, granules otherwise); 2 another cluster's body (granules only); 3 no body (ground).
            // publish: the body's NEXT update in solve order (next colour in use at the body, cyclically) belongs to a
            // remote row, so this update must be written to the granules for it
            const unsigned long long bitc = 1ull << color_m;
            const uint32_t ha = cluster_home(cluster_slot, a, cluster_slots);
            const uint32_t hb = has_b ? cluster_home(cluster_slot, b, cluster_slots) : kNoHome;
            const uint32_t owner = cluster_row_owner(a, ha, hb, cluster_count);
            // one side: {slot | publish | mode} of body x whose home is `hx`
            auto side = [&](uint32_t x, uint32_t hx) -> uint32_t {
                if (hx != owner) return 2u << 14;  // homeless, or at home elsewhere
                const unsigned long long rem = ((unsigned long long)body_shared[2 * (size_t)x + 1] << 32) | body_shared[2 * (size_t)x];
                uint32_t pub = 0;
                if (rem) {
                    const unsigned long long ux = used[x];
                    const unsigned long long above = ux & ~(bitc | (bitc - 1ull));
                    const unsigned long long next = above ? (above & (~above + 1ull)) : (ux & (~ux + 1ull));  // lowest colour above, else the first
                    pub = (rem & next) ? 1u : 0u;
                }
                return (cluster_slot[x] - hx * cluster_slots) | (pub << 13) | ((rem ? 1u : 0u) << 14);
            };
            info = side(a, ha);
            const uint32_t ib = has_b ? side(b, hb) : (3u << 14);
            info |= ib << 16;
        }
        rows.n[d] = make_float4(g.normal.x, g.normal.y, g.normal.z, __uint_as_float(info));
        if (cluster_slots) {
            // compact rows of the cluster solver (cluster.hip): the contact points themselves and the bias; lever arms and
            // row masses are remade there from the body positions it keeps in LDS. Nothing of the bodies is read here.
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < g.count)
                    rows.all[(size_t)(2 + k) * cap + d] = make_float4(g.pt[k].x, g.pt[k].y, g.pt[k].z, contact_bias(g.depth[k], &sp));
            if (warm.man_prev) {  // the starting impulses, packed like the solver packs them (planes 6-8; read in sweep 0)
                rows.all[(size_t)6 * cap + d] = make_float4(w_pn[0], w_pt0[0], w_pt1[0], w_pn[1]);
                rows.all[(size_t)7 * cap + d] = make_float4(w_pt0[1], w_pt1[1], w_pn[2], w_pt0[2]);
                rows.all[(size_t)8 * cap + d] = make_float4(w_pt1[2], w_pn[3], w_pt0[3], w_pt1[3]);
            }
            if (((info >> 30) & 3u) == 2u) {
                // body B belongs to another cluster: what does not change during the solve rides with the row (planes 12,
                // 13), or fetching it by body id would be a second dependent round trip in every colour step of the solver
                const v3 xb = ld3(pos, b);
                rows.all[(size_t)12 * cap + d] = make_float4(xb.x, xb.y, xb.z, vel[8 * (size_t)b + 3]);
                if (DIAG) rows.all[(size_t)13 * cap + d] = reinterpret_cast<const float4*>(inv_inertia)[b * inertia_stride];
            }
            if (((info >> 14) & 3u) == 2u) {
                // body A has no home (dynamic clusters beyond their capacity): the same for A in planes 14, 15
                const v3 xa = ld3(pos, a);
                rows.all[(size_t)14 * cap + d] = make_float4(xa.x, xa.y, xa.z, vel[8 * (size_t)a + 3]);
                if (DIAG) rows.all[(size_t)15 * cap + d] = reinterpret_cast<const float4*>(inv_inertia)[a * inertia_stride];
            }
            continue;
        }
        // DIAG: 16 bytes per body instead of 36 (and none at all when every body shares one tensor: stride 0);
        // the zero off-diagonals are put back, so solver_prep's arithmetic is the general path's
        const m33 IA = ld_inertia<DIAG>(inv_inertia, a * inertia_stride);
        m33 IB;
#pragma unroll
        for (int k = 0; k < 9; ++k) IB.m[k] = 0.0f;
        float imb = 0.0f;
        v3 xB = v3_make(0.0f, 0.0f, 0.0f);
        if (has_b) { IB = ld_inertia<DIAG>(inv_inertia, b * inertia_stride); imb = vel[8 * (size_t)b + 3]; xB = ld3(pos, b); }
        solver_manifold_t sm;
        solver_prep(&g, has_b, ld3(pos, a), xB, vel[8 * (size_t)a + 3], &IA, imb, &IB, &sp, &sm);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < sm.count) {
                const contact_row_t& c = sm.row[k];
                rows.pt[(size_t)(2 * k) * cap + d] = make_float4(c.rA.x, c.rA.y, c.rA.z, c.normal_mass);
                rows.pt[(size_t)(2 * k + 1) * cap + d] = make_float4(c.rB.x, c.rB.y, c.rB.z, c.tangent_mass[0]);
                // accumulated impulses of point k: zero, or where the previous update left off. (The dataflow kernels read
                // these with plain loads in their first sweep only when warm starting is on; tag 0 = no epoch ever.)
                if (!flow || warm.man_prev) rows.acc[(size_t)k * cap + d] = make_float4(w_pn[k], w_pt0[k], w_pt1[k], 0.0f);
            }
        }
        rows.tb[d] = make_float4(sm.row[0].tangent_mass[1], sm.row[0].bias, sm.row[1].tangent_mass[1], sm.row[1].bias);
        if (sm.count > 2)
            rows.tb[cap + d] = make_float4(sm.row[2].tangent_mass[1], sm.row[2].bias, sm.row[3].tangent_mass[1], sm.row[3].bias);
    }
}

// one manifold row d of the colour-major numbering, in registers
struct RowRegs {
    uint32_t a, b, ticket;
    solver_manifold_t sm;
};

// everything of a row that does NOT depend on body velocities. ACC: also the accumulated impulses (plain loads;
// k_solve_flow receives them as tagged granules instead)
// EAGER: every plane of the row is fetched at once, whatever the point count turns out to be (planes beyond it hold
// stale but readable data that is then ignored): no load waits for the header. For the per-colour kernel of large
// scenes, where a launch is one wave per SIMD and a second dependent round trip to memory is ~1.5 us of every launch.
template <bool ACC, bool EAGER = false>
__device__ __forceinline__ void load_row(RowRegs& R, uint32_t d, const RowArrays& rows) {
    const uint64_t cap = rows.cap;
    float4 e_p0[4], e_p1[4], e_acc[4], e_t23;
    if (EAGER) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e_p0[k] = rows.pt[(size_t)(2 * k) * cap + d];
            e_p1[k] = rows.pt[(size_t)(2 * k + 1) * cap + d];
            if (ACC) e_acc[k] = rows.acc[(size_t)k * cap + d];
        }
        e_t23 = rows.tb[cap + d];
    }
    const uint4 h = rows.hdr[d];
    R.a = h.x; R.b = h.y; R.ticket = h.w;
    solver_manifold_t& sm = R.sm;
    sm.count = (int)h.z;
    sm.has_b = R.b != PHYS_GROUND_ID;
    const float4 nn = rows.n[d];
    sm.n = v3_make(nn.x, nn.y, nn.z);
    tangent_basis(sm.n, &sm.t1, &sm.t2);  // same inputs as solver_prep => same bits as the basis used there
    const float4 t01 = rows.tb[d];
    float4 t23 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (sm.count > 2) t23 = EAGER ? e_t23 : rows.tb[cap + d];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        contact_row_t& c = sm.row[k];
        if (k < sm.count) {
            const float4 p0 = EAGER ? e_p0[k] : rows.pt[(size_t)(2 * k) * cap + d];
            const float4 p1 = EAGER ? e_p1[k] : rows.pt[(size_t)(2 * k + 1) * cap + d];
            c.rA = v3_make(p0.x, p0.y, p0.z); c.normal_mass = p0.w;
            c.rB = v3_make(p1.x, p1.y, p1.z); c.tangent_mass[0] = p1.w;
            const float4 t = k < 2 ? t01 : t23;
            c.tangent_mass[1] = (k & 1) ? t.z : t.x;
            c.bias = (k & 1) ? t.w : t.y;
            if (ACC) {
                const float4 acc = EAGER ? e_acc[k] : rows.acc[(size_t)k * cap + d];
                c.pn = acc.x; c.pt[0] = acc.y; c.pt[1] = acc.z;
            } else {
                c.pn = 0.0f; c.pt[0] = 0.0f; c.pt[1] = 0.0f;
            }
        } else {
            c.rA = v3_make(0.0f, 0.0f, 0.0f); c.rB = v3_make(0.0f, 0.0f, 0.0f);
            c.normal_mass = 0.0f; c.tangent_mass[0] = 0.0f; c.tangent_mass[1] = 0.0f; c.bias = 0.0f;
            c.pn = 0.0f; c.pt[0] = 0.0f; c.pt[1] = 0.0f;
        }
    }
}

// gather the two bodies, solve_manifold, write velocities and accumulated impulses back
template <bool DIAG, bool EAGER = false>
__device__ __forceinline__ void solve_row(uint32_t d, const RowArrays& rows, float friction,
                                          const float* __restrict__ inv_inertia,
                                          uint32_t inertia_stride /* 0: one tensor shared by every body */,
                                          float* __restrict__ vel, int apply_only /* sweep 0 of a warm-started solve */,
                                          int last /* the solve's last sweep: the impulses are remembered */, const WarmJob& wj,
                                          uint32_t ablate = 0, uint32_t n_bodies = 1) {
    RowRegs R;
    load_row<true, EAGER>(R, d, rows);
    solver_manifold_t& sm = R.sm;
    // PHYS_DEBUG_ABLATE (timing diagnosis, WRONG results): bit 0 - body records gathered at consecutive indices instead
    // of the row's bodies (what the gather's scatter costs); bit 1 - no row arithmetic; bit 2 - no velocity write-back
    const uint32_t a = (ablate & 1u) ? d % n_bodies : R.a, b = (ablate & 1u) ? (d + 7u) % n_bodies : R.b;
    const m33 IA = ld_inertia<DIAG>(inv_inertia, a * inertia_stride);
    BodyVel A = ld_vel(vel, a);
    const float ima = A.inv_mass;
    v3 vA = A.v, wA = A.w;
    m33 IB;
#pragma unroll
    for (int k = 0; k < 9; ++k) IB.m[k] = 0.0f;
    float imb = 0.0f;
    v3 vB = v3_make(0.0f, 0.0f, 0.0f), wB = v3_make(0.0f, 0.0f, 0.0f);
    BodyVel B = A;
    if (sm.has_b) { IB = ld_inertia<DIAG>(inv_inertia, b * inertia_stride); B = ld_vel(vel, b); imb = B.inv_mass; vB = B.v; wB = B.w; }
    // rows made on the way: these kernels are throughput-bound and want the registers (k_solve_flow makes them all
    // beforehand, while it waits; same arithmetic)
    if (!(ablate & 2u)) solve_manifold_lazy(&sm, friction, ima, &IA, imb, &IB, &vA, &wA, &vB, &wB, apply_only);
    A.v = vA; A.w = wA;
    if (!(ablate & 4u)) {
        st_vel(vel, a, A);
        if (sm.has_b) { B.v = vB; B.w = wB; st_vel(vel, b, B); }
    }
    if (!apply_only) {  // (sweep 0 changes no impulse)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < sm.count)
                rows.acc[(size_t)k * rows.cap + d] = make_float4(sm.row[k].pn, sm.row[k].pt[0], sm.row[k].pt[1], 0.0f);
    }
    if (last && wj.man_prev) {
        const float pn[4] = {sm.row[0].pn, sm.row[1].pn, sm.row[2].pn, sm.row[3].pn};
        const float p0[4] = {sm.row[0].pt[0], sm.row[1].pt[0], sm.row[2].pt[0], sm.row[3].pt[0]};
        const float p1[4] = {sm.row[0].pt[1], sm.row[1].pt[1], sm.row[2].pt[1], sm.row[3].pt[1]};
        store_final_impulses(wj, d, pn, p0, p1);
    }
}

// Tiles of `tile_rows` consecutive items handed to the workgroups of a launch so that the workgroups sharing an XCD
// (equal blockIdx % 8 under the dispatcher's round-robin placement: a label, never relied on for correctness) cover
// one contiguous eighth of the range. Every tile is visited exactly once for any grid size.
struct XcdTiles {
    uint32_t tile, last, step;
    __device__ __forceinline__ XcdTiles(uint32_t items, uint32_t tile_rows) {
        const uint32_t tiles = (items + tile_rows - 1) / tile_rows;
        if (gridDim.x < 8u) {  // fewer workgroups than labels: plain striding (some labels would have nobody)
            tile = blockIdx.x; last = tiles; step = gridDim.x;
            return;
        }
        const uint32_t per = (tiles + 7u) / 8u;                       // tiles per label
        const uint32_t label = blockIdx.x & 7u, j = blockIdx.x >> 3;  // j-th workgroup of its label
        step = (gridDim.x - label + 7u) / 8u;                         // workgroups carrying this label
        tile = label * per + j;
        const uint32_t stop = (label + 1u) * per;
        last = stop < tiles ? stop : tiles;
    }
    __device__ __forceinline__ bool valid() const { return tile < last; }
    __device__ __forceinline__ void next() { tile += step; }
};

// one colour of one iteration; the row range comes from the device-side colour table
template <bool DIAG>
__global__ __launch_bounds__(256) void k_solve_color(const StepCounters* __restrict__ ctr, uint32_t col, RowArrays rows,
                                                     float friction, const float* __restrict__ inv_inertia,
                                                     uint32_t inertia_stride, float* __restrict__ vel, uint32_t ablate,
                                                     uint32_t n_bodies, int apply_only, int last, WarmJob wj) {
    if (ctr->overflow) return;
    const uint32_t start = ctr->color_start[col], end = ctr->color_start[col + 1];
    // XCD-aware tiles: the rows of a colour are in emission (= spatial) order, so a contiguous range of them touches a
    // compact set of bodies. Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the L2 they share),
    // so the workgroups of one label take ONE contiguous eighth of the colour: the 128-byte lines of the gathered body
    // records are then fetched into one L2 instead of into all eight (PMC, C5: 22 MB of line fetches per launch for
    // 5.6 MB of gathered records when tiles were dealt in plain blockIdx order). Speed only: any mapping is correct.
    for (XcdTiles t(end - start, blockDim.x); t.valid(); t.next()) {
        const uint32_t d = start + t.tile * blockDim.x + threadIdx.x;
        if (d < end) solve_row<DIAG, true>(d, rows, friction, inv_inertia, inertia_stride, vel, apply_only, last, wj, ablate, n_bodies);
    }
}

// The colour classes [first, n_colours) of one iteration in ONE launch of ONE workgroup: colours in
// ascending order with a workgroup barrier between them (all waves share this CU's L1, so a barrier orders
// the body-velocity writes of one colour before the reads of the next). Same order of work as one launch per
// colour, without paying a ~8 us launch for a few hundred manifolds. `first` is a host HINT (the small
// colours of the previous step); any value gives the same result, only the speed changes.
constexpr int kTailThreads = 512;  // 2 waves per SIMD: the row solve needs >128 VGPRs, 1024 threads would spill
template <bool DIAG>
__global__ __launch_bounds__(kTailThreads) void k_solve_tail(const StepCounters* __restrict__ ctr, uint32_t first, RowArrays rows,
                                                            float friction, const float* __restrict__ inv_inertia,
                                                            uint32_t inertia_stride, float* __restrict__ vel, int apply_only,
                                                            int last_sweep, WarmJob wj) {
    if (ctr->overflow) return;
    const uint32_t last = ctr->n_colors;
    if (first >= last) return;
    // the colour table goes to LDS once: a scalar load from global memory per colour phase would put a full
    // memory round trip in front of every phase
    __shared__ uint32_t s_start[PHYS_MAX_COLORS + 1];
    if (threadIdx.x <= (uint32_t)PHYS_MAX_COLORS) s_start[threadIdx.x] = ctr->color_start[threadIdx.x];
    __syncthreads();
    for (uint32_t col = first; col < last; ++col) {
        const uint32_t start = s_start[col], end = s_start[col + 1];
        for (uint32_t d = start + threadIdx.x; d < end; d += kTailThreads)
            solve_row<DIAG>(d, rows, friction, inv_inertia, inertia_stride, vel, apply_only, last_sweep, wj);
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Single-launch dataflow solver. Same arithmetic and the same ORDER OF UPDATES PER BODY as the per-colour
// launches (iterations outermost, colours ascending), so the results are bit-identical; what changes is how
// the order is enforced. Instead of a kernel boundary after every colour of every iteration (8 x n_colours
// dependent launches of 5-20 us), every body carries a ticket: the k-th update of body A in solve order may
// only be made by the row holding ticket k for A (k = iteration * deg(A) + rank of the row's colour among A's
// colours; k_rows_build). In-flight velocities live in `flow_vel` as two 16-byte granules {x, y, z, tag} whose
// tag = (epoch << 16) | number of updates applied, i.e. THE DATA IS ITS OWN READY FLAG: a row polls its bodies'
// granules (sc1 loads: served past the CU's L1) until both tags equal its ticket, solves, and stores them back
// (sc1 = write-through stores) with tag + 1. One hop costs ~0.7-1.0 us (tools/hop_bench.hip) instead of a
// launch, and rows of different colours / iterations overlap wherever the contact graph allows.
//   * the FIRST update of a body reads the plain `vel` record (written by the previous kernel), the LAST one
//     writes it (read by the next kernel): flow_vel never needs initialising; stale tags of earlier steps carry
//     another epoch;
//   * accumulated impulses cross iterations the same way (row_acc {pn, pt0, pt1, tag = iteration});
//   * work items (iteration, chunk of blockDim rows) are handed out by an atomic ticket in solve order, so every
//     dependency of an item was taken EARLIER by a workgroup that is running: no co-residency assumption, no
//     deadlock by construction; and every spin is bounded (timeout -> overflow bit 4 -> PHYS_ERR_HIP at sync).
// Granule discipline follows the guide's data-tagged hand-off: each granule is written by ONE 16-byte sc1
// store and only ever read by 16-byte sc1 loads. Waiting waves cost issue slots and L2 bandwidth, so the launch
// is sized to about one wave per SIMD (launch_solver) and waiters back off by their distance in hops.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld_granule(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    // aux: 16 = sc1, bit 31 = volatile (the compiler must re-issue the load in every sweep)
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, (int)0x80000010);
}
__device__ __forceinline__ void st_granule(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, v3 v, uint32_t tag) {
    u32x4 g;
    g.x = __float_as_uint(v.x); g.y = __float_as_uint(v.y); g.z = __float_as_uint(v.z); g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, r, byte_off, 0, 16);
}
__device__ __forceinline__ v3 granule_v3(u32x4 g) {
    return v3_make(__uint_as_float(g.x), __uint_as_float(g.y), __uint_as_float(g.z));
}

constexpr long long kFlowTimeoutTicks = 300000000ll;  // 3 s of the 100 MHz wall clock (fault injection: 20 ms)
constexpr uint64_t kFlowMaxManifolds = 400000;        // above: one launch per colour streams better
// below: four lanes per manifold (k_solve_flow_quad). Measured again with its statically dealt items (tools/cluster_crossover.py
// --path flow, solve ms quad / one lane): towers 92k manifolds 0.371 / 0.466, 182k 0.645 / 0.670, 256k 0.914 / 0.903; mixed piles
// 91k 0.308 / 0.405, 155k 0.444 / 0.502 (round 2, with tickets: 45k +13 %, 108k -32 %)
constexpr uint64_t kFlowQuadMaxManifolds = 200000;

// Work items are SOFTWARE-PIPELINED inside a workgroup: while item k waits for its bodies and is solved, the rows of
// item k + 1 (its ticket is taken one item ahead) are already on their way, and so are - issued once those rows have
// arrived, i.e. behind the Jacobians of item k - the records of its bodies. At one wave per SIMD nobody else hides
// these two dependent round trips (~4.5 us of the ~14 us an item cost on the 1M-cube scene: 11.5k items over 256
// workgroups). A workgroup still finishes its items in ticket order, so the earliest unfinished item of the solve is
// always somebody's CURRENT item: the no-deadlock argument above holds. `pipeline` = 0 takes the next item only when
// the current one is done: holding an item ahead means it cannot go to whichever workgroup is free first, and where a
// colour class is much smaller than the launch (C3: 64 items per colour, 256 workgroups) the solve is bound by the
// chain of hand-offs, not by item throughput - there the look-ahead costs 10 % instead of saving it (launch_solver).
struct FlowRowRaw { uint4 h; float4 nn, t01, t23, p[8]; };
struct FlowBodies { m33 IA, IB; float ima, imb, massA, massB; v3 vA, wA, vB, wB; };

template <bool DIAG>
__global__ __launch_bounds__(256) void k_solve_flow(StepCounters* __restrict__ ctr, uint32_t iterations, uint32_t epoch,
                                                    RowArrays rows, float friction,
                                                    const float* __restrict__ inv_inertia, uint32_t inertia_stride,
                                                    float* vel, float* flow_vel, uint32_t n_bodies, long long timeout_ticks,
                                                    uint32_t pipeline, uint32_t warm_sweep /* sweep 0 applies the starting impulses;
                                                    `iterations` counts it */, WarmJob wj) {
    __shared__ uint32_t s_item;
    if (ctr->overflow) return;
    const uint32_t M = ctr->n_manifolds;
    const uint32_t nchunks = (M + blockDim.x - 1) / blockDim.x;
    const uint32_t total = nchunks * iterations;
    const uint32_t cap = (uint32_t)rows.cap;  // collision_alloc: 64 * cap < 4 GiB (32-bit buffer offsets)
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(flow_vel, 0, n_bodies * 32u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(rows.acc, 0, cap * 64u, 0x00020000);
    const uint32_t etag = epoch << 16;
    const long long t_start = wall_clock64();
    auto take = [&]() -> uint32_t {  // next work item in solve order (block-uniform)
        __syncthreads();
        if (threadIdx.x == 0)
            s_item = (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u)
                         ? 0xFFFFFFFFu : atomicAdd(&ctr->flow_ticket, 1u);
        __syncthreads();
        return s_item;
    };
    auto row_of = [&](uint32_t L) -> uint32_t {  // this lane's row of item L (>= M: none)
        if (L >= total) return 0xFFFFFFFFu;
        const uint32_t it = L / nchunks;
        return (L - it * nchunks) * blockDim.x + threadIdx.x;
    };
    auto fetch_row = [&](uint32_t d, FlowRowRaw& r) {
        // every plane at once, whatever the point count turns out to be: ONE round trip
        r.h = rows.hdr[d];
        r.nn = rows.n[d];
        r.t01 = rows.tb[d];
        r.t23 = rows.tb[cap + d];
#pragma unroll
        for (int k = 0; k < 8; ++k) r.p[k] = rows.pt[(size_t)k * cap + d];
    };
    auto fetch_bodies = [&](const FlowRowRaw& r, FlowBodies& g) {
        // everything of the two bodies that is constant during the solve; v, w are the body's state only for ticket 0
        g.IA = ld_inertia<DIAG>(inv_inertia, r.h.x * inertia_stride);
        const BodyVel A0 = ld_vel(vel, r.h.x);
        g.ima = A0.inv_mass; g.massA = A0.mass; g.vA = A0.v; g.wA = A0.w;
        if (r.h.y != PHYS_GROUND_ID) {
            g.IB = ld_inertia<DIAG>(inv_inertia, r.h.y * inertia_stride);
            const BodyVel B0 = ld_vel(vel, r.h.y);
            g.imb = B0.inv_mass; g.massB = B0.mass; g.vB = B0.v; g.wB = B0.w;
        }
    };
    auto clear_bodies = [&](FlowBodies& g) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { g.IA.m[k] = 0.0f; g.IB.m[k] = 0.0f; }
        g.ima = 0.0f; g.imb = 0.0f; g.massA = 0.0f; g.massB = 0.0f;
        g.vA = v3_make(0.0f, 0.0f, 0.0f); g.wA = g.vA; g.vB = g.vA; g.wB = g.vA;
    };
    auto clear_row = [&](FlowRowRaw& r) {
        r.h = make_uint4(0u, PHYS_GROUND_ID, 0u, 0u);
        r.nn = make_float4(0.0f, 0.0f, 0.0f, 0.0f); r.t01 = r.nn; r.t23 = r.nn;
#pragma unroll
        for (int k = 0; k < 8; ++k) r.p[k] = r.nn;
    };
    // prologue: the first item with its rows and bodies
    uint32_t L = take();
    FlowRowRaw raw, raw_next;
    FlowBodies bod, bod_next;
    clear_row(raw); clear_bodies(bod);
    {
        const uint32_t d0 = row_of(L);
        if (d0 < M) { fetch_row(d0, raw); fetch_bodies(raw, bod); }
    }
    while (L < total) {
        uint32_t L_next = 0xFFFFFFFFu, d_next = 0xFFFFFFFFu;
        clear_row(raw_next); clear_bodies(bod_next);
        if (pipeline) {
            L_next = take();
            d_next = row_of(L_next);
            if (d_next < M) fetch_row(d_next, raw_next);
        }
        // ---- item L: unpack the row (same values as load_row)
        const uint32_t it = L / nchunks;
        const bool last_it = it + 1 == iterations;
        const uint32_t d = (L - it * nchunks) * blockDim.x + threadIdx.x;
        bool done = d >= M;
        RowRegs R;
        R.a = raw.h.x; R.b = raw.h.y; R.ticket = raw.h.w;
        {
            solver_manifold_t& sm = R.sm;
            sm.count = done ? 0 : (int)raw.h.z;
            sm.has_b = !done && R.b != PHYS_GROUND_ID;
            sm.n = v3_make(raw.nn.x, raw.nn.y, raw.nn.z);
            if (done) sm.n = v3_make(0.0f, 1.0f, 0.0f);
            tangent_basis(sm.n, &sm.t1, &sm.t2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                contact_row_t& c = sm.row[k];
                c.rA = v3_make(0.0f, 0.0f, 0.0f); c.rB = c.rA;
                c.normal_mass = 0.0f; c.tangent_mass[0] = 0.0f; c.tangent_mass[1] = 0.0f; c.bias = 0.0f;
                c.pn = 0.0f; c.pt[0] = 0.0f; c.pt[1] = 0.0f;
                if (k < sm.count) {
                    const float4 p0 = raw.p[2 * k], p1 = raw.p[2 * k + 1];
                    c.rA = v3_make(p0.x, p0.y, p0.z); c.normal_mass = p0.w;
                    c.rB = v3_make(p1.x, p1.y, p1.z); c.tangent_mass[0] = p1.w;
                    const float4 t = k < 2 ? raw.t01 : raw.t23;
                    c.tangent_mass[1] = (k & 1) ? t.z : t.x;
                    c.bias = (k & 1) ? t.w : t.y;
                }
            }
        }
        const bool apply_only = warm_sweep != 0u && it == 0u;
        if (apply_only && !done) {  // the starting impulses k_rows_build left in the impulse planes (plain loads: an earlier kernel wrote them)
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < R.sm.count) {
                const float4 a0 = rows.acc[(size_t)k * cap + d];
                R.sm.row[k].pn = a0.x; R.sm.row[k].pt[0] = a0.y; R.sm.row[k].pt[1] = a0.z;
            }
        }
        const uint32_t rankA = R.ticket & 0xFFu, degA = (R.ticket >> 8) & 0xFFu;
        const uint32_t rankB = (R.ticket >> 16) & 0xFFu, degB = R.ticket >> 24;
        const uint32_t tA = done ? 0u : it * degA + rankA, tB = (done || !R.sm.has_b) ? 0u : it * degB + rankB;
        const bool finalA = !done && last_it && rankA + 1 == degA, finalB = !done && R.sm.has_b && last_it && rankB + 1 == degB;
        const m33 IA = bod.IA, IB = bod.IB;
        const float ima = bod.ima, imb = bod.imb, massA = bod.massA, massB = bod.massB;
        v3 vA = bod.vA, wA = bod.wA, vB = bod.vB, wB = bod.wB;
        // the velocity-independent part of every row, while the row waits: the chain behind the wait is short
        solver_jac_t J;
        solver_jacobians(&R.sm, ima, &IA, imb, &IB, &J);
        // the rows of the next item have arrived by now (they were asked for before all of the above): its bodies
        if (pipeline && d_next < M) fetch_bodies(raw_next, bod_next);
        // what is still missing (a matched granule cannot change any more: this row is its next writer)
        bool needA = !done && tA != 0, needB = !done && R.sm.has_b && tB != 0, needAcc = !done && it != 0;
        uint32_t sweeps = 0;
        for (;;) {
            uint32_t gap = 0;
            if (!done) {
                u32x4 a0, a1, b0, b1, p[4];
                if (needA) { a0 = ld_granule(rv, R.a * 32u); a1 = ld_granule(rv, R.a * 32u + 16u); }
                if (needB) { b0 = ld_granule(rv, R.b * 32u); b1 = ld_granule(rv, R.b * 32u + 16u); }
                if (needAcc) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < R.sm.count) p[k] = ld_granule(ra, (k * cap + d) * 16u);
                }
                if (needA) {
                    const uint32_t want = etag | tA;
                    if (a0.w == want && a1.w == want) { vA = granule_v3(a0); wA = granule_v3(a1); needA = false; }
                    else {
                        const uint32_t seen = (a0.w >> 16) == epoch ? (a0.w & 0xFFFFu) : 0u;
                        gap = tA > seen ? tA - seen : 1u;
                    }
                }
                if (needB) {
                    const uint32_t want = etag | tB;
                    if (b0.w == want && b1.w == want) { vB = granule_v3(b0); wB = granule_v3(b1); needB = false; }
                    else {
                        const uint32_t seen = (b0.w >> 16) == epoch ? (b0.w & 0xFFFFu) : 0u;
                        const uint32_t g = tB > seen ? tB - seen : 1u;
                        gap = g > gap ? g : gap;
                    }
                }
                if (needAcc) {
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < R.sm.count) all = all && p[k].w == (etag | it);
                    if (all) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < R.sm.count) {
                            R.sm.row[k].pn = __uint_as_float(p[k].x);
                            R.sm.row[k].pt[0] = __uint_as_float(p[k].y);
                            R.sm.row[k].pt[1] = __uint_as_float(p[k].z);
                        }
                        needAcc = false;
                    } else if (gap == 0) {
                        gap = 1;
                    }
                }
                if (!needA && !needB && !needAcc) {
                    solve_manifold(&R.sm, &J, friction, &vA, &wA, &vB, &wB, apply_only ? 1 : 0);
                    // publish: bodies first (they are what other rows wait for)
                    if (finalA) { BodyVel o; o.v = vA; o.inv_mass = ima; o.w = wA; o.mass = massA; st_vel(vel, R.a, o); }
                    else { st_granule(rv, R.a * 32u, vA, etag | (tA + 1u)); st_granule(rv, R.a * 32u + 16u, wA, etag | (tA + 1u)); }
                    if (R.sm.has_b) {
                        if (finalB) { BodyVel o; o.v = vB; o.inv_mass = imb; o.w = wB; o.mass = massB; st_vel(vel, R.b, o); }
                        else { st_granule(rv, R.b * 32u, vB, etag | (tB + 1u)); st_granule(rv, R.b * 32u + 16u, wB, etag | (tB + 1u)); }
                    }
                    if (!last_it) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < R.sm.count)
                            st_granule(ra, (k * cap + d) * 16u, v3_make(R.sm.row[k].pn, R.sm.row[k].pt[0], R.sm.row[k].pt[1]),
                                       etag | (it + 1u));
                    } else if (wj.man_prev) {  // the solve's last sweep: remembered for the next update
                        const float fpn[4] = {R.sm.row[0].pn, R.sm.row[1].pn, R.sm.row[2].pn, R.sm.row[3].pn};
                        const float fp0[4] = {R.sm.row[0].pt[0], R.sm.row[1].pt[0], R.sm.row[2].pt[0], R.sm.row[3].pt[0]};
                        const float fp1[4] = {R.sm.row[0].pt[1], R.sm.row[1].pt[1], R.sm.row[2].pt[1], R.sm.row[3].pt[1]};
                        store_final_impulses(wj, d, fpn, fp0, fp1);
                    }
                    done = true;
                }
            }
            if (__all(done)) break;
            // back off in proportion to how many hops away the nearest waiting lane is
            if (__any(!done && gap <= 1u)) __builtin_amdgcn_s_sleep(1);
            else if (__any(!done && gap <= 4u)) __builtin_amdgcn_s_sleep(40);
            else __builtin_amdgcn_s_sleep(127);
            if ((++sweeps & 63u) == 0u) {
                const bool dead = (wall_clock64() - t_start > timeout_ticks) ||
                                  (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u);
                if (dead) {  // wave-uniform: both inputs are
                    if ((threadIdx.x & 63u) == 0u) flag_overflow(ctr, 16u);
                    done = true;
                }
            }
        }
        if (!pipeline) {
            L_next = take();
            d_next = row_of(L_next);
            if (d_next < M) { fetch_row(d_next, raw_next); fetch_bodies(raw_next, bod_next); }
        }
        L = L_next;
        raw = raw_next;
        bod = bod_next;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same dataflow solver with FOUR LANES PER MANIFOLD, for scenes where the hop latency is everything. At one
// wave per SIMD the row solve is VALU-issue-bound (~65 instructions per row x 12 rows, ~2/3 of a hop). A row is
// four dot products, a scalar update and four axpys over {vA, wA, vB, wB}: lane q of a quad owns ONE of those
// vectors with its Jacobian column and response (made beforehand, while waiting), so a row costs each lane one
// dot product, two cross-lane adds (DPP inside the quad: no LDS), the scalar update and one axpy - about a third
// of the instructions. The partial sums are combined in the order of the spec ((dir.vB + aB.wB) - (dir.vA +
// aA.wA)), so the bits are those of the one-lane kernels. The granule protocol maps one to one: lane q polls and
// publishes exactly its own 16-byte granule (v or w of A or B), and the impulse granule of contact point q.
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
constexpr int kQuadXor1 = 0xB1;  // quad_perm [1, 0, 3, 2]
constexpr int kQuadXor2 = 0x4E;  // quad_perm [2, 3, 0, 1]

template <bool DIAG>
__global__ __launch_bounds__(256) void k_solve_flow_quad(StepCounters* __restrict__ ctr, uint32_t iterations, uint32_t epoch,
                                                         RowArrays rows, float friction,
                                                         const float* __restrict__ inv_inertia, uint32_t inertia_stride,
                                                         float* vel, float* flow_vel, uint32_t n_bodies, long long timeout_ticks,
                                                         uint32_t warm_sweep, WarmJob wj) {
    if (ctr->overflow) return;
    const uint32_t M = ctr->n_manifolds;
    const uint32_t rows_per_item = blockDim.x >> 2;
    const uint32_t nchunks = (M + rows_per_item - 1) / rows_per_item;
    const uint32_t total = nchunks * iterations;
    const uint32_t cap = (uint32_t)rows.cap;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(flow_vel, 0, n_bodies * 32u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(rows.acc, 0, cap * 64u, 0x00020000);
    const uint32_t etag = epoch << 16;
    const uint32_t q = threadIdx.x & 3u;      // 0: vA, 1: wA, 2: vB, 3: wB
    const bool side_a = q < 2u, angular = (q & 1u) != 0u;
    const long long t_start = wall_clock64();
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    // Items are dealt STATICALLY here: workgroup g takes items g, g + G, g + 2G, ... in that order. A ticket from a global
    // counter (k_solve_flow keeps it) was one more dependent round trip and two barriers in front of every item of a
    // launch whose whole time is such chains: C2 0.078 -> 0.054 ms per solve, 6400 -> 7700 steps/s. (Measured and dropped:
    // the ticket of the next item asked for when an item begins - an item reserved by a workgroup that is still busy is an
    // item nobody else may start: C2 0.095 ms.)
    // No deadlock: an item waits only for items before it in the global order, and every workgroup walks its own items in
    // increasing order, so the earliest unfinished item of the launch belongs to a workgroup with nothing older left to
    // wait for - PROVIDED every workgroup of the launch gets to run. The grid is at most 224 workgroups of 256 threads
    // at <= 160 registers: three fit a CU, so a third of the chip holds the whole launch even beside other streams'
    // kernels; a workgroup that starts late delays its items (bounded spins: time-out after 3 s, never a silent hang).
    // k_solve_flow (one workgroup per CU at 370 registers) needs the whole chip for that and keeps its tickets.
    for (uint32_t L = blockIdx.x;; L += gridDim.x) {
        if (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u) return;  // somebody gave up
        if (L >= total) return;
        const uint32_t it = L / nchunks, chunk = L - it * nchunks;
        const bool last_it = it + 1 == iterations;
        const uint32_t d = chunk * rows_per_item + (threadIdx.x >> 2);
        bool done = d >= M;  // the same for the four lanes of a quad
        // per lane: its vector, its Jacobian column and (signed) response for the 12 rows; replicated: the scalars
        v3 x = zero, Jv[4][3], Rs[4][3];
        float nm[4], tm0[4], tm1[4], bias[4], pn[4], pt0[4], pt1[4];
        float keep_w = 0.0f;
        uint32_t body = 0, count = 0, ticket = 0;
        bool has_body = false, final_update = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nm[k] = 0.0f; tm0[k] = 0.0f; tm1[k] = 0.0f; bias[k] = 0.0f; pn[k] = 0.0f; pt0[k] = 0.0f; pt1[k] = 0.0f;
#pragma unroll
            for (int t = 0; t < 3; ++t) { Jv[k][t] = zero; Rs[k][t] = zero; }
        }
        if (!done) {
            const uint4 h = rows.hdr[d];
            count = h.z;
            const bool has_b = h.y != PHYS_GROUND_ID;
            body = side_a ? h.x : h.y;
            has_body = side_a || has_b;
            const uint32_t tk = side_a ? h.w : (h.w >> 16);
            const uint32_t rank = tk & 0xFFu, deg = (tk >> 8) & 0xFFu;
            ticket = it * deg + rank;
            final_update = last_it && rank + 1 == deg;
            const float4 nn = rows.n[d];
            v3 dir[3];
            dir[2] = v3_make(nn.x, nn.y, nn.z);
            tangent_basis(dir[2], &dir[0], &dir[1]);
            const float4 t01 = rows.tb[d];
            float4 t23 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (count > 2) t23 = rows.tb[cap + d];
            m33 I;
#pragma unroll
            for (int k = 0; k < 9; ++k) I.m[k] = 0.0f;
            float inv_m = 0.0f;
            if (has_body) {
                // the body's half of the plain velocity record: the state itself for ticket 0, the masses always
                const float4 h0 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body + (angular ? 1 : 0)];
                x = v3_make(h0.x, h0.y, h0.z);
                keep_w = h0.w;
                if (angular) I = ld_inertia<DIAG>(inv_inertia, body * inertia_stride); else inv_m = h0.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < (int)count) {
                    const float4 p0 = rows.pt[(size_t)(2 * k) * cap + d];      // rA, normal mass
                    const float4 p1 = rows.pt[(size_t)(2 * k + 1) * cap + d];  // rB, tangent mass 0
                    nm[k] = p0.w; tm0[k] = p1.w;
                    const float4 tt = k < 2 ? t01 : t23;
                    tm1[k] = (k & 1) ? tt.z : tt.x;
                    bias[k] = (k & 1) ? tt.w : tt.y;
                    const v3 r = side_a ? v3_make(p0.x, p0.y, p0.z) : v3_make(p1.x, p1.y, p1.z);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if (has_body) {
                            // solver_jacobians: lA = dir * invM; aA = r x dir, mA = I aA. The A side is subtracted
                            // by the spec: its response is stored negated (exact)
                            v3 jv, rs;
                            if (angular) { jv = v3_cross_f(r, dir[t]); rs = inertia_mul(&I, jv); }
                            else { jv = dir[t]; rs = v3_scale(dir[t], inv_m); }
                            Jv[k][t] = jv;
                            Rs[k][t] = side_a ? v3_neg(rs) : rs;
                        }
                    }
                }
            }
            if (!has_body) x = zero;
        }
        const bool apply_only = warm_sweep != 0u && it == 0u;
        bool need = !done && has_body && ticket != 0;
        bool need_acc = !done && it != 0 && q < count;  // lane q fetches the impulses of contact point q
        v3 my_acc = zero;
        if (apply_only && !done && q < count) {  // the starting impulses of point q (plain load: k_rows_build wrote them)
            const float4 a0 = rows.acc[(size_t)q * cap + d];
            my_acc = v3_make(a0.x, a0.y, a0.z);
        }
        uint32_t sweeps = 0;
        for (;;) {
            uint32_t gap = 0;
            if (!done) {
                u32x4 g, p;
                if (need) g = ld_granule(rv, body * 32u + (angular ? 16u : 0u));
                if (need_acc) p = ld_granule(ra, (q * cap + d) * 16u);
                if (need) {
                    const uint32_t want = etag | ticket;
                    if (g.w == want) { x = granule_v3(g); need = false; }
                    else {
                        const uint32_t seen = (g.w >> 16) == epoch ? (g.w & 0xFFFFu) : 0u;
                        gap = ticket > seen ? ticket - seen : 1u;
                    }
                }
                if (need_acc) {
                    if (p.w == (etag | it)) { my_acc = granule_v3(p); need_acc = false; }
                    else if (gap == 0) gap = 1;
                }
            }
            // the manifold goes when its four lanes have what they need
            int ready = (!need && !need_acc) ? 1 : 0;
            ready &= quad_perm_i<kQuadXor1>(ready);
            ready &= quad_perm_i<kQuadXor2>(ready);
            if (!done && ready) {
                if (it != 0 || apply_only) {  // everybody needs every point's accumulated impulses
                    pn[0] = quad_perm<0x00>(my_acc.x); pt0[0] = quad_perm<0x00>(my_acc.y); pt1[0] = quad_perm<0x00>(my_acc.z);
                    pn[1] = quad_perm<0x55>(my_acc.x); pt0[1] = quad_perm<0x55>(my_acc.y); pt1[1] = quad_perm<0x55>(my_acc.z);
                    pn[2] = quad_perm<0xAA>(my_acc.x); pt0[2] = quad_perm<0xAA>(my_acc.y); pt1[2] = quad_perm<0xAA>(my_acc.z);
                    pn[3] = quad_perm<0xFF>(my_acc.x); pt0[3] = quad_perm<0xFF>(my_acc.y); pt1[3] = quad_perm<0xFF>(my_acc.z);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < (int)count) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            // row_velocity: (dir.vB + aB.wB) - (dir.vA + aA.wA), the two sums made inside each pair
                            float lambda;
                            if (apply_only) {  // sweep 0 of a warm-started solve: the starting impulse reaches this lane's vector
                                lambda = t == 0 ? pt0[k] : (t == 1 ? pt1[k] : pn[k]);
                                x = v3_madd(x, Rs[k][t], lambda);
                                continue;
                            }
                            const float part = v3_dot_f(Jv[k][t], x);
                            const float mine = part + quad_perm<kQuadXor1>(part);
                            const float other = quad_perm<kQuadXor2>(mine);
                            const float vrel = side_a ? other - mine : mine - other;
                            if (t < 2) {  // solve_row_dir, friction
                                const float mass = t == 0 ? tm0[k] : tm1[k];
                                float& acc = t == 0 ? pt0[k] : pt1[k];
                                lambda = -mass * vrel;
                                const float maxf = friction * pn[k];
                                const float old = acc;
                                const float np = det_maxf(-maxf, det_minf(old + lambda, maxf));
                                lambda = np - old;
                                acc = np;
                            } else {      // normal
                                lambda = nm[k] * (bias[k] - vrel);
                                const float old = pn[k];
                                const float np = det_maxf(old + lambda, 0.0f);
                                lambda = np - old;
                                pn[k] = np;
                            }
                            x = v3_madd(x, Rs[k][t], lambda);  // row_apply
                        }
                    }
                }
                if (has_body) {
                    if (final_update)
                        reinterpret_cast<float4*>(vel)[2 * (size_t)body + (angular ? 1 : 0)] = make_float4(x.x, x.y, x.z, keep_w);
                    else
                        st_granule(rv, body * 32u + (angular ? 16u : 0u), x, etag | (ticket + 1u));
                }
                if (!last_it && q < count) {
                    const v3 mine = q == 0 ? v3_make(pn[0], pt0[0], pt1[0])
                                  : q == 1 ? v3_make(pn[1], pt0[1], pt1[1])
                                  : q == 2 ? v3_make(pn[2], pt0[2], pt1[2]) : v3_make(pn[3], pt0[3], pt1[3]);
                    st_granule(ra, (q * cap + d) * 16u, mine, etag | (it + 1u));
                }
                if (last_it && wj.man_prev) {  // lane q remembers point q's impulses (zeros beyond the count)
                    const v3 fin = q == 0 ? v3_make(pn[0], pt0[0], pt1[0])
                                 : q == 1 ? v3_make(pn[1], pt0[1], pt1[1])
                                 : q == 2 ? v3_make(pn[2], pt0[2], pt1[2]) : v3_make(pn[3], pt0[3], pt1[3]);
                    st3(wj.imp + 12 * (size_t)wj.row_src[d] + 3u * q, 0, fin);
                }
                done = true;
            }
            if (__all(done)) break;
            if (__any(!done && gap <= 1u)) __builtin_amdgcn_s_sleep(1);
            else if (__any(!done && gap <= 4u)) __builtin_amdgcn_s_sleep(40);
            else __builtin_amdgcn_s_sleep(127);
            if ((++sweeps & 63u) == 0u) {
                const bool dead = (wall_clock64() - t_start > timeout_ticks) ||
                                  (__hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u);
                if (dead) {
                    if ((threadIdx.x & 63u) == 0u) flag_overflow(ctr, 16u);
                    done = true;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// One colour of one iteration with FOUR LANES PER MANIFOLD and the rows staged through LDS: the per-colour
// kernel for scenes whose colour classes fill the chip (above kFlowMaxManifolds). k_solve_color runs one lane
// per manifold: a colour of 60-70k manifolds is then ~270 workgroups = one wave per SIMD, each lane a chain of
// header load -> ~20 dependent dwordx4 loads -> ~1200 VALU instructions -> stores with nothing to overlap it
// (17 us per launch for 26-33 MB, 19-24 % of the HBM rate). Here
//   * a workgroup owns 64 consecutive rows; every 16-byte element of the 16 row planes of those rows is loaded
//     ONCE (wave w fetches planes 4w..4w+3, 1 KiB contiguous per instruction, all issued before anything
//     else: no load depends on the header any more) and parked in LDS (16 KiB);
//   * lane q of a quad owns one of {vA, wA, vB, wB} exactly as in k_solve_flow_quad (same arithmetic, same
//     DPP order of the partial sums => same bits): a third of the VALU chain per lane, four times the waves
//     to overlap memory with arithmetic, and a gather of ONE 16-byte half record per lane.
// Rows of one colour share no body, so the plain loads / stores of `vel` need no ordering inside a launch.
constexpr int kQuadRowsPerGroup = 64;
template <bool DIAG>
__global__ __launch_bounds__(256) void k_solve_color_quad(StepCounters* ctr, uint32_t col, RowArrays rows,
                                                          float friction, const float* __restrict__ inv_inertia,
                                                          uint32_t inertia_stride, float* __restrict__ vel, uint32_t n_bodies,
                                                          int apply_only, int last, WarmJob wj) {
    __shared__ float4 s_rows[16][kQuadRowsPerGroup];  // [plane][row of this workgroup]
    if (ctr->overflow) return;
    const uint32_t start = ctr->color_start[col], end = ctr->color_start[col + 1];
    const uint32_t cap = (uint32_t)rows.cap;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q = threadIdx.x & 3u, r = threadIdx.x >> 2;  // q: 0 vA, 1 wA, 2 vB, 3 wB
    const bool side_a = q < 2u, angular = (q & 1u) != 0u;
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    // the four planes this wave stages: 4 * wave + j of the ONE row allocation (0 hdr, 1 n, 2-3 tb, 4-11 pt, 12-15 acc).
    // Plain arithmetic on purpose: written as `if (wave == 0) {pointers..} else if ..` over the five arrays, hipcc
    // (ROCm 7.2) lost one case of the pointer selection (wave 3 staged from a null pointer: memory access fault)
    const float4* pl = rows.all + (size_t)(4u * wave) * cap;
    for (XcdTiles tl(end - start, kQuadRowsPerGroup); tl.valid(); tl.next()) {  // XCD-aware: see k_solve_color
        const uint32_t base = start + tl.tile * kQuadRowsPerGroup;
        {
            // rows beyond `end` belong to later colours (read only, never used); beyond the arrays: clamp
            uint32_t e = base + lane;
            e = e < cap ? e : cap - 1u;
            const float4 t0 = pl[e], t1 = pl[(size_t)cap + e], t2 = pl[2 * (size_t)cap + e], t3 = pl[3 * (size_t)cap + e];
            s_rows[4 * wave + 0][lane] = t0; s_rows[4 * wave + 1][lane] = t1;
            s_rows[4 * wave + 2][lane] = t2; s_rows[4 * wave + 3][lane] = t3;
        }
        __syncthreads();
        const uint32_t d = base + r;
        const bool live = d < end;  // the same for the four lanes of a quad
        v3 x = zero, Jv[4][3], Rs[4][3];
        float nm[4], tm0[4], tm1[4], bias[4], pn[4], pt0[4], pt1[4];
        float keep_w = 0.0f;
        uint32_t body = 0, count = 0;
        bool has_body = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nm[k] = 0.0f; tm0[k] = 0.0f; tm1[k] = 0.0f; bias[k] = 0.0f; pn[k] = 0.0f; pt0[k] = 0.0f; pt1[k] = 0.0f;
#pragma unroll
            for (int t = 0; t < 3; ++t) { Jv[k][t] = zero; Rs[k][t] = zero; }
        }
        if (live) {
            const float4 hraw = s_rows[0][r];
            const uint32_t ha = __float_as_uint(hraw.x), hb = __float_as_uint(hraw.y);
            count = __float_as_uint(hraw.z);
            const bool has_b = hb != PHYS_GROUND_ID;
            body = side_a ? ha : hb;
            has_body = side_a || has_b;
            // a row header that names no body of this world must never become an address (a faulting kernel can
            // take the whole node down): flag the step (bit 5) and skip the row
            if (count > 4u || (has_body && body >= n_bodies)) {
                flag_overflow(ctr, 32u);
                ctr->debug[0] = d; ctr->debug[1] = ha; ctr->debug[2] = hb; ctr->debug[3] = count;
                ctr->debug[4] = start; ctr->debug[5] = end; ctr->debug[6] = col; ctr->debug[7] = base;
                count = 0; has_body = false;
            }
            m33 I;
#pragma unroll
            for (int k = 0; k < 9; ++k) I.m[k] = 0.0f;
            float inv_m = 0.0f;
            if (has_body) {
                // this lane's half of the velocity record: {v, 1/m} or {w, m}
                const float4 h0 = reinterpret_cast<const float4*>(vel)[2 * (size_t)body + (angular ? 1 : 0)];
                x = v3_make(h0.x, h0.y, h0.z);
                keep_w = h0.w;
                if (angular) I = ld_inertia<DIAG>(inv_inertia, body * inertia_stride); else inv_m = h0.w;
            }
            const float4 nn = s_rows[1][r];
            v3 dir[3];
            dir[2] = v3_make(nn.x, nn.y, nn.z);
            tangent_basis(dir[2], &dir[0], &dir[1]);
            const float4 t01 = s_rows[2][r];
            float4 t23 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (count > 2) t23 = s_rows[3][r];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < (int)count) {
                    const float4 p0 = s_rows[4 + 2 * k][r];      // rA, normal mass
                    const float4 p1 = s_rows[4 + 2 * k + 1][r];  // rB, tangent mass 0
                    const float4 ac = s_rows[12 + k][r];         // accumulated impulses of point k
                    nm[k] = p0.w; tm0[k] = p1.w;
                    const float4 tt = k < 2 ? t01 : t23;
                    tm1[k] = (k & 1) ? tt.z : tt.x;
                    bias[k] = (k & 1) ? tt.w : tt.y;
                    pn[k] = ac.x; pt0[k] = ac.y; pt1[k] = ac.z;
                    const v3 rr = side_a ? v3_make(p0.x, p0.y, p0.z) : v3_make(p1.x, p1.y, p1.z);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if (has_body) {
                            // solver_jacobians: lA = dir * invM; aA = r x dir, mA = I aA; the A side is subtracted by
                            // the spec, so its response is stored negated (exact)
                            v3 jv, rs;
                            if (angular) { jv = v3_cross_f(rr, dir[t]); rs = inertia_mul(&I, jv); }
                            else { jv = dir[t]; rs = v3_scale(dir[t], inv_m); }
                            Jv[k][t] = jv;
                            Rs[k][t] = side_a ? v3_neg(rs) : rs;
                        }
                    }
                }
            }
            if (!has_body) x = zero;
        }
        // every lane takes part in the DPP exchanges (dead quads carry zeros)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < (int)count) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    // row_velocity: (dir.vB + aB.wB) - (dir.vA + aA.wA), the two sums made inside each pair
                    float lambda;
                    if (apply_only) {  // sweep 0 of a warm-started solve: the starting impulse reaches this lane's vector
                        lambda = t == 0 ? pt0[k] : (t == 1 ? pt1[k] : pn[k]);
                        x = v3_madd(x, Rs[k][t], lambda);
                        continue;
                    }
                    const float part = v3_dot_f(Jv[k][t], x);
                    const float mine = part + quad_perm<kQuadXor1>(part);
                    const float other = quad_perm<kQuadXor2>(mine);
                    const float vrel = side_a ? other - mine : mine - other;
                    if (t < 2) {  // solve_row_dir, friction
                        const float mass = t == 0 ? tm0[k] : tm1[k];
                        float& acc = t == 0 ? pt0[k] : pt1[k];
                        lambda = -mass * vrel;
                        const float maxf = friction * pn[k];
                        const float old = acc;
                        const float np = det_maxf(-maxf, det_minf(old + lambda, maxf));
                        lambda = np - old;
                        acc = np;
                    } else {      // normal
                        lambda = nm[k] * (bias[k] - vrel);
                        const float old = pn[k];
                        const float np = det_maxf(old + lambda, 0.0f);
                        lambda = np - old;
                        pn[k] = np;
                    }
                    x = v3_madd(x, Rs[k][t], lambda);  // row_apply
                }
            }
        }
        if (live) {
            if (has_body)
                reinterpret_cast<float4*>(vel)[2 * (size_t)body + (angular ? 1 : 0)] = make_float4(x.x, x.y, x.z, keep_w);
            if (q < count) {
                const float4 mine = q == 0 ? make_float4(pn[0], pt0[0], pt1[0], 0.0f)
                                  : q == 1 ? make_float4(pn[1], pt0[1], pt1[1], 0.0f)
                                  : q == 2 ? make_float4(pn[2], pt0[2], pt1[2], 0.0f) : make_float4(pn[3], pt0[3], pt1[3], 0.0f);
                rows.acc[(size_t)q * cap + d] = mine;
            }
            if (last && wj.man_prev) {  // lane q remembers point q's impulses (zeros beyond the count)
                float* o = wj.imp + 12 * (size_t)wj.row_src[d] + 3u * q;
                st3(o, 0, q == 0 ? v3_make(pn[0], pt0[0], pt1[0]) : q == 1 ? v3_make(pn[1], pt0[1], pt1[1])
                        : q == 2 ? v3_make(pn[2], pt0[2], pt1[2]) : v3_make(pn[3], pt0[3], pt1[3]));
            }
        }
        __syncthreads();  // the LDS tile is restaged by the next trip
    }
}

// Launch sizes come from the HINT (counters of an earlier step, read back asynchronously); every kernel
// takes its real ranges from the device-side counters, so a stale hint costs speed, never correctness.
void launch_solver(phys_world* w, float dt) {
    if (w->n == 0) return;
    const StepHint& h = w->hint;
    solve_params_t sp;
    sp.dt = dt;
    sp.baumgarte = w->cfg.baumgarte;
    sp.slop = w->cfg.slop;
    sp.friction = w->cfg.friction;
    sp.max_bias = w->cfg.max_bias;
    hipStream_t s = w->stream;
    const uint64_t cap = w->max_manifolds;
    const dim3 tb(256);
    const bool diag = w->all_diag_inertia;
    // every body shares one diagonal tensor (the reference's only case, identity): all lanes read entry 0
    const float* inertia = diag ? w->inv_inertia_diag.p : w->inv_inertia.p;
    const uint32_t stride = diag && w->uniform_inertia ? 0u : 1u;
    RowArrays rows;
    rows.all = reinterpret_cast<float4*>(w->row_all.p);
    rows.hdr = reinterpret_cast<uint4*>(w->row_hdr.p);
    rows.n = reinterpret_cast<float4*>(w->row_n.p);
    rows.pt = reinterpret_cast<float4*>(w->row_pt.p);
    rows.tb = reinterpret_cast<float4*>(w->row_tb.p);
    rows.acc = reinterpret_cast<float4*>(w->row_acc.p);
    rows.cap = cap;
    auto grid_for_count = [&](uint64_t count) {
        uint64_t b = (count * 5 / 4 + 255) / 256 + 1;
        const uint64_t hi = (cap + 255) / 256;
        if (b > hi) b = hi;
        if (b > 4096) b = 4096;
        return dim3((unsigned)(b ? b : 1));
    };
    const uint64_t m_hint = h.valid ? h.n_manifolds : cap;
    // the dataflow kernel wins while a colour class is too small to fill the chip (launch / latency bound);
    // beyond that the per-colour launches stream better. Both give the same bits, so the choice may change
    // from step to step.
    // (tickets are 16-bit: iterations x 64 colours must stay below 65536)
    // PHYS_DEBUG_FLOW_MAX=<manifolds>: move the dataflow / per-colour crossover (measurements only; same bits either way)
    static const char* flow_max_env = getenv("PHYS_DEBUG_FLOW_MAX");
    const uint64_t flow_max = flow_max_env ? strtoull(flow_max_env, nullptr, 10) : kFlowMaxManifolds;
    const bool cluster = w->cluster_step;  // decided by launch_coloring: this update's rows are in (cluster, colour) order
    const bool flow = cluster || (w->flow_vel.p && h.valid && m_hint <= flow_max && w->cfg.solver_iterations > 0 &&
                                  w->cfg.solver_iterations < 1000);
    // fault injection for tests/test_gpu_full_size.py: one row gets a ticket nobody will ever publish, so the bounded
    // spin of the dataflow kernels must give up, flag the step (overflow bit 4) and let the launch end
    static const bool stall = getenv("PHYS_DEBUG_FLOW_STALL") != nullptr;
    const long long timeout_ticks = stall ? 2000000ll : kFlowTimeoutTicks;
    // warm starting (contact_solve.h): one sweep more, in front - it applies the impulses the rows start from
    WarmJob warm{};
    if (w->warm) {
        warm.man_prev = w->man_prev.p; warm.geo_prev = w->man_geo_prev.p; warm.imp_prev = w->man_imp_prev.p;
        warm.imp = w->man_imp.p; warm.row_src = w->row_src.p;
    }
    const uint32_t warm_sweep = w->warm ? 1u : 0u;
    const uint32_t sweeps = w->cfg.solver_iterations + warm_sweep;
    ColorTableJob table{};
    if (w->ctab_job_pending) {
        table.tab = reinterpret_cast<ulonglong2*>(w->ctab.p);
        table.mask = w->ctab_mask;
        table.stamp = w->ctab_job_stamp;
        table.all = w->ctab_job_all ? 1u : 0u;
        table.man_color = w->man_color.p; table.man_prio = w->man_prio.p;
        w->ctab_job_pending = false;
    }
    if (flow && ++w->flow_epoch > 0xFFFFu) {
        // tags would repeat: forget every old one. AHEAD of k_rows_build: on a cluster step that kernel writes the constants
        // of foreign bodies into planes 12-15 = row_acc (with flow != 0 it never writes the impulses there), and a memset
        // behind it wiped them - every 65535th solve ran with x = 0, 1/m = 0, I^-1 = 0 for those rows
        PHYS_PROF(w, PHYS_STAGE_MISC);
        (void)hipMemsetAsync(w->flow_vel.p, 0, 8 * w->n * sizeof(float), s);
        (void)hipMemsetAsync(w->row_acc.p, 0, 16 * cap * sizeof(float), s);
        w->flow_epoch = 1;
    }
    { PHYS_PROF(w, PHYS_STAGE_ROWS);
      if (diag)
          hipLaunchKernelGGL(k_rows_build<true>, grid_for_count(m_hint), tb, 0, s, w->counters.p, rows, sp, w->row_src.p, w->man_geo.p,
                             w->pos.p, w->vel.p, inertia, stride, w->man_color.p,
                             w->color_state.p, flow ? (stall ? 2 : 1) : 0, table, w->cluster_slot.p, w->body_shared.p,
                             w->cluster_step ? w->cluster_slots : 0u, w->cluster_count, warm);
      else
          hipLaunchKernelGGL(k_rows_build<false>, grid_for_count(m_hint), tb, 0, s, w->counters.p, rows, sp, w->row_src.p, w->man_geo.p,
                             w->pos.p, w->vel.p, inertia, stride, w->man_color.p,
                             w->color_state.p, flow ? (stall ? 2 : 1) : 0, table, w->cluster_slot.p, w->body_shared.p,
                             w->cluster_step ? w->cluster_slots : 0u, w->cluster_count, warm); }
    if (flow) {
        if (cluster) {
            PHYS_PROF(w, PHYS_STAGE_SOLVE_CLUSTER);
            launch_solve_cluster(w, rows.all, cap, sp.friction, inertia, stride, diag, timeout_ticks);
            return;
        }
        // about one wave per SIMD or less: waiting waves must not crowd out the ones that can run
        static const uint64_t quad_max_env = getenv("PHYS_DEBUG_FLOW_QUAD_MAX") ? strtoull(getenv("PHYS_DEBUG_FLOW_QUAD_MAX"), nullptr, 10) : 0;  // measurements
        // four lanes per manifold while the hop latency is everything - and, where the launch may take the whole chip
        // (w->flow_wide: three workgroups per CU, 672 of them), all the way up: 155k manifolds 0.250 ms against 0.446 with
        // 224 workgroups, C3's 216k 0.325 (cluster kernel 0.513), the 1M cubes' 379k 0.405 (cluster kernel 0.235)
        const bool quad = m_hint <= (quad_max_env ? quad_max_env : (w->flow_wide ? kFlowMaxManifolds : kFlowQuadMaxManifolds));
        const uint32_t threads = 256u;
        const uint32_t rows_per_item = quad ? threads / 4 : threads;
        uint64_t items = (uint64_t)sweeps * ((m_hint * 5 / 4 + rows_per_item - 1) / rows_per_item) + 1;
        static const uint64_t wgs_env = getenv("PHYS_DEBUG_FLOW_WGS") ? strtoull(getenv("PHYS_DEBUG_FLOW_WGS"), nullptr, 10) : 0;  // measurements (quad)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w->device);
        // statically dealt items need every workgroup running: a third of the chip's slots by default (beside other
        // streams' kernels), seven eighths of them - the cluster kernel's share - where the GPU is this world's alone
        // (small scenes are a chain of hand-offs, not throughput: C2's 10k manifolds 0.053 ms at 224 workgroups, 0.058 at 672)
        const uint64_t most = quad ? (wgs_env ? wgs_env : (w->flow_wide && m_hint > 32768u ? (uint64_t)(3 * (cus - cus / 8)) : 224)) : 256;
        if (items > most) items = most;  // the remaining items are taken by the same workgroups
        PHYS_PROF(w, PHYS_STAGE_SOLVE_FLOW);
        // look one work item ahead (k_solve_flow) while a colour class keeps a good part of the launch busy; below that the
        // solve is a chain of hand-offs and an item held ahead only waits (C3: 16k rows per colour, 65k lanes: +10 %;
        // 1M cubes: 41k rows per colour: -10 %). PHYS_DEBUG_FLOW_PIPELINE=0/1 forces it (measurements; same bits).
        static const char* pipe_env = getenv("PHYS_DEBUG_FLOW_PIPELINE");
        const uint64_t per_color = m_hint / (h.valid && h.n_colors ? h.n_colors : 1u);
        const uint32_t pipeline = pipe_env ? (uint32_t)(pipe_env[0] == '1') : (uint32_t)(4 * per_color >= threads * items);
#define PHYS_FLOW_ARGS dim3((unsigned)items), dim3(threads), 0, s, w->counters.p, sweeps, \
                       w->flow_epoch, rows, sp.friction, inertia, stride, w->vel.p, w->flow_vel.p, (uint32_t)w->n, timeout_ticks
        if (quad) { if (diag) hipLaunchKernelGGL(k_solve_flow_quad<true>, PHYS_FLOW_ARGS, warm_sweep, warm); else hipLaunchKernelGGL(k_solve_flow_quad<false>, PHYS_FLOW_ARGS, warm_sweep, warm); }
        else { if (diag) hipLaunchKernelGGL(k_solve_flow<true>, PHYS_FLOW_ARGS, pipeline, warm_sweep, warm); else hipLaunchKernelGGL(k_solve_flow<false>, PHYS_FLOW_ARGS, pipeline, warm_sweep, warm); }
#undef PHYS_FLOW_ARGS
        return;
    }
    // colours [0, big) get a launch each; [big, n_colours) go through the single-workgroup tail
    constexpr uint32_t kTailMax = 512;  // manifolds per colour the tail should take: one trip of the workgroup
    uint32_t big = 0;
    if (h.valid) {
        big = h.n_colors;
        while (big > 0 && h.color_count[big - 1] <= kTailMax) --big;
        if (h.n_colors - big < 2) big = h.n_colors;  // a tail of one colour is just a slower launch
    }
    // PHYS_DEBUG_COLOR_KERNEL=lane: the one-lane-per-manifold kernel for every colour (A/B measurements, parity tests)
    static const char* color_kernel_env = getenv("PHYS_DEBUG_COLOR_KERNEL");
    static const uint32_t ablate = getenv("PHYS_DEBUG_ABLATE") ? (uint32_t)atoi(getenv("PHYS_DEBUG_ABLATE")) : 0u;
    // four lanes per manifold while a colour is too small to fill the chip with one lane per manifold (measured
    // crossover ~30k rows: 15k rows 10.2 vs 11.9 us per launch, 53k rows 18.3 vs 16.7, 85k rows 21.5 vs 18.7)
    constexpr uint32_t kQuadColorMaxRows = 32768;
    const int color_kernel_mode = !color_kernel_env ? 0 : (color_kernel_env[0] == 'l' ? 1 : 2);  // 0 auto, 1 lane, 2 quad
    auto grid_for_quads = [&](uint64_t count) {
        uint64_t b = (count * 5 / 4 + kQuadRowsPerGroup - 1) / kQuadRowsPerGroup + 1;
        const uint64_t hi = (cap + kQuadRowsPerGroup - 1) / kQuadRowsPerGroup;
        if (b > hi) b = hi;
        if (b > 16384) b = 16384;
        return dim3((unsigned)(b ? b : 1));
    };
    for (uint32_t it = 0; it < sweeps; ++it) {
        const int apply_only = warm_sweep && it == 0 ? 1 : 0, last = it + 1 == sweeps ? 1 : 0;
        for (uint32_t col = 0; col < big; ++col) {
            PHYS_PROF(w, PHYS_STAGE_SOLVE);
            if (color_kernel_mode == 2 || (color_kernel_mode == 0 && h.color_count[col] <= kQuadColorMaxRows)) {
                if (diag)
                    hipLaunchKernelGGL(k_solve_color_quad<true>, grid_for_quads(h.color_count[col]), tb, 0, s, w->counters.p, col, rows,
                                       sp.friction, inertia, stride, w->vel.p, (uint32_t)w->n, apply_only, last, warm);
                else
                    hipLaunchKernelGGL(k_solve_color_quad<false>, grid_for_quads(h.color_count[col]), tb, 0, s, w->counters.p, col, rows,
                                       sp.friction, inertia, stride, w->vel.p, (uint32_t)w->n, apply_only, last, warm);
                continue;
            }
            if (diag)
                hipLaunchKernelGGL(k_solve_color<true>, grid_for_count(h.color_count[col]), tb, 0, s, w->counters.p, col, rows,
                                   sp.friction, inertia, stride, w->vel.p, ablate, (uint32_t)w->n, apply_only, last, warm);
            else
                hipLaunchKernelGGL(k_solve_color<false>, grid_for_count(h.color_count[col]), tb, 0, s, w->counters.p, col, rows,
                                   sp.friction, inertia, stride, w->vel.p, ablate, (uint32_t)w->n, apply_only, last, warm);
        }
        PHYS_PROF(w, PHYS_STAGE_SOLVE_TAIL);
        if (diag)
            hipLaunchKernelGGL(k_solve_tail<true>, dim3(1), dim3(kTailThreads), 0, s, w->counters.p, big, rows, sp.friction,
                               inertia, stride, w->vel.p, apply_only, last, warm);
        else
            hipLaunchKernelGGL(k_solve_tail<false>, dim3(1), dim3(kTailThreads), 0, s, w->counters.p, big, rows, sp.friction,
                               inertia, stride, w->vel.p, apply_only, last, warm);
    }
}

}  // namespace phys
