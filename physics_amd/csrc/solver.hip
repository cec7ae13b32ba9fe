// solver.hip — sequential-impulse contact solver for gfx950. No reference
// counterpart; the arithmetic is include/spec/contact_solve.h (solver_prep / solve_manifold).
//
// Order of work = the spec's: iterations outermost, colours ascending, manifolds of one colour in
// parallel (they share no body, so there are no float atomics and the result is independent of the
// schedule), points of a manifold in index order inside one lane.
//
// Data layout: after colouring, manifolds are renumbered colour-major (row d = colour start + slot in
// colour). Solver rows are plane-major SoA over d, so each of the 8 x n_colours solve launches streams
// its rows with unit stride; only the body velocities (24 B read + 24 B written per body) and the
// inverse mass / inertia are gathered by body id.
// Algorithmic bytes per manifold per iteration: row planes 12 + 12 (ids, count, normal) +
// per point 40 (rA, rB, masses, bias) + 12 R + 12 W (accumulated impulses), + per body 48 (v, w R+W) +
// 4 (inv mass) + 36 (inverse inertia).
#include "kernels.hpp"

namespace phys {

struct ColorTable {
    uint32_t start[kMaxColors + 1];
};

constexpr int kRowPlanesPerPoint = 10;  // rA xyz, rB xyz, normal mass, tangent mass 0/1, bias
constexpr int kAccPlanesPerPoint = 3;   // pn, pt0, pt1

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

__global__ __launch_bounds__(256) void k_rows_build(const StepCounters* __restrict__ ctr, uint64_t cap, solve_params_t sp,
                                                    const uint32_t* __restrict__ row_src,
                                                    const uint32_t* __restrict__ man_a, const uint32_t* __restrict__ man_b,
                                                    const uint32_t* __restrict__ man_count,
                                                    const float* __restrict__ man_normal,
                                                    const float* __restrict__ man_points, const float* __restrict__ pos,
                                                    const float* __restrict__ vel,
                                                    const float* __restrict__ inv_inertia, uint32_t* __restrict__ row_a,
                                                    uint32_t* __restrict__ row_b, uint32_t* __restrict__ row_count,
                                                    float* __restrict__ row_normal, float* __restrict__ row_data,
                                                    float* __restrict__ row_acc) {
    if (ctr->overflow) return;  // never solve a truncated set; phys_sync / phys_get_stats report it
    const uint32_t M = ctr->n_manifolds;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < M; d += gridDim.x * blockDim.x) {
    const uint32_t m = row_src[d];
    manifold_t g;
    const uint32_t a = man_a[m], b = man_b[m];
    g.count = (int)man_count[m];
    g.normal = ld3(man_normal, m);
    const float4* pp = reinterpret_cast<const float4*>(man_points) + 4 * (size_t)m;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 p = pp[k];
        g.pt[k] = v3_make(p.x, p.y, p.z);
        g.depth[k] = p.w;
    }
    const int has_b = b != PHYS_GROUND_ID;
    const m33 IA = ld_m33(inv_inertia, a);
    m33 IB;
#pragma unroll
    for (int k = 0; k < 9; ++k) IB.m[k] = 0.0f;
    float imb = 0.0f;
    v3 xB = v3_make(0.0f, 0.0f, 0.0f);
    if (has_b) { IB = ld_m33(inv_inertia, b); imb = vel[8 * (size_t)b + 3]; xB = ld3(pos, b); }
    solver_manifold_t sm;
    solver_prep(&g, has_b, ld3(pos, a), xB, vel[8 * (size_t)a + 3], &IA, imb, &IB, &sp, &sm);
    row_a[d] = a; row_b[d] = b; row_count[d] = (uint32_t)sm.count;
    row_normal[0 * cap + d] = sm.n.x; row_normal[1 * cap + d] = sm.n.y; row_normal[2 * cap + d] = sm.n.z;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < sm.count) {
            float* r = row_data + (size_t)(k * kRowPlanesPerPoint) * cap + d;
            const contact_row_t& c = sm.row[k];
            r[0 * cap] = c.rA.x; r[1 * cap] = c.rA.y; r[2 * cap] = c.rA.z;
            r[3 * cap] = c.rB.x; r[4 * cap] = c.rB.y; r[5 * cap] = c.rB.z;
            r[6 * cap] = c.normal_mass; r[7 * cap] = c.tangent_mass[0]; r[8 * cap] = c.tangent_mass[1];
            r[9 * cap] = c.bias;
            float* acc = row_acc + (size_t)(k * kAccPlanesPerPoint) * cap + d;
            acc[0 * cap] = 0.0f; acc[1 * cap] = 0.0f; acc[2 * cap] = 0.0f;
        }
    }
    }
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

// one manifold row d of the colour-major numbering, in registers
struct RowRegs {
    uint32_t a, b;
    solver_manifold_t sm;
};

// part 1: everything that does NOT depend on body velocities (can be fetched a phase ahead)
__device__ __forceinline__ void load_row(RowRegs& R, uint32_t d, uint64_t cap, const uint32_t* __restrict__ row_a,
                                         const uint32_t* __restrict__ row_b, const uint32_t* __restrict__ row_count,
                                         const float* __restrict__ row_normal, const float* __restrict__ row_data,
                                         const float* __restrict__ row_acc) {
    R.a = row_a[d]; R.b = row_b[d];
    solver_manifold_t& sm = R.sm;
    sm.count = (int)row_count[d];
    sm.has_b = R.b != PHYS_GROUND_ID;
    sm.n = v3_make(row_normal[0 * cap + d], row_normal[1 * cap + d], row_normal[2 * cap + d]);
    tangent_basis(sm.n, &sm.t1, &sm.t2);  // same inputs as solver_prep => same bits as the stored basis
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        contact_row_t& c = sm.row[k];
        if (k < sm.count) {
            const float* r = row_data + (size_t)(k * kRowPlanesPerPoint) * cap + d;
            c.rA = v3_make(r[0 * cap], r[1 * cap], r[2 * cap]);
            c.rB = v3_make(r[3 * cap], r[4 * cap], r[5 * cap]);
            c.normal_mass = r[6 * cap]; c.tangent_mass[0] = r[7 * cap]; c.tangent_mass[1] = r[8 * cap];
            c.bias = r[9 * cap];
            const float* acc = row_acc + (size_t)(k * kAccPlanesPerPoint) * cap + d;
            c.pn = acc[0 * cap]; c.pt[0] = acc[1 * cap]; c.pt[1] = acc[2 * cap];
        } else {
            c.rA = v3_make(0.0f, 0.0f, 0.0f); c.rB = v3_make(0.0f, 0.0f, 0.0f);
            c.normal_mass = 0.0f; c.tangent_mass[0] = 0.0f; c.tangent_mass[1] = 0.0f; c.bias = 0.0f;
            c.pn = 0.0f; c.pt[0] = 0.0f; c.pt[1] = 0.0f;
        }
    }
}

// part 2: gather the two bodies, solve_manifold, write velocities and accumulated impulses back
template <bool DIAG>
__device__ __forceinline__ void solve_loaded_row(RowRegs& R, uint32_t d, uint64_t cap, float friction,
                                                 float* __restrict__ row_acc, const float* __restrict__ inv_inertia,
                                                 uint32_t inertia_stride /* 0: one tensor shared by every body */,
                                                 float* __restrict__ vel) {
    solver_manifold_t& sm = R.sm;
    const uint32_t a = R.a, b = R.b;
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
    solve_manifold(&sm, friction, ima, &IA, imb, &IB, &vA, &wA, &vB, &wB);
    A.v = vA; A.w = wA;
    st_vel(vel, a, A);
    if (sm.has_b) { B.v = vB; B.w = wB; st_vel(vel, b, B); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < sm.count) {
            float* acc = row_acc + (size_t)(k * kAccPlanesPerPoint) * cap + d;
            acc[0 * cap] = sm.row[k].pn; acc[1 * cap] = sm.row[k].pt[0]; acc[2 * cap] = sm.row[k].pt[1];
        }
    }
}

template <bool DIAG>
__device__ __forceinline__ void solve_row(uint32_t d, uint64_t cap, float friction, const uint32_t* __restrict__ row_a,
                                          const uint32_t* __restrict__ row_b, const uint32_t* __restrict__ row_count,
                                          const float* __restrict__ row_normal, const float* __restrict__ row_data,
                                          float* __restrict__ row_acc, const float* __restrict__ inv_inertia,
                                          uint32_t inertia_stride, float* __restrict__ vel) {
    RowRegs R;
    load_row(R, d, cap, row_a, row_b, row_count, row_normal, row_data, row_acc);
    solve_loaded_row<DIAG>(R, d, cap, friction, row_acc, inv_inertia, inertia_stride, vel);
}

// one colour of one iteration; the row range comes from the device-side colour table
template <bool DIAG>
__global__ __launch_bounds__(256) void k_solve_color(const StepCounters* __restrict__ ctr, uint32_t col, uint64_t cap,
                                                     float friction, const uint32_t* __restrict__ row_a,
                                                     const uint32_t* __restrict__ row_b,
                                                     const uint32_t* __restrict__ row_count,
                                                     const float* __restrict__ row_normal,
                                                     const float* __restrict__ row_data, float* __restrict__ row_acc,
                                                     const float* __restrict__ inv_inertia, uint32_t inertia_stride, float* __restrict__ vel) {
    if (ctr->overflow) return;
    const uint32_t start = ctr->color_start[col], end = ctr->color_start[col + 1];
    for (uint32_t d = start + blockIdx.x * blockDim.x + threadIdx.x; d < end; d += gridDim.x * blockDim.x)
        solve_row<DIAG>(d, cap, friction, row_a, row_b, row_count, row_normal, row_data, row_acc, inv_inertia, inertia_stride, vel);
}

// The colour classes [first, n_colours) of one iteration in ONE launch of ONE workgroup: colours in
// ascending order with a workgroup barrier between them (all waves share this CU's L1, so a barrier orders
// the body-velocity writes of one colour before the reads of the next). Same order of work as one launch per
// colour, without paying a ~8 us launch for a few hundred manifolds. `first` is a host HINT (the small
// colours of the previous step); any value gives the same result, only the speed changes.
constexpr int kTailThreads = 512;  // 2 waves per SIMD: the row solve needs ~144 VGPRs, 1024 threads would spill
template <bool DIAG>
__global__ __launch_bounds__(kTailThreads) void k_solve_tail(const StepCounters* __restrict__ ctr, uint32_t first, uint64_t cap,
                                                            float friction, const uint32_t* __restrict__ row_a,
                                                            const uint32_t* __restrict__ row_b,
                                                            const uint32_t* __restrict__ row_count,
                                                            const float* __restrict__ row_normal,
                                                            const float* __restrict__ row_data, float* __restrict__ row_acc,
                                                            const float* __restrict__ inv_inertia, uint32_t inertia_stride, float* __restrict__ vel) {
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
            solve_row<DIAG>(d, cap, friction, row_a, row_b, row_count, row_normal, row_data, row_acc, inv_inertia, inertia_stride, vel);
        __threadfence_block();
        __syncthreads();
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
    const uint32_t istride = w->uniform_inertia ? 0u : 1u;
    auto grid_for_count = [&](uint64_t count) {
        uint64_t b = (count * 5 / 4 + 255) / 256 + 1;
        const uint64_t hi = (cap + 255) / 256;
        if (b > hi) b = hi;
        if (b > 4096) b = 4096;
        return dim3((unsigned)(b ? b : 1));
    };
    const uint64_t m_hint = h.valid ? h.n_manifolds : cap;
    { PHYS_PROF(w, PHYS_STAGE_ROWS); hipLaunchKernelGGL(k_rows_build, grid_for_count(m_hint), tb, 0, s, w->counters.p, cap, sp, w->row_src.p, w->man_a.p, w->man_b.p, w->man_count.p,
                       w->man_normal.p, w->man_points.p, w->pos.p, w->vel.p, w->inv_inertia.p, w->row_a.p,
                       w->row_b.p, w->row_count.p, w->row_normal.p, w->row_data.p, w->row_acc.p); }
    // colours [0, big) get a launch each; [big, n_colours) go through the single-workgroup tail
    constexpr uint32_t kTailMax = 512;  // manifolds per colour the tail should take: one trip of the workgroup
    uint32_t big = 0;
    if (h.valid) {
        big = h.n_colors;
        while (big > 0 && h.color_count[big - 1] <= kTailMax) --big;
        if (h.n_colors - big < 2) big = h.n_colors;  // a tail of one colour is just a slower launch
    }
    for (uint32_t it = 0; it < w->cfg.solver_iterations; ++it) {
        for (uint32_t col = 0; col < big; ++col) {
            PHYS_PROF(w, PHYS_STAGE_SOLVE);
            if (diag)
                hipLaunchKernelGGL(k_solve_color<true>, grid_for_count(h.color_count[col]), tb, 0, s, w->counters.p, col, cap, sp.friction,
                                   w->row_a.p, w->row_b.p, w->row_count.p, w->row_normal.p, w->row_data.p, w->row_acc.p,
                                   w->inv_inertia_diag.p, istride, w->vel.p);
            else
                hipLaunchKernelGGL(k_solve_color<false>, grid_for_count(h.color_count[col]), tb, 0, s, w->counters.p, col, cap, sp.friction,
                                   w->row_a.p, w->row_b.p, w->row_count.p, w->row_normal.p, w->row_data.p, w->row_acc.p,
                                   w->inv_inertia.p, 1u, w->vel.p);
        }
        PHYS_PROF(w, PHYS_STAGE_SOLVE_TAIL);
        if (diag)
            hipLaunchKernelGGL(k_solve_tail<true>, dim3(1), dim3(kTailThreads), 0, s, w->counters.p, big, cap, sp.friction,
                               w->row_a.p, w->row_b.p, w->row_count.p, w->row_normal.p, w->row_data.p, w->row_acc.p,
                               w->inv_inertia_diag.p, istride, w->vel.p);
        else
            hipLaunchKernelGGL(k_solve_tail<false>, dim3(1), dim3(kTailThreads), 0, s, w->counters.p, big, cap, sp.friction,
                               w->row_a.p, w->row_b.p, w->row_count.p, w->row_normal.p, w->row_data.p, w->row_acc.p,
                               w->inv_inertia.p, 1u, w->vel.p);
    }
}

}  // namespace phys
