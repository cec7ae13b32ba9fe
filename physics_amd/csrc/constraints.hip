// constraints.hip — the reference's constrained-dynamics path for gfx950:
//   ConstraintSolver::solve_constraints   reference src/physics/constraints.rs:67-169
//   FixToPointConstraint::calculate       src/physics/constraints/fixed_position_constraint.rs:13-27
//   FixedOrientationConstraint::calculate src/physics/constraints/fixed_orientation_constraint.rs:15-30
//   SparseMatrix::{multiply,tr_multiply}_vector  src/physics/sparse_matrix.rs:25-50
//   solve_conjugate_gradient              src/physics/sle_solver.rs:21-46
//   the quirk-Q3 scatter                  src/physics.rs:45-51
//
// Both reference constraint kinds have J blocks that are 3x3 identity selectors (rows r -> column
// 6*body + 3*kind + r), J-dot = 0, ks = 10, kd = 1, and the constraint-space mass is 1/mass for all six
// columns (quirk Q4). So J*v gathers, J^T*v scatters-with-accumulation, and A = J W J^T applied to p is
//   t[col] = W[col] * (sum of p over the rows selecting col, IN CONSTRAINT ORDER)  ;  (A p)[row] = t[col(row)]
// which is exactly what the reference's block loops compute (the other block entries multiply by 0).
// The whole solve - right-hand side, warm start, CG loop, convergence test, scatter - is ONE launch of
// one workgroup: every reduction follows nalgebra's operation order (8-accumulator dot, left-to-right
// column sums), so lambda is bit-identical to the CPU oracle; a tree reduction would be faster for huge
// systems but would change the iteration at which CG stops.
#include <algorithm>
#include <vector>

#include "kernels.hpp"

namespace phys {

struct CgParams {
    uint32_t n_constraints;
    uint32_t n_cols;  // distinct selected columns
    uint32_t max_iterations;
    float max_error, min_error;
    // apply_gravity (physics.rs:87-94) still pending for this update: Q = accumulator + gravity is formed here for the
    // constrained bodies (the same addition apply_gravity makes), instead of a pass over all N bodies that materialises
    // gravity in the accumulators only for this kernel to read a handful of them back
    uint32_t gravity_pending;
    float g_force[3], g_torque[3];
};

constexpr int kCgThreads = 1024;

// nalgebra dotc on Dyn vectors: 8 strided accumulators, then (acc0+acc4) + (acc1+acc5) + ..., then the tail.
// Lanes 0..7 of the calling wave each run one accumulator chain; the result is returned to every lane.
// The chains are n/8 DEPENDENT additions each and cannot be shortened without changing the sum (and with it the iteration
// at which CG stops); what can be taken out of them is everything else: the products are made by the whole workgroup,
// kDotChunk at a time, into LDS (rounded once, as `acc += a * b` rounds them without contraction), and a chain then reads
// eight of them per trip from LDS instead of waiting for two global loads per addition (24k rows: 3072 links per chain).
// The stage is CHAIN-major: chain c's products sit side by side (element 8k + c of the chunk at prod[c * kDotStride + k]),
// so a chain reads sixteen links with four ds_read_b128 and adds them while the next sixteen are on their way; the
// stride is 4 words off a multiple of the 32 banks, so the eight chains' reads fall on disjoint banks. (Element-major,
// one ds_read_b32 per link: 62 cycles per link, 243 us per CG iteration at 24k rows; this layout: see DESIGN.md.)
constexpr uint32_t kDotChunk = 8192;                 // products staged per trip
constexpr uint32_t kDotStride = kDotChunk / 8 + 4;   // words between two chains' runs (1028)
constexpr uint32_t kDotLdsWords = 8 * kDotStride;    // 32.1 KiB
__device__ __forceinline__ float dyn_dot8(const float* __restrict__ a, const float* __restrict__ b, uint32_t n,
                                          float* __restrict__ lds8, float* __restrict__ prod) {
    const uint32_t lane = threadIdx.x;
    const uint32_t full = n & ~7u;
    float acc = 0.0f;
    for (uint32_t base = 0; base < full; base += kDotChunk) {
        const uint32_t m = full - base < kDotChunk ? full - base : kDotChunk;  // a multiple of 8
        for (uint32_t i = threadIdx.x; i < m; i += kCgThreads) prod[(i & 7u) * kDotStride + (i >> 3)] = a[base + i] * b[base + i];
        __syncthreads();
        if (lane < 8) {
            const float* run = prod + lane * kDotStride;  // this chain's m / 8 links, in order
            const uint32_t links = m >> 3;
            uint32_t k = 0;
            for (; k + 16 <= links; k += 16) {
                const float4 q0 = *reinterpret_cast<const float4*>(run + k), q1 = *reinterpret_cast<const float4*>(run + k + 4),
                             q2 = *reinterpret_cast<const float4*>(run + k + 8), q3 = *reinterpret_cast<const float4*>(run + k + 12);
                acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w; acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
                acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w; acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
            }
            for (; k < links; ++k) acc += run[k];
        }
        __syncthreads();
    }
    if (lane < 8) lds8[lane] = acc;
    __syncthreads();
    float res = 0.0f;
    res += lds8[0] + lds8[4];
    res += lds8[1] + lds8[5];
    res += lds8[2] + lds8[6];
    res += lds8[3] + lds8[7];
    for (uint32_t k = n & ~7u; k < n; ++k) res += a[k] * b[k];
    __syncthreads();
    return res;
}

// nalgebra amax = fold of f32::max over |e|: a NaN operand is ignored. `e > m ? e : m` keeps m when e is NaN.
__device__ __forceinline__ float block_amax(const float* __restrict__ v, uint32_t n, float* __restrict__ lds16) {
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) { const float e = det_absf(v[i]); m = e > m ? e : m; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) lds16[threadIdx.x >> 6] = m;
    __syncthreads();
    float r = 0.0f;
    for (int k = 0; k < kCgThreads / 64; ++k) { const float o = lds16[k]; r = o > r ? o : r; }
    __syncthreads();
    return r;
}

// UnitQuaternion::euler_angles (nalgebra; Slabaugh) with the deterministic trig set
__device__ __forceinline__ void quat_euler_angles(quat q, float out[3]) {
    m33 R;
    quat_to_m33(q, &R);
    const float r20 = R.m[6];
    if (det_absf(r20) < 1.0f) {
        const float pitch = -det_asinf(r20);
        const float theta_cos = det_cosf(pitch);
        out[0] = det_atan2f(R.m[7] / theta_cos, R.m[8] / theta_cos);
        out[1] = pitch;
        out[2] = det_atan2f(R.m[3] / theta_cos, R.m[0] / theta_cos);
    } else if (r20 <= -1.0f) {
        out[0] = det_atan2f(R.m[1], R.m[2]);
        out[1] = 1.57079632679489661923f;
        out[2] = 0.0f;
    } else {
        out[0] = det_atan2f(-R.m[1], -R.m[2]);
        out[1] = -1.57079632679489661923f;
        out[2] = 0.0f;
    }
}

// (A v)[row] for every row: t[col] = W[col] * ordered sum, then gather. Two barriers.
__device__ __forceinline__ void apply_A(const float* __restrict__ v, float* __restrict__ out, float* __restrict__ tcol,
                                        const CgParams& cp, const uint32_t* __restrict__ col_ptr,
                                        const uint32_t* __restrict__ col_rows, const float* __restrict__ col_w,
                                        const uint32_t* __restrict__ row_cidx) {
    for (uint32_t u = threadIdx.x; u < cp.n_cols; u += kCgThreads) {
        float s = 0.0f;  // tr_multiply_vector: res[col] += result, block after block (sparse_matrix.rs:39-50)
        for (uint32_t k = col_ptr[u]; k < col_ptr[u + 1]; ++k) s += v[col_rows[k]];
        tcol[u] = s * col_w[u];  // component_mul(inv_masses) (sle_solver.rs:50)
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < 3 * cp.n_constraints; r += kCgThreads) out[r] = tcol[row_cidx[r]];
    __syncthreads();
}

__global__ __launch_bounds__(kCgThreads) void k_constraint_solve(
    CgParams cp, const Constraint* __restrict__ cons, const uint32_t* __restrict__ col_ptr,
    const uint32_t* __restrict__ col_rows, const uint32_t* __restrict__ col_id, const uint32_t* __restrict__ row_cidx,
    float* __restrict__ col_w, const float* __restrict__ pos, const float* __restrict__ rot,
    const float* __restrict__ vel, const float* __restrict__ force, const float* __restrict__ torque, float* __restrict__ x_prev, float* __restrict__ x, float* __restrict__ r,
    float* __restrict__ p, float* __restrict__ ap, float* __restrict__ rhs, float* __restrict__ tcol,
    uint32_t* __restrict__ status /* [0] converged, [1] iterations, [2] previous_solution.is_some() */,
    float* __restrict__ jl /* 6: J^T lambda of entity 0, added to its accumulators by the step kernel when status[0] */) {
    __shared__ float lds8[8];
    __shared__ __attribute__((aligned(16))) float prod[kDotLdsWords];
    __shared__ float lds16[kCgThreads / 64];
    __shared__ int s_done;
    const uint32_t n = 3 * cp.n_constraints;

    // constraint-space inverse masses of the selected columns: 1/mass for all six (quirk Q4, constraints.rs:72-78)
    for (uint32_t u = threadIdx.x; u < cp.n_cols; u += kCgThreads) col_w[u] = vel[8 * (size_t)(col_id[u] / 6) + 3];
    // rhs = -Jdot*qdot - J(Q o W) - ks o C - kd o (J qdot)   (constraints.rs:153-160), Jdot = 0
    for (uint32_t c = threadIdx.x; c < cp.n_constraints; c += kCgThreads) {
        const Constraint con = cons[c];
        const uint32_t b = con.body;
        float cv[3], qd[3], Q[3];
        if (con.kind == 0u) {
            const v3 x0 = ld3(pos, b);
            cv[0] = x0.x - con.target[0]; cv[1] = x0.y - con.target[1]; cv[2] = x0.z - con.target[2];
            const v3 v = ld_vel(vel, b).v;
            v3 F = ld3(force, b);
            if (cp.gravity_pending) F = v3_add(F, v3_make(cp.g_force[0], cp.g_force[1], cp.g_force[2]));  // apply_force_at_offset: force += F
            qd[0] = v.x; qd[1] = v.y; qd[2] = v.z;
            Q[0] = F.x; Q[1] = F.y; Q[2] = F.z;
        } else {
            const float4 qq = reinterpret_cast<const float4*>(rot)[b];
            quat q; q.i = qq.x; q.j = qq.y; q.k = qq.z; q.w = qq.w;
            float rpy[3];
            quat_euler_angles(q, rpy);
            cv[0] = rpy[0] - con.target[0]; cv[1] = rpy[1] - con.target[1]; cv[2] = rpy[2] - con.target[2];
            const v3 wv = ld_vel(vel, b).w;
            v3 T = ld3(torque, b);
            if (cp.gravity_pending) T = v3_add(T, v3_make(cp.g_torque[0], cp.g_torque[1], cp.g_torque[2]));  // torque += offset x F
            qd[0] = wv.x; qd[1] = wv.y; qd[2] = wv.z;
            Q[0] = T.x; Q[1] = T.y; Q[2] = T.z;
        }
        const float W = vel[8 * (size_t)b + 3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float kd = 1.0f * qd[k];   // k_d o c_dot, KD = 1
            const float ks = 10.0f * cv[k];  // k_s o c,     KS = 10
            rhs[3 * c + k] = ((-0.0f - Q[k] * W) - ks) - kd;
        }
    }
    // x = previous_solution or zeros (sle_solver.rs:22-26)
    const bool warm = status[2] != 0u;
    for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) x[i] = warm ? x_prev[i] : 0.0f;
    __syncthreads();
    // r = rhs - A x; p = r (:28-29)
    apply_A(x, ap, tcol, cp, col_ptr, col_rows, col_w, row_cidx);
    for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) { const float ri = rhs[i] - ap[i]; r[i] = ri; p[i] = ri; }
    __syncthreads();
    const float rhs_amax = block_amax(rhs, n, lds16);
    const float b0 = rhs_amax * cp.max_error;
    const float bound = b0 > cp.min_error ? b0 : cp.min_error;
    uint32_t it = 0;
    bool converged = false;
    for (; it < cp.max_iterations; ++it) {
        apply_A(p, ap, tcol, cp, col_ptr, col_rows, col_w, row_cidx);  // :32
        const float rk = dyn_dot8(r, r, n, lds8, prod);                     // :33
        const float alpha = rk / dyn_dot8(p, ap, n, lds8, prod);            // :34
        for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) {
            x[i] = x[i] + alpha * p[i];   // :35
            r[i] = r[i] - alpha * ap[i];  // :37
        }
        __syncthreads();
        const float ramax = block_amax(r, n, lds16);
        if (threadIdx.x == 0) s_done = ramax < bound ? 1 : 0;  // :38
        __syncthreads();
        if (s_done) { converged = true; ++it; break; }
        const float beta = dyn_dot8(r, r, n, lds8, prod) / rk;  // :42
        for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) p[i] = r[i] + beta * p[i];  // :43
        __syncthreads();
    }
    if (converged) {
        // previous_solution = Some(solution) (physics.rs:46)
        for (uint32_t i = threadIdx.x; i < n; i += kCgThreads) x_prev[i] = x[i];
        // J^T lambda restricted to body 0 for the quirk-Q3 scatter: entities[0] only (physics.rs:47-50). The addition
        // itself is made by the step kernel, behind gravity as in the reference (force = (force + gravity) + J^T lambda)
        if (threadIdx.x < 6) {
            float s = 0.0f;
            for (uint32_t u = 0; u < cp.n_cols; ++u)
                if (col_id[u] == threadIdx.x)
                    for (uint32_t k = col_ptr[u]; k < col_ptr[u + 1]; ++k) s += x[col_rows[k]];
            jl[threadIdx.x] = s;
        }
    }
    if (threadIdx.x == 0) {
        status[0] = converged ? 1u : 0u;
        status[1] = it;
        if (converged) status[2] = 1u;
    }
}

// rebuild the device-side constraint tables when the constraint list changed
int32_t constraints_alloc(phys_world* w) {
    if (!w->constraints_dirty && w->d_constraints.p) return PHYS_OK;
    const size_t C = w->constraints.size();
    const size_t n = 3 * C;
    // distinct selected columns, each with its selecting rows in constraint order
    std::vector<std::pair<uint32_t, uint32_t>> sel(n);  // (column, row)
    for (size_t c = 0; c < C; ++c)
        for (uint32_t k = 0; k < 3; ++k) sel[3 * c + k] = {6u * w->constraints[c].body + 3u * w->constraints[c].kind + k, (uint32_t)(3 * c + k)};
    std::vector<std::pair<uint32_t, uint32_t>> sorted = sel;
    std::stable_sort(sorted.begin(), sorted.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<uint32_t> col_id, col_ptr, col_rows(n), row_cidx(n);
    for (size_t k = 0; k < n; ++k) {
        if (k == 0 || sorted[k].first != sorted[k - 1].first) { col_id.push_back(sorted[k].first); col_ptr.push_back((uint32_t)k); }
        col_rows[k] = sorted[k].second;
        row_cidx[sorted[k].second] = (uint32_t)col_id.size() - 1;
    }
    col_ptr.push_back((uint32_t)n);
    const size_t U = col_id.size();
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    PHYS_HIP_TRY(w->d_constraints.resize(C));
    PHYS_HIP_TRY(w->cg_cols.resize(U + (U + 1) + n + n));
    PHYS_HIP_TRY(w->cg_x.resize(2 * n));  // x_prev | x
    PHYS_HIP_TRY(w->cg_r.resize(n)); PHYS_HIP_TRY(w->cg_p.resize(n)); PHYS_HIP_TRY(w->cg_ap.resize(n));
    PHYS_HIP_TRY(w->cg_rhs.resize(n)); PHYS_HIP_TRY(w->cg_scratch.resize(2 * U + 1));  // tcol | col_w
    PHYS_HIP_TRY(w->cg_status.resize(4));
    PHYS_HIP_TRY(w->cg_jl.resize(8));
    uint32_t* base = w->cg_cols.p;
    PHYS_HIP_TRY(hipMemcpy(w->d_constraints.p, w->constraints.data(), C * sizeof(Constraint), hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(base, col_id.data(), U * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(base + U, col_ptr.data(), (U + 1) * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(base + 2 * U + 1, col_rows.data(), n * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(base + 2 * U + 1 + n, row_cidx.data(), n * 4, hipMemcpyHostToDevice));
    // a changed constraint list resets the warm start (the reference would panic on the shape mismatch)
    PHYS_HIP_TRY(hipMemset(w->cg_status.p, 0, 16));
    w->cg_n_cols = (uint32_t)U;
    w->constraints_dirty = false;
    return PHYS_OK;
}

void launch_constraint_phase(phys_world* w, bool gravity_pending) {
    const uint32_t C = (uint32_t)w->constraints.size();
    if (C == 0) return;
    CgParams cp;
    cp.gravity_pending = gravity_pending ? 1u : 0u;
    {
        const float* F = w->cfg.gravity_force;
        const float* o = w->cfg.gravity_offset;
        for (int k = 0; k < 3; ++k) cp.g_force[k] = F[k];
        cp.g_torque[0] = o[1] * F[2] - o[2] * F[1];  // offset.cross(&force), rigid_body.rs:60 (as integrate.hip make_params)
        cp.g_torque[1] = o[2] * F[0] - o[0] * F[2];
        cp.g_torque[2] = o[0] * F[1] - o[1] * F[0];
    }
    cp.n_constraints = C;
    cp.n_cols = w->cg_n_cols;
    cp.max_iterations = w->cfg.cg_max_iterations;
    cp.max_error = w->cfg.cg_max_error;
    cp.min_error = w->cfg.cg_min_error;
    const uint32_t U = cp.n_cols, n = 3 * C;
    uint32_t* base = w->cg_cols.p;
    PHYS_PROF(w, PHYS_STAGE_CONSTRAINTS);
    hipLaunchKernelGGL(k_constraint_solve, dim3(1), dim3(kCgThreads), 0, w->stream, cp, w->d_constraints.p, base + U,
                       base + 2 * U + 1, base, base + 2 * U + 1 + n, w->cg_scratch.p + U, w->pos.p, w->rot.p, w->vel.p,
                       w->force.p, w->torque.p, w->cg_x.p, w->cg_x.p + n, w->cg_r.p, w->cg_p.p,
                       w->cg_ap.p, w->cg_rhs.p, w->cg_scratch.p, w->cg_status.p, w->cg_jl.p);
}

// ---- the general block-sparse product (sparse_matrix.rs:16-50) -----------------------------------------------------
// One lane per OUTPUT entry. entry_ptr / entry_blk / entry_loc list, for every output entry, the (block, local row or
// column) pairs that add to it, in add_block order - so overlapping blocks accumulate in the reference's order - and the
// inner product of a block row (or column) runs left to right: `result = t` for the first term, `t + result` after
// (nalgebra's gemv is an axpy per column, the way the oracle restates it).
struct BlockDesc { uint32_t i, j, il, jl, off; };
__global__ __launch_bounds__(256) void k_block_spmv(uint32_t n_out, int transpose, const BlockDesc* __restrict__ blk,
                                                    const uint32_t* __restrict__ entry_ptr, const uint32_t* __restrict__ entry_blk,
                                                    const uint32_t* __restrict__ entry_loc, const float* __restrict__ data,
                                                    const float* __restrict__ vec, float* __restrict__ out) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    float res = 0.0f;  // OVector::repeat(n, 0.0)
    for (uint32_t e = entry_ptr[o]; e < entry_ptr[o + 1]; ++e) {
        const BlockDesc b = blk[entry_blk[e]];
        const uint32_t l = entry_loc[e];
        float result = 0.0f;  // OVector::zeros(1)
        if (!transpose) {
            for (uint32_t c = 0; c < b.jl; ++c) {  // row(l) . vector.rows(j, j_length)
                const float t = data[b.off + l * b.jl + c] * vec[b.j + c];
                result = c == 0 ? t : t + result;
            }
        } else {
            for (uint32_t r = 0; r < b.il; ++r) {  // column(l)^T . vector.rows(i, i_length)
                const float t = data[b.off + r * b.jl + l] * vec[b.i + r];
                result = r == 0 ? t : t + result;
            }
        }
        res += result;  // res[(block.i + row, 0)] += result[(0, 0)]
    }
    out[o] = res;
}

}  // namespace phys

extern "C" int32_t phys_block_spmv(int32_t device, uint64_t nrows, uint64_t ncols, uint64_t nblocks, const uint64_t* block_desc,
                                   const float* data, const float* vec, uint64_t vec_len, int32_t transpose, float* out) {
    using namespace phys;
    auto bad = [](const char* msg) { set_error(msg); return (int32_t)PHYS_ERR_INVALID_ARG; };
    if ((nblocks && (!block_desc || !data)) || !vec || !out) return bad("null argument");
    if (nrows >= 0x7FFFFFFFull || ncols >= 0x7FFFFFFFull || nblocks >= 0x7FFFFFFFull) return bad("matrix too large (u32 indices)");
    if (vec_len != (transpose ? nrows : ncols)) return bad("vector length does not match the matrix (reference: assert_eq, sparse_matrix.rs:26,40)");
    const uint64_t n_out = transpose ? ncols : nrows;
    std::vector<BlockDesc> blk(nblocks);
    std::vector<uint32_t> entry_ptr(n_out + 1, 0u);
    uint64_t off = 0;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t i = block_desc[4 * b], j = block_desc[4 * b + 1], il = block_desc[4 * b + 2], jl = block_desc[4 * b + 3];
        if (i + il > nrows || j + jl > ncols) return bad("a block reaches outside the matrix (reference: index panic)");
        if (off + il * jl >= 0xFFFFFFFFull) return bad("block data too large (u32 offsets)");
        blk[b] = BlockDesc{(uint32_t)i, (uint32_t)j, (uint32_t)il, (uint32_t)jl, (uint32_t)off};
        off += il * jl;
        const uint64_t first = transpose ? j : i, count = transpose ? jl : il;
        for (uint64_t k = 0; k < count; ++k) entry_ptr[first + k + 1] += 1u;
    }
    for (uint64_t o = 0; o < n_out; ++o) entry_ptr[o + 1] += entry_ptr[o];
    const uint32_t n_entries = entry_ptr[n_out];
    std::vector<uint32_t> entry_blk(n_entries ? n_entries : 1), entry_loc(n_entries ? n_entries : 1), cursor(entry_ptr.begin(), entry_ptr.end() - 1);
    for (uint64_t b = 0; b < nblocks; ++b) {  // blocks in list order: every entry's list comes out in that order
        const uint32_t first = transpose ? blk[b].j : blk[b].i, count = transpose ? blk[b].jl : blk[b].il;
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t at = cursor[first + k]++;
            entry_blk[at] = (uint32_t)b;
            entry_loc[at] = k;
        }
    }
    if (n_out == 0) return PHYS_OK;
    int count_dev = 0;
    if (hipGetDeviceCount(&count_dev) != hipSuccess || device < 0 || device >= count_dev) {
        set_error("no HIP device visible: libphysics_hip has no CPU fallback");
        return PHYS_ERR_NO_DEVICE;
    }
    PHYS_HIP_TRY(hipSetDevice(device));
    DevBuf<uint8_t> d_blk, d_data, d_vec, d_out;
    DevBuf<uint32_t> d_ptr, d_eb, d_el;
    struct Free { DevBuf<uint8_t>*a, *b, *c, *d; DevBuf<uint32_t>*e, *f, *g; ~Free() { a->free(); b->free(); c->free(); d->free(); e->free(); f->free(); g->free(); } }
        cleanup{&d_blk, &d_data, &d_vec, &d_out, &d_ptr, &d_eb, &d_el};
    PHYS_HIP_TRY(d_blk.resize(std::max<size_t>(blk.size(), 1) * sizeof(BlockDesc)));
    PHYS_HIP_TRY(d_data.resize(std::max<size_t>(off, 1) * 4)); PHYS_HIP_TRY(d_vec.resize(std::max<size_t>(vec_len, 1) * 4)); PHYS_HIP_TRY(d_out.resize(n_out * 4));
    PHYS_HIP_TRY(d_ptr.resize(n_out + 1)); PHYS_HIP_TRY(d_eb.resize(entry_blk.size())); PHYS_HIP_TRY(d_el.resize(entry_loc.size()));
    if (!blk.empty()) PHYS_HIP_TRY(hipMemcpy(d_blk.p, blk.data(), blk.size() * sizeof(BlockDesc), hipMemcpyHostToDevice));
    if (off) PHYS_HIP_TRY(hipMemcpy(d_data.p, data, off * 4, hipMemcpyHostToDevice));
    if (vec_len) PHYS_HIP_TRY(hipMemcpy(d_vec.p, vec, vec_len * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(d_ptr.p, entry_ptr.data(), (n_out + 1) * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(d_eb.p, entry_blk.data(), entry_blk.size() * 4, hipMemcpyHostToDevice));
    PHYS_HIP_TRY(hipMemcpy(d_el.p, entry_loc.data(), entry_loc.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_block_spmv, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, nullptr, (uint32_t)n_out, transpose,
                       reinterpret_cast<const BlockDesc*>(d_blk.p), d_ptr.p, d_eb.p, d_el.p, reinterpret_cast<const float*>(d_data.p),
                       reinterpret_cast<const float*>(d_vec.p), reinterpret_cast<float*>(d_out.p));
    PHYS_HIP_TRY(hipGetLastError());
    PHYS_HIP_TRY(hipMemcpy(out, d_out.p, n_out * 4, hipMemcpyDeviceToHost));
    return PHYS_OK;
}
