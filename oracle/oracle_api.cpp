// oracle_api.cpp — C entry points of liboracle.so. CPU ORACLE: test infrastructure, NOT product
// code. Loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// (through oracle/binding.py). Mirrors the argument conventions of include/physics_hip.h so the
// same seeded inputs can be fed to both.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

#include "../include/physics_hip.h"
#include "../include/spec/det_math.h"
#include "collide_oracle.hpp"
#include "ref_physics.hpp"

using namespace oracle;

struct oracle_world {
    PhysicsState state;
    CollisionWorld col;
    phys_config cfg;
    uint64_t steps = 0;
};

static thread_local std::string g_err;
static int32_t fail(int32_t code, const char* msg) { g_err = msg; return code; }

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int32_t oracle_create(const phys_config* cfg, int32_t trig /*0 libm, 1 det*/, oracle_world** out) {
    if (!cfg || !out) return fail(PHYS_ERR_INVALID_ARG, "null argument");
    oracle_world* w = new oracle_world();
    w->cfg = *cfg;
    for (int k = 0; k < 3; ++k) {
        w->state.gravity_force[k] = cfg->gravity_force[k];
        w->state.gravity_offset[k] = cfg->gravity_offset[k];
    }
    w->state.cg.max_iterations = cfg->cg_max_iterations;
    w->state.cg.max_error = cfg->cg_max_error;
    w->state.cg.min_error = cfg->cg_min_error;
    w->state.trig = trig ? Trig::Det : Trig::Libm;
    w->state.exact_rotation = (cfg->flags & PHYS_FLAG_EXACT_ROTATION) != 0;
    w->col.configure(*cfg);
    *out = w;
    return PHYS_OK;
}

int32_t oracle_destroy(oracle_world* w) { delete w; return PHYS_OK; }

// OpenMP variant of the CPU baseline: `threads` > 1 runs the loops of the collision stages
// whose iterations are independent (narrow phase per pair, row preparation, the manifolds of one colour) on that
// many threads. The results do not depend on it (tests/test_oracle_golden.py).
int32_t oracle_set_threads(oracle_world* w, int32_t threads) {
    if (!w || threads < 1) return fail(PHYS_ERR_INVALID_ARG, "threads must be >= 1");
    w->col.threads = threads;
    return PHYS_OK;
}

int32_t oracle_set_bodies(oracle_world* w, uint64_t n, const float* pos, const float* rot, const float* lin,
                          const float* ang, const float* mass, const float* inertia, const uint32_t* shape_type,
                          const float* half_extent) {
    if (!w || (n && !pos)) return fail(PHYS_ERR_INVALID_ARG, "null argument");
    w->state.entities.clear();
    w->state.entities.reserve(n);
    w->col.shape_type.assign(n, PHYS_SHAPE_NONE);
    w->col.half_extent.assign(3 * n, 0.0f);
    for (uint64_t i = 0; i < n; ++i) {
        RigidBody b = RigidBody::new_(i);
        for (int k = 0; k < 3; ++k) b.position[k] = pos[3 * i + k];
        if (rot) for (int k = 0; k < 4; ++k) b.rotation[k] = rot[4 * i + k];
        if (lin) for (int k = 0; k < 3; ++k) b.lin_velocity[k] = lin[3 * i + k];
        if (ang) for (int k = 0; k < 3; ++k) b.angular_velocity[k] = ang[3 * i + k];
        if (mass) b.mass = mass[i];
        if (inertia) for (int k = 0; k < 9; ++k) b.inertia_tensor[k] = inertia[9 * i + k];
        if (shape_type) w->col.shape_type[i] = shape_type[i];
        if (half_extent) for (int k = 0; k < 3; ++k) w->col.half_extent[3 * i + k] = half_extent[3 * i + k];
        w->state.entities.push_back(b);
    }
    w->state.previous_solution.reset();
    w->col.color_cache.clear();
    w->col.warm_cache.clear();
    w->col.color_epoch = 0;
    return PHYS_OK;
}

static int32_t add_constraint(oracle_world* w, Constraint::Kind kind, uint64_t body, const float t[3]) {
    if (!w || !t) return fail(PHYS_ERR_INVALID_ARG, "null argument");
    if (body >= w->state.entities.size()) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    Constraint c{kind, (size_t)body, {t[0], t[1], t[2]}};
    w->state.constraints.push_back(c);
    return PHYS_OK;
}
int32_t oracle_add_constraint_fix_point(oracle_world* w, uint64_t body, const float t[3]) {
    return add_constraint(w, Constraint::FixedPosition, body, t);
}
int32_t oracle_add_constraint_fix_orientation(oracle_world* w, uint64_t body, const float t[3]) {
    return add_constraint(w, Constraint::FixedOrientation, body, t);
}
int32_t oracle_clear_constraints(oracle_world* w) {
    w->state.constraints.clear();
    w->state.previous_solution.reset();
    return PHYS_OK;
}

int32_t oracle_apply_force_centre_of_gravity(oracle_world* w, uint64_t body, const float f[3]) {
    if (body >= w->state.entities.size()) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    w->state.entities[body].apply_force_centre_of_gravity(f);
    return PHYS_OK;
}
int32_t oracle_apply_force_at_position(oracle_world* w, uint64_t body, const float f[3], const float p[3]) {
    if (body >= w->state.entities.size()) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    w->state.entities[body].apply_force_at_position(f, p);
    return PHYS_OK;
}
int32_t oracle_apply_force_at_offset(oracle_world* w, uint64_t body, const float f[3], const float o[3]) {
    if (body >= w->state.entities.size()) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    w->state.entities[body].apply_force_at_offset(f, o);
    return PHYS_OK;
}

int32_t oracle_apply_gravity(oracle_world* w) { w->state.apply_gravity(); return PHYS_OK; }

int32_t oracle_step(oracle_world* w, uint64_t dt_nanos) {
    if (!w->state.step(dt_nanos)) return fail(PHYS_ERR_SINGULAR_INERTIA, "singular inertia tensor");
    return PHYS_OK;
}

int32_t oracle_update(oracle_world* w, uint64_t dt_nanos) {
    PhysicsState& s = w->state;
    if (s.entities.empty()) return fail(PHYS_ERR_NO_BODIES, "update with no bodies");
    // (a zero-length step has no contact problem to solve - the bias terms divide by dt - and is a plain update)
    if (!(w->cfg.flags & PHYS_FLAG_COLLISIONS) || !(duration_as_secs_f32(dt_nanos) > 0.0f)) {
        if (!s.update(dt_nanos)) return fail(PHYS_ERR_SINGULAR_INERTIA, "singular inertia tensor");
    } else {
        // collision mode: physics.rs:41-55 with the contact stages between the velocity and the
        // position half of RigidBody::step (DESIGN.md "step order")
        const float dt = duration_as_secs_f32(dt_nanos);
        s.apply_gravity();
        s.constraint_phase();
        for (RigidBody& b : s.entities)
            if (!b.step_velocity(dt)) return fail(PHYS_ERR_SINGULAR_INERTIA, "singular inertia tensor");
        w->col.collide_and_solve(s.entities, dt);
        for (RigidBody& b : s.entities) b.step_position(dt, s.trig, s.exact_rotation);
    }
    w->steps++;
    return PHYS_OK;
}

int32_t oracle_get_transforms(oracle_world* w, float* pos, float* rot) {
    for (size_t i = 0; i < w->state.entities.size(); ++i) {
        const RigidBody& b = w->state.entities[i];
        if (pos) for (int k = 0; k < 3; ++k) pos[3 * i + k] = b.position[k];
        if (rot) for (int k = 0; k < 4; ++k) rot[4 * i + k] = b.rotation[k];
    }
    return PHYS_OK;
}
int32_t oracle_get_velocities(oracle_world* w, float* lin, float* ang) {
    for (size_t i = 0; i < w->state.entities.size(); ++i) {
        const RigidBody& b = w->state.entities[i];
        if (lin) for (int k = 0; k < 3; ++k) lin[3 * i + k] = b.lin_velocity[k];
        if (ang) for (int k = 0; k < 3; ++k) ang[3 * i + k] = b.angular_velocity[k];
    }
    return PHYS_OK;
}
int32_t oracle_get_forces(oracle_world* w, float* f, float* t) {
    for (size_t i = 0; i < w->state.entities.size(); ++i) {
        const RigidBody& b = w->state.entities[i];
        if (f) for (int k = 0; k < 3; ++k) f[3 * i + k] = b.force[k];
        if (t) for (int k = 0; k < 3; ++k) t[3 * i + k] = b.torque[k];
    }
    return PHYS_OK;
}

// Instance::to_raw (graphics.rs:13-21): Matrix4::new_translation(p) * rotation.to_homogeneous(),
// column-major 16 floats
int32_t oracle_get_instance_matrices(oracle_world* w, float* out) {
    for (size_t i = 0; i < w->state.entities.size(); ++i) {
        const RigidBody& b = w->state.entities[i];
        const float qi = b.rotation[0], qj = b.rotation[1], qk = b.rotation[2], qw = b.rotation[3];
        const float ww = qw * qw, ii = qi * qi, jj = qj * qj, kk = qk * qk;
        const float ij = qi * qj * 2.0f, wk = qw * qk * 2.0f, wj = qw * qj * 2.0f;
        const float ik = qi * qk * 2.0f, jk = qj * qk * 2.0f, wi = qw * qi * 2.0f;
        const float R[9] = {ww + ii - jj - kk, ij - wk, wj + ik, wk + ij, ww - ii + jj - kk, jk - wi,
                            ik - wj, wi + jk, ww - ii - jj + kk};
        float* m = out + 16 * i;
        // T * R_h: upper 3x3 = R, last column = translation (T has identity rotation part so the
        // product's entries are exact copies plus additions of zero products)
        for (int c = 0; c < 3; ++c) {
            for (int r = 0; r < 3; ++r) m[4 * c + r] = R[3 * r + c];
            m[4 * c + 3] = 0.0f;
        }
        m[12] = b.position[0]; m[13] = b.position[1]; m[14] = b.position[2]; m[15] = 1.0f;
    }
    return PHYS_OK;
}

int32_t oracle_get_lambda(oracle_world* w, float* out, uint64_t cap, uint64_t* n_rows) {
    if (!w->state.previous_solution) { *n_rows = 0; return PHYS_OK; }
    const auto& l = *w->state.previous_solution;
    *n_rows = l.size();
    if (out) for (size_t k = 0; k < l.size() && k < cap; ++k) out[k] = l[k];
    return PHYS_OK;
}

int32_t oracle_get_stats(oracle_world* w, phys_stats* out) {
    std::memset(out, 0, sizeof(*out));
    out->n_bodies = w->state.entities.size();
    out->n_pairs = w->col.pairs.size();
    out->n_manifolds = w->col.manifolds.size();
    out->n_contacts = w->col.n_contacts;
    out->n_colors = w->col.n_colors;
    out->color_rounds = w->col.color_rounds;
    out->cg_iterations = w->state.last_cg_iterations;
    out->cg_converged = w->state.last_cg_converged ? 1 : 0;
    out->steps = w->steps;
    for (size_t i = 0; i + 5 < w->col.aabb.size(); i += 6)
        if (w->col.shape_type[i / 6] != PHYS_SHAPE_NONE)
            for (int k = 0; k < 3; ++k) out->max_extent = std::max(out->max_extent, w->col.aabb[i + 3 + k] - w->col.aabb[i + k]);
    for (const auto& m : w->col.manifolds) out->n_ground_manifolds += (m.b == PHYS_GROUND_ID) ? 1u : 0u;
    return PHYS_OK;
}

// ---- collision-stage read-outs (oracle of SURVEY §8 A10-A12; no reference counterpart)
int32_t oracle_broadphase(oracle_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs) {
    w->col.compute_aabbs(w->state.entities);
    w->col.broadphase_sweep();
    *n_pairs = w->col.pairs.size();
    if (pairs_out)
        for (size_t k = 0; k < w->col.pairs.size() && k < cap; ++k) {
            pairs_out[2 * k] = w->col.pairs[k].first;
            pairs_out[2 * k + 1] = w->col.pairs[k].second;
        }
    return PHYS_OK;
}
// broad phase by the uniform-grid driver (the timed baseline path); oracle_broadphase uses sort-and-sweep
int32_t oracle_broadphase_grid(oracle_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs) {
    w->col.compute_aabbs(w->state.entities);
    w->col.broadphase_grid();
    *n_pairs = w->col.pairs.size();
    if (pairs_out)
        for (size_t k = 0; k < w->col.pairs.size() && k < cap; ++k) {
            pairs_out[2 * k] = w->col.pairs[k].first;
            pairs_out[2 * k + 1] = w->col.pairs[k].second;
        }
    return PHYS_OK;
}
// AABBs + broad phase + narrow phase + colouring of the CURRENT poses, without stepping (for the KATs)
int32_t oracle_collide_now(oracle_world* w) {
    w->col.compute_aabbs(w->state.entities);
    w->col.broadphase_sweep();
    w->col.narrowphase(w->state.entities);
    w->col.color_manifolds(w->state.entities.size(), false);
    return PHYS_OK;
}
int32_t oracle_get_aabbs(oracle_world* w, float* out) {
    w->col.compute_aabbs(w->state.entities);
    std::memcpy(out, w->col.aabb.data(), w->col.aabb.size() * sizeof(float));
    return PHYS_OK;
}
int32_t oracle_get_manifolds(oracle_world* w, uint32_t* ids, uint32_t* counts, float* normals, float* points,
                             uint64_t cap, uint64_t* n) {
    *n = w->col.manifolds.size();
    const auto order = w->col.sorted_manifold_order();
    for (size_t k = 0; k < order.size() && k < cap; ++k) {
        const auto& m = w->col.manifolds[order[k]];
        if (ids) { ids[2 * k] = m.a; ids[2 * k + 1] = m.b; }
        if (counts) counts[k] = (uint32_t)m.count;
        if (normals) { normals[3 * k] = m.normal.x; normals[3 * k + 1] = m.normal.y; normals[3 * k + 2] = m.normal.z; }
        if (points)
            for (int p = 0; p < 4; ++p) {
                points[16 * k + 4 * p + 0] = p < m.count ? m.pt[p].x : 0.0f;
                points[16 * k + 4 * p + 1] = p < m.count ? m.pt[p].y : 0.0f;
                points[16 * k + 4 * p + 2] = p < m.count ? m.pt[p].z : 0.0f;
                points[16 * k + 4 * p + 3] = p < m.count ? m.depth[p] : 0.0f;
            }
    }
    return PHYS_OK;
}
int32_t oracle_get_colors(oracle_world* w, uint32_t* colors_out, uint64_t cap) {
    const auto order = w->col.sorted_manifold_order();
    for (size_t k = 0; k < order.size() && k < cap; ++k) colors_out[k] = w->col.color[order[k]];
    return PHYS_OK;
}

// ---- unit hooks for the golden vectors
// G3: block SpMV (sparse_matrix.rs tests). blocks: per block {i, j, i_len, j_len}; data row-major, concatenated
int32_t oracle_spmv(uint64_t nrows, uint64_t ncols, uint64_t nblocks, const uint64_t* block_desc, const float* data,
                    const float* vec, int32_t transpose, float* out) {
    SparseMatrix m(nrows, ncols);
    size_t off = 0;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const size_t i = block_desc[4 * b], j = block_desc[4 * b + 1], il = block_desc[4 * b + 2], jl = block_desc[4 * b + 3];
        m.add_block(i, j, il, jl, std::vector<float>(data + off, data + off + il * jl));
        off += il * jl;
    }
    if (!transpose) {
        std::vector<float> v(vec, vec + ncols);
        auto r = m.multiply_vector(v);
        std::memcpy(out, r.data(), r.size() * sizeof(float));
    } else {
        std::vector<float> v(vec, vec + nrows);
        auto r = m.tr_multiply_vector(v);
        std::memcpy(out, r.data(), r.size() * sizeof(float));
    }
    return PHYS_OK;
}
float oracle_dyn_dot(const float* a, const float* b, uint64_t n) {
    return dyn_dot(std::vector<float>(a, a + n), std::vector<float>(b, b + n));
}
float oracle_duration_as_secs_f32(uint64_t nanos) { return duration_as_secs_f32(nanos); }
void oracle_quat_from_euler(float r, float p, float y, int32_t trig, float* out) {
    quat_from_euler_angles(r, p, y, trig ? Trig::Det : Trig::Libm, out);
}
void oracle_quat_euler_angles(const float* q, int32_t trig, float* out) {
    quat_euler_angles(q, trig ? Trig::Det : Trig::Libm, out);
}

// det_math probes (tests/test_det_math.py compares them with glibc)
void oracle_det_sincos(const float* x, uint64_t n, float* s, float* c) {
    for (uint64_t k = 0; k < n; ++k) { s[k] = det_sinf(x[k]); c[k] = det_cosf(x[k]); }
}
// the host libm's own float routines (glibc sinf/cosf/asinf/atan2f: what Rust's f32::sin etc. call)
void oracle_libm_sincos(const float* x, uint64_t n, float* s, float* c) {
    for (uint64_t k = 0; k < n; ++k) { s[k] = sinf(x[k]); c[k] = cosf(x[k]); }
}
void oracle_libm_asin(const float* x, uint64_t n, float* out) {
    for (uint64_t k = 0; k < n; ++k) out[k] = asinf(x[k]);
}
void oracle_libm_atan2(const float* y, const float* x, uint64_t n, float* out) {
    for (uint64_t k = 0; k < n; ++k) out[k] = atan2f(y[k], x[k]);
}
void oracle_det_asin(const float* x, uint64_t n, float* out) {
    for (uint64_t k = 0; k < n; ++k) out[k] = det_asinf(x[k]);
}
void oracle_det_atan2(const float* y, const float* x, uint64_t n, float* out) {
    for (uint64_t k = 0; k < n; ++k) out[k] = det_atan2f(y[k], x[k]);
}

// contact_solve.h has three drivers of the row arithmetic (Jacobians made beforehand / on the way / everything
// remade from the contact geometry); the HIP kernels use all of them. Runs `iters` sweeps of each on one manifold described by flat arrays; returns the number of output
// floats whose BITS differ (must be 0). in: normal 3, count, has_b, 4 x (rA 3, rB 3, depth) = 28 floats + invMA,
// invMB, IA 9, IB 9, vA 3, wA 3, vB 3, wB 3, friction; see tests/test_collide_kat.py.
int32_t oracle_solve_drivers_mismatch(const float* in, int32_t count, int32_t has_b, int32_t iters) {
    manifold_t g;
    g.normal = v3_make(in[0], in[1], in[2]);
    g.count = count;
    const float* p = in + 3;
    v3 xA = v3_make(0.25f * p[29], -0.5f * p[30], 0.125f * p[28]), xB = v3_make(p[28], p[29], p[30]);
    for (int k = 0; k < 4; ++k) { g.pt[k] = v3_make(p[7 * k], p[7 * k + 1], p[7 * k + 2]); g.depth[k] = p[7 * k + 6]; }
    const float invMA = p[31], invMB = p[32];
    m33 IA, IB;
    for (int k = 0; k < 9; ++k) { IA.m[k] = p[33 + k]; IB.m[k] = p[42 + k]; }
    const float friction = p[63];
    solve_params_t sp;
    sp.dt = 0.016666668f; sp.baumgarte = 0.2f; sp.slop = 0.01f; sp.friction = friction; sp.max_bias = 3.0f;
    solver_manifold_t a, b;
    solver_prep(&g, has_b, xA, xB, invMA, &IA, invMB, &IB, &sp, &a);
    b = a;
    geo_manifold_t c;
    c.n = a.n; c.t1 = a.t1; c.t2 = a.t2; c.count = count; c.has_b = has_b;
    for (int k = 0; k < 4; ++k) {
        c.pt[k] = g.pt[k];
        c.bias[k] = contact_bias(g.depth[k], &sp);
        c.pn[k] = 0.0f; c.pt0[k] = 0.0f; c.pt1[k] = 0.0f;
    }
    v3 va[4], vb[4], vc[4];
    for (int k = 0; k < 4; ++k) va[k] = vb[k] = vc[k] = v3_make(p[51 + 3 * k], p[52 + 3 * k], p[53 + 3 * k]);
    solver_jac_t J;
    solver_jacobians(&a, invMA, &IA, invMB, &IB, &J);
    for (int it = 0; it < iters; ++it) {
        solve_manifold(&a, &J, friction, &va[0], &va[1], &va[2], &va[3], 0);
        solve_manifold_lazy(&b, friction, invMA, &IA, invMB, &IB, &vb[0], &vb[1], &vb[2], &vb[3], 0);
        solve_manifold_geo(&c, it == 0, 0, friction, xA, invMA, &IA, xB, invMB, &IB, &vc[0], &vc[1], &vc[2], &vc[3]);
    }
    int32_t bad = 0;
    for (int k = 0; k < 4; ++k) bad += std::memcmp(&va[k], &vb[k], sizeof(v3)) != 0;
    for (int k = 0; k < 4; ++k) bad += std::memcmp(&va[k], &vc[k], sizeof(v3)) != 0;
    for (int k = 0; k < 4; ++k) {
        bad += std::memcmp(&a.row[k].pn, &b.row[k].pn, 4) != 0;
        bad += std::memcmp(a.row[k].pt, b.row[k].pt, 8) != 0;
        if (k < count) {
            bad += std::memcmp(&a.row[k].pn, &c.pn[k], 4) != 0;
            bad += std::memcmp(&a.row[k].pt[0], &c.pt0[k], 4) != 0;
            bad += std::memcmp(&a.row[k].pt[1], &c.pt1[k], 4) != 0;
        }
    }
    return bad;
}

}  // extern "C"
