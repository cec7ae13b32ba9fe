// ref_physics.cpp — CPU ORACLE (test infrastructure, NOT product code). See ref_physics.hpp.
// Build with -O2 -ffp-contract=off (rustc never contracts a*b+c into fma).
#include "ref_physics.hpp"

#include <cmath>
#include <limits>

#include "../include/spec/det_math.h"

namespace oracle {

static inline float t_sin(float x, Trig t) { return t == Trig::Libm ? sinf(x) : det_sinf(x); }
static inline float t_cos(float x, Trig t) { return t == Trig::Libm ? cosf(x) : det_cosf(x); }
static inline float t_asin(float x, Trig t) { return t == Trig::Libm ? asinf(x) : det_asinf(x); }
static inline float t_atan2(float y, float x, Trig t) { return t == Trig::Libm ? atan2f(y, x) : det_atan2f(y, x); }

// std::time::Duration::as_secs_f32: (secs as f32) + (nanos as f32) / (NANOS_PER_SEC as f32)
float duration_as_secs_f32(uint64_t nanos_total) {
    const uint64_t secs = nanos_total / 1000000000ull;
    const uint32_t nanos = (uint32_t)(nanos_total % 1000000000ull);
    return (float)secs + (float)nanos / 1.0e9f;
}

// ---------------------------------------------------------------- rigid_body.rs
RigidBody RigidBody::new_(size_t index) {  // rigid_body.rs:64-76
    RigidBody b{};
    b.mass = 1.0f;
    for (int k = 0; k < 3; ++k) {
        b.lin_velocity[k] = 0.0f;
        b.angular_velocity[k] = 0.0f;
        b.force[k] = 0.0f;
        b.torque[k] = 0.0f;
        b.position[k] = 0.0f;
    }
    for (int k = 0; k < 9; ++k) b.inertia_tensor[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    // from_axis_angle(x_axis, 0.0): (sin(0)*axis, cos(0)) = identity
    b.rotation[0] = 0.0f; b.rotation[1] = 0.0f; b.rotation[2] = 0.0f; b.rotation[3] = 1.0f;
    b.index = index;
    return b;
}

// nalgebra Matrix3::try_inverse: adjugate over determinant
static bool try_inverse3(const float A[9], float out[9]) {
    const float m11 = A[0], m12 = A[1], m13 = A[2];
    const float m21 = A[3], m22 = A[4], m23 = A[5];
    const float m31 = A[6], m32 = A[7], m33 = A[8];
    const float minor_m12_m23 = m22 * m33 - m32 * m23;
    const float minor_m11_m23 = m21 * m33 - m31 * m23;
    const float minor_m11_m22 = m21 * m32 - m31 * m22;
    const float determinant = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (determinant == 0.0f) return false;
    out[0] = minor_m12_m23 / determinant;
    out[1] = (m13 * m32 - m33 * m12) / determinant;
    out[2] = (m12 * m23 - m22 * m13) / determinant;
    out[3] = -minor_m11_m23 / determinant;
    out[4] = (m11 * m33 - m31 * m13) / determinant;
    out[5] = (m13 * m21 - m23 * m11) / determinant;
    out[6] = minor_m11_m22 / determinant;
    out[7] = (m12 * m31 - m32 * m11) / determinant;
    out[8] = (m11 * m22 - m21 * m12) / determinant;
    return true;
}

bool RigidBody::step_velocity(float dt) {
    // Euler Translation (velocity part) — rigid_body.rs:27
    for (int k = 0; k < 3; ++k) lin_velocity[k] = lin_velocity[k] + force[k] / mass * dt;
    // rigid_body.rs:30-31
    float angular_momentum[3];
    for (int k = 0; k < 3; ++k) angular_momentum[k] = torque[k] * dt;
    float inv[9];
    if (!try_inverse3(inertia_tensor, inv)) return false;  // .unwrap() panic in the reference
    for (int r = 0; r < 3; ++r) {
        // Matrix3 * Vector3 = gemv: axpy over columns, i.e. left-to-right row sum
        float y = inv[3 * r + 0] * angular_momentum[0];
        y = inv[3 * r + 1] * angular_momentum[1] + y;
        y = inv[3 * r + 2] * angular_momentum[2] + y;
        angular_velocity[r] = angular_velocity[r] + y;
    }
    return true;
}

void RigidBody::step_position(float dt, Trig trig, bool exact_rotation) {
    // rigid_body.rs:28
    for (int k = 0; k < 3; ++k) position[k] = position[k] + lin_velocity[k] * dt;
    // rigid_body.rs:32-37
    if (angular_velocity[0] != 0.0f || angular_velocity[1] != 0.0f || angular_velocity[2] != 0.0f) {
        const float nrm = sqrtf(angular_velocity[0] * angular_velocity[0] +
                                angular_velocity[1] * angular_velocity[1] +
                                angular_velocity[2] * angular_velocity[2]);
        // DELIBERATE DIVERGENCE (Q10): when w != 0 but |w|^2 underflows to 0 the reference
        // divides by zero in normalize() and the quaternion becomes NaN for good. Contact impulses
        // produce such w (~1e-27); the rotation update is skipped instead. Unreachable in reference
        // scenes: apply_gravity adds 14.715*dt to w.x every frame.
        if (!(nrm > 0.0f)) goto rotation_done;
        float a[3];
        for (int k = 0; k < 3; ++k) a[k] = angular_velocity[k] / nrm;  // normalize()
        const float theta = nrm * dt;                                   // magnitude() * dt
        // UnitQuaternion::new(axisangle) = Quaternion::from_imag(axisangle / 2).exp()   (quirk Q1)
        const float scale = exact_rotation ? theta : t_sin(theta * 0.5f, trig);
        float u[3];
        for (int k = 0; k < 3; ++k) u[k] = (a[k] * scale) / 2.0f;
        const float nn = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        const float eps = std::numeric_limits<float>::epsilon();
        float dq[4];
        if (nn <= eps * eps) {
            dq[0] = 0.0f; dq[1] = 0.0f; dq[2] = 0.0f; dq[3] = 1.0f;
        } else {
            const float w_exp = 1.0f;  // exp(scalar part = 0)
            const float n = sqrtf(nn);
            const float f = w_exp * t_sin(n, trig) / n;
            dq[0] = u[0] * f; dq[1] = u[1] * f; dq[2] = u[2] * f;
            dq[3] = w_exp * t_cos(n, trig);
        }
        // rotation = dq * rotation (Hamilton product, no renormalisation: quirk Q6)
        const float ai = dq[0], aj = dq[1], ak = dq[2], aw = dq[3];
        const float bi = rotation[0], bj = rotation[1], bk = rotation[2], bw = rotation[3];
        const float w = aw * bw - ai * bi - aj * bj - ak * bk;
        const float i = aw * bi + ai * bw + aj * bk - ak * bj;
        const float j = aw * bj - ai * bk + aj * bw + ak * bi;
        const float k = aw * bk + ai * bj - aj * bi + ak * bw;
        rotation[0] = i; rotation[1] = j; rotation[2] = k; rotation[3] = w;
    }
rotation_done:
    // rigid_body.rs:38-39
    for (int k = 0; k < 3; ++k) { force[k] = 0.0f; torque[k] = 0.0f; }
}

bool RigidBody::step(float dt, Trig trig, bool exact_rotation) {
    // The reference interleaves: v, x, w, q (rigid_body.rs:27-37). x depends only on v, and q only
    // on w, so velocity-then-position is the same arithmetic in a different statement order.
    if (!step_velocity(dt)) return false;
    step_position(dt, trig, exact_rotation);
    return true;
}

void RigidBody::apply_force_centre_of_gravity(const float f[3]) {
    for (int k = 0; k < 3; ++k) force[k] = force[k] + f[k];
}
static inline void cross3(const float a[3], const float b[3], float out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
void RigidBody::apply_force_at_position(const float f[3], const float point[3]) {
    float d[3], c[3];
    for (int k = 0; k < 3; ++k) d[k] = point[k] - position[k];
    cross3(d, f, c);
    for (int k = 0; k < 3; ++k) torque[k] = torque[k] + c[k];
    for (int k = 0; k < 3; ++k) force[k] = force[k] + f[k];
}
void RigidBody::apply_force_at_offset(const float f[3], const float offset[3]) {
    float c[3];
    cross3(offset, f, c);
    for (int k = 0; k < 3; ++k) torque[k] = torque[k] + c[k];
    for (int k = 0; k < 3; ++k) force[k] = force[k] + f[k];
}

// ---------------------------------------------------------------- sparse_matrix.rs
void SparseMatrix::add_block(size_t row, size_t column, size_t nr, size_t nc, std::vector<float> data) {
    blocks.push_back(SparseMatrixBlock{row, column, nr, nc, std::move(data)});
}

std::vector<float> SparseMatrix::multiply_vector(const std::vector<float>& v) const {
    std::vector<float> res(nrows, 0.0f);
    for (const auto& b : blocks) {
        for (size_t row = 0; row < b.i_length; ++row) {
            float result = 0.0f;  // OVector::zeros(1); row(row).mul_to(rows(j, j_length))
            for (size_t c = 0; c < b.j_length; ++c) {
                const float t = b.data[row * b.j_length + c] * v[b.j + c];
                result = (c == 0) ? t : t + result;
            }
            res[b.i + row] += result;
        }
    }
    return res;
}

std::vector<float> SparseMatrix::tr_multiply_vector(const std::vector<float>& v) const {
    std::vector<float> res(ncols, 0.0f);
    for (const auto& b : blocks) {
        for (size_t col = 0; col < b.j_length; ++col) {
            float result = 0.0f;
            for (size_t r = 0; r < b.i_length; ++r) {
                const float t = b.data[r * b.j_length + col] * v[b.i + r];
                result = (r == 0) ? t : t + result;
            }
            res[b.j + col] += result;
        }
    }
    return res;
}

// ---------------------------------------------------------------- nalgebra Dyn reductions
float dyn_dot(const std::vector<float>& a, const std::vector<float>& b) {
    const size_t n = a.size();
    float res = 0.0f;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f, acc4 = 0.0f, acc5 = 0.0f, acc6 = 0.0f, acc7 = 0.0f;
    size_t i = 0;
    while (n - i >= 8) {
        acc0 += a[i] * b[i];
        acc1 += a[i + 1] * b[i + 1];
        acc2 += a[i + 2] * b[i + 2];
        acc3 += a[i + 3] * b[i + 3];
        acc4 += a[i + 4] * b[i + 4];
        acc5 += a[i + 5] * b[i + 5];
        acc6 += a[i + 6] * b[i + 6];
        acc7 += a[i + 7] * b[i + 7];
        i += 8;
    }
    res += acc0 + acc4;
    res += acc1 + acc5;
    res += acc2 + acc6;
    res += acc3 + acc7;
    for (size_t k = i; k < n; ++k) res += a[k] * b[k];
    return res;
}

float dyn_amax(const std::vector<float>& a) {
    // nalgebra amax folds f32::max over |e|, which ignores a NaN operand; starting from 0 is the same
    // value for every input that is not all-NaN (NaN outcomes are undefined, never tested)
    float m = 0.0f;
    for (size_t k = 0; k < a.size(); ++k) {
        const float e = fabsf(a[k]);
        m = (e > m) ? e : m;
    }
    return m;
}

// ---------------------------------------------------------------- sle_solver.rs
static std::vector<float> calculate_lhs_multiplied_to_vec(const SparseMatrix& j, const std::vector<float>& inv_masses,
                                                          const std::vector<float>& factor) {
    std::vector<float> j_factor = j.tr_multiply_vector(factor);  // sle_solver.rs:49
    for (size_t k = 0; k < j_factor.size(); ++k) j_factor[k] = j_factor[k] * inv_masses[k];  // component_mul
    return j.multiply_vector(j_factor);  // :50
}

std::optional<std::vector<float>> solve_conjugate_gradient(const SparseMatrix& j, const std::vector<float>& inv_masses,
                                                           const std::vector<float>& rhs,
                                                           const std::optional<std::vector<float>>& previous,
                                                           const CgConfig& cfg, uint32_t* iterations_out) {
    const size_t n = rhs.size();
    std::vector<float> x = previous ? *previous : std::vector<float>(n, 0.0f);  // :22-26
    if (x.size() != n) x.assign(n, 0.0f);  // nalgebra would panic on the shape mismatch; callers reset first
    std::vector<float> r(n);
    {
        const std::vector<float> ax = calculate_lhs_multiplied_to_vec(j, inv_masses, x);
        for (size_t k = 0; k < n; ++k) r[k] = rhs[k] - ax[k];  // :28
    }
    std::vector<float> p = r;  // :29
    const float rhs_amax = dyn_amax(rhs);
    const float bound0 = rhs_amax * cfg.max_error;
    const float bound = bound0 > cfg.min_error ? bound0 : cfg.min_error;  // f32::max
    uint32_t it = 0;
    for (; it < cfg.max_iterations; ++it) {
        const std::vector<float> j_p = calculate_lhs_multiplied_to_vec(j, inv_masses, p);  // :32
        const float rk_magnitude = dyn_dot(r, r);                                          // :33
        const float alpha = rk_magnitude / dyn_dot(p, j_p);                                // :34
        for (size_t k = 0; k < n; ++k) x[k] = x[k] + alpha * p[k];                         // :35
        for (size_t k = 0; k < n; ++k) r[k] = r[k] - alpha * j_p[k];                       // :37
        if (dyn_amax(r) < bound) {                                                         // :38
            if (iterations_out) *iterations_out = it + 1;
            return x;
        }
        const float beta = dyn_dot(r, r) / rk_magnitude;            // :42
        for (size_t k = 0; k < n; ++k) p[k] = r[k] + beta * p[k];   // :43
    }
    if (iterations_out) *iterations_out = it;
    return std::nullopt;  // :45
}

// ---------------------------------------------------------------- nalgebra quaternion <-> euler
void quat_from_euler_angles(float roll, float pitch, float yaw, Trig trig, float out[4]) {
    const float sr = t_sin(roll * 0.5f, trig), cr = t_cos(roll * 0.5f, trig);
    const float sp = t_sin(pitch * 0.5f, trig), cp = t_cos(pitch * 0.5f, trig);
    const float sy = t_sin(yaw * 0.5f, trig), cy = t_cos(yaw * 0.5f, trig);
    out[3] = cr * cp * cy + sr * sp * sy;  // w
    out[0] = sr * cp * cy - cr * sp * sy;  // i
    out[1] = cr * sp * cy + sr * cp * sy;  // j
    out[2] = cr * cp * sy - sr * sp * cy;  // k
}

void quat_euler_angles(const float q[4], Trig trig, float out[3]) {
    // to_rotation_matrix
    const float i = q[0], j = q[1], k = q[2], w = q[3];
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    const float r00 = ww + ii - jj - kk, r01 = ij - wk, r02 = wj + ik;
    const float r10 = wk + ij;
    const float r20 = ik - wj, r21 = wi + jk, r22 = ww - ii - jj + kk;
    // Rotation3::euler_angles (Slabaugh)
    if (fabsf(r20) < 1.0f) {
        const float pitch = -t_asin(r20, trig);
        const float theta_cos = t_cos(pitch, trig);
        out[0] = t_atan2(r21 / theta_cos, r22 / theta_cos, trig);
        out[1] = pitch;
        out[2] = t_atan2(r10 / theta_cos, r00 / theta_cos, trig);
    } else if (r20 <= -1.0f) {
        out[0] = t_atan2(r01, r02, trig);
        out[1] = 1.57079632679489661923f;
        out[2] = 0.0f;
    } else {
        out[0] = t_atan2(-r01, -r02, trig);
        out[1] = -1.57079632679489661923f;
        out[2] = 0.0f;
    }
}

// ---------------------------------------------------------------- constraints
ConstraintOutput PhysicsState::calculate(const Constraint& con) const {
    ConstraintOutput o{};
    const RigidBody& b = entities[con.rigid_body];
    if (con.kind == Constraint::FixedPosition) {  // fixed_position_constraint.rs:13-27
        for (int k = 0; k < 3; ++k) o.c[k] = b.position[k] - con.position[k];
        o.j[0 * 12 + 0] = 1.0f;
        o.j[1 * 12 + 1] = 1.0f;
        o.j[2 * 12 + 2] = 1.0f;
    } else {  // fixed_orientation_constraint.rs:15-30
        float rpy[3];
        quat_euler_angles(b.rotation, trig, rpy);
        for (int k = 0; k < 3; ++k) o.c[k] = rpy[k] - con.position[k];
        o.j[0 * 12 + 3] = 1.0f;
        o.j[1 * 12 + 4] = 1.0f;
        o.j[2 * 12 + 5] = 1.0f;
    }
    for (int k = 0; k < 3; ++k) { o.kd[k] = 1.0f; o.ks[k] = 10.0f; }  // KD, KS (:5-7 in both files)
    return o;
}

std::optional<std::pair<std::vector<float>, std::vector<float>>> PhysicsState::solve_constraints() {
    const size_t n6 = entities.size() * 6;
    std::vector<float> inv_masses(n6), q_dot(n6), existing_forces(n6);
    for (size_t b = 0; b < entities.size(); ++b) {  // constraints.rs:72-104
        const RigidBody& body = entities[b];
        const float inv_mass = 1.0f / body.mass;
        for (int k = 0; k < 6; ++k) inv_masses[6 * b + k] = inv_mass;  // quirk Q4
        for (int k = 0; k < 3; ++k) {
            q_dot[6 * b + k] = body.lin_velocity[k];
            q_dot[6 * b + 3 + k] = body.angular_velocity[k];
            existing_forces[6 * b + k] = body.force[k];
            existing_forces[6 * b + 3 + k] = body.torque[k];
        }
    }
    const size_t constraint_count = constraints.size() * 3;  // get_full_constraint_count (:172-176)
    SparseMatrix j(constraint_count, n6), j_dot(constraint_count, n6);
    std::vector<float> k_d(constraint_count, 0.0f), k_s(constraint_count, 0.0f), c(constraint_count, 0.0f);
    size_t constraint_index = 0;
    for (const Constraint& con : constraints) {  // :115-151
        const ConstraintOutput output = calculate(con);
        const size_t single = 3;
        for (size_t i = 0; i < single; ++i) {
            k_d[constraint_index + i] = output.kd[i];
            k_s[constraint_index + i] = output.ks[i];
            c[constraint_index + i] = output.c[i];
        }
        const size_t bodies[1] = {con.rigid_body};  // get_rigid_bodies() returns one index
        for (size_t i = 0; i < 1; ++i) {
            std::vector<float> j_slice(single * 6), j_dot_slice(single * 6);
            for (size_t r = 0; r < single; ++r)
                for (size_t cc = 0; cc < 6; ++cc) {
                    j_slice[r * 6 + cc] = output.j[r * 12 + i * 6 + cc];
                    j_dot_slice[r * 6 + cc] = output.j_dot[r * 12 + i * 6 + cc];
                }
            j.add_block(constraint_index, bodies[i] * 6, single, 6, std::move(j_slice));
            j_dot.add_block(constraint_index, bodies[i] * 6, single, 6, std::move(j_dot_slice));
        }
        constraint_index += single;
    }
    std::vector<float> j_dot_times_q_dot = j_dot.multiply_vector(q_dot);  // :153
    for (float& e : j_dot_times_q_dot) e = -e;
    const std::vector<float> c_dot = j.multiply_vector(q_dot);  // :155
    for (size_t k = 0; k < constraint_count; ++k) k_d[k] = k_d[k] * c_dot[k];
    for (size_t k = 0; k < constraint_count; ++k) k_s[k] = k_s[k] * c[k];
    std::vector<float> fw(n6);
    for (size_t k = 0; k < n6; ++k) fw[k] = existing_forces[k] * inv_masses[k];
    const std::vector<float> jfw = j.multiply_vector(fw);
    std::vector<float> rhs(constraint_count);
    for (size_t k = 0; k < constraint_count; ++k) rhs[k] = j_dot_times_q_dot[k] - jfw[k] - k_s[k] - k_d[k];  // :159-160

    uint32_t iters = 0;
    auto lambda = solve_conjugate_gradient(j, inv_masses, rhs, previous_solution, cg, &iters);  // :162
    last_cg_iterations = iters;
    last_cg_converged = lambda.has_value();
    if (lambda) {
        std::vector<float> matrix = j.tr_multiply_vector(*lambda);  // :165
        return std::make_pair(std::move(*lambda), std::move(matrix));
    }
    return std::nullopt;
}

// ---------------------------------------------------------------- physics.rs
void PhysicsState::apply_gravity() {
    for (RigidBody& b : entities) b.apply_force_at_offset(gravity_force, gravity_offset);
}

void PhysicsState::constraint_phase() {
    auto lambda = solve_constraints();  // physics.rs:43
    if (lambda) {
        previous_solution = std::move(lambda->first);
        // physics.rs:47-50: column_iter() over a 6N x 1 vector yields ONE column => only
        // entities[0] receives rows 0..3 / 3..6 (quirk Q3).
        const std::vector<float>& matrix = lambda->second;
        RigidBody& b0 = entities[0];
        for (int k = 0; k < 3; ++k) b0.force[k] = b0.force[k] + matrix[k];
        for (int k = 0; k < 3; ++k) b0.torque[k] = b0.torque[k] + matrix[3 + k];
    }
}

bool PhysicsState::step(uint64_t dt_nanos) {
    const float dt = duration_as_secs_f32(dt_nanos);
    for (RigidBody& b : entities)
        if (!b.step(dt, trig, exact_rotation)) return false;
    return true;
}

bool PhysicsState::update(uint64_t dt_nanos) {
    if (entities.empty()) return false;  // the reference panics indexing entities[0] / view
    apply_gravity();
    constraint_phase();
    return step(dt_nanos);
}

}  // namespace oracle
