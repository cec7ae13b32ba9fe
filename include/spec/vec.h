/* vec.h — f32 3-vector / quaternion / 3x3 helpers with the exact operation order of
 * nalgebra 0.32.2 (the reference's arithmetic crate, Cargo.toml:19), as relied on by
 * /root/reference/src/physics/rigid_body.rs:24-62. nalgebra's source is not in the container;
 * the orders below restate its published algorithms (flagged "from knowledge" in SURVEY.md §8c).
 * Shared by HIP kernels and the collision-stage spec; build with -ffp-contract=off.
 */
#ifndef PHYS_SPEC_VEC_H
#define PHYS_SPEC_VEC_H

#include "det_math.h"

typedef struct { float x, y, z; } v3;
typedef struct { float i, j, k, w; } quat; /* nalgebra storage order [i, j, k, w] */
typedef struct { float m[9]; } m33;        /* row-major: m[3*r + c] */

PHYS_HD v3 v3_make(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PHYS_HD v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
PHYS_HD v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
PHYS_HD v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
PHYS_HD v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
PHYS_HD v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* nalgebra dot for U3: a + b + c, left to right */
PHYS_HD float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* nalgebra cross for 3-vectors */
PHYS_HD v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PHYS_HD float v3_norm(v3 a) { return det_sqrtf(v3_dot(a, a)); }

/* Matrix3 * Vector3 as nalgebra gemv: axpy over columns => left-to-right row sums */
PHYS_HD v3 m33_mul_v3(const m33* M, v3 v) {
    v3 r;
    r.x = (M->m[0] * v.x + M->m[1] * v.y) + M->m[2] * v.z;
    r.y = (M->m[3] * v.x + M->m[4] * v.y) + M->m[5] * v.z;
    r.z = (M->m[6] * v.x + M->m[7] * v.y) + M->m[8] * v.z;
    return r;
}

/* nalgebra Matrix3::try_inverse (adjugate / determinant). Returns 0 when det == 0. */
PHYS_HD int m33_try_inverse(const m33* A, m33* out) {
    const float m11 = A->m[0], m12 = A->m[1], m13 = A->m[2];
    const float m21 = A->m[3], m22 = A->m[4], m23 = A->m[5];
    const float m31 = A->m[6], m32 = A->m[7], m33_ = A->m[8];
    const float minor_m12_m23 = m22 * m33_ - m32 * m23;
    const float minor_m11_m23 = m21 * m33_ - m31 * m23;
    const float minor_m11_m22 = m21 * m32 - m31 * m22;
    const float det = (m11 * minor_m12_m23 - m12 * minor_m11_m23) + m13 * minor_m11_m22;
    if (det == 0.0f) return 0;
    out->m[0] = minor_m12_m23 / det;
    out->m[1] = (m13 * m32 - m33_ * m12) / det;
    out->m[2] = (m12 * m23 - m22 * m13) / det;
    out->m[3] = -minor_m11_m23 / det;
    out->m[4] = (m11 * m33_ - m31 * m13) / det;
    out->m[5] = (m13 * m21 - m23 * m11) / det;
    out->m[6] = minor_m11_m22 / det;
    out->m[7] = (m12 * m31 - m32 * m11) / det;
    out->m[8] = (m11 * m22 - m21 * m12) / det;
    return 1;
}

/* Hamilton product, nalgebra Quaternion * Quaternion term order; no renormalisation */
PHYS_HD quat quat_mul(quat a, quat b) {
    quat r;
    r.w = ((a.w * b.w - a.i * b.i) - a.j * b.j) - a.k * b.k;
    r.i = ((a.w * b.i + a.i * b.w) + a.j * b.k) - a.k * b.j;
    r.j = ((a.w * b.j - a.i * b.k) + a.j * b.w) + a.k * b.i;
    r.k = ((a.w * b.k + a.i * b.j) - a.j * b.i) + a.k * b.w;
    return r;
}

/* rotation matrix of a (not necessarily unit) quaternion, nalgebra to_rotation_matrix form */
PHYS_HD void quat_to_m33(quat q, m33* R) {
    const float i = q.i, j = q.j, k = q.k, w = q.w;
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    R->m[0] = ((ww + ii) - jj) - kk; R->m[1] = ij - wk;               R->m[2] = wj + ik;
    R->m[3] = wk + ij;               R->m[4] = ((ww - ii) + jj) - kk; R->m[5] = jk - wi;
    R->m[6] = ik - wj;               R->m[7] = wi + jk;               R->m[8] = ((ww - ii) - jj) + kk;
}

#endif /* PHYS_SPEC_VEC_H */
