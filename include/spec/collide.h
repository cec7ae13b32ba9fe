/* collide.h — normative scalar definition of the collision-stage arithmetic.
 *
 * The reference (martingoe/physics) has NO shapes, broad-phase or narrow-phase ("parity unpinned"
 * for these rows). This header is therefore the specification itself: per-body AABB, per-pair
 * overlap test and per-pair contact manifold, written as scalar f32 functions with a fixed
 * operation order. The HIP kernels (physics_amd/csrc) and the CPU oracle (oracle/) both compile
 * these functions (-ffp-contract=off) and drive them with independent loops, so a parallel GPU run
 * and a sequential CPU run agree bit for bit. tests/test_collide_kat.py pins the functions
 * themselves with closed-form known answers.
 *
 * Conventions: manifold normal points from body A to body B; depth > 0 is penetration, depth < 0 a
 * speculative gap inside the contact margin. Body B may be the ground plane (PHYS_GROUND_ID).
 */
#ifndef PHYS_SPEC_COLLIDE_H
#define PHYS_SPEC_COLLIDE_H

#include <stdint.h>

#include "vec.h"

#define PHYS_GROUND_ID 0xFFFFFFFFu
#define PHYS_SPEC_SHAPE_NONE 0u
#define PHYS_SPEC_SHAPE_SPHERE 1u
#define PHYS_SPEC_SHAPE_BOX 2u

typedef struct { v3 lo, hi; } aabb_t;

typedef struct {
    v3 c;          /* centre (world) */
    m33 R;         /* rotation matrix of the (unnormalised, quirk Q6) quaternion */
    v3 h;          /* half extents; sphere radius in h.x */
    uint32_t type;
} geom_t;

typedef struct {
    v3 normal;     /* A -> B */
    int count;     /* 0..4 */
    v3 pt[4];      /* world contact points */
    float depth[4];
} manifold_t;

/* per-lane scratch of the polygon clipper: 128 bytes - ONE polygon of up to 8 vertices, clipped in place (clip_poly), and
 * the depths of the candidate points. The HIP narrow phase keeps it in LDS (one slice per lane, odd dword stride)
 * instead of private scratch memory; the oracle puts it on the stack. At 33 dwords per lane four 256-thread workgroups
 * fit a CU's LDS (two at 192 bytes and two polygons, three at 48 dwords): the narrow phase is bound by how many dependent
 * memory round trips its resident waves can overlap. */
typedef struct {
    v3 poly[8];
    float depth[8];
} clip_ws_t;

PHYS_HD geom_t geom_make(v3 c, quat q, v3 h, uint32_t type) {
    geom_t g;
    g.c = c; g.h = h; g.type = type;
    quat_to_m33(q, &g.R);
    return g;
}

/* column c of a row-major 3x3. Selections, not indexing: a matrix (or any small array) indexed by a run-time value
 * is put into per-lane scratch memory by the GPU compiler, and nothing of this header may live there (DESIGN.md:
 * kernels that used scratch lost data whenever a scratch-using kernel of another stream ran beside them). */
PHYS_HD v3 m33_col(const m33* R, int c) {
    const float m0 = R->m[0], m1 = R->m[1], m2 = R->m[2], m3 = R->m[3], m4 = R->m[4], m5 = R->m[5], m6 = R->m[6],
                m7 = R->m[7], m8 = R->m[8]; /* all nine read first: the selections below are between values */
    return v3_make(c == 0 ? m0 : (c == 1 ? m1 : m2), c == 0 ? m3 : (c == 1 ? m4 : m5),
                   c == 0 ? m6 : (c == 1 ? m7 : m8));
}
PHYS_HD float sel3(const float* a, int i) {
    const float a0 = a[0], a1 = a[1], a2 = a[2];
    return i == 0 ? a0 : (i == 1 ? a1 : a2);
}
/* R^T * v (world -> local for an orthonormal R) */
PHYS_HD v3 m33_tmul_v3(const m33* R, v3 v) {
    return v3_make((R->m[0] * v.x + R->m[3] * v.y) + R->m[6] * v.z,
                   (R->m[1] * v.x + R->m[4] * v.y) + R->m[7] * v.z,
                   (R->m[2] * v.x + R->m[5] * v.y) + R->m[8] * v.z);
}
PHYS_HD float v3_get(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* ---- A10: fattened world AABB of one body. NONE shapes get an inverted box (never overlaps). */
PHYS_HD aabb_t body_aabb(v3 c, quat q, v3 h, uint32_t type, float margin) {
    aabb_t b;
    v3 e;
    if (type == PHYS_SPEC_SHAPE_SPHERE) {
        e = v3_make(h.x, h.x, h.x);
    } else if (type == PHYS_SPEC_SHAPE_BOX) {
        m33 R;
        quat_to_m33(q, &R);
        e.x = (det_absf(R.m[0]) * h.x + det_absf(R.m[1]) * h.y) + det_absf(R.m[2]) * h.z;
        e.y = (det_absf(R.m[3]) * h.x + det_absf(R.m[4]) * h.y) + det_absf(R.m[5]) * h.z;
        e.z = (det_absf(R.m[6]) * h.x + det_absf(R.m[7]) * h.y) + det_absf(R.m[8]) * h.z;
    } else {
        b.lo = v3_make(3.0e38f, 3.0e38f, 3.0e38f);
        b.hi = v3_make(-3.0e38f, -3.0e38f, -3.0e38f);
        return b;
    }
    b.lo = v3_make((c.x - e.x) - margin, (c.y - e.y) - margin, (c.z - e.z) - margin);
    b.hi = v3_make((c.x + e.x) + margin, (c.y + e.y) + margin, (c.z + e.z) + margin);
    return b;
}

PHYS_HD int aabb_overlap(aabb_t a, aabb_t b) {
    return a.lo.x <= b.hi.x && b.lo.x <= a.hi.x && a.lo.y <= b.hi.y && b.lo.y <= a.hi.y && a.lo.z <= b.hi.z &&
           b.lo.z <= a.hi.z;
}

/* ---- A11: narrow phase -------------------------------------------------------------------- */

PHYS_HD void manifold_clear(manifold_t* m) {
    m->count = 0;
    m->normal = v3_make(0.0f, 1.0f, 0.0f);
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) { m->pt[k] = v3_make(0.0f, 0.0f, 0.0f); m->depth[k] = 0.0f; }
}

PHYS_HD void collide_sphere_sphere(const geom_t* A, const geom_t* B, float margin, manifold_t* m) {
    const v3 d = v3_sub(B->c, A->c);
    const float dist2 = v3_dot(d, d);
    const float r = A->h.x + B->h.x;
    const float reach = r + margin;
    if (dist2 > reach * reach) return;
    const float dist = det_sqrtf(dist2);
    v3 n = v3_make(0.0f, 1.0f, 0.0f);
    if (dist > 1.0e-12f) n = v3_div(d, dist);
    const float depth = r - dist;
    m->normal = n;
    m->count = 1;
    m->pt[0] = v3_add(A->c, v3_scale(n, A->h.x - 0.5f * depth));
    m->depth[0] = depth;
}

/* sphere S against box X; normal written S -> X */
PHYS_HD void collide_sphere_box_raw(const geom_t* S, const geom_t* X, float margin, manifold_t* m) {
    const v3 p = m33_tmul_v3(&X->R, v3_sub(S->c, X->c)); /* sphere centre in box frame */
    v3 q;
    q.x = det_maxf(-X->h.x, det_minf(p.x, X->h.x));
    q.y = det_maxf(-X->h.y, det_minf(p.y, X->h.y));
    q.z = det_maxf(-X->h.z, det_minf(p.z, X->h.z));
    const float r = S->h.x;
    v3 nl; /* box -> sphere, box frame */
    float depth;
    if (q.x == p.x && q.y == p.y && q.z == p.z) {
        /* centre inside the box: leave through the nearest face */
        const float fx = X->h.x - det_absf(p.x), fy = X->h.y - det_absf(p.y), fz = X->h.z - det_absf(p.z);
        if (fx <= fy && fx <= fz) { nl = v3_make(p.x < 0.0f ? -1.0f : 1.0f, 0.0f, 0.0f); depth = fx + r; q.x = p.x < 0.0f ? -X->h.x : X->h.x; }
        else if (fy <= fz)        { nl = v3_make(0.0f, p.y < 0.0f ? -1.0f : 1.0f, 0.0f); depth = fy + r; q.y = p.y < 0.0f ? -X->h.y : X->h.y; }
        else                      { nl = v3_make(0.0f, 0.0f, p.z < 0.0f ? -1.0f : 1.0f); depth = fz + r; q.z = p.z < 0.0f ? -X->h.z : X->h.z; }
    } else {
        const v3 dl = v3_sub(p, q);
        const float dist2 = v3_dot(dl, dl);
        const float reach = r + margin;
        if (dist2 > reach * reach) return;
        const float dist = det_sqrtf(dist2);
        nl = v3_div(dl, dist);
        depth = r - dist;
    }
    const v3 nw = m33_mul_v3(&X->R, nl);           /* box -> sphere, world */
    m->normal = v3_neg(nw);                        /* sphere -> box */
    m->count = 1;
    m->pt[0] = v3_add(X->c, m33_mul_v3(&X->R, q)); /* on the box surface */
    m->depth[0] = depth;
}

/* Sutherland-Hodgman clip of a convex polygon (<= 8 verts) against  dot(p - c, u) <= lim, IN PLACE. Vertex k is the
 * `a` of edge k and the `b` of edge k - 1: it travels in registers from one edge to the next, and so does its distance (the
 * same expression on the same vertex: the same value), and the first vertex, which the last edge needs again. The output
 * never passes the vertex that is read next: before edge k at most k points are out - or k + 1 right behind a LEAVING edge
 * (two points out for one vertex in), whose next vertex is outside and puts out at most one. A convex polygon leaves a
 * half-plane once; should rounding ever make a degenerate one leave twice, a vertex is overwritten before it is read -
 * deterministically, inside the array, and alike in the oracle and on the device. */
PHYS_HD int clip_poly(v3* p, int n_in, v3 c, v3 u, float lim) {
    int n_out = 0;
    if (n_in <= 0) return 0;
    const v3 first = p[0];
    v3 a = first;
    float da = v3_dot(v3_sub(a, c), u) - lim;
    for (int k = 0; k < 8; ++k) {
        if (k >= n_in) break;
        v3 b = first;
        if (k + 1 != n_in) b = p[k + 1];
        const float db = v3_dot(v3_sub(b, c), u) - lim;
        if (da <= 0.0f) {
            if (n_out < 8) p[n_out++] = a;
            if (db > 0.0f) {
                const float t = da / (da - db);
                if (n_out < 8) p[n_out++] = v3_add(a, v3_scale(v3_sub(b, a), t));
            }
        } else if (db <= 0.0f) {
            const float t = da / (da - db);
            if (n_out < 8) p[n_out++] = v3_add(a, v3_scale(v3_sub(b, a), t));
        }
        a = b; da = db;
    }
    return n_out;
}

/* keep at most 4 of n candidate points: deepest, farthest from it, then the two extreme signed
 * areas. Ties resolve to the lowest index (strict comparisons). */
PHYS_HD void manifold_reduce(const v3* p, const float* depth, int n, v3 normal, manifold_t* m) {
    if (n <= 4) {
        m->count = n;
        PHYS_UNROLL
        for (int k = 0; k < 4; ++k) if (k < n) { m->pt[k] = p[k]; m->depth[k] = depth[k]; }
        return;
    }
    int i0 = 0;
    for (int k = 1; k < 8; ++k) if (k < n && depth[k] > depth[i0]) i0 = k;
    int i1 = -1; float best = -1.0f;
    for (int k = 0; k < 8; ++k) if (k < n && k != i0) {
        const v3 d = v3_sub(p[k], p[i0]);
        const float dd = v3_dot(d, d);
        if (dd > best) { best = dd; i1 = k; }
    }
    int i2 = -1, i3 = -1; float amax = 0.0f, amin = 0.0f;
    const v3 e = v3_sub(p[i1], p[i0]);
    for (int k = 0; k < 8; ++k) if (k < n && k != i0 && k != i1) {
        const float area = v3_dot(v3_cross(e, v3_sub(p[k], p[i0])), normal);
        if (area > amax) { amax = area; i2 = k; }
        if (area < amin) { amin = area; i3 = k; }
    }
    m->pt[0] = p[i0]; m->depth[0] = depth[i0];
    m->pt[1] = p[i1]; m->depth[1] = depth[i1];
    /* the area extremes follow in that order; each manifold slot is written at a fixed index (no run-time
     * indexing of the manifold: it stays in registers on the GPU) */
    const int has2 = i2 >= 0, has3 = i3 >= 0;
    const int s2 = has2 ? i2 : i3;
    if (has2 || has3) { m->pt[2] = p[s2]; m->depth[2] = depth[s2]; }
    if (has2 && has3) { m->pt[3] = p[i3]; m->depth[3] = depth[i3]; }
    m->count = 2 + has2 + has3;
}

/* face contact: Ref's face (axis r, side sgn) against the most anti-parallel face of Inc.
 * Writes points and depths; manifold normal is set by the caller. */
PHYS_HD void box_face_contact(const geom_t* Ref, const geom_t* Inc, int r, float sgn, float margin, v3 n_ab,
                              manifold_t* m, clip_ws_t* ws) {
    const v3 nref = v3_scale(m33_col(&Ref->R, r), sgn);
    /* incident face: axis of Inc most anti-parallel to nref */
    const float d0 = v3_dot(nref, m33_col(&Inc->R, 0));
    const float d1 = v3_dot(nref, m33_col(&Inc->R, 1));
    const float d2 = v3_dot(nref, m33_col(&Inc->R, 2));
    int j = 0; float dj = d0;
    if (det_absf(d1) > det_absf(dj)) { j = 1; dj = d1; }
    if (det_absf(d2) > det_absf(dj)) { j = 2; dj = d2; }
    const float jsgn = dj > 0.0f ? -1.0f : 1.0f;
    const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    const v3 fc = v3_add(Inc->c, v3_scale(m33_col(&Inc->R, j), jsgn * v3_get(Inc->h, j)));
    const v3 e1 = v3_scale(m33_col(&Inc->R, j1), v3_get(Inc->h, j1));
    const v3 e2 = v3_scale(m33_col(&Inc->R, j2), v3_get(Inc->h, j2));
    v3* polyA = ws->poly;
    polyA[0] = v3_add(v3_add(fc, e1), e2);
    polyA[1] = v3_add(v3_sub(fc, e1), e2);
    polyA[2] = v3_sub(v3_sub(fc, e1), e2);
    polyA[3] = v3_sub(v3_add(fc, e1), e2);
    int n = 4;
    const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
    const v3 u1 = m33_col(&Ref->R, r1), u2 = m33_col(&Ref->R, r2);
    const float l1 = v3_get(Ref->h, r1), l2 = v3_get(Ref->h, r2);
    n = clip_poly(polyA, n, Ref->c, u1, l1);
    n = clip_poly(polyA, n, Ref->c, v3_neg(u1), l1);
    n = clip_poly(polyA, n, Ref->c, u2, l2);
    n = clip_poly(polyA, n, Ref->c, v3_neg(u2), l2);
    /* extent of the contact patch (the incident face inside the reference face's side planes) along the two
     * in-plane axes of the reference face: the sliver rule below */
    float lo1 = 3.0e38f, hi1 = -3.0e38f, lo2 = 3.0e38f, hi2 = -3.0e38f;
    /* keep points at or below the reference face (+ margin): compacted in place (nc <= k) */
    v3* cand = ws->poly; float* cdep = ws->depth; int nc = 0;
    const float hr = v3_get(Ref->h, r);
    float deepest = -3.0e38f;
    for (int k = 0; k < 8; ++k) {
        if (k >= n) break;
        const v3 rel = v3_sub(polyA[k], Ref->c);
        const float c1 = v3_dot(rel, u1), c2 = v3_dot(rel, u2);
        lo1 = det_minf(lo1, c1); hi1 = det_maxf(hi1, c1);
        lo2 = det_minf(lo2, c2); hi2 = det_maxf(hi2, c2);
        const float dep = hr - v3_dot(rel, nref);
        if (dep >= -margin) { cand[nc] = polyA[k]; cdep[nc] = dep; deepest = det_maxf(deepest, dep); ++nc; }
    }
    /* SLIVER RULE. Two boxes that touch along an edge or at a corner (diagonal neighbours of a stack: face separation 0
     * on two or three axes at once) come out of the clipper with a patch of no width - up to four points on a line or on
     * one spot, all at depth ~0. Such a manifold carries no load (bias 0, and friction is bounded by a normal impulse
     * that stays 0), yet a stack of boxes has four of them for every manifold that does (C5: 11.4 manifolds and 45
     * points per box, of which 3 x 4 bear weight; 33 solver colours instead of 8). So: a face contact whose patch is
     * narrower than kSliverRel of the smallest half extent of the two faces, and whose deepest point penetrates by no
     * more than kSliverDepth margins, is NOT a contact. A sliver that does penetrate (a box balanced on a thin strip
     * sinks by a quarter of the margin first) is kept, and so is every line contact of a tilted edge on a face (its
     * patch, taken before the depth filter, is as wide as the incident face). */
    {
        const float kSliverRel = 0.1f, kSliverDepth = 0.25f;
        const float width = det_minf(hi1 - lo1, hi2 - lo2);
        const float face = det_minf(det_minf(l1, l2), det_minf(v3_get(Inc->h, j1), v3_get(Inc->h, j2)));
        if (nc > 0 && width < kSliverRel * face && deepest <= kSliverDepth * margin) nc = 0;
    }
    manifold_reduce(cand, cdep, nc, n_ab, m);
}

PHYS_HD void collide_box_box(const geom_t* A, const geom_t* B, float margin, manifold_t* m, clip_ws_t* ws) {
    const v3 d = v3_sub(B->c, A->c);
    float C[3][3], absC[3][3], tA[3], tB[3], hA[3], hB[3];
    hA[0] = A->h.x; hA[1] = A->h.y; hA[2] = A->h.z;
    hB[0] = B->h.x; hB[1] = B->h.y; hB[2] = B->h.z;
    PHYS_UNROLL
    for (int i = 0; i < 3; ++i) {
        const v3 ai = m33_col(&A->R, i);
        tA[i] = v3_dot(d, ai);
        PHYS_UNROLL
        for (int j = 0; j < 3; ++j) {
            C[i][j] = v3_dot(ai, m33_col(&B->R, j));
            absC[i][j] = det_absf(C[i][j]) + 1.0e-6f;
        }
    }
    PHYS_UNROLL
    for (int j = 0; j < 3; ++j) tB[j] = v3_dot(d, m33_col(&B->R, j));

    /* face axes of A */
    float aMax = -3.0e38f; int aAxis = 0;
    PHYS_UNROLL
    for (int i = 0; i < 3; ++i) {
        const float s = det_absf(tA[i]) - (hA[i] + ((absC[i][0] * hB[0] + absC[i][1] * hB[1]) + absC[i][2] * hB[2]));
        if (s > margin) return;
        if (s > aMax) { aMax = s; aAxis = i; }
    }
    /* face axes of B */
    float bMax = -3.0e38f; int bAxis = 0;
    PHYS_UNROLL
    for (int j = 0; j < 3; ++j) {
        const float s = det_absf(tB[j]) - (hB[j] + ((absC[0][j] * hA[0] + absC[1][j] * hA[1]) + absC[2][j] * hA[2]));
        if (s > margin) return;
        if (s > bMax) { bMax = s; bAxis = j; }
    }
    /* edge axes a_i x b_j */
    float eMax = -3.0e38f; int eI = -1, eJ = -1;
    PHYS_UNROLL
    for (int i = 0; i < 3; ++i) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        PHYS_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const float len2 = 1.0f - C[i][j] * C[i][j];
            if (len2 < 1.0e-4f) continue; /* near-parallel edges: covered by the face axes */
            const float ra = hA[i1] * absC[i2][j] + hA[i2] * absC[i1][j];
            const float rb = hB[j1] * absC[i][j2] + hB[j2] * absC[i][j1];
            const float dist = det_absf(tA[i2] * C[i1][j] - tA[i1] * C[i2][j]);
            const float s = (dist - (ra + rb)) / det_sqrtf(len2);
            if (s > margin) return;
            if (s > eMax) { eMax = s; eI = i; eJ = j; }
        }
    }
    /* prefer faces (frame coherence): an edge axis must beat the faces clearly. The tolerance scales the FACE side:
     * with separations negative (penetration) `eMax > kRel * faceMax + kAbs` asks the edge axis to be shallower than
     * the shallowest face by 5 % + 0.01. (Round 1 had the factor on the edge side, `kRel * eMax > faceMax + kAbs`,
     * which for penetrations beyond 0.2 let an edge axis EQUAL to a face axis win - two axis-aligned boxes in deep
     * face contact got one edge-edge point instead of a four-point face manifold; found by the brute-force SAT check
     * of tests/test_gpu_independent.py.) */
    const float kRel = 0.95f, kAbs = 0.01f;
    const float faceMax = det_maxf(aMax, bMax);
    if (eI >= 0 && eMax > kRel * faceMax + kAbs) {
        /* edge-edge */
        const v3 ai = m33_col(&A->R, eI), bj = m33_col(&B->R, eJ);
        v3 n = v3_cross(ai, bj);
        const float nl = v3_norm(n);
        n = v3_div(n, nl);
        if (v3_dot(n, d) < 0.0f) n = v3_neg(n);
        /* supporting edges */
        v3 pA = A->c, pB = B->c;
        PHYS_UNROLL
        for (int k = 0; k < 3; ++k) {
            if (k != eI) {
                const v3 ak = m33_col(&A->R, k);
                const float sg = v3_dot(n, ak) > 0.0f ? 1.0f : -1.0f;
                pA = v3_add(pA, v3_scale(ak, sg * hA[k]));
            }
            if (k != eJ) {
                const v3 bk = m33_col(&B->R, k);
                const float sg = v3_dot(n, bk) > 0.0f ? -1.0f : 1.0f;
                pB = v3_add(pB, v3_scale(bk, sg * hB[k]));
            }
        }
        /* closest points of the two lines pA + s*ai, pB + t*bj */
        const v3 w = v3_sub(pA, pB);
        const float b = v3_dot(ai, bj); /* == C[eI][eJ], recomputed so that C is never indexed at run time */
        const float dd = v3_dot(ai, w), ee = v3_dot(bj, w);
        const float den = 1.0f - b * b;
        const float sa = (b * ee - dd) / den;
        const float tb = (ee - b * dd) / den;
        const v3 qa = v3_add(pA, v3_scale(ai, sa));
        const v3 qb = v3_add(pB, v3_scale(bj, tb));
        m->normal = n;
        m->count = 1;
        m->pt[0] = v3_scale(v3_add(qa, qb), 0.5f);
        m->depth[0] = -eMax;
        return;
    }
    if (bMax > kRel * aMax + kAbs) {
        /* reference = B; its face looks back toward A */
        const float sgn = sel3(tB, bAxis) > 0.0f ? -1.0f : 1.0f;
        const v3 nref = v3_scale(m33_col(&B->R, bAxis), sgn); /* B -> A */
        m->normal = v3_neg(nref);
        box_face_contact(B, A, bAxis, sgn, margin, m->normal, m, ws);
    } else {
        const float sgn = sel3(tA, aAxis) < 0.0f ? -1.0f : 1.0f;
        m->normal = v3_scale(m33_col(&A->R, aAxis), sgn); /* A -> B */
        box_face_contact(A, B, aAxis, sgn, margin, m->normal, m, ws);
    }
}

/* body-body manifold for the ordered pair (A = lower index, B = higher index) */
PHYS_HD void collide_pair(const geom_t* A, const geom_t* B, float margin, manifold_t* m, clip_ws_t* ws) {
    manifold_clear(m);
    if (A->type == PHYS_SPEC_SHAPE_SPHERE && B->type == PHYS_SPEC_SHAPE_SPHERE) {
        collide_sphere_sphere(A, B, margin, m);
    } else if (A->type == PHYS_SPEC_SHAPE_SPHERE && B->type == PHYS_SPEC_SHAPE_BOX) {
        collide_sphere_box_raw(A, B, margin, m);
    } else if (A->type == PHYS_SPEC_SHAPE_BOX && B->type == PHYS_SPEC_SHAPE_SPHERE) {
        collide_sphere_box_raw(B, A, margin, m);
        m->normal = v3_neg(m->normal);
    } else if (A->type == PHYS_SPEC_SHAPE_BOX && B->type == PHYS_SPEC_SHAPE_BOX) {
        collide_box_box(A, B, margin, m, ws);
    }
}

/* body against the ground plane y = ground (normal +y). Manifold has A = body, B = ground, so the
 * A -> B normal is (0,-1,0). */
PHYS_HD void collide_ground(const geom_t* A, float ground, float margin, manifold_t* m, clip_ws_t* ws) {
    manifold_clear(m);
    m->normal = v3_make(0.0f, -1.0f, 0.0f);
    if (A->type == PHYS_SPEC_SHAPE_SPHERE) {
        const float bottom = A->c.y - A->h.x;
        const float depth = ground - bottom;
        if (depth < -margin) return;
        m->count = 1;
        m->pt[0] = v3_make(A->c.x, bottom + 0.5f * depth, A->c.z);
        m->depth[0] = depth;
    } else if (A->type == PHYS_SPEC_SHAPE_BOX) {
        v3* cand = ws->poly; float* cdep = ws->depth; int nc = 0;
        const v3 ex = v3_scale(m33_col(&A->R, 0), A->h.x);
        const v3 ey = v3_scale(m33_col(&A->R, 1), A->h.y);
        const v3 ez = v3_scale(m33_col(&A->R, 2), A->h.z);
        for (int k = 0; k < 8; ++k) {
            v3 p = A->c;
            p = (k & 1) ? v3_add(p, ex) : v3_sub(p, ex);
            p = (k & 2) ? v3_add(p, ey) : v3_sub(p, ey);
            p = (k & 4) ? v3_add(p, ez) : v3_sub(p, ez);
            const float depth = ground - p.y;
            if (depth >= -margin) { cand[nc] = p; cdep[nc] = depth; ++nc; }
        }
        if (nc == 0) return;
        manifold_reduce(cand, cdep, nc, v3_make(0.0f, 1.0f, 0.0f), m);
    }
}

#endif /* PHYS_SPEC_COLLIDE_H */
