/* contact_solve.h — normative scalar definition of the sequential-impulse contact solver
 *  and of the deterministic manifold colouring that orders it.
 *
 * No reference counterpart exists ("parity unpinned"): the reference's solver is the global CG
 * of constraints.rs / sle_solver.rs, which stays in scope separately. This header is the
 * specification; HIP kernels and the CPU oracle both compile it (see collide.h header).
 *
 * Solver order (what makes a parallel run equal a sequential one):
 *   for iteration in 0..solver_iterations:
 *     for colour in 0..n_colours (ascending):
 *       every manifold of that colour, in any order (they share no body)
 *   inside a manifold: points in index order; per point tangent 1, tangent 2, then normal.
 *
 * A row (point k, direction d) is solved in Jacobian form: with aA = rA x d, aB = rB x d (angular Jacobians),
 * mA = IA aA, mB = IB aB and lA = d invMA, lB = d invMB - all independent of the velocities, computed once per
 * solve of a manifold (solver_jacobians) - the relative velocity along d is (d.vB + aB.wB) - (d.vA + aA.wA) and
 * an impulse lambda changes the velocities by -lA lambda, -mA lambda, +lB lambda, +mB lambda. The sequential
 * part of a row is therefore four dot products, a clamp and four axpys (the latency of a parallel solve is
 * the length of exactly this chain, times the rows of a manifold, times colours, times iterations).
 *
 * Colouring (Jones-Plassmann on the line graph, synchronous rounds; a pure function of the SET of
 * manifolds, independent of their storage order):
 *   priority(m) = mix64(a << 32 | b)           (bijective, so priorities are distinct)
 *   round: top[body] = max priority over that body's uncoloured manifolds;
 *          an uncoloured manifold that is top at every dynamic body it touches takes the lowest
 *          colour not yet used at those bodies, and marks it used there.
 *   repeat until all are coloured. At most PHYS_MAX_COLORS colours.
 *
 * Warm starting (what makes a stack stand): the accumulated impulses a manifold ends an update with are remembered
 * with its contact points, and a manifold of the same pair in the next update starts from them - point k takes the
 * impulses of the first remembered point within PHYS_WARM_DIST of it that no earlier point took (normal impulse when the
 * normals agree to PHYS_WARM_DOT; friction impulses only when the tangent bases do too). The solve then has one sweep
 * more, in front of the others: sweep 0 APPLIES the starting impulses of every manifold to its bodies, colour by colour
 * in solve order (the same row_apply as a relaxation, no clamping, nothing changed in the rows); sweeps 1..iterations
 * relax as before. Eight iterations never carried a tall tower from zero (rounds 1-2: the bottom layer of the 256k tower
 * sank 0.12 in 40 updates and kept sinking); started from the previous update's answer they only have to follow.
 * PHYS_FLAG_NO_WARM_START runs the solver cold (no sweep 0, rows start from zero: the behaviour of rounds 1-2).
 *
 * Persistent colouring (what makes a steady scene cheap): a manifold (a, b) that also existed in the previous
 * update KEEPS its colour; the rounds above run over the NEW manifolds only, with `used` pre-seeded by the kept
 * colours. The colouring is thus a pure function of (previous colouring, manifold set). A new manifold takes the
 * lowest colour free at its two bodies, so colour numbers stay below deg(a) + deg(b) without ever being re-compacted
 * (rounds 1-2 re-coloured everything every 64th update: a 13.8 ms hitch once a second on the 256k tower, for nothing).
 * PHYS_COLOR_CACHE_PERIOD is what is left of that: the period at which the DEVICE's colour table is rebuilt from the
 * live manifolds (dead entries purged) - storage housekeeping that changes no colour.
 */
#ifndef PHYS_SPEC_CONTACT_SOLVE_H
#define PHYS_SPEC_CONTACT_SOLVE_H

#include "collide.h"

#define PHYS_MAX_COLORS 64
#define PHYS_COLOR_CACHE_PERIOD 64
#define PHYS_WARM_DIST 0.05f /* a remembered contact point within this distance (world units) is the same point */
#define PHYS_WARM_DOT 0.999f /* cosine of the angle (2.5 degrees) within which two directions count as the same */

typedef struct {
    float dt;
    float baumgarte;
    float slop;
    float friction;
    float max_bias; /* cap on the push-out velocity */
} solve_params_t;

typedef struct {
    v3 rA, rB;
    float normal_mass;
    float tangent_mass[2];
    float bias;
    float pn;
    float pt[2];
} contact_row_t;

typedef struct {
    v3 n, t1, t2;
    int count;
    int has_b; /* 0: body B is the static ground */
    contact_row_t row[4];
} solver_manifold_t;

PHYS_HD uint64_t color_priority(uint32_t a, uint32_t b) {
    uint64_t z = ((uint64_t)a << 32) | (uint64_t)b;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* orthonormal tangents of a unit normal */
PHYS_HD void tangent_basis(v3 n, v3* t1, v3* t2) {
    v3 t;
    if (det_absf(n.x) >= 0.57735f) t = v3_make(n.y, -n.x, 0.0f);
    else                           t = v3_make(0.0f, n.z, -n.y);
    t = v3_div(t, v3_norm(t));
    *t1 = t;
    *t2 = v3_cross(n, t);
}

/* inverse inertia times a vector in the contact solver. A DIAGONAL tensor (the reference's only case: identity,
 * rigid_body.rs:71) is applied as its three products; the six products with the zero off-diagonals that the general
 * product adds could only change the sign of a zero result (for finite operands). The rule is per tensor, so the
 * oracle and every solver kernel apply it alike, whatever else the world holds. */
/* ---- fused multiply-add in the solver's row arithmetic (round 3). The reference has no contact solver, so nothing of
 * nalgebra's operation order binds these rows (vec.h keeps that order for everything the reference does have); a row is
 * ~1000 dependent f32 operations and the solver kernels are bound by exactly that chain. a*b + c with ONE rounding is
 * v_fma_f32 on the device and the x86 FMA instruction on the host (the oracle is built with -mfma; without it the
 * compiler calls libm's fmaf, which is the same correctly rounded value): the same bits on both sides, a third fewer
 * operations. Everything below that multiplies and adds says so explicitly through these four helpers; the build keeps
 * -ffp-contract=off, so nothing else is ever fused. */
PHYS_HD float spec_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
/* x first, then y, then z - each product joins the sum with one rounding */
PHYS_HD float v3_dot_f(v3 a, v3 b) { return spec_fma(a.z, b.z, spec_fma(a.y, b.y, a.x * b.x)); }
PHYS_HD v3 v3_cross_f(v3 a, v3 b) {
    return v3_make(spec_fma(a.y, b.z, -(a.z * b.y)), spec_fma(a.z, b.x, -(a.x * b.z)), spec_fma(a.x, b.y, -(a.y * b.x)));
}
/* a + b * s */
PHYS_HD v3 v3_madd(v3 a, v3 b, float s) { return v3_make(spec_fma(b.x, s, a.x), spec_fma(b.y, s, a.y), spec_fma(b.z, s, a.z)); }

PHYS_HD int m33_is_diagonal(const m33* M) {
    return M->m[1] == 0.0f && M->m[2] == 0.0f && M->m[3] == 0.0f && M->m[5] == 0.0f && M->m[6] == 0.0f && M->m[7] == 0.0f;
}
PHYS_HD v3 inertia_mul(const m33* I, v3 a) {
    if (m33_is_diagonal(I)) return v3_make(I->m[0] * a.x, I->m[4] * a.y, I->m[8] * a.z);
    return m33_mul_v3(I, a);
}

PHYS_HD float direction_mass(v3 dir, v3 rA, v3 rB, float invMA, const m33* IA, float invMB, const m33* IB, int has_b) {
    const v3 ra = v3_cross_f(rA, dir);
    float k = invMA + v3_dot_f(inertia_mul(IA, ra), ra);
    if (has_b) {
        const v3 rb = v3_cross_f(rB, dir);
        k = (k + invMB) + v3_dot_f(inertia_mul(IB, rb), rb);
    }
    return k > 0.0f ? 1.0f / k : 0.0f;
}

/* velocity the normal row of a point asks for: push out beyond the slop, or let a speculative contact close its gap */
PHYS_HD float contact_bias(float depth, const solve_params_t* sp) {
    float bias = 0.0f;
    if (depth > sp->slop) bias = det_minf((sp->baumgarte / sp->dt) * (depth - sp->slop), sp->max_bias);
    else if (depth < 0.0f) bias = depth / sp->dt; /* speculative: may close the gap, not more */
    return bias;
}

/* build the solver rows of one manifold. xB / invMB / IB are ignored when has_b == 0. */
PHYS_HD void solver_prep(const manifold_t* m, int has_b, v3 xA, v3 xB, float invMA, const m33* IA, float invMB,
                         const m33* IB, const solve_params_t* sp, solver_manifold_t* out) {
    out->n = m->normal;
    tangent_basis(m->normal, &out->t1, &out->t2);
    out->count = m->count;
    out->has_b = has_b;
    for (int k = 0; k < 4; ++k) {
        contact_row_t* r = &out->row[k];
        if (k < m->count) {
            r->rA = v3_sub(m->pt[k], xA);
            r->rB = has_b ? v3_sub(m->pt[k], xB) : v3_make(0.0f, 0.0f, 0.0f);
            r->normal_mass = direction_mass(out->n, r->rA, r->rB, invMA, IA, invMB, IB, has_b);
            r->tangent_mass[0] = direction_mass(out->t1, r->rA, r->rB, invMA, IA, invMB, IB, has_b);
            r->tangent_mass[1] = direction_mass(out->t2, r->rA, r->rB, invMA, IA, invMB, IB, has_b);
            r->bias = contact_bias(m->depth[k], sp);
        } else {
            r->rA = v3_make(0.0f, 0.0f, 0.0f);
            r->rB = v3_make(0.0f, 0.0f, 0.0f);
            r->normal_mass = 0.0f; r->tangent_mass[0] = 0.0f; r->tangent_mass[1] = 0.0f; r->bias = 0.0f;
        }
        r->pn = 0.0f; r->pt[0] = 0.0f; r->pt[1] = 0.0f;
    }
}

/* velocity-independent part of the rows of one manifold; d = 0: t1, 1: t2, 2: n */
typedef struct {
    v3 aA, aB; /* r x dir */
    v3 mA, mB; /* I^-1 (r x dir) */
} jac_row_t;

typedef struct {
    v3 lA[3], lB[3]; /* dir * inverse mass */
    jac_row_t j[4][3];
} solver_jac_t;

PHYS_HD void jac_row_make(const solver_manifold_t* sm, int k, v3 dir, const m33* IA, const m33* IB, jac_row_t* j) {
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    j->aA = v3_cross_f(sm->row[k].rA, dir);
    j->mA = inertia_mul(IA, j->aA);
    if (sm->has_b) {
        j->aB = v3_cross_f(sm->row[k].rB, dir);
        j->mB = inertia_mul(IB, j->aB);
    } else {
        j->aB = zero; j->mB = zero;
    }
}

PHYS_HD void solver_jacobians(const solver_manifold_t* sm, float invMA, const m33* IA, float invMB, const m33* IB,
                              solver_jac_t* J) {
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    PHYS_UNROLL
    for (int d = 0; d < 3; ++d) {
        const v3 dir = d == 0 ? sm->t1 : (d == 1 ? sm->t2 : sm->n);
        J->lA[d] = v3_scale(dir, invMA);
        J->lB[d] = sm->has_b ? v3_scale(dir, invMB) : zero;
    }
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) {
        PHYS_UNROLL
        for (int d = 0; d < 3; ++d) {
            jac_row_t* j = &J->j[k][d];
            if (k < sm->count) {
                jac_row_make(sm, k, d == 0 ? sm->t1 : (d == 1 ? sm->t2 : sm->n), IA, IB, j);
            } else {
                j->aA = zero; j->aB = zero; j->mA = zero; j->mB = zero;
            }
        }
    }
}

/* relative velocity of the contact point along dir (body B minus body A) */
PHYS_HD float row_velocity(v3 dir, const jac_row_t* j, int has_b, v3 vA, v3 wA, v3 vB, v3 wB) {
    /* four separate dots, summed in pairs: the kernels that give a row four lanes keep one of them per lane */
    const float ua = v3_dot_f(dir, vA) + v3_dot_f(j->aA, wA);
    const float ub = has_b ? v3_dot_f(dir, vB) + v3_dot_f(j->aB, wB) : 0.0f; /* the ground does not move */
    return ub - ua;
}

PHYS_HD void row_apply(float lambda, v3 lA, v3 lB, const jac_row_t* j, int has_b, v3* vA, v3* wA, v3* vB, v3* wB) {
    *vA = v3_madd(*vA, lA, -lambda); /* = fma(-lA, lambda, vA): negation is exact */
    *wA = v3_madd(*wA, j->mA, -lambda);
    if (has_b) {
        *vB = v3_madd(*vB, lB, lambda);
        *wB = v3_madd(*wB, j->mB, lambda);
    }
}

/* ---- warm starting: what is remembered of a manifold, and the impulses a new manifold of the same pair starts from */
typedef struct {
    v3 normal;
    int count; /* 0: nothing remembered */
    v3 pt[4];  /* world-space contact points of that update */
    float pn[4], pt0[4], pt1[4]; /* accumulated impulses its solve ended with */
} warm_t;

PHYS_HD void warm_match(const manifold_t* m, const warm_t* w, float* pn, float* pt0, float* pt1) {
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) { pn[k] = 0.0f; pt0[k] = 0.0f; pt1[k] = 0.0f; }
    if (w->count <= 0 || !(v3_dot(m->normal, w->normal) >= PHYS_WARM_DOT)) return;
    /* friction impulses live in the tangent basis of their normal: carried over only if that basis is (nearly) the same */
    v3 a1, a2, b1, b2;
    tangent_basis(m->normal, &a1, &a2);
    tangent_basis(w->normal, &b1, &b2);
    const int same_basis = v3_dot(a1, b1) >= PHYS_WARM_DOT && v3_dot(a2, b2) >= PHYS_WARM_DOT;
    int taken0 = 0, taken1 = 0, taken2 = 0, taken3 = 0; /* (flags, not a mask indexed at run time) */
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) {
        if (k < m->count) {
            int found = 0;
            PHYS_UNROLL
            for (int j = 0; j < 4; ++j) {
                const int taken = j == 0 ? taken0 : (j == 1 ? taken1 : (j == 2 ? taken2 : taken3));
                if (!found && j < w->count && !taken) {
                    const v3 d = v3_sub(m->pt[k], w->pt[j]);
                    if (v3_dot(d, d) <= PHYS_WARM_DIST * PHYS_WARM_DIST) {
                        found = 1;
                        if (j == 0) taken0 = 1; else if (j == 1) taken1 = 1; else if (j == 2) taken2 = 1; else taken3 = 1;
                        pn[k] = w->pn[j];
                        if (same_basis) { pt0[k] = w->pt0[j]; pt1[k] = w->pt1[j]; }
                    }
                }
            }
        }
    }
}

/* one row: friction direction t (0, 1) or the normal (t = 2) of point k */
PHYS_HD void solve_row_dir(solver_manifold_t* sm, int k, int t, const jac_row_t* j, v3 lA, v3 lB, float friction, v3* vA,
                           v3* wA, v3* vB, v3* wB, int apply_only) {
    contact_row_t* r = &sm->row[k];
    const int has_b = sm->has_b;
    if (apply_only) { /* sweep 0 of a warm-started solve: the row's starting impulse reaches the bodies, nothing else moves */
        row_apply(t == 0 ? r->pt[0] : (t == 1 ? r->pt[1] : r->pn), lA, lB, j, has_b, vA, wA, vB, wB);
        return;
    }
    if (t < 2) {
        const v3 dir = t == 0 ? sm->t1 : sm->t2;
        const float vt = row_velocity(dir, j, has_b, *vA, *wA, *vB, *wB);
        float lambda = -r->tangent_mass[t] * vt;
        const float maxf = friction * r->pn;
        const float old = r->pt[t];
        const float np = det_maxf(-maxf, det_minf(old + lambda, maxf));
        lambda = np - old;
        r->pt[t] = np;
        row_apply(lambda, lA, lB, j, has_b, vA, wA, vB, wB);
    } else {
        const float vn = row_velocity(sm->n, j, has_b, *vA, *wA, *vB, *wB);
        float lambda = r->normal_mass * (r->bias - vn);
        const float old = r->pn;
        const float np = det_maxf(old + lambda, 0.0f);
        lambda = np - old;
        r->pn = np;
        row_apply(lambda, lA, lB, j, has_b, vA, wA, vB, wB);
    }
}

/* one Gauss-Seidel sweep over the points of one manifold; velocities are updated in place. Two drivers of the
 * SAME arithmetic: with the Jacobians made beforehand (the sequential part is then as short as it gets) ... */
PHYS_HD void solve_manifold(solver_manifold_t* sm, const solver_jac_t* J, float friction, v3* vA, v3* wA, v3* vB, v3* wB,
                            int apply_only) {
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) {
        if (k < sm->count) {
            PHYS_UNROLL
            for (int t = 0; t < 3; ++t)
                solve_row_dir(sm, k, t, &J->j[k][t], J->lA[t], J->lB[t], friction, vA, wA, vB, wB, apply_only);
        }
    }
}

/* ... or made row by row on the way (one quarter of the registers; same values, same results) */
PHYS_HD void solve_manifold_lazy(solver_manifold_t* sm, float friction, float invMA, const m33* IA, float invMB,
                                 const m33* IB, v3* vA, v3* wA, v3* vB, v3* wB, int apply_only) {
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    v3 lA[3], lB[3];
    PHYS_UNROLL
    for (int d = 0; d < 3; ++d) {
        const v3 dir = d == 0 ? sm->t1 : (d == 1 ? sm->t2 : sm->n);
        lA[d] = v3_scale(dir, invMA);
        lB[d] = sm->has_b ? v3_scale(dir, invMB) : zero;
    }
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) {
        if (k < sm->count) {
            PHYS_UNROLL
            for (int t = 0; t < 3; ++t) {
                jac_row_t j;
                jac_row_make(sm, k, t == 0 ? sm->t1 : (t == 1 ? sm->t2 : sm->n), IA, IB, &j);
                solve_row_dir(sm, k, t, &j, lA[t], lB[t], friction, vA, wA, vB, wB, apply_only);
            }
        }
    }
}

/* ... or with NOTHING made beforehand but the bias: lever arms and Jacobians of a point are remade from the contact
 * point, the two body positions and the mass properties whenever the point is solved, and so are the three row masses
 * in a sweep with make_masses != 0 (the first one), which leaves them in gm->mass for the later sweeps. For a solver
 * that streams its rows from memory once per iteration this is 28 bytes per point instead of 40 and no pass over the
 * bodies to prepare them. The operations are those of solver_prep / direction_mass / jac_row_make on the same inputs,
 * so the values are the same. */
typedef struct {
    v3 n, t1, t2;
    int count;
    int has_b;
    v3 pt[4];      /* world-space contact points */
    float bias[4]; /* contact_bias(depth) */
    float pn[4], pt0[4], pt1[4]; /* accumulated impulses */
    float mass[4][3];            /* row masses of t1, t2, n per point */
} geo_manifold_t;

PHYS_HD void solve_manifold_geo(geo_manifold_t* gm, int make_masses, int apply_only, float friction, v3 xA, float invMA, const m33* IA,
                                v3 xB, float invMB, const m33* IB, v3* vA, v3* wA, v3* vB, v3* wB) {
    const v3 zero = v3_make(0.0f, 0.0f, 0.0f);
    const int has_b = gm->has_b;
    v3 lA[3], lB[3];
    PHYS_UNROLL
    for (int d = 0; d < 3; ++d) {
        const v3 dir = d == 0 ? gm->t1 : (d == 1 ? gm->t2 : gm->n);
        lA[d] = v3_scale(dir, invMA);
        lB[d] = has_b ? v3_scale(dir, invMB) : zero;
    }
    PHYS_UNROLL
    for (int k = 0; k < 4; ++k) {
        if (k < gm->count) {
            const v3 rA = v3_sub(gm->pt[k], xA);                 /* solver_prep */
            const v3 rB = has_b ? v3_sub(gm->pt[k], xB) : zero;
            PHYS_UNROLL
            for (int t = 0; t < 3; ++t) {
                const v3 dir = t == 0 ? gm->t1 : (t == 1 ? gm->t2 : gm->n);
                jac_row_t j;
                j.aA = v3_cross_f(rA, dir);                      /* jac_row_make */
                j.mA = inertia_mul(IA, j.aA);
                if (has_b) {
                    j.aB = v3_cross_f(rB, dir);
                    j.mB = inertia_mul(IB, j.aB);
                } else {
                    j.aB = zero; j.mB = zero;
                }
                if (make_masses) {                               /* direction_mass: its ra, IA ra are aA, mA */
                    float km = invMA + v3_dot_f(j.mA, j.aA);
                    if (has_b) km = (km + invMB) + v3_dot_f(j.mB, j.aB);
                    gm->mass[k][t] = km > 0.0f ? 1.0f / km : 0.0f;
                }
                if (apply_only) {                                /* solve_row_dir, sweep 0 */
                    row_apply(t == 0 ? gm->pt0[k] : (t == 1 ? gm->pt1[k] : gm->pn[k]), lA[t], lB[t], &j, has_b, vA, wA, vB, wB);
                    continue;
                }
                const float mass = gm->mass[k][t];
                const float vrel = row_velocity(dir, &j, has_b, *vA, *wA, *vB, *wB);
                float lambda;
                if (t < 2) {                                     /* solve_row_dir */
                    float* acc = t == 0 ? &gm->pt0[k] : &gm->pt1[k];
                    lambda = -mass * vrel;
                    const float maxf = friction * gm->pn[k];
                    const float old = *acc;
                    const float np = det_maxf(-maxf, det_minf(old + lambda, maxf));
                    lambda = np - old;
                    *acc = np;
                } else {
                    lambda = mass * (gm->bias[k] - vrel);
                    const float old = gm->pn[k];
                    const float np = det_maxf(old + lambda, 0.0f);
                    lambda = np - old;
                    gm->pn[k] = np;
                }
                row_apply(lambda, lA[t], lB[t], &j, has_b, vA, wA, vB, wB);
            }
        }
    }
}

#endif /* PHYS_SPEC_CONTACT_SOLVE_H */
