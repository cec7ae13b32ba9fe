// collide_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code). See collide_oracle.hpp.
#include "collide_oracle.hpp"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <numeric>

namespace oracle {

void CollisionWorld::configure(const phys_config& cfg) {
    flags = cfg.flags;
    margin = cfg.contact_margin;
    ground = cfg.ground_height;
    iterations = cfg.solver_iterations;
    sp.dt = 0.0f;
    sp.baumgarte = cfg.baumgarte;
    sp.slop = cfg.slop;
    sp.friction = cfg.friction;
    sp.max_bias = cfg.max_bias;
    warm_start = !(cfg.flags & PHYS_FLAG_NO_WARM_START);
}

static inline v3 V(const float* p) { return v3_make(p[0], p[1], p[2]); }
static inline quat Q(const float* p) { quat q; q.i = p[0]; q.j = p[1]; q.k = p[2]; q.w = p[3]; return q; }

void CollisionWorld::compute_aabbs(const std::vector<RigidBody>& bodies) {
    const size_t n = bodies.size();
    aabb.resize(6 * n);
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (long long ii = 0; ii < (long long)n; ++ii) {
        const size_t i = (size_t)ii;
        const aabb_t b = body_aabb(V(bodies[i].position), Q(bodies[i].rotation), V(&half_extent[3 * i]), shape_type[i], margin);
        aabb[6 * i + 0] = b.lo.x; aabb[6 * i + 1] = b.lo.y; aabb[6 * i + 2] = b.lo.z;
        aabb[6 * i + 3] = b.hi.x; aabb[6 * i + 4] = b.hi.y; aabb[6 * i + 5] = b.hi.z;
    }
}

static inline aabb_t box_at(const std::vector<float>& aabb, size_t i) {
    aabb_t b;
    b.lo = v3_make(aabb[6 * i], aabb[6 * i + 1], aabb[6 * i + 2]);
    b.hi = v3_make(aabb[6 * i + 3], aabb[6 * i + 4], aabb[6 * i + 5]);
    return b;
}

void CollisionWorld::broadphase_sweep() {
    const size_t n = aabb.size() / 6;
    pairs.clear();
    std::vector<uint32_t> order;
    for (size_t i = 0; i < n; ++i)
        if (shape_type[i] != PHYS_SHAPE_NONE) order.push_back((uint32_t)i);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const float lx = aabb[6 * x], ly = aabb[6 * y];
        return lx < ly || (lx == ly && x < y);
    });
    for (size_t s = 0; s < order.size(); ++s) {
        const uint32_t i = order[s];
        const aabb_t bi = box_at(aabb, i);
        for (size_t t = s + 1; t < order.size(); ++t) {
            const uint32_t j = order[t];
            if (aabb[6 * j] > bi.hi.x) break;
            if (aabb_overlap(bi, box_at(aabb, j))) pairs.emplace_back(std::min(i, j), std::max(i, j));
        }
    }
    std::sort(pairs.begin(), pairs.end());
}

void CollisionWorld::broadphase_grid() {
    const size_t n = aabb.size() / 6;
    pairs.clear();
    // cell = largest AABB extent: overlapping boxes then have centres in adjacent cells
    float cell = 0.0f;
    std::vector<uint32_t> ids;
    for (size_t i = 0; i < n; ++i) {
        if (shape_type[i] == PHYS_SHAPE_NONE) continue;
        ids.push_back((uint32_t)i);
        for (int k = 0; k < 3; ++k) cell = std::max(cell, aabb[6 * i + 3 + k] - aabb[6 * i + k]);
    }
    if (ids.empty()) return;
    if (!(cell > 0.0f)) cell = 1.0f;
    const double inv = 1.0 / (double)cell;
    auto cell_of = [&](uint32_t i, int k) -> int64_t {
        const double c = 0.5 * ((double)aabb[6 * i + k] + (double)aabb[6 * i + 3 + k]);
        return (int64_t)std::floor(c * inv);
    };
    auto key_of = [](int64_t x, int64_t y, int64_t z) -> uint64_t {
        const uint64_t B = 1ull << 20;
        return ((uint64_t)(x + (int64_t)B) << 42) | ((uint64_t)(y + (int64_t)B) << 21) | (uint64_t)(z + (int64_t)B);
    };
    std::vector<std::pair<uint64_t, uint32_t>> keyed;
    keyed.reserve(ids.size());
    for (uint32_t i : ids) keyed.emplace_back(key_of(cell_of(i, 0), cell_of(i, 1), cell_of(i, 2)), i);
    std::sort(keyed.begin(), keyed.end());
    // neighbour search per body is independent; the final sort fixes the order whatever the thread count
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> found((size_t)(threads > 1 ? threads : 1));
#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        std::vector<std::pair<uint32_t, uint32_t>>& out = found[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
        for (long long q = 0; q < (long long)ids.size(); ++q) {
            const uint32_t i = ids[(size_t)q];
            const int64_t cx = cell_of(i, 0), cy = cell_of(i, 1), cz = cell_of(i, 2);
            const aabb_t bi = box_at(aabb, i);
            for (int64_t dx = -1; dx <= 1; ++dx)
                for (int64_t dy = -1; dy <= 1; ++dy)
                    for (int64_t dz = -1; dz <= 1; ++dz) {
                        const uint64_t key = key_of(cx + dx, cy + dy, cz + dz);
                        auto it = std::lower_bound(keyed.begin(), keyed.end(), std::make_pair(key, (uint32_t)0));
                        for (; it != keyed.end() && it->first == key; ++it) {
                            const uint32_t j = it->second;
                            if (j > i && aabb_overlap(bi, box_at(aabb, j))) out.emplace_back(i, j);
                        }
                    }
        }
    }
    for (const auto& part : found) pairs.insert(pairs.end(), part.begin(), part.end());
    std::sort(pairs.begin(), pairs.end());
}

static inline geom_t geom_of(const RigidBody& b, const float* h, uint32_t type) {
    return geom_make(V(b.position), Q(b.rotation), V(h), type);
}

void CollisionWorld::narrowphase(const std::vector<RigidBody>& bodies) {
    manifolds.clear();
    n_contacts = 0;
    // every pair / ground test is independent: results go to a slot of their own and are compacted in list order
    // afterwards, so the manifold order (and everything downstream) is the same for any thread count
    const size_t P = pairs.size();
    const size_t G = (flags & PHYS_FLAG_GROUND_PLANE) ? bodies.size() : 0;
    std::vector<manifold_t> found(P + G);
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (long long k = 0; k < (long long)(P + G); ++k) {
        manifold_t& m = found[(size_t)k];
        clip_ws_t ws;
        m.count = 0;
        if ((size_t)k < P) {
            const uint32_t a = pairs[(size_t)k].first, b = pairs[(size_t)k].second;
            const geom_t ga = geom_of(bodies[a], &half_extent[3 * a], shape_type[a]);
            const geom_t gb = geom_of(bodies[b], &half_extent[3 * b], shape_type[b]);
            collide_pair(&ga, &gb, margin, &m, &ws);
        } else {
            const size_t i = (size_t)k - P;
            if (shape_type[i] == PHYS_SHAPE_NONE) continue;
            const geom_t ga = geom_of(bodies[i], &half_extent[3 * i], shape_type[i]);
            collide_ground(&ga, ground, margin, &m, &ws);
        }
    }
    for (size_t k = 0; k < P + G; ++k) {
        const manifold_t& m = found[k];
        if (m.count <= 0) continue;
        const uint32_t a = k < P ? pairs[k].first : (uint32_t)(k - P);
        const uint32_t b = k < P ? pairs[k].second : PHYS_GROUND_ID;
        Manifold M{a, b, m.normal, m.count, {m.pt[0], m.pt[1], m.pt[2], m.pt[3]}, {m.depth[0], m.depth[1], m.depth[2], m.depth[3]}};
        manifolds.push_back(M);
        n_contacts += (uint64_t)m.count;
    }
}

// Colouring of one step (include/spec/contact_solve.h "persistent colouring"): a manifold that existed in
// the previous update keeps its colour; only the new ones go through the Jones-Plassmann rounds, against
// `used` masks pre-seeded with the kept colours (never re-compacted: contact_solve.h). `persistent = false` (the
// collide_now test hook) neither reads nor updates the cache.
void CollisionWorld::color_manifolds(size_t n_bodies, bool persistent) {
    const size_t M = manifolds.size();
    const uint32_t UNCOLORED = 0xFFFFFFFFu;
    color.assign(M, UNCOLORED);
    std::vector<uint64_t> used(n_bodies, 0ull), top(n_bodies, 0ull), prio(M);
    for (size_t m = 0; m < M; ++m) prio[m] = color_priority(manifolds[m].a, manifolds[m].b);
    size_t remaining = M;
    n_colors = 0;
    color_rounds = 0;
    const bool keep = persistent;  // the cache of the first update is empty
    if (keep) {
        for (size_t m = 0; m < M; ++m) {
            const Manifold& mf = manifolds[m];
            auto it = color_cache.find(((uint64_t)mf.a << 32) | mf.b);
            if (it == color_cache.end()) continue;
            color[m] = it->second;
            used[mf.a] |= 1ull << it->second;
            if (mf.b != PHYS_GROUND_ID) used[mf.b] |= 1ull << it->second;
            n_colors = std::max(n_colors, it->second + 1);
            --remaining;
        }
    }
    std::vector<size_t> winners;
    while (remaining > 0) {
        std::fill(top.begin(), top.end(), 0ull);
        for (size_t m = 0; m < M; ++m) {
            if (color[m] != UNCOLORED) continue;
            const Manifold& mf = manifolds[m];
            top[mf.a] = std::max(top[mf.a], prio[m]);
            if (mf.b != PHYS_GROUND_ID) top[mf.b] = std::max(top[mf.b], prio[m]);
        }
        winners.clear();
        for (size_t m = 0; m < M; ++m) {
            if (color[m] != UNCOLORED) continue;
            const Manifold& mf = manifolds[m];
            if (prio[m] == top[mf.a] && (mf.b == PHYS_GROUND_ID || prio[m] == top[mf.b])) winners.push_back(m);
        }
        for (size_t m : winners) {
            const Manifold& mf = manifolds[m];
            uint64_t mask = used[mf.a];
            if (mf.b != PHYS_GROUND_ID) mask |= used[mf.b];
            uint32_t c = 0;
            while (c < PHYS_MAX_COLORS - 1 && ((mask >> c) & 1ull)) ++c;
            color[m] = c;
            used[mf.a] |= 1ull << c;
            if (mf.b != PHYS_GROUND_ID) used[mf.b] |= 1ull << c;
            n_colors = std::max(n_colors, c + 1);
        }
        remaining -= winners.size();
        ++color_rounds;
    }
    if (persistent) {
        color_cache.clear();
        color_cache.reserve(2 * M);
        for (size_t m = 0; m < M; ++m) color_cache[((uint64_t)manifolds[m].a << 32) | manifolds[m].b] = color[m];
        ++color_epoch;
    }
}

void CollisionWorld::solve(std::vector<RigidBody>& bodies, float dt) {
    const size_t M = manifolds.size();
    sp.dt = dt;
    // world-frame inverse inertia, the reference's convention (quirk Q5): constant, never rotated
    std::vector<m33> inv_inertia(bodies.size());
    std::vector<float> inv_mass(bodies.size());
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (long long ii = 0; ii < (long long)bodies.size(); ++ii) {
        const size_t i = (size_t)ii;
        m33 I;
        for (int k = 0; k < 9; ++k) I.m[k] = bodies[i].inertia_tensor[k];
        if (!m33_try_inverse(&I, &inv_inertia[i]))
            for (int k = 0; k < 9; ++k) inv_inertia[i].m[k] = 0.0f;
        inv_mass[i] = 1.0f / bodies[i].mass;
    }
    std::vector<solver_manifold_t> rows(M);
    const m33 zero{};
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (long long mm = 0; mm < (long long)M; ++mm) {
        const size_t m = (size_t)mm;
        const Manifold& mf = manifolds[m];
        manifold_t g;
        g.normal = mf.normal; g.count = mf.count;
        for (int k = 0; k < 4; ++k) { g.pt[k] = mf.pt[k]; g.depth[k] = mf.depth[k]; }
        const int has_b = mf.b != PHYS_GROUND_ID;
        solver_prep(&g, has_b, V(bodies[mf.a].position), has_b ? V(bodies[mf.b].position) : v3_make(0, 0, 0),
                    inv_mass[mf.a], &inv_inertia[mf.a], has_b ? inv_mass[mf.b] : 0.0f,
                    has_b ? &inv_inertia[mf.b] : &zero, &sp, &rows[m]);
        if (warm_start) {  // a manifold of the same pair in the previous update: start from what its solve ended with
            auto it = warm_cache.find(((uint64_t)mf.a << 32) | mf.b);
            if (it != warm_cache.end()) {
                float pn[4], pt0[4], pt1[4];
                warm_match(&g, &it->second, pn, pt0, pt1);
                for (int k = 0; k < 4; ++k) { rows[m].row[k].pn = pn[k]; rows[m].row[k].pt[0] = pt0[k]; rows[m].row[k].pt[1] = pt1[k]; }
            }
        }
    }
    // colour-major order; inside a colour the order is irrelevant (disjoint bodies)
    std::vector<std::vector<size_t>> by_color(n_colors);
    for (size_t m = 0; m < M; ++m) by_color[color[m]].push_back(m);
    // sweep 0 of a warm-started solve applies the starting impulses; sweeps 1..iterations relax (contact_solve.h)
    const uint32_t first_sweep = warm_start ? 0u : 1u;
    for (uint32_t it = first_sweep; it <= iterations; ++it)
        for (uint32_t c = 0; c < n_colors; ++c) {
            // the manifolds of one colour touch disjoint bodies: any order, any number of threads, same bits
            const std::vector<size_t>& cls = by_color[c];
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1 && cls.size() >= 64)
            for (long long q = 0; q < (long long)cls.size(); ++q) {
                const size_t m = cls[(size_t)q];
                const Manifold& mf = manifolds[m];
                const int has_b = mf.b != PHYS_GROUND_ID;
                RigidBody& A = bodies[mf.a];
                v3 vA = V(A.lin_velocity), wA = V(A.angular_velocity);
                v3 vB = v3_make(0, 0, 0), wB = v3_make(0, 0, 0);
                if (has_b) { vB = V(bodies[mf.b].lin_velocity); wB = V(bodies[mf.b].angular_velocity); }
                solve_manifold_lazy(&rows[m], sp.friction, inv_mass[mf.a], &inv_inertia[mf.a], has_b ? inv_mass[mf.b] : 0.0f,
                                    has_b ? &inv_inertia[mf.b] : &zero, &vA, &wA, &vB, &wB, /*apply_only=*/it == 0u);
                A.lin_velocity[0] = vA.x; A.lin_velocity[1] = vA.y; A.lin_velocity[2] = vA.z;
                A.angular_velocity[0] = wA.x; A.angular_velocity[1] = wA.y; A.angular_velocity[2] = wA.z;
                if (has_b) {
                    RigidBody& B = bodies[mf.b];
                    B.lin_velocity[0] = vB.x; B.lin_velocity[1] = vB.y; B.lin_velocity[2] = vB.z;
                    B.angular_velocity[0] = wB.x; B.angular_velocity[1] = wB.y; B.angular_velocity[2] = wB.z;
                }
            }
        }
    remember_impulses(rows);
}

// what the next update's manifolds of the same pairs start from
void CollisionWorld::remember_impulses(const std::vector<solver_manifold_t>& rows) {
    warm_cache.clear();
    if (!warm_start) return;
    warm_cache.reserve(2 * manifolds.size());
    for (size_t m = 0; m < manifolds.size(); ++m) {
        const Manifold& mf = manifolds[m];
        warm_t w;
        w.normal = mf.normal; w.count = mf.count;
        for (int k = 0; k < 4; ++k) {
            w.pt[k] = mf.pt[k];
            w.pn[k] = rows[m].row[k].pn; w.pt0[k] = rows[m].row[k].pt[0]; w.pt1[k] = rows[m].row[k].pt[1];
        }
        warm_cache[((uint64_t)mf.a << 32) | mf.b] = w;
    }
}

void CollisionWorld::collide_and_solve(std::vector<RigidBody>& bodies, float dt) {
    compute_aabbs(bodies);
    broadphase_grid();
    if (flags & PHYS_FLAG_BROADPHASE_ONLY) { manifolds.clear(); color.clear(); n_colors = 0; n_contacts = 0; return; }
    narrowphase(bodies);
    color_manifolds(bodies.size(), true);
    solve(bodies, dt);
}

std::vector<size_t> CollisionWorld::sorted_manifold_order() const {
    std::vector<size_t> order(manifolds.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        const Manifold& a = manifolds[x];
        const Manifold& b = manifolds[y];
        return a.a < b.a || (a.a == b.a && a.b < b.b);
    });
    return order;
}

}  // namespace oracle
