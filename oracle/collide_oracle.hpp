// collide_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// Sequential drivers for the collision stages. These stages do not exist in the
// reference ("parity unpinned"); the per-body / per-pair / per-manifold arithmetic is the normative
// scalar spec in include/spec/{collide,contact_solve}.h, and this file drives it with plain
// sequential loops and its own broad-phase algorithms (sort-and-sweep, and a std::map-free grid),
// so the HIP pipeline (hashed grid + ballot compaction + parallel colouring) is checked against an
// independently organised computation of the same specification.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../include/physics_hip.h"
#include "../include/spec/contact_solve.h"
#include "ref_physics.hpp"

namespace oracle {

struct Manifold {
    uint32_t a, b;
    v3 normal;
    int count;
    v3 pt[4];
    float depth[4];
};

struct CollisionWorld {
    std::vector<uint32_t> shape_type;
    std::vector<float> half_extent;  // 3n
    uint32_t flags = 0;
    float margin = 0.02f, ground = 0.0f;
    uint32_t iterations = 8;
    int threads = 1;  // > 1: OpenMP over the loops whose iterations are independent (same results, bit for bit)
    solve_params_t sp{};

    std::vector<float> aabb;  // 6n
    std::vector<std::pair<uint32_t, uint32_t>> pairs;  // sorted (i < j)
    std::vector<Manifold> manifolds;
    std::vector<uint32_t> color;
    uint32_t n_colors = 0, color_rounds = 0;
    uint64_t n_contacts = 0;

    void configure(const phys_config& cfg);
    void compute_aabbs(const std::vector<RigidBody>& bodies);
    void broadphase_sweep();  // sort-and-sweep on x (independent check, O(N * layer))
    void broadphase_grid();   // uniform grid, O(N); same pair set
    void narrowphase(const std::vector<RigidBody>& bodies);
    void color_manifolds(size_t n_bodies, bool persistent);
    std::unordered_map<uint64_t, uint32_t> color_cache;  // (a << 32 | b) -> colour of the previous update
    std::unordered_map<uint64_t, warm_t> warm_cache;     // (a << 32 | b) -> what the solve of the previous update ended with
    bool warm_start = true;                              // contact_solve.h "warm starting" (off: PHYS_FLAG_NO_WARM_START)
    uint64_t color_epoch = 0;                            // updates since the bodies were set
    void solve(std::vector<RigidBody>& bodies, float dt);
    void remember_impulses(const std::vector<solver_manifold_t>& rows);
    void collide_and_solve(std::vector<RigidBody>& bodies, float dt);
    std::vector<size_t> sorted_manifold_order() const;
};

}  // namespace oracle
