// demo_scene.cpp — the reference's only scene (src/lib.rs:20-42) driven through the C++ host mirror
// (include/physics_state.hpp) exactly as lib.rs:55-59 drives PhysicsState::update. Prints the state as
// JSON; tests/test_gpu_host_mirror.py compares it with golden vector G1 and with the oracle.
#include <cstdio>

#include "physics_state.hpp"

using namespace physics;

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 1;
    try {
        PhysicsState physics_state;
        rigid_body::RigidBody rigid_body = rigid_body::RigidBody::new_(0);
        rigid_body.position = Vector3(1.0f, 0.0f, 0.0f);
        rigid_body.rotation = UnitQuaternion::from_euler_angles(1.0f, 0.0f, 0.0f);
        physics_state.entities.push_back(Entity{rigid_body, 0});
        physics_state.constraint_solver.constraints.push_back(
            constraints::Constraints::FixedPositionOf({rigid_body.index, Vector3(0, 0, 0)}));
        physics_state.constraint_solver.constraints.push_back(
            constraints::Constraints::FixedOrientationOf({rigid_body.index, Vector3(0, 0, 0)}));
        const Duration dt(16666667);
        for (int f = 0; f < frames; ++f) physics_state.update(dt);
        const auto& b = physics_state.entities[0].body;
        const auto lam = physics_state.previous_solution();
        const auto m = physics_state.instance_matrices();
        std::printf("{\"pos\": [%.9g, %.9g, %.9g], \"rot_ijkw\": [%.9g, %.9g, %.9g, %.9g], \"lambda\": [", b.position.x,
                    b.position.y, b.position.z, b.rotation.i, b.rotation.j, b.rotation.k, b.rotation.w);
        for (size_t k = 0; k < lam.size(); ++k) std::printf("%s%.9g", k ? ", " : "", lam[k]);
        std::printf("], \"model_col3\": [%.9g, %.9g, %.9g, %.9g]", m[0][12], m[0][13], m[0][14], m[0][15]);
        // mutate through the pub field like a caller of the reference would, then step again
        physics_state.entities[0].body.position = Vector3(2.0f, 0.0f, 0.0f);
        physics_state.entities[0].body.apply_force_at_offset(Vector3(0, 1, 0), Vector3(1, 0, 0));
        physics_state.update(dt);
        std::printf(", \"after_edit_pos\": [%.9g, %.9g, %.9g]}\n", physics_state.entities[0].body.position.x,
                    physics_state.entities[0].body.position.y, physics_state.entities[0].body.position.z);
        // error behaviour: an index past the end is the reference's Vec-index panic
        PhysicsState empty;
        try { empty.update(dt); std::printf("ERROR: update with no bodies did not fail\n"); return 2; }
        catch (const Panic& p) { if (p.code != PHYS_ERR_NO_BODIES) return 3; }
    } catch (const Panic& p) {
        std::fprintf(stderr, "panic %d: %s\n", p.code, p.what());
        return 1;
    }
    return 0;
}
