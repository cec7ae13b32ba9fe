// physics_state.hpp — C++ host-side mirror of the reference's physics module surface, over the C ABI
// of libphysics_hip.so (include/physics_hip.h). Header-only; link with -lphysics_hip.
//
// The reference is Rust and no Rust toolchain exists on the build machines, so the host side above the
// C ABI is written in C++ with the reference's own names, argument meaning and error behaviour:
//
//   reference (src/physics.rs, src/physics/*.rs)            here
//   -------------------------------------------------       ------------------------------------------
//   rigid_body::RigidBody {pub position, rotation, ...}     physics::rigid_body::RigidBody (same fields)
//   RigidBody::new(index)                                   RigidBody::new_(index)
//   RigidBody::apply_force_{centre_of_gravity,at_position,at_offset}   same names
//   Entity { body, instance }                               physics::Entity
//   constraints::Constraints::{FixedPosition,FixedOrientation}          physics::constraints::Constraints
//   ConstraintSolver { constraints: Vec<Constraints> }      physics::constraints::ConstraintSolver
//   PhysicsState { entities, constraint_solver, .. }        physics::PhysicsState
//   PhysicsState::update(&mut self, dt: &Duration)          update(const Duration&)
//   PhysicsState::apply_gravity / step                      same names
//   panics (unwrap / index / assert)                        physics::Panic exception (never crosses the C ABI)
//
// Callers of the reference mutate bodies through pub fields between frames (lib.rs:21-22). The mirror
// keeps that contract: before each device call it compares the host-visible bodies with the snapshot it
// downloaded last and re-uploads what changed; after the call it writes the new state back into
// `entities`, so a renderer reading body.position / body.rotation (physics.rs:64-65) stays untouched.
#pragma once
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "physics_hip.h"

namespace physics {

using Duration = std::chrono::nanoseconds;  // std::time::Duration; whole nanoseconds cross the ABI (quirk Q7)

struct Panic : std::runtime_error {
    int32_t code;
    Panic(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Vector3 {
    float x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    bool operator==(const Vector3& o) const { return x == o.x && y == o.y && z == o.z; }
};

struct UnitQuaternion {  // nalgebra storage order [i, j, k, w]
    float i = 0, j = 0, k = 0, w = 1;
    // UnitQuaternion::from_euler_angles(roll, pitch, yaw) (lib.rs:22)
    static UnitQuaternion from_euler_angles(float roll, float pitch, float yaw) {
        const float sr = std::sin(roll * 0.5f), cr = std::cos(roll * 0.5f);
        const float sp = std::sin(pitch * 0.5f), cp = std::cos(pitch * 0.5f);
        const float sy = std::sin(yaw * 0.5f), cy = std::cos(yaw * 0.5f);
        UnitQuaternion q;
        q.w = cr * cp * cy + sr * sp * sy;
        q.i = sr * cp * cy - cr * sp * sy;
        q.j = cr * sp * cy + sr * cp * sy;
        q.k = cr * cp * sy - sr * sp * cy;
        return q;
    }
};

namespace rigid_body {
struct RigidBody {  // rigid_body.rs:5-21
    float mass = 1.0f;
    Vector3 lin_velocity, angular_velocity;
    Vector3 force, torque;
    std::array<float, 9> inertia_tensor{{1, 0, 0, 0, 1, 0, 0, 0, 1}};  // row-major
    Vector3 position;
    UnitQuaternion rotation;
    size_t index = 0;
    // new: collision shape (the reference has none; PHYS_SHAPE_NONE keeps reference behaviour)
    uint32_t shape_type = PHYS_SHAPE_NONE;
    Vector3 half_extent;

    static RigidBody new_(size_t index) { RigidBody b; b.index = index; return b; }  // rigid_body.rs:64-76
    void apply_force_centre_of_gravity(const Vector3& f) { force = add(force, f); }       // :43-45
    void apply_force_at_position(const Vector3& f, const Vector3& point) {                  // :47-54
        torque = add(torque, cross(Vector3(point.x - position.x, point.y - position.y, point.z - position.z), f));
        force = add(force, f);
    }
    void apply_force_at_offset(const Vector3& f, const Vector3& offset) {                   // :55-62
        torque = add(torque, cross(offset, f));
        force = add(force, f);
    }

  private:
    static Vector3 add(const Vector3& a, const Vector3& b) { return Vector3(a.x + b.x, a.y + b.y, a.z + b.z); }
    static Vector3 cross(const Vector3& a, const Vector3& b) {
        return Vector3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    }
};
}  // namespace rigid_body

namespace constraints {
struct FixToPointConstraint { size_t rigid_body; Vector3 position; };        // fixed_position_constraint.rs:8-11
struct FixedOrientationConstraint { size_t rigid_body; Vector3 position; };  // fixed_orientation_constraint.rs:9-12
struct Constraints {                                                         // constraints.rs:33-37
    enum Kind { FixedPosition, FixedOrientation } kind;
    size_t rigid_body;
    Vector3 position;
    static Constraints FixedPositionOf(const FixToPointConstraint& c) { return {FixedPosition, c.rigid_body, c.position}; }
    static Constraints FixedOrientationOf(const FixedOrientationConstraint& c) { return {FixedOrientation, c.rigid_body, c.position}; }
};
struct ConstraintSolver { std::vector<Constraints> constraints; };           // constraints.rs:61-63
}  // namespace constraints

struct Entity {  // physics.rs:16-19
    rigid_body::RigidBody body;
    uint32_t instance = 0;
};

class PhysicsState {  // physics.rs:25-31
  public:
    std::vector<Entity> entities;
    constraints::ConstraintSolver constraint_solver;

    explicit PhysicsState(const phys_config* cfg = nullptr) {
        phys_config c;
        if (cfg) c = *cfg; else phys_config_default(&c);
        check(phys_create(&c, &w_));
    }
    ~PhysicsState() { if (w_) phys_destroy(w_); }
    PhysicsState(const PhysicsState&) = delete;
    PhysicsState& operator=(const PhysicsState&) = delete;

    // physics.rs:41-55
    void update(const Duration& dt) {
        push();
        check(phys_update(w_, (uint64_t)dt.count()));
        pull();
    }
    // physics.rs:87-94
    void apply_gravity() {
        push();
        check(phys_apply_gravity(w_));
        pull_forces();
    }
    // physics.rs:95-99
    void step(const Duration& dt) {
        push();
        check(phys_step(w_, (uint64_t)dt.count()));
        pull();
    }
    // previous_solution (physics.rs:30): empty vector == None
    std::vector<float> previous_solution() {
        uint64_t n = 0;
        check(phys_get_lambda(w_, nullptr, 0, &n));
        std::vector<float> l(n);
        if (n) check(phys_get_lambda(w_, l.data(), n, &n));
        return l;
    }
    // what get_render_data feeds the renderer: Instance::to_raw per entity (physics.rs:61-69, graphics.rs:13-21)
    std::vector<std::array<float, 16>> instance_matrices() {
        push();
        std::vector<std::array<float, 16>> m(entities.size());
        if (!m.empty()) check(phys_get_instance_matrices(w_, &m[0][0]));
        return m;
    }
    phys_world* raw() { return w_; }

  private:
    phys_world* w_ = nullptr;
    std::vector<Entity> snapshot_;
    std::vector<constraints::Constraints> con_snapshot_;
    bool uploaded_ = false;

    static void check(int32_t rc) {
        if (rc != PHYS_OK) throw Panic(rc, phys_last_error());
    }
    static bool same_body(const rigid_body::RigidBody& a, const rigid_body::RigidBody& b) {
        return a.mass == b.mass && a.lin_velocity == b.lin_velocity && a.angular_velocity == b.angular_velocity &&
               a.force == b.force && a.torque == b.torque && a.inertia_tensor == b.inertia_tensor &&
               a.position == b.position && a.rotation.i == b.rotation.i && a.rotation.j == b.rotation.j &&
               a.rotation.k == b.rotation.k && a.rotation.w == b.rotation.w && a.shape_type == b.shape_type &&
               a.half_extent == b.half_extent;
    }
    // host -> device for whatever the caller changed since the last pull
    void push() {
        const size_t n = entities.size();
        bool bodies_changed = !uploaded_ || snapshot_.size() != n;
        bool forces_changed = false;
        for (size_t i = 0; i < n && !bodies_changed; ++i) {
            rigid_body::RigidBody a = entities[i].body, b = snapshot_[i].body;
            if (!(a.force == b.force) || !(a.torque == b.torque)) forces_changed = true;
            a.force = b.force; a.torque = b.torque;
            if (!same_body(a, b)) bodies_changed = true;
        }
        if (bodies_changed) {
            std::vector<float> pos(3 * n), rot(4 * n), lin(3 * n), ang(3 * n), mass(n), inertia(9 * n), he(3 * n);
            std::vector<uint32_t> st(n);
            for (size_t i = 0; i < n; ++i) {
                const rigid_body::RigidBody& b = entities[i].body;
                pos[3 * i] = b.position.x; pos[3 * i + 1] = b.position.y; pos[3 * i + 2] = b.position.z;
                rot[4 * i] = b.rotation.i; rot[4 * i + 1] = b.rotation.j; rot[4 * i + 2] = b.rotation.k; rot[4 * i + 3] = b.rotation.w;
                lin[3 * i] = b.lin_velocity.x; lin[3 * i + 1] = b.lin_velocity.y; lin[3 * i + 2] = b.lin_velocity.z;
                ang[3 * i] = b.angular_velocity.x; ang[3 * i + 1] = b.angular_velocity.y; ang[3 * i + 2] = b.angular_velocity.z;
                mass[i] = b.mass;
                for (int k = 0; k < 9; ++k) inertia[9 * i + k] = b.inertia_tensor[k];
                st[i] = b.shape_type;
                he[3 * i] = b.half_extent.x; he[3 * i + 1] = b.half_extent.y; he[3 * i + 2] = b.half_extent.z;
            }
            check(phys_set_bodies(w_, n, pos.data(), rot.data(), lin.data(), ang.data(), mass.data(), inertia.data(), st.data(), he.data()));
            forces_changed = true;
            con_snapshot_.clear();
            uploaded_ = true;
        }
        if (forces_changed && n) {
            std::vector<float> f(3 * n), t(3 * n);
            for (size_t i = 0; i < n; ++i) {
                const rigid_body::RigidBody& b = entities[i].body;
                f[3 * i] = b.force.x; f[3 * i + 1] = b.force.y; f[3 * i + 2] = b.force.z;
                t[3 * i] = b.torque.x; t[3 * i + 1] = b.torque.y; t[3 * i + 2] = b.torque.z;
            }
            check(phys_set_forces(w_, f.data(), t.data()));
        }
        // constraint list
        const auto& cs = constraint_solver.constraints;
        bool same = bodies_changed ? false : cs.size() == con_snapshot_.size();
        for (size_t k = 0; same && k < cs.size(); ++k)
            same = cs[k].kind == con_snapshot_[k].kind && cs[k].rigid_body == con_snapshot_[k].rigid_body &&
                   cs[k].position == con_snapshot_[k].position;
        if (!same) {
            check(phys_clear_constraints(w_));
            for (const auto& c : cs) {
                const float t[3] = {c.position.x, c.position.y, c.position.z};
                if (c.kind == constraints::Constraints::FixedPosition) check(phys_add_constraint_fix_point(w_, c.rigid_body, t));
                else check(phys_add_constraint_fix_orientation(w_, c.rigid_body, t));
            }
            con_snapshot_ = cs;
        }
        snapshot_ = entities;
    }
    void pull_forces() {
        const size_t n = entities.size();
        if (!n) return;
        std::vector<float> f(3 * n), t(3 * n);
        check(phys_get_forces(w_, f.data(), t.data()));
        for (size_t i = 0; i < n; ++i) {
            entities[i].body.force = Vector3(f[3 * i], f[3 * i + 1], f[3 * i + 2]);
            entities[i].body.torque = Vector3(t[3 * i], t[3 * i + 1], t[3 * i + 2]);
        }
        snapshot_ = entities;
    }
    void pull() {
        const size_t n = entities.size();
        if (!n) return;
        std::vector<float> pos(3 * n), rot(4 * n), lin(3 * n), ang(3 * n);
        check(phys_get_transforms(w_, pos.data(), rot.data()));
        check(phys_get_velocities(w_, lin.data(), ang.data()));
        for (size_t i = 0; i < n; ++i) {
            rigid_body::RigidBody& b = entities[i].body;
            b.position = Vector3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
            b.rotation.i = rot[4 * i]; b.rotation.j = rot[4 * i + 1]; b.rotation.k = rot[4 * i + 2]; b.rotation.w = rot[4 * i + 3];
            b.lin_velocity = Vector3(lin[3 * i], lin[3 * i + 1], lin[3 * i + 2]);
            b.angular_velocity = Vector3(ang[3 * i], ang[3 * i + 1], ang[3 * i + 2]);
            b.force = Vector3();   // rigid_body.rs:38-39
            b.torque = Vector3();
        }
        snapshot_ = entities;
    }
};

}  // namespace physics
