// ref_physics.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// Scalar f32 restatement of the reference's per-frame physics path, AoS and single-threaded like
// the reference, quirks included. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this. The product (libphysics_hip.so) never links it.
//
// PARITY STATUS: the reference is Rust and no Rust toolchain exists in this container, so it was
// never run here. Its own tests pin only the block-SpMV (sparse_matrix.rs:65-119; G3 in
// tests/golden). Gravity, integration, constraint assembly and CG are "parity unpinned": they follow
// the reference source line by line plus nalgebra 0.32.2's published operation order (Cargo.toml:19;
// crate source absent), and are cross-checked against hand-derived vectors G1/G2.
//
// Follows, line by line:
//   /root/reference/src/physics.rs:41-55, 87-99
//   /root/reference/src/physics/rigid_body.rs:24-76
//   /root/reference/src/physics/constraints.rs:67-176
//   /root/reference/src/physics/constraints/fixed_position_constraint.rs:13-35
//   /root/reference/src/physics/constraints/fixed_orientation_constraint.rs:15-38
//   /root/reference/src/physics/sparse_matrix.rs:16-50
//   /root/reference/src/physics/sle_solver.rs:21-51
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>
#include <vector>

namespace oracle {

// trig provider: libm (what the reference links) or the deterministic double-polynomial set that
// the HIP kernels use (include/spec/det_math.h).
enum class Trig { Libm = 0, Det = 1 };

// rigid_body.rs:5-21
struct RigidBody {
    float mass;
    float lin_velocity[3];
    float angular_velocity[3];
    float force[3];
    float torque[3];
    float inertia_tensor[9];  // row-major
    float position[3];
    float rotation[4];  // [i, j, k, w]
    size_t index;

    static RigidBody new_(size_t index);  // rigid_body.rs:64-76
    // rigid_body.rs:24-40; returns false where the reference panics (singular inertia)
    bool step(float dt, Trig trig, bool exact_rotation);
    // the two halves of step, split so contact impulses can act between them (collision mode);
    // step == step_velocity then step_position, operation for operation
    bool step_velocity(float dt);
    void step_position(float dt, Trig trig, bool exact_rotation);
    void apply_force_centre_of_gravity(const float f[3]);                  // rigid_body.rs:43-45
    void apply_force_at_position(const float f[3], const float point[3]);  // rigid_body.rs:47-54
    void apply_force_at_offset(const float f[3], const float offset[3]);   // rigid_body.rs:55-62
};

// Duration::as_secs_f32 (used at rigid_body.rs:25)
float duration_as_secs_f32(uint64_t nanos);

// sparse_matrix.rs:3-58. Block data is stored row-major here (nalgebra's from_vec is column-major;
// the golden tests transpose accordingly).
struct SparseMatrixBlock {
    size_t i, j, i_length, j_length;
    std::vector<float> data;  // row-major i_length x j_length
};
struct SparseMatrix {
    std::vector<SparseMatrixBlock> blocks;
    size_t nrows, ncols;
    SparseMatrix(size_t r, size_t c) : nrows(r), ncols(c) {}
    void add_block(size_t row, size_t column, size_t nr, size_t nc, std::vector<float> data);
    std::vector<float> multiply_vector(const std::vector<float>& v) const;     // :25-37
    std::vector<float> tr_multiply_vector(const std::vector<float>& v) const;  // :39-50
};

// nalgebra reductions on Dyn vectors
float dyn_dot(const std::vector<float>& a, const std::vector<float>& b);  // 8-accumulator dotc
float dyn_amax(const std::vector<float>& a);

struct CgConfig {
    uint32_t max_iterations = 1000;  // sle_solver.rs:5
    float max_error = 1e-2f;         // :6
    float min_error = 1e-3f;         // :7
};
// sle_solver.rs:21-46. nullopt == None. iterations_out = loop trips executed.
std::optional<std::vector<float>> solve_conjugate_gradient(const SparseMatrix& j,
                                                           const std::vector<float>& inv_masses,
                                                           const std::vector<float>& rhs,
                                                           const std::optional<std::vector<float>>& previous,
                                                           const CgConfig& cfg, uint32_t* iterations_out);

// constraints.rs:33-59 with the two concrete kinds
struct Constraint {
    enum Kind { FixedPosition = 0, FixedOrientation = 1 } kind;
    size_t rigid_body;
    float position[3];
};

struct ConstraintOutput {  // constraints.rs:19-25 (3 x 12 blocks, row-major)
    float c[3];
    float j[3 * 12];
    float j_dot[3 * 12];
    float ks[3];
    float kd[3];
};

// UnitQuaternion::euler_angles (nalgebra) -> (roll, pitch, yaw)
void quat_euler_angles(const float rot_ijkw[4], Trig trig, float out_rpy[3]);
// UnitQuaternion::from_euler_angles (lib.rs:22)
void quat_from_euler_angles(float roll, float pitch, float yaw, Trig trig, float out_ijkw[4]);

struct PhysicsState {  // physics.rs:25-31 without the render-only members
    std::vector<RigidBody> entities;
    std::vector<Constraint> constraints;
    std::optional<std::vector<float>> previous_solution;
    float gravity_force[3] = {0.0f, -9.81f, 0.0f};  // physics.rs:90
    float gravity_offset[3] = {0.0f, 0.0f, 1.5f};   // physics.rs:91
    CgConfig cg;
    Trig trig = Trig::Libm;
    bool exact_rotation = false;
    uint32_t last_cg_iterations = 0;
    bool last_cg_converged = true;

    ConstraintOutput calculate(const Constraint& c) const;  // fixed_*_constraint.rs calculate()
    // constraints.rs:67-169; returns (lambda, J^T lambda) or nullopt
    std::optional<std::pair<std::vector<float>, std::vector<float>>> solve_constraints();
    void apply_gravity();         // physics.rs:87-94
    void constraint_phase();      // physics.rs:43-51 (solve + quirk-Q3 scatter)
    bool step(uint64_t dt_nanos); // physics.rs:95-99
    bool update(uint64_t dt_nanos);  // physics.rs:41-55
};

}  // namespace oracle
