/* physics_hip.h — C ABI of libphysics_hip.so, the MI355X (gfx950) backend for the per-frame
 * rigid-body step of martingoe/physics.
 *
 * The reference has no FFI of its own; the surface replaced is the inherent-method surface of
 * `PhysicsState` (reference src/physics.rs:40-100) and `RigidBody` (src/physics/rigid_body.rs).
 * Each entry point cites the reference item it stands in for. A Rust shim (rust/physics_hip_sys,
 * shown in INTEGRATION.md) binds exactly these symbols.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - every call returns int32_t: 0 = PHYS_OK, < 0 = error; phys_last_error() gives the text
 *     (thread-local). The reference panics instead (unwrap at rigid_body.rs:31, assert_eq at
 *     sparse_matrix.rs:26,40); a panic cannot cross FFI, so the same conditions become codes.
 *   - a phys_world* is used from one thread at a time (mirrors `&mut self`).
 *   - host arrays passed in are copied before the call returns; output arrays are caller-allocated.
 *   - quaternions are [i, j, k, w] (nalgebra storage order), matrices row-major unless stated.
 *   - there is NO CPU fallback: phys_create fails with PHYS_ERR_NO_DEVICE when no gfx950 device
 *     is usable.
 */
#ifndef PHYSICS_HIP_H
#define PHYSICS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHYS_ABI_VERSION 2u

/* status codes */
#define PHYS_OK 0
#define PHYS_ERR_INVALID_ARG (-1)
#define PHYS_ERR_NO_DEVICE (-2)
#define PHYS_ERR_HIP (-3)
#define PHYS_ERR_SINGULAR_INERTIA (-4) /* reference: try_inverse().unwrap() panic, rigid_body.rs:31 */
#define PHYS_ERR_CAPACITY (-5)         /* pair / manifold / halo buffer overflow, or more than 64 manifolds at one body;
                                          raised in ANY step since the last phys_sync (sticky), cleared by that phys_sync */
#define PHYS_ERR_OUT_OF_RANGE (-6)     /* body index out of range (reference: Vec index panic) */
#define PHYS_ERR_UNSUPPORTED (-7)
#define PHYS_ERR_NO_BODIES (-8)        /* reference: view() panic on N = 0 */

/* shape types (new: the reference has no shapes) */
#define PHYS_SHAPE_NONE 0u   /* takes part in integration only */
#define PHYS_SHAPE_SPHERE 1u /* radius = half_extent[0] */
#define PHYS_SHAPE_BOX 2u    /* half extents along the body axes */

/* phys_config.flags */
#define PHYS_FLAG_COLLISIONS 0x1u     /* run broad-phase + narrow-phase + sequential impulses */
#define PHYS_FLAG_GROUND_PLANE 0x2u   /* static plane y = ground_height, normal +y */
#define PHYS_FLAG_EXACT_ROTATION 0x4u /* OFF (default) = reference quirk Q1: dq = exp(a*sin(th/2)/2) */
#define PHYS_FLAG_BROADPHASE_ONLY 0x8u /* with COLLISIONS: stop after the candidate-pair list */
#define PHYS_FLAG_SOLVER_PER_COLOR 0x10u /* contact solver as one launch per colour class instead of the single-launch
                                            dataflow kernel; same order of updates per body, bit-identical results */
#define PHYS_FLAG_SHARED_GPU 0x20u       /* kept for callers of ABI 2: the guarded start it asked for is the DEFAULT now */
#define PHYS_FLAG_EXCLUSIVE_GPU 0x80u    /* nothing else runs on this GPU while phys_update does (no other stream of the
                                            application, no other process). The cluster solver - one launch whose
                                            workgroups must all be resident - by default starts all-or-nothing (every
                                            workgroup is counted in before anything is written; a launch that does not
                                            fit beside other streams' kernels is called off and tried again: ~0.04
                                            ms per update); with this flag it skips the count, and the dataflow solver of
                                            mid-size scenes may fill the chip as well (three workgroups per CU instead of
                                            a third of that: up to 1.8x faster at 100k-250k manifolds). A world that sets
                                            it on a GPU that IS shared may spin into the solver's 3 s time-out
                                            (PHYS_ERR_HIP at phys_sync). Several worlds of one process on a device are
                                            always guarded. Same results either way. */
#define PHYS_FLAG_SOLVER_CLUSTER 0x40u    /* contact solver: the cluster kernel (body velocities resident in LDS per spatial
                                            cluster, one launch) wherever the scene admits it (>= 32768 bodies, > 40k
                                            manifolds), instead of only where it is the fastest path (>= 170k manifolds).
                                            Bit-identical results; for tests and measurements */
#define PHYS_FLAG_NO_WARM_START 0x100u   /* contact solver: start every update from zero impulses (rounds 1-2). Default: a manifold
                                            that persists starts from the impulses it ended the previous update with
                                            (include/spec/contact_solve.h: warm starting; one sweep more per update) */

typedef struct phys_config {
    uint32_t abi_version;       /* PHYS_ABI_VERSION */
    int32_t device;             /* HIP device ordinal */
    uint32_t flags;             /* default 0: pure reference semantics (no collisions) */
    float gravity_force[3];     /* default (0,-9.81,0): a FORCE, not m*g (physics.rs:90, quirk Q2) */
    float gravity_offset[3];    /* default (0,0,1.5) world offset (physics.rs:91, quirk Q2) */
    uint32_t cg_max_iterations; /* default 1000 (sle_solver.rs:5) */
    float cg_max_error;         /* default 1e-2 (sle_solver.rs:6) */
    float cg_min_error;         /* default 1e-3 (sle_solver.rs:7) */
    uint32_t solver_iterations; /* sequential-impulse iterations, default 8 */
    float baumgarte;            /* default 0.2 */
    float slop;                 /* penetration allowance, default 0.01 */
    float friction;             /* Coulomb coefficient, default 0.5 */
    float contact_margin;       /* AABB fattening / speculative distance, default 0.02 */
    float ground_height;        /* default 0 */
    float max_bias;             /* cap on the contact push-out velocity, default 3.0 */
    uint64_t max_pairs;         /* 0 = auto (24 per body) */
    uint64_t max_manifolds;     /* 0 = auto (17 per body) */
    uint64_t max_ghosts;        /* sharded worlds: room for remote boundary bodies ("ghosts": kinematic copies of bodies owned
                                   by neighbouring ranks, refreshed by phys_halo_exchange before every update); 0 = none */
} phys_config;

typedef struct phys_world phys_world;

typedef struct phys_stats {
    uint64_t n_bodies;     /* owned bodies (ghost slots not counted) */
    uint64_t n_pairs;      /* candidate pairs of the last update */
    uint64_t n_manifolds;  /* body-body + body-ground manifolds */
    uint64_t n_contacts;   /* contact points */
    uint32_t n_colors;     /* solver colours */
    uint32_t color_rounds; /* colouring rounds */
    uint32_t cg_iterations; /* CG iterations of the last constraint solve */
    int32_t cg_converged;   /* 1 = Some(lambda), 0 = None (sle_solver.rs:45) */
    uint64_t steps;         /* updates since creation */
    uint32_t overflow;      /* bit 0 pairs, 1 manifolds, 2 colours (> 64 manifolds at one body), 3 halo / cross pairs,
                               4 solver hand-off timeout, 6 colour table full: the last update's bits OR every bit raised since
                               the last phys_sync */
    uint32_t n_ground_manifolds; /* manifolds against the ground plane (subset of n_manifolds) */
    float max_extent;       /* largest fattened-AABB edge of the last broad phase (the grid cell is 1.001x this) */
    uint32_t n_halo_records; /* records written by the last phys_halo_pack */
    uint64_t n_cross_pairs;  /* cross-rank pairs found by the last phys_halo_pairs */
    uint32_t n_ghosts;       /* ghost slots filled by the last phys_halo_unpack_ghosts */
    uint32_t n_new_manifolds; /* manifolds of the last update whose pair had none in the update before (the colouring's work) */
} phys_stats;

/* reference defaults (see phys_config field comments) */
void phys_config_default(phys_config* cfg);
const char* phys_last_error(void);
uint32_t phys_abi_version(void);

/* PhysicsState { .. } literal at lib.rs:34-42 / drop */
int32_t phys_create(const phys_config* cfg, phys_world** out);
int32_t phys_destroy(phys_world* w);

/* entities: Vec<Entity> (physics.rs:26). One call replaces the whole body set.
 * Any pointer except pos may be NULL -> RigidBody::new defaults (rigid_body.rs:64-76):
 * rot identity, velocities 0, mass 1, inertia identity, shape NONE. */
int32_t phys_set_bodies(phys_world* w, uint64_t n, const float* pos /*3n*/, const float* rot_ijkw /*4n*/,
                        const float* lin_vel /*3n*/, const float* ang_vel /*3n*/, const float* mass /*n*/,
                        const float* inertia /*9n row-major*/, const uint32_t* shape_type /*n*/,
                        const float* half_extent /*3n*/);

/* ConstraintSolver.constraints.push(Constraints::FixedPosition(..)) (lib.rs:24, fixed_position_constraint.rs) */
int32_t phys_add_constraint_fix_point(phys_world* w, uint64_t body, const float target[3]);
/* Constraints::FixedOrientation (lib.rs:25, fixed_orientation_constraint.rs); target = (roll,pitch,yaw) */
int32_t phys_add_constraint_fix_orientation(phys_world* w, uint64_t body, const float target_rpy[3]);
int32_t phys_clear_constraints(phys_world* w);

/* RigidBody::apply_force_centre_of_gravity / apply_force_at_position / apply_force_at_offset
 * (rigid_body.rs:43-62) */
int32_t phys_apply_force_centre_of_gravity(phys_world* w, uint64_t body, const float force[3]);
int32_t phys_apply_force_at_position(phys_world* w, uint64_t body, const float force[3], const float point[3]);
int32_t phys_apply_force_at_offset(phys_world* w, uint64_t body, const float force[3], const float offset[3]);

/* direct writes of body.force / body.torque (pub(crate) fields, rigid_body.rs:12-13): replaces the
 * accumulators of every body; either pointer may be NULL (= leave that array as it is) */
int32_t phys_set_forces(phys_world* w, const float* force /*3n*/, const float* torque /*3n*/);

/* PhysicsState::update(&mut self, dt: &Duration) (physics.rs:41-55). dt is passed as whole
 * nanoseconds so Duration::as_secs_f32 (rigid_body.rs:25) is reproduced bit for bit. */
int32_t phys_update(phys_world* w, uint64_t dt_nanos);
/* PhysicsState::apply_gravity (physics.rs:87-94) */
int32_t phys_apply_gravity(phys_world* w);
/* PhysicsState::step(&mut self, dt) (physics.rs:95-99) = RigidBody::step for every body */
int32_t phys_step(phys_world* w, uint64_t dt_nanos);
/* n updates back to back without returning to the host in between (same result as n phys_update) */
int32_t phys_update_n(phys_world* w, uint64_t dt_nanos, uint32_t n);
/* block until all queued device work of this world is done. Device-side errors are sticky: a capacity overflow or a
 * solver time-out in ANY update since the previous phys_sync (also an early one of a phys_update_n batch) is reported
 * here once - PHYS_ERR_CAPACITY / PHYS_ERR_HIP - and then cleared. The contact solve of an overflowing update is
 * skipped (never run on a truncated contact set); bodies are still integrated.
 * Limit: at most 64 contact manifolds per body (one solver colour each). */
int32_t phys_sync(phys_world* w);

/* body.position / body.rotation reads (physics.rs:64-65); synchronous device-to-host */
int32_t phys_get_transforms(phys_world* w, float* pos_out /*3n*/, float* rot_ijkw_out /*4n*/);
/* body.lin_velocity / body.angular_velocity (pub fields, rigid_body.rs:9-10) */
int32_t phys_get_velocities(phys_world* w, float* lin_out /*3n*/, float* ang_out /*3n*/);
/* body.force / body.torque accumulators (pub(crate), rigid_body.rs:12-13) */
int32_t phys_get_forces(phys_world* w, float* force_out /*3n*/, float* torque_out /*3n*/);
/* Instance::to_raw for every entity (graphics.rs:13-21; consumed at physics.rs:61-69):
 * 16 f32 per body, column-major T(position) * R(rotation) */
int32_t phys_get_instance_matrices(phys_world* w, float* out /*16n*/);
/* previous_solution (physics.rs:30): warm-start lambda; *n_rows = 0 when None */
int32_t phys_get_lambda(phys_world* w, float* lambda_out, uint64_t cap, uint64_t* n_rows);

/* SparseMatrix { add_block, multiply_vector, tr_multiply_vector } (sparse_matrix.rs:16-50) in its GENERAL form, on the
 * device: a block-sparse nrows x ncols matrix given as a list of dense blocks - block b covers rows [i, i + i_length) and
 * columns [j, j + j_length), block_desc = {i, j, i_length, j_length} per block, data = the blocks one after the other,
 * each ROW-major (the reference's from_vec is column-major: the caller transposes) - times a vector: out = M v
 * (transpose == 0: v has ncols entries, out nrows; sparse_matrix.rs:25-37) or out = M^T v (transpose != 0: v has nrows,
 * out ncols; :39-50). Overlapping blocks accumulate IN LIST ORDER and the inner sum of a block row runs left to right, as
 * the reference's loops do, so the result equals the reference's bit for bit (one lane per output entry walks the blocks
 * that touch it, in list order). All pointers are HOST memory; the call uploads, runs the kernel on `device` and
 * downloads. The constraint solve inside phys_update uses the specialised identity-selector form of the same product
 * (the reference's two constraint kinds have nothing else); this entry point is the general one, held to the
 * reference's own unit tests (sparse_matrix.rs:65-119). A block reaching outside the matrix, or a vector of the wrong
 * length (the reference: assert_eq panic, sparse_matrix.rs:26,40) is PHYS_ERR_INVALID_ARG. */
int32_t phys_block_spmv(int32_t device, uint64_t nrows, uint64_t ncols, uint64_t nblocks, const uint64_t* block_desc /*4 per block*/,
                        const float* data, const float* vec, uint64_t vec_len, int32_t transpose, float* out);

/* broad-phase of the current poses: candidate pairs (i < j) with overlapping fattened AABBs,
 * sorted by (i, j). pairs_out may be NULL to query the count. (new: SURVEY §8 A10) */
int32_t phys_broadphase(phys_world* w, uint32_t* pairs_out /*2*cap*/, uint64_t cap, uint64_t* n_pairs);
/* AABBs as computed by the device: min xyz, max xyz per body */
int32_t phys_get_aabbs(phys_world* w, float* out /*6n*/);
/* contact manifolds of the last update, sorted by (body_a, body_b); body_b = 0xFFFFFFFF = ground.
 * rows: per manifold 2 u32 ids + u32 count; per point (4 slots): position xyz + depth. NULLs allowed. */
int32_t phys_get_manifolds(phys_world* w, uint32_t* ids_out /*2*cap*/, uint32_t* counts_out /*cap*/,
                           float* normals_out /*3*cap*/, float* points_out /*16*cap*/, uint64_t cap,
                           uint64_t* n_manifolds);
int32_t phys_get_stats(phys_world* w, phys_stats* out);
/* manifolds per solver colour of the last update (64 entries) */
int32_t phys_get_color_counts(phys_world* w, uint32_t* counts_out /*64*/);

/* --- per-stage device timing (HIP events on the world's stream), for bench.py's roofline --- */
#define PHYS_STAGE_STEP_FULL 0u     /* gravity + RigidBody::step, one kernel (no collisions) */
#define PHYS_STAGE_VELOCITY_AABB 1u /* gravity + velocity half + AABB */
#define PHYS_STAGE_GRID 2u          /* cell assign, scan, scatter */
#define PHYS_STAGE_PAIRS 3u         /* k_find_pairs */
#define PHYS_STAGE_NARROW 4u        /* k_narrowphase */
#define PHYS_STAGE_COLOR 5u         /* colouring rounds */
#define PHYS_STAGE_ROWS 6u          /* colour-major renumbering + solver_prep */
#define PHYS_STAGE_SOLVE 7u         /* k_solve_color */
#define PHYS_STAGE_POSITION 8u      /* position half */
#define PHYS_STAGE_CONSTRAINTS 9u   /* constraint assembly + CG (A3-A7) */
#define PHYS_STAGE_MISC 10u         /* memsets, halo */
#define PHYS_STAGE_SOLVE_TAIL 11u   /* k_solve_tail: the small colours of one iteration in one workgroup */
#define PHYS_STAGE_SOLVE_FLOW 12u   /* k_solve_flow: all iterations and colours in one launch */
#define PHYS_STAGE_SOLVE_CLUSTER 13u /* k_solve_cluster: one launch, body velocities resident in LDS per spatial cluster */
#define PHYS_STAGE_COUNT 14u
typedef struct phys_profile {
    double ms[PHYS_STAGE_COUNT];         /* summed device time per stage since enable */
    uint64_t launches[PHYS_STAGE_COUNT]; /* kernel launches (memsets included) per stage */
    uint64_t steps;                      /* updates profiled */
} phys_profile;
/* on != 0: bracket every launch with events (slower; never inside a timed region). Resets the sums. */
int32_t phys_profile_enable(phys_world* w, int32_t on);
int32_t phys_profile_get(phys_world* w, phys_profile* out);

/* --- device-side access for zero-copy callers (torch / RCCL plumbing) and multi-GPU halos --- */
typedef struct phys_device_view {
    uint64_t n;
    float* pos;     /* 3n */
    float* rot;     /* 4n */
    float* lin_vel; /* 3 floats per body at a stride of vel_stride floats */
    float* ang_vel; /* 3 floats per body at a stride of vel_stride floats */
    float* aabb;    /* 6n: min xyz max xyz, valid after an update with COLLISIONS or phys_broadphase */
    void* stream;   /* hipStream_t the world launches on */
    uint64_t vel_stride; /* = 8: velocities live in 32-byte records {v.xyz, 1/m, w.xyz, m} */
} phys_device_view;
int32_t phys_get_device_view(phys_world* w, phys_device_view* out);

/* Sharded broad-phase, AABB records only (config C4). Each rank owns the bodies it was given; a halo
 * record is 32 B: {min xyz, max xyz, global id, pad}. phys_halo_pack writes to DEVICE memory the records of
 * owned bodies whose fattened AABB reaches outside [x_lo + reach, x_hi - reach] and returns the count;
 * reach must be >= the largest AABB edge on ANY rank (all-reduce phys_stats.max_extent), reach <= 0
 * means this rank's own grid cell. Both calls use the AABBs / grid of the last update or phys_broadphase.
 * phys_halo_pack first fills the whole buffer with 0xFF (empty slot = global id 0xFFFFFFFF). phys_halo_pairs
 * ignores records [skip_first, skip_first + skip_count): the caller's own block of an all-gathered buffer.
 * With n_records / n_cross_pairs == NULL both calls only ENQUEUE work on the world's stream and return
 * (no host synchronisation; phys_get_stats reports the counts later): the per-step exchange then costs no
 * host round trip when the collective is enqueued on the same stream (phys_device_view.stream). phys_halo_pairs takes the gathered records of the OTHER ranks (device
 * memory) and appends owned-vs-remote candidate pairs (local index, global id of the remote body)
 * under the ownership rule "emitted by the rank owning the body with the smaller global id". */
int32_t phys_set_global_ids(phys_world* w, const uint32_t* global_ids /*n*/);
int32_t phys_halo_pack(phys_world* w, float x_lo, float x_hi, float reach, void* dev_records_out, uint64_t cap,
                       uint64_t* n_records);
int32_t phys_halo_pairs(phys_world* w, const void* dev_remote_records, uint64_t n_remote, uint64_t skip_first,
                        uint64_t skip_count, uint64_t* n_cross_pairs);
int32_t phys_get_cross_pairs(phys_world* w, uint32_t* pairs_out /*2*cap*/, uint64_t cap, uint64_t* n_pairs);

/* Sharded worlds with contacts across the cut planes (SURVEY §8 rows E + N4; no reference counterpart: the reference is
 * one thread on one CPU). A world created with phys_config.max_ghosts > 0 keeps max_ghosts GHOST slots behind its owned
 * bodies. Before every update the ranks exchange the full state of their boundary bodies - a 96-byte record {pos, rot,
 * lin vel, ang vel, half extent, shape, global id, inverse mass, inverse inertia diagonal} of every owned body within
 * `reach` of a slab face - and every rank places the neighbours' records that lie within `reach` of ITS slab into its
 * ghost slots. A ghost is a DYNAMIC body of the receiving world for the length of one update: its owner's mass and
 * inertia, this update's gravity, contacts with the ground, with other ghosts and with owned bodies. A contact between an
 * owned body and a body across the plane is therefore solved as the two-body contact it is, on BOTH sides of the plane
 * from the same state, and each side keeps its own body's half of the outcome (what the update did to the ghost is
 * forgotten at the next exchange): for an isolated pair the two halves are equal and opposite up to float rounding
 * (momentum across the plane conserved to ~1e-6), in a pile each side also sees its body's other contacts, which the
 * neighbour sees only as far as its ghosts reach - the N-rank run approximates the single-world run, it does not
 * reproduce its bits. (Round 2 placed ghosts as KINEMATIC bodies - an impact on a ghost was an impact on a moving wall,
 * 2 m v instead of the two-body impulse; a body whose world-frame inertia tensor is not diagonal still crosses that way:
 * the record holds a diagonal.) Slot order is a function of the records alone (both compactions are prefix sums, not
 * atomics), so a sharded run repeats bit for bit.
 *   phys_set_slab            this rank's x-interval and the reach (>= the largest bounding diameter + margin on any rank)
 *   phys_halo_pack_bodies    owned boundary bodies -> 96-byte records in DEVICE memory (unused slots: global id 0xFFFFFFFF)
 *   phys_halo_unpack_ghosts  gathered records (device) -> ghost slots; [skip_first, skip_first+skip_count) = own block
 *   phys_get_global_ids      global ids of the owned bodies followed by those of the ghost slots (0xFFFFFFFF = empty);
 *                            manifold ids >= phys_stats.n_bodies name ghost slots
 * All three device calls only enqueue on the world's stream. */
#define PHYS_HALO_BODY_RECORD_BYTES 96u
int32_t phys_set_slab(phys_world* w, float x_lo, float x_hi, float reach);
int32_t phys_halo_pack_bodies(phys_world* w, void* dev_records_out, uint64_t cap);
/* the bodies within reach of ONE slab face only: face < 0 the low face (what rank r - 1 may need), face > 0 the high
 * face (rank r + 1), 0 both (= phys_halo_pack_bodies). What a neighbour exchange sends (phys_comm_set_neighbours). */
int32_t phys_halo_pack_bodies_face(phys_world* w, void* dev_records_out, uint64_t cap, int32_t face);
int32_t phys_halo_unpack_ghosts(phys_world* w, const void* dev_records, uint64_t n_records, uint64_t skip_first,
                                uint64_t skip_count);
int32_t phys_get_global_ids(phys_world* w, uint32_t* out /* n_bodies + max_ghosts */);

/* The exchange itself behind the C ABI: RCCL (librccl.so, loaded on first use; PHYS_RCCL_PATH overrides the search) over
 * xGMI, one communicator rank per world / GPU, every call enqueued on the world's own stream - a Rust or C++ host shards
 * without Python. Message shape: ONE ncclAllGather of a fixed-size block per rank (latency-bound: no count exchange).
 *   phys_comm_unique_id   rank 0 makes the 128-byte id and ships it to the other ranks by any means
 *   phys_comm_create      ncclCommInitRank on the world's device; `capacity` = records per rank and step
 *   phys_comm_create_local  ONE process driving n worlds on n different devices: ncclCommInitAll (+ phys_halo_exchange_all)
 *   phys_halo_exchange    max_ghosts > 0:  pack bodies -> all-gather -> unpack ghosts   (call BEFORE phys_update)
 *                         otherwise:       pack AABBs  -> all-gather -> cross pairs     (call AFTER phys_update; C4) */
#define PHYS_COMM_ID_BYTES 128u
typedef struct phys_comm phys_comm;
int32_t phys_comm_unique_id(uint8_t id_out[PHYS_COMM_ID_BYTES]);
int32_t phys_comm_create(phys_world* w, const uint8_t id[PHYS_COMM_ID_BYTES], int32_t rank, int32_t n_ranks, uint64_t capacity,
                         phys_comm** out);
int32_t phys_comm_create_local(phys_world** worlds, int32_t n, uint64_t capacity, phys_comm** comms_out /*n*/);
int32_t phys_comm_destroy(phys_comm* c);
/* enable != 0: the ranks are x-slabs ORDERED BY RANK and none is thinner than the reach, so only ranks r - 1 and r + 1 can
 * hold bodies near this rank's faces: the exchange becomes one grouped ncclSend / ncclRecv pair per neighbour (direct xGMI
 * links, two blocks to scan) instead of an all-gather of every rank's block. Every rank of the communicator must choose
 * the same. Default: all-gather (right for any partition). */
int32_t phys_comm_set_neighbours(phys_comm* c, int32_t enable);
int32_t phys_halo_exchange(phys_world* w, phys_comm* c);
/* the n collectives of a step as ONE group: required when one thread drives the comms of phys_comm_create_local */
int32_t phys_halo_exchange_all(phys_world** worlds, phys_comm** comms, int32_t n);

/* Slab partition (host arithmetic only, no device, no communication: the caller sums the histograms of the ranks with
 * whatever transport it has - RCCL, MPI, a pipe). Equal-count cut planes along x, SURVEY §8 row E: histogram of body x
 * over `bins` equal bins of [x_min, x_max] -> prefix sums -> n_ranks + 1 cut planes (cuts[0] = -inf side = x_min,
 * cuts[n_ranks] = x_max; interior planes at bin boundaries... interpolated inside the bin that crosses k/n_ranks of the
 * bodies) -> owner of every body (the rank r with cuts[r] <= x < cuts[r+1]; bodies outside [x_min, x_max] go to the
 * first / last rank). Re-cut every k steps and hand a body that changed owner over to its new rank. */
int32_t phys_slab_histogram(const float* pos /*3n*/, uint64_t n, float x_min, float x_max, uint32_t bins,
                            uint64_t* hist /*bins, added to*/);
int32_t phys_slab_cuts(const uint64_t* hist /*bins, summed over ranks*/, uint32_t bins, float x_min, float x_max, int32_t n_ranks,
                       float* cuts_out /*n_ranks + 1*/);
int32_t phys_slab_owners(const float* pos /*3n*/, uint64_t n, const float* cuts /*n_ranks + 1*/, int32_t n_ranks,
                         int32_t* owner_out /*n*/);

#ifdef __cplusplus
}
#endif
#endif /* PHYSICS_HIP_H */
