// abi.hip — the extern "C" surface of libphysics_hip.so (include/physics_hip.h). Host code only:
// argument checking, uploads, and the per-frame launch sequence. All compute is in HIP kernels; there
// is no CPU fallback anywhere in this library.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace phys {
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* get_error() { return g_error.c_str(); }
}  // namespace phys

using namespace phys;

namespace phys {
void poll_snapshots(phys_world* w);
// Launch-size hints: snapshots of the step counters in a ring of pinned host slots. A slot is filled either by
// an asynchronous copy (snapshot_counters_async) or by a kernel writing the host-mapped slot itself
// (snapshot_acquire -> kernel -> snapshot_commit); an event behind it tells poll_snapshots when it is complete.
StepCounters* snapshot_acquire(phys_world* w) {
    const uint32_t k = w->snap_next;
    if (w->snap_pending[k]) {
        // ring full: the host is kSnapRing steps ahead of the device. Wait for the oldest sample, so the
        // launch-size hints never lag by more than the ring depth.
        (void)hipEventSynchronize(w->snap_event[k]);
        poll_snapshots(w);
    }
    if (!w->h_snap[k]) {
        if (hipHostMalloc((void**)&w->h_snap[k], sizeof(StepCounters), hipHostMallocDefault) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&w->snap_event[k], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return w->h_snap[k];
}
void snapshot_commit(phys_world* w) {
    const uint32_t k = w->snap_next;
    (void)hipEventRecord(w->snap_event[k], w->stream);
    w->snap_pending[k] = true;
    w->snap_full[k] = w->snap_tag_full;  // was this update a full re-colouring?
    w->snap_next = (k + 1) % phys_world::kSnapRing;
}
void snapshot_counters_async(phys_world* w) {
    StepCounters* slot = snapshot_acquire(w);
    if (!slot) return;
    (void)hipMemcpyAsync(slot, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, w->stream);
    snapshot_commit(w);
}

// adopt every snapshot whose copy has completed (oldest first, so the newest complete one wins)
void poll_snapshots(phys_world* w) {
    for (int i = 0; i < phys_world::kSnapRing; ++i) {
        const uint32_t k = (w->snap_next + i) % phys_world::kSnapRing;
        if (!w->snap_pending[k]) continue;
        if (hipEventQuery(w->snap_event[k]) != hipSuccess) {
            (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: do not leave it behind as the
            continue;                 // thread's sticky error for whoever calls HIP next (the host application)
        }
        const StepCounters& c = *w->h_snap[k];
        w->snap_pending[k] = false;
        w->host_sticky_overflow |= c.overflow | c.sticky_overflow;  // latched until phys_sync reports it
        if (c.overflow) continue;
        w->hint.valid = true;
        w->hint.n_manifolds = c.n_manifolds;
        w->hint.n_pairs = c.n_pairs;
        w->hint.n_contacts = c.n_contacts;
        if (c.max_region) w->hint.max_region = c.max_region;
        w->hint.n_used_buckets = c.n_used_buckets;
        w->hint.n_colors = c.n_colors;
        if (c.n_active) w->hint.n_active = c.n_active;  // counted only in the updates that deal out the dynamic homes
        // colouring rounds: a full re-colouring and an incremental update need very different counts, and the
        // incremental count fluctuates: remember the full count, and the maximum of the recent incremental ones
        if (w->snap_full[k]) {
            w->hint.full_rounds = c.color_rounds;
        } else {
            w->hint.recent_new[w->hint.recent_pos % 8] = c.n_new_manifolds;
            w->hint.recent_rounds[w->hint.recent_pos++ % 8] = c.color_rounds;
            uint32_t mx = 0, mn = 0;
            for (int q = 0; q < 8; ++q) {
                mx = w->hint.recent_rounds[q] > mx ? w->hint.recent_rounds[q] : mx;
                mn = w->hint.recent_new[q] > mn ? w->hint.recent_new[q] : mn;
            }
            w->hint.color_rounds = mx;
            w->hint.n_new = mn;
        }
        for (int q = 0; q < kMaxColors; ++q) w->hint.color_count[q] = c.color_count[q];
    }
}
}  // namespace phys

namespace phys {
static std::atomic<int> g_worlds[64];
int worlds_on_device(int device) { return g_worlds[device & 63].load(std::memory_order_relaxed); }
}  // namespace phys

static int32_t fail(int32_t code, const char* msg) {
    set_error(msg);
    return code;
}

// std::time::Duration::as_secs_f32 (used at rigid_body.rs:25)
static float duration_as_secs_f32(uint64_t nanos_total) {
    const uint64_t secs = nanos_total / 1000000000ull;
    const uint32_t nanos = (uint32_t)(nanos_total % 1000000000ull);
    return (float)secs + (float)nanos / 1.0e9f;
}

#define ENTER(w)                                                                    \
    do {                                                                            \
        if (!(w)) return fail(PHYS_ERR_INVALID_ARG, "null world");                  \
        PHYS_HIP_TRY(hipSetDevice((w)->device));                                    \
    } while (0)

extern "C" {

void phys_config_default(phys_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = PHYS_ABI_VERSION;
    cfg->device = 0;
    cfg->flags = 0;
    cfg->gravity_force[0] = 0.0f; cfg->gravity_force[1] = -9.81f; cfg->gravity_force[2] = 0.0f;  // physics.rs:90
    cfg->gravity_offset[0] = 0.0f; cfg->gravity_offset[1] = 0.0f; cfg->gravity_offset[2] = 1.5f;  // physics.rs:91
    cfg->cg_max_iterations = 1000;  // sle_solver.rs:5
    cfg->cg_max_error = 1e-2f;      // sle_solver.rs:6
    cfg->cg_min_error = 1e-3f;      // sle_solver.rs:7
    cfg->solver_iterations = 8;
    cfg->baumgarte = 0.2f;
    cfg->slop = 0.01f;
    cfg->friction = 0.5f;
    cfg->contact_margin = 0.02f;
    cfg->ground_height = 0.0f;
    cfg->max_bias = 3.0f;
    cfg->max_pairs = 0;
    cfg->max_manifolds = 0;
    cfg->max_ghosts = 0;
}

const char* phys_last_error(void) { return get_error(); }
uint32_t phys_abi_version(void) { return PHYS_ABI_VERSION; }

int32_t phys_create(const phys_config* cfg, phys_world** out) {
    if (!cfg || !out) return fail(PHYS_ERR_INVALID_ARG, "null argument");
    if (cfg->abi_version != PHYS_ABI_VERSION) return fail(PHYS_ERR_INVALID_ARG, "phys_config.abi_version mismatch");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PHYS_ERR_NO_DEVICE, "no HIP device visible: libphysics_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= count) return fail(PHYS_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    PHYS_HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PHYS_ERR_NO_DEVICE, "device is not gfx950 (MI355X); this library carries gfx950 code only");
    PHYS_HIP_TRY(hipSetDevice(cfg->device));
    phys_world* w = new phys_world();
    w->cfg = *cfg;
    w->device = cfg->device;
    hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete w; return fail(PHYS_ERR_HIP, "hipStreamCreate failed"); }
    e = w->counters.resize(1);
    if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_counters, sizeof(StepCounters), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemsetAsync(w->counters.p, 0, sizeof(StepCounters), w->stream);
    if (e != hipSuccess) { delete w; return fail(PHYS_ERR_HIP, "counter allocation failed"); }
    std::memset(w->h_counters, 0, sizeof(StepCounters));
    phys::g_worlds[w->device & 63].fetch_add(1);
    *out = w;
    return PHYS_OK;
}

int32_t phys_destroy(phys_world* w) {
    if (!w) return PHYS_OK;
    phys::g_worlds[w->device & 63].fetch_sub(1);
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    DevBuf<float>* fb[] = {&w->pos, &w->rot, &w->vel, &w->force, &w->torque, &w->inv_inertia_diag,
                           &w->inv_inertia, &w->half_extent, &w->aabb, &w->cg_x, &w->cg_r, &w->cg_p, &w->cg_ap,
                           &w->cg_rhs, &w->cg_c, &w->cg_scratch, &w->cg_jl, &w->geo, &w->man_geo_prev, &w->man_imp, &w->man_imp_prev, &w->man_geo, &w->row_n,
                           &w->row_pt, &w->row_tb, &w->row_acc, &w->row_all, &w->flow_vel, &w->sorted_box, &w->slot_box};
    for (auto* b : fb) b->free();
    DevBuf<uint32_t>* ub[] = {&w->shape, &w->global_id, &w->cg_status, &w->bucket_of, &w->bucket_count,
                              &w->bucket_start, &w->bucket_cursor, &w->sorted_ids, &w->slot_ids, &w->grid_ovf, &w->scan_block_sums, &w->pairs,
                              &w->man_a, &w->man_b, &w->man_color, &w->row_hdr, &w->halo_block_counts,
                              &w->man_prev, &w->cluster_slot, &w->cluster_body, &w->body_shared, &w->active_flag, &w->active_rank, &w->seg_count, &w->seg_start, &w->man_rank,
                              &w->row_src, &w->cross_pairs, &w->color_block_hist, &w->cg_cols};
    for (auto* b : ub) b->free();
    w->man_prio.free(); w->color_state.free(); w->bucket_count.free(); w->step_zero.free();
    w->d_constraints.free(); w->counters.free();
    w->ctab.free(); w->unc_list.free();
    w->prof.destroy();
    for (int k = 0; k < phys_world::kSnapRing; ++k) {
        if (w->h_snap[k]) (void)hipHostFree(w->h_snap[k]);
        if (w->snap_event[k]) (void)hipEventDestroy(w->snap_event[k]);
    }
    if (w->h_counters) (void)hipHostFree(w->h_counters);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
    return PHYS_OK;
}

int32_t phys_set_bodies(phys_world* w, uint64_t n, const float* pos, const float* rot, const float* lin,
                        const float* ang, const float* mass, const float* inertia, const uint32_t* shape_type,
                        const float* half_extent) {
    ENTER(w);
    if (n && !pos) return fail(PHYS_ERR_INVALID_ARG, "pos is required");
    if (n >= 0x7FFFFFFFull) return fail(PHYS_ERR_INVALID_ARG, "too many bodies (u32 indices)");
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    // sharded worlds: max_ghosts kinematic slots behind the owned bodies (filled by phys_halo_unpack_ghosts)
    const uint64_t G = (w->cfg.flags & PHYS_FLAG_COLLISIONS) && !(w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY) ? w->cfg.max_ghosts : 0;
    const uint64_t nt = n ? n + G : 0;
    if (nt >= 0x7FFFFFFFull) return fail(PHYS_ERR_INVALID_ARG, "too many bodies + ghosts (u32 indices)");
    PHYS_HIP_TRY(w->pos.resize(3 * nt)); PHYS_HIP_TRY(w->rot.resize(4 * nt)); PHYS_HIP_TRY(w->vel.resize(8 * nt));
    PHYS_HIP_TRY(w->force.resize(3 * nt)); PHYS_HIP_TRY(w->torque.resize(3 * nt));
    PHYS_HIP_TRY(w->inv_inertia.resize(9 * nt)); PHYS_HIP_TRY(w->inv_inertia_diag.resize(4 * nt));
    PHYS_HIP_TRY(w->half_extent.resize(3 * nt)); PHYS_HIP_TRY(w->aabb.resize(6 * nt)); PHYS_HIP_TRY(w->shape.resize(nt));
    PHYS_HIP_TRY(w->global_id.resize(nt));
    if ((w->cfg.flags & PHYS_FLAG_COLLISIONS) && !(w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY)) PHYS_HIP_TRY(w->geo.resize(16 * nt));
    w->n = nt;
    w->n_owned = n;
    w->max_ghosts = nt - n;
    w->forces_dirty = false;
    w->have_lambda = false;  // previous_solution: None
    w->aabbs_valid = false;
    w->grid_valid = false;
    w->sorted_grid_valid = false;
    w->hint = StepHint();
    for (int k = 0; k < phys_world::kSnapRing; ++k) w->snap_pending[k] = false;  // the stream was synchronised above
    if (n == 0) return PHYS_OK;

    // host staging with RigidBody::new defaults (rigid_body.rs:64-76); ghost slots: no shape, immovable
    std::vector<float> h_pos(3 * nt, 0.0f), h_rot(4 * nt), h_vel(8 * nt), h_inv(9 * nt, 0.0f), h_diag(4 * nt, 0.0f), h_he(3 * nt, 0.0f);
    std::vector<uint32_t> h_shape(nt, PHYS_SHAPE_NONE), h_gid(nt, 0xFFFFFFFFu);
    std::memcpy(h_pos.data(), pos, 12 * n);
    w->singular_inertia = false;
    w->all_diag_inertia = true;
    w->uniform_inertia = true;
    for (uint64_t i = 0; i < n; ++i) {
        if (rot) std::memcpy(&h_rot[4 * i], rot + 4 * i, 16);
        else { h_rot[4 * i] = 0.0f; h_rot[4 * i + 1] = 0.0f; h_rot[4 * i + 2] = 0.0f; h_rot[4 * i + 3] = 1.0f; }
        const float m_i = mass ? mass[i] : 1.0f;
        for (int k = 0; k < 3; ++k) { h_vel[8 * i + k] = lin ? lin[3 * i + k] : 0.0f; h_vel[8 * i + 4 + k] = ang ? ang[3 * i + k] : 0.0f; }
        h_vel[8 * i + 3] = 1.0f / m_i;  // constraints.rs:75
        h_vel[8 * i + 7] = m_i;
        m33 I, inv;
        for (int k = 0; k < 9; ++k) I.m[k] = inertia ? inertia[9 * i + k] : ((k % 4 == 0) ? 1.0f : 0.0f);
        // The reference inverts the (constant, world-frame) tensor every step (rigid_body.rs:31, quirk Q5);
        // inverting once gives the same bits.
        if (!m33_try_inverse(&I, &inv)) {
            w->singular_inertia = true;
            for (int k = 0; k < 9; ++k) inv.m[k] = 0.0f;
        }
        for (int k = 0; k < 9; ++k) {
            h_inv[9 * i + k] = inv.m[k];
            if (k % 4 != 0 && inv.m[k] != 0.0f) w->all_diag_inertia = false;
        }
        h_diag[4 * i] = inv.m[0]; h_diag[4 * i + 1] = inv.m[4]; h_diag[4 * i + 2] = inv.m[8];
        if (h_diag[4 * i] != h_diag[0] || h_diag[4 * i + 1] != h_diag[1] || h_diag[4 * i + 2] != h_diag[2]) w->uniform_inertia = false;
        if (shape_type) h_shape[i] = shape_type[i];
        if (half_extent) std::memcpy(&h_he[3 * i], half_extent + 3 * i, 12);
        h_gid[i] = (uint32_t)i;
    }
    for (uint64_t i = n; i < nt; ++i) {  // ghost slots: identity pose, inverse mass 0, mass +inf (F / m = 0), inverse inertia 0
        h_rot[4 * i + 3] = 1.0f;
        h_vel[8 * i + 3] = 0.0f;
        h_vel[8 * i + 7] = std::numeric_limits<float>::infinity();
    }
    if (nt > n) w->uniform_inertia = false;  // the ghosts' zero tensors differ from everybody's
    hipStream_t s = w->stream;
    PHYS_HIP_TRY(hipMemcpyAsync(w->pos.p, h_pos.data(), 12 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->rot.p, h_rot.data(), 16 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->vel.p, h_vel.data(), 32 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemsetAsync(w->force.p, 0, 12 * nt, s));
    PHYS_HIP_TRY(hipMemsetAsync(w->torque.p, 0, 12 * nt, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->inv_inertia.p, h_inv.data(), 36 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->inv_inertia_diag.p, h_diag.data(), 16 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->half_extent.p, h_he.data(), 12 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->shape.p, h_shape.data(), 4 * nt, hipMemcpyHostToDevice, s));
    PHYS_HIP_TRY(hipMemcpyAsync(w->global_id.p, h_gid.data(), 4 * nt, hipMemcpyHostToDevice, s));
    if (!w->all_diag_inertia) w->uniform_inertia = false;
    PHYS_HIP_TRY(hipStreamSynchronize(s));  // staging vectors die here
    if (w->cfg.flags & PHYS_FLAG_COLLISIONS) {
        grid_plan(w, pos, half_extent);
        int32_t rc = collision_alloc(w);
        if (rc != PHYS_OK) return rc;
        if (!(w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY)) {
            rc = cluster_assign(w, pos);  // spatial clusters of the cluster solver (large scenes only)
            if (rc != PHYS_OK) return rc;
        }
    }
    return PHYS_OK;
}

static int32_t add_constraint(phys_world* w, uint32_t kind, uint64_t body, const float t[3]) {
    ENTER(w);
    if (!t) return fail(PHYS_ERR_INVALID_ARG, "null target");
    if (body >= w->n_owned) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    Constraint c;
    c.kind = kind; c.body = (uint32_t)body;
    c.target[0] = t[0]; c.target[1] = t[1]; c.target[2] = t[2];
    w->constraints.push_back(c);
    w->constraints_dirty = true;
    return PHYS_OK;
}
int32_t phys_add_constraint_fix_point(phys_world* w, uint64_t body, const float target[3]) {
    return add_constraint(w, 0u, body, target);
}
int32_t phys_add_constraint_fix_orientation(phys_world* w, uint64_t body, const float target_rpy[3]) {
    return add_constraint(w, 1u, body, target_rpy);
}
int32_t phys_clear_constraints(phys_world* w) {
    ENTER(w);
    w->constraints.clear();
    w->constraints_dirty = true;
    w->have_lambda = false;
    return PHYS_OK;
}

static int32_t apply_force(phys_world* w, uint64_t body, int mode, const float f[3], const float arg[3]) {
    ENTER(w);
    if (!f || (mode != 0 && !arg)) return fail(PHYS_ERR_INVALID_ARG, "null argument");
    if (body >= w->n_owned) return fail(PHYS_ERR_OUT_OF_RANGE, "body index out of range");
    launch_apply_force_one(w, (uint32_t)body, mode, f, arg);
    PHYS_HIP_TRY(hipGetLastError());
    return PHYS_OK;
}
int32_t phys_apply_force_centre_of_gravity(phys_world* w, uint64_t body, const float force[3]) {
    return apply_force(w, body, 0, force, nullptr);
}
int32_t phys_apply_force_at_position(phys_world* w, uint64_t body, const float force[3], const float point[3]) {
    return apply_force(w, body, 1, force, point);
}
int32_t phys_apply_force_at_offset(phys_world* w, uint64_t body, const float force[3], const float offset[3]) {
    return apply_force(w, body, 2, force, offset);
}

int32_t phys_set_forces(phys_world* w, const float* force, const float* torque) {
    ENTER(w);
    if (force) PHYS_HIP_TRY(hipMemcpyAsync(w->force.p, force, 12 * w->n_owned, hipMemcpyHostToDevice, w->stream));
    if (torque) PHYS_HIP_TRY(hipMemcpyAsync(w->torque.p, torque, 12 * w->n_owned, hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    if (force || torque) w->forces_dirty = true;
    return PHYS_OK;
}

int32_t phys_apply_gravity(phys_world* w) {
    ENTER(w);
    launch_apply_gravity(w);
    PHYS_HIP_TRY(hipGetLastError());
    return PHYS_OK;
}

int32_t phys_step(phys_world* w, uint64_t dt_nanos) {
    ENTER(w);
    if (w->singular_inertia) return fail(PHYS_ERR_SINGULAR_INERTIA, "singular inertia tensor (reference: unwrap panic, rigid_body.rs:31)");
    launch_step_full(w, duration_as_secs_f32(dt_nanos), /*gravity=*/false);
    PHYS_HIP_TRY(hipGetLastError());
    return PHYS_OK;
}

// one PhysicsState::update (physics.rs:41-55), enqueued without synchronising
static int32_t enqueue_update(phys_world* w, float dt) {
    // a zero-length step has no contact problem to solve (the bias terms divide by dt): plain RigidBody::step then
    const bool collisions = (w->cfg.flags & PHYS_FLAG_COLLISIONS) != 0 && dt > 0.0f;
    if (collisions) poll_snapshots(w);
    const bool have_constraints = !w->constraints.empty();
    bool gravity_pending = true;
    if (have_constraints) {
        // the constraint right-hand side reads Q = force/torque accumulators including gravity (constraints.rs:92-104):
        // the constraint kernel adds gravity to the accumulators of the bodies it reads, and the step kernel adds
        // J^T lambda to entity 0 behind its own gravity addition - the reference's order, without a pass over all bodies
        const int32_t rc = constraints_alloc(w);
        if (rc != PHYS_OK) return rc;
        launch_constraint_phase(w, gravity_pending);
    }
    if (!collisions) {
        launch_step_full(w, dt, gravity_pending, have_constraints);
    } else {
        // per-step state: zeroed by the first kernel of the step itself; every 32nd step a memset in front of it
        // also restarts the running extent bound (which that kernel raises, so it cannot zero it)
        const bool restart_extent = w->steps % 32 == 0 || w->step_zero_reset_bytes % 16 != 0 || (w->step_zero_reset_bytes >> 36) != 0;
        if (restart_extent) zero_step_state(w, /*including_extent=*/true);
        launch_step_velocity_aabb(w, dt, gravity_pending, /*zero_step=*/!restart_extent, have_constraints);
        launch_broadphase(w);
        if (!(w->cfg.flags & PHYS_FLAG_BROADPHASE_ONLY)) {
            launch_narrowphase(w);
            launch_coloring(w);
            launch_solver(w, dt);
        } else {
            snapshot_counters_async(w);  // launch-size hints of later updates (the colouring stage takes it otherwise)
        }
        launch_step_position(w, dt);
    }
    PHYS_HIP_TRY(hipGetLastError());
    w->steps++;
    if (w->prof.on) { w->prof.steps++; if (w->prof.used > 4096) w->prof.collect(w->stream); }
    return PHYS_OK;
}

int32_t phys_update_n(phys_world* w, uint64_t dt_nanos, uint32_t n) {
    ENTER(w);
    if (w->n == 0) return fail(PHYS_ERR_NO_BODIES, "update with no bodies (reference: index panic, physics.rs:48)");
    if (w->singular_inertia) return fail(PHYS_ERR_SINGULAR_INERTIA, "singular inertia tensor (reference: unwrap panic, rigid_body.rs:31)");
    const float dt = duration_as_secs_f32(dt_nanos);
    for (uint32_t k = 0; k < n; ++k) {
        const int32_t rc = enqueue_update(w, dt);
        if (rc != PHYS_OK) return rc;
    }
    return PHYS_OK;
}

int32_t phys_update(phys_world* w, uint64_t dt_nanos) { return phys_update_n(w, dt_nanos, 1); }

static int32_t fetch_counters(phys_world* w) {
    PHYS_HIP_TRY(hipMemcpyAsync(w->h_counters, w->counters.p, sizeof(StepCounters), hipMemcpyDeviceToHost, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}

// Device-side errors are STICKY: every overflow bit raised by any step since the last phys_sync is reported here
// (StepCounters::sticky_overflow; the per-step word is zeroed by the next step), then cleared.
int32_t phys_sync(phys_world* w) {
    ENTER(w);
    const int32_t rc = fetch_counters(w);
    if (rc != PHYS_OK) return rc;
    poll_snapshots(w);  // the stream is idle: every snapshot in flight is adopted now, none can bring reported bits back later
    const uint32_t bits = w->h_counters->overflow | w->h_counters->sticky_overflow | w->host_sticky_overflow;
    w->host_sticky_overflow = 0;
    if (w->h_counters->sticky_overflow) {
        PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->sticky_overflow, 0, sizeof(uint32_t), w->stream));
        PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    }
    if (bits & 32u) {
        const uint32_t* g = w->h_counters->debug;
        return fail(PHYS_ERR_HIP, ("internal error: a solver row names no body of this world and was refused (row " +
                                   std::to_string(g[0]) + ": a " + std::to_string(g[1]) + ", b " + std::to_string(g[2]) + ", points " +
                                   std::to_string(g[3]) + "; colour " + std::to_string(g[6]) + " rows [" + std::to_string(g[4]) + ", " +
                                   std::to_string(g[5]) + "), tile base " + std::to_string(g[7]) + ")").c_str());
    }
    if (w->h_counters->debug[0] != 0u) {  // reported below: the next event may leave its own note
        PHYS_HIP_TRY(hipMemsetAsync(w->counters.p->debug, 0, sizeof(w->counters.p->debug), w->stream));
        PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    }
    if ((bits & 16u) && w->h_counters->debug[0] == 0xC1u) {
        const uint32_t* g = w->h_counters->debug;  // what the first lane of k_solve_cluster to give up was waiting for
        return fail(PHYS_ERR_HIP, ("contact solver hand-off timed out (k_solve_cluster) in a step since the last phys_sync; velocities "
                                   "are invalid from that step on. First lane to give up: cluster " + std::to_string(g[1] & 0xFFFFu) + " of " + std::to_string(g[1] >> 16) + ", row " +
                                   std::to_string(g[2]) + ", bodies " + std::to_string(g[3]) + " / " + std::to_string(g[4]) +
                                   ", tickets " + std::to_string(g[5] & 0xFFFFu) + " / " + std::to_string(g[5] >> 16) + ", waiting A/B " +
                                   std::to_string(g[6] & 1u) + "/" + std::to_string((g[6] >> 1) & 1u) + ", modes " +
                                   std::to_string((g[6] >> 4) & 3u) + "/" + std::to_string((g[6] >> 8) & 3u) + ", iteration " +
                                   std::to_string((g[7] >> 8) & 0xFFu) + ", colour " + std::to_string(g[7] & 0xFFu) +
                                   ". If other work shares this GPU with phys_update, create the world WITHOUT PHYS_FLAG_EXCLUSIVE_GPU").c_str());
    }
    if (bits & 16u)
        return fail(PHYS_ERR_HIP, "contact solver hand-off timed out (k_solve_flow) in a step since the last phys_sync; "
                                  "velocities are invalid from that step on");
    if (bits & 4u)
        return fail(PHYS_ERR_CAPACITY, "a body has more than 64 contact manifolds (PHYS_MAX_COLORS): the contact solve of "
                                       "that step was skipped. This limit is not configurable");
    if (bits & 8u)
        return fail(PHYS_ERR_CAPACITY, "halo record / cross-pair capacity exceeded in a step since the last phys_sync");
    if (bits & 64u)
        return fail(PHYS_ERR_CAPACITY, "the persistent colour table is full (a look-up or an insert gave up after thousands of "
                                       "slots): the contact solve of that step was skipped. Raise phys_config.max_manifolds");
    if (bits)
        return fail(PHYS_ERR_CAPACITY, "pair / manifold capacity exceeded in a step since the last phys_sync (the contact "
                                       "solve of that step was skipped): raise phys_config.max_pairs / max_manifolds");
    return PHYS_OK;
}

static int32_t d2h(phys_world* w, void* dst, const void* src, size_t bytes) {
    if (!dst || bytes == 0) return PHYS_OK;
    PHYS_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, w->stream));
    return PHYS_OK;
}

int32_t phys_get_transforms(phys_world* w, float* pos_out, float* rot_out) {
    ENTER(w);
    int32_t rc = d2h(w, pos_out, w->pos.p, 12 * w->n_owned); if (rc) return rc;
    rc = d2h(w, rot_out, w->rot.p, 16 * w->n_owned); if (rc) return rc;
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}
int32_t phys_get_velocities(phys_world* w, float* lin_out, float* ang_out) {
    ENTER(w);
    // strided read-out of the 32-byte velocity records: 12 bytes per body from a 32-byte pitch
    if (lin_out && w->n_owned) PHYS_HIP_TRY(hipMemcpy2DAsync(lin_out, 12, w->vel.p, 32, 12, w->n_owned, hipMemcpyDeviceToHost, w->stream));
    if (ang_out && w->n_owned) PHYS_HIP_TRY(hipMemcpy2DAsync(ang_out, 12, w->vel.p + 4, 32, 12, w->n_owned, hipMemcpyDeviceToHost, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}
int32_t phys_get_forces(phys_world* w, float* force_out, float* torque_out) {
    ENTER(w);
    int32_t rc = d2h(w, force_out, w->force.p, 12 * w->n_owned); if (rc) return rc;
    rc = d2h(w, torque_out, w->torque.p, 12 * w->n_owned); if (rc) return rc;
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}

int32_t phys_get_instance_matrices(phys_world* w, float* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    if (w->n == 0) return PHYS_OK;
    float* d = nullptr;
    PHYS_HIP_TRY(hipMalloc((void**)&d, 64 * w->n));
    launch_instance_matrices(w, d);
    hipError_t e = hipMemcpyAsync(out, d, 64 * w->n_owned, hipMemcpyDeviceToHost, w->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
    (void)hipFree(d);
    PHYS_HIP_TRY(e);
    return PHYS_OK;
}

int32_t phys_get_lambda(phys_world* w, float* lambda_out, uint64_t cap, uint64_t* n_rows) {
    ENTER(w);
    if (!n_rows) return fail(PHYS_ERR_INVALID_ARG, "null n_rows");
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    uint32_t st[4] = {0, 0, 0, 0};
    if (w->cg_status.p && !w->constraints_dirty) PHYS_HIP_TRY(hipMemcpy(st, w->cg_status.p, 16, hipMemcpyDeviceToHost));
    const uint64_t rows = st[2] ? 3 * (uint64_t)w->constraints.size() : 0;  // previous_solution: Option
    *n_rows = rows;
    if (lambda_out && rows) {
        const uint64_t m = rows < cap ? rows : cap;
        PHYS_HIP_TRY(hipMemcpy(lambda_out, w->cg_x.p, 4 * m, hipMemcpyDeviceToHost));
    }
    return PHYS_OK;
}

int32_t phys_get_aabbs(phys_world* w, float* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    launch_aabb_only(w);
    int32_t rc = d2h(w, out, w->aabb.p, 24 * w->n_owned); if (rc) return rc;
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}

int32_t phys_broadphase(phys_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs) {
    ENTER(w);
    if (!n_pairs) return fail(PHYS_ERR_INVALID_ARG, "null n_pairs");
    if (!(w->cfg.flags & PHYS_FLAG_COLLISIONS)) return fail(PHYS_ERR_UNSUPPORTED, "world created without PHYS_FLAG_COLLISIONS");
    if (w->n == 0) { *n_pairs = 0; return PHYS_OK; }
    zero_step_state(w, true);
    launch_aabb_only(w);
    launch_broadphase(w);
    PHYS_HIP_TRY(hipGetLastError());
    return sorted_pairs_to_host(w, pairs_out, cap, n_pairs);
}

int32_t phys_get_manifolds(phys_world* w, uint32_t* ids_out, uint32_t* counts_out, float* normals_out,
                           float* points_out, uint64_t cap, uint64_t* n_manifolds) {
    ENTER(w);
    if (!n_manifolds) return fail(PHYS_ERR_INVALID_ARG, "null n_manifolds");
    int32_t rc = fetch_counters(w); if (rc) return rc;
    const uint64_t m = w->h_counters->n_manifolds < w->max_manifolds ? w->h_counters->n_manifolds : w->max_manifolds;
    *n_manifolds = m;
    if (m == 0 || (!ids_out && !counts_out && !normals_out && !points_out)) return PHYS_OK;
    // read back in storage order, sort by (a, b) on the host (a read-out convenience, not the hot path)
    std::vector<uint32_t> a(m), b(m), c(m);
    std::vector<float> nrm(3 * m), pts(16 * m);
    {
        std::vector<float> geo(32 * m);  // 128-byte records: {a, b, count, -} {normal, -} 4 x {point, depth} + 32 spare bytes
        PHYS_HIP_TRY(hipMemcpy(geo.data(), w->man_geo.p, 128 * m, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < m; ++k) {
            const float* g = &geo[32 * k];
            std::memcpy(&a[k], g, 4); std::memcpy(&b[k], g + 1, 4); std::memcpy(&c[k], g + 2, 4);
            std::memcpy(&nrm[3 * k], g + 4, 12);
            std::memcpy(&pts[16 * k], g + 8, 64);
        }
    }
    std::vector<uint64_t> order(m);
    for (uint64_t k = 0; k < m; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return a[x] < a[y] || (a[x] == a[y] && b[x] < b[y]); });
    for (uint64_t k = 0; k < m && k < cap; ++k) {
        const uint64_t s = order[k];
        if (ids_out) { ids_out[2 * k] = a[s]; ids_out[2 * k + 1] = b[s]; }
        if (counts_out) counts_out[k] = c[s];
        if (normals_out) std::memcpy(normals_out + 3 * k, &nrm[3 * s], 12);
        if (points_out) std::memcpy(points_out + 16 * k, &pts[16 * s], 64);
    }
    return PHYS_OK;
}

int32_t phys_get_stats(phys_world* w, phys_stats* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    int32_t rc = fetch_counters(w); if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    const StepCounters& c = *w->h_counters;
    out->n_bodies = w->n_owned;
    out->n_pairs = c.n_pairs;
    out->n_manifolds = c.n_manifolds;
    out->n_contacts = c.n_contacts;
    out->n_colors = c.n_colors;
    out->color_rounds = c.color_rounds;
    out->n_new_manifolds = c.n_new_manifolds;
    if (w->cg_status.p && !w->constraints.empty()) {
        uint32_t st[2] = {1, 0};
        PHYS_HIP_TRY(hipMemcpy(st, w->cg_status.p, 8, hipMemcpyDeviceToHost));
        out->cg_converged = (int32_t)st[0];
        out->cg_iterations = st[1];
    } else {
        out->cg_converged = 1;  // quirk Q8: CG on the empty system returns Some(empty) on its first check
        out->cg_iterations = w->steps ? 1u : 0u;
    }
    out->steps = w->steps;
    out->overflow = c.overflow | c.sticky_overflow | w->host_sticky_overflow;  // last step's bits + everything since the last phys_sync
    out->n_ground_manifolds = c.n_ground_manifolds;
    std::memcpy(&out->max_extent, &c.max_extent_bits, 4);
    out->n_halo_records = c.n_halo + c.n_halo_low;  // both faces of a neighbour exchange
    out->n_cross_pairs = c.n_cross_pairs;
    out->n_ghosts = c.n_ghosts;
    return PHYS_OK;
}

int32_t phys_get_color_counts(phys_world* w, uint32_t* counts_out) {
    ENTER(w);
    if (!counts_out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    int32_t rc = fetch_counters(w); if (rc) return rc;
    for (int k = 0; k < kMaxColors; ++k) counts_out[k] = k < (int)w->h_counters->n_colors ? w->h_counters->color_count[k] : 0u;
    return PHYS_OK;
}

int32_t phys_profile_enable(phys_world* w, int32_t on) {
    ENTER(w);
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    w->prof.reset();
    w->prof.on = on != 0;
    return PHYS_OK;
}

int32_t phys_profile_get(phys_world* w, phys_profile* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    w->prof.collect(w->stream);
    for (uint32_t k = 0; k < PHYS_STAGE_COUNT; ++k) { out->ms[k] = w->prof.ms[k]; out->launches[k] = w->prof.launches[k]; }
    out->steps = w->prof.steps;
    return PHYS_OK;
}

int32_t phys_get_device_view(phys_world* w, phys_device_view* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    out->n = w->n_owned;
    out->pos = w->pos.p; out->rot = w->rot.p; out->lin_vel = w->vel.p; out->ang_vel = w->vel.p + 4;  // both with a stride of 8 floats
    out->aabb = w->aabb.p;
    out->stream = (void*)w->stream;
    out->vel_stride = 8;
    return PHYS_OK;
}

int32_t phys_set_global_ids(phys_world* w, const uint32_t* global_ids) {
    ENTER(w);
    if (!global_ids) return fail(PHYS_ERR_INVALID_ARG, "null ids");
    PHYS_HIP_TRY(hipMemcpyAsync(w->global_id.p, global_ids, 4 * w->n_owned, hipMemcpyHostToDevice, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}

int32_t phys_halo_pack(phys_world* w, float x_lo, float x_hi, float reach, void* dev_records_out, uint64_t cap,
                       uint64_t* n_records) {
    ENTER(w);
    return halo_pack(w, x_lo, x_hi, reach, dev_records_out, cap, n_records);
}
int32_t phys_halo_pairs(phys_world* w, const void* dev_remote_records, uint64_t n_remote, uint64_t skip_first,
                        uint64_t skip_count, uint64_t* n_cross_pairs) {
    ENTER(w);
    return halo_pairs(w, dev_remote_records, n_remote, skip_first, skip_count, n_cross_pairs);
}
int32_t phys_set_slab(phys_world* w, float x_lo, float x_hi, float reach) {
    ENTER(w);
    if (!(x_lo < x_hi) || !(reach > 0.0f)) return fail(PHYS_ERR_INVALID_ARG, "slab needs x_lo < x_hi and reach > 0");
    w->slab_lo = x_lo; w->slab_hi = x_hi; w->slab_reach = reach;
    return PHYS_OK;
}
int32_t phys_halo_pack_bodies(phys_world* w, void* dev_records_out, uint64_t cap) {
    ENTER(w);
    return halo_pack_bodies(w, dev_records_out, cap);
}
int32_t phys_halo_pack_bodies_face(phys_world* w, void* dev_records_out, uint64_t cap, int32_t face) {
    ENTER(w);
    const float far = 3.0e38f;
    return halo_pack_bodies_faces(w, dev_records_out, cap, face > 0 ? -far : w->slab_lo, face < 0 ? far : w->slab_hi);
}
int32_t phys_halo_unpack_ghosts(phys_world* w, const void* dev_records, uint64_t n_records, uint64_t skip_first,
                                uint64_t skip_count) {
    ENTER(w);
    return halo_unpack_ghosts(w, dev_records, n_records, skip_first, skip_count);
}
int32_t phys_get_global_ids(phys_world* w, uint32_t* out) {
    ENTER(w);
    if (!out) return fail(PHYS_ERR_INVALID_ARG, "null output");
    if (w->n) PHYS_HIP_TRY(hipMemcpyAsync(out, w->global_id.p, 4 * w->n, hipMemcpyDeviceToHost, w->stream));
    PHYS_HIP_TRY(hipStreamSynchronize(w->stream));
    return PHYS_OK;
}
int32_t phys_get_cross_pairs(phys_world* w, uint32_t* pairs_out, uint64_t cap, uint64_t* n_pairs) {
    ENTER(w);
    if (!n_pairs) return fail(PHYS_ERR_INVALID_ARG, "null n_pairs");
    int32_t rc = fetch_counters(w); if (rc) return rc;
    const uint64_t m = w->h_counters->n_cross_pairs < w->max_cross_pairs ? w->h_counters->n_cross_pairs : w->max_cross_pairs;
    *n_pairs = m;
    if (pairs_out && m) {
        std::vector<uint64_t> keys(m);
        std::vector<uint32_t> raw(2 * m);
        PHYS_HIP_TRY(hipMemcpy(raw.data(), w->cross_pairs.p, 8 * m, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < m; ++k) keys[k] = ((uint64_t)raw[2 * k] << 32) | raw[2 * k + 1];
        std::sort(keys.begin(), keys.end());
        for (uint64_t k = 0; k < m && k < cap; ++k) { pairs_out[2 * k] = (uint32_t)(keys[k] >> 32); pairs_out[2 * k + 1] = (uint32_t)keys[k]; }
    }
    return PHYS_OK;
}

}  // extern "C"
