// world.hpp — device-resident state of one phys_world (MI355X / gfx950).
//
// Layout in HBM: structure-of-arrays, one array per body attribute, so a wave reading attribute k of
// bodies [64w, 64w+64) touches one contiguous span (3-float attributes: 768 B per wave, quaternions:
// 1 KiB as dwordx4). Everything stays resident across steps; the host only sees what it asks for.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/physics_hip.h"

namespace phys {

constexpr int kMaxColors = 64;  // == PHYS_MAX_COLORS of include/spec/contact_solve.h
constexpr uint64_t kClusterMinBodies = 32768;  // below: the dataflow kernels win anyway (few launches' worth of rows)
constexpr uint32_t kClusterDynamicPeriod = 8;  // cluster steps between two deals of the dynamic homes (a body that became active
                                               // since has none and is served as another cluster's body: slower, never wrong)
// measured with tools/cluster_crossover.py (solve + rows, ms: cluster / four-lane dataflow kernel with statically dealt items):
// mixed piles 91k manifolds 0.518 / 0.353, 145k 0.519 / 0.486, 155k 0.535 / 0.494, 216k (C3) 0.576 / 0.696; towers 92k 0.647 /
// 0.415, 182k 0.699 / 0.700, 256k 0.712 / 0.984 - the cluster kernel's time is its chain (nearly the same at every size), the
// dataflow kernel's grows with the rows
constexpr uint64_t kClusterMinManifolds = 170000;
constexpr uint64_t kFlowWideMaxManifolds = 400000;  // the dataflow kernels' upper end (solver.hip kFlowMaxManifolds)
constexpr uint32_t kClusterMaxSlots = 2496;    // bodies per cluster whose {v, w, x, I^-1} fit one CU's LDS (64 B each: 156 KiB; 13-bit slot field)

void set_error(const std::string& msg);
const char* get_error();

#define PHYS_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            phys::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
            return PHYS_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t resize(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void free() {
        if (p && !view) (void)hipFree(p);
        p = nullptr;
        n = 0;
        view = false;
    }
    // non-owning window into another allocation (the per-step zeroed block)
    bool view = false;
    void point_at(T* ptr, size_t count) { free(); p = ptr; n = count; view = true; }
};

// device-side counters of the collision pipeline (one 256-B block, zeroed per step by one memset)
struct StepCounters {
    // the first two words are reserved TOGETHER by the narrow phase: one 64-bit atomic per workgroup trip adds the trip's
    // manifolds to the low word and its uncoloured manifolds to the high one (same-address atomics serialise chip-wide;
    // two of them per trip was two places in that queue)
    uint32_t n_manifolds;    // manifolds written (body-body + ground)
    uint32_t unc_count[3];   // colouring rounds: length of the list of uncoloured manifolds read / written / cleared (rotating)
    uint32_t n_pairs;        // candidate pairs written
    uint32_t n_contacts;     // contact points
    uint32_t n_uncolored;    // manifolds still uncoloured (colouring loop)
    uint32_t n_colors;       // colours in use
    uint32_t color_rounds;
    uint32_t overflow;       // bit 0 pairs, bit 1 manifolds, bit 2 colours, bit 3 cross pairs, bit 4 solver hand-off timeout, bit 5 corrupt solver row refused, bit 6 colour table walk given up
    uint32_t n_halo;         // halo records packed
    uint32_t n_cross_pairs;
    uint32_t n_ground_manifolds;
    uint32_t flow_ticket;    // k_solve_flow: next (iteration, row chunk) item to hand to a workgroup
    uint32_t n_grid_ovf;     // slot grid: bodies that found their bucket's four slots taken
    uint32_t n_active;       // owned bodies with at least one manifold in this update (dynamic clusters, cluster.hip)
    uint32_t n_used_buckets; // buckets of the sorted grid holding at least one body (k_cell_assign)
    uint32_t max_region;     // k_find_pairs_brick: most records in the region of one brick (sizes the LDS stage of later updates)
    uint32_t n_new_manifolds;  // manifolds that kept no colour in this update (= the colouring's work; never counted down)
    uint32_t cluster_arrived[2][8];  // k_solve_cluster, per attempt: workgroups that have begun (eight counters: same-address
                                     // atomics serialise chip-wide) ...
    uint32_t cluster_state[2];       // ... and the launch's one decision: 0 undecided, 1 go (all are resident), 2 called off
    uint32_t color_count[kMaxColors];  // manifolds per colour
    uint32_t color_start[kMaxColors + 1];
    // LAST member: survives the per-step reset (only the bytes before it are zeroed), so a wave issues the
    // same-address atomicMax only when it RAISES the bound. It is a running upper bound of the largest
    // fattened-AABB edge (float bits; positive floats order as uints), re-derived from zero every 32 steps.
    // Any upper bound is a valid grid cell size: the pair SET does not depend on it.
    alignas(16) uint32_t max_extent_bits;
    // Every overflow bit ever raised since the host last looked (phys_sync reports and clears it). `overflow` above
    // is per step - the first kernel of the next step zeroes it - so a capacity miss or a hand-off timeout in an
    // EARLY step of a phys_update_n batch would otherwise be gone by the time the host synchronises. Zeroed by
    // neither the per-step reset nor the extent restart (both stop short of it).
    uint32_t sticky_overflow;
    uint32_t n_ghosts;   // ghost slots filled by the last phys_halo_unpack_ghosts (set before the update: not part of the per-step reset)
    uint32_t n_halo_low; // neighbour exchange: records of the LOW-face block (n_halo then counts the high-face block; the stats add them)
    uint32_t debug[8];  // what a kernel that refused a corrupt row saw (overflow bit 5); never read by device code
};
static_assert(offsetof(StepCounters, n_manifolds) % 8 == 0 && offsetof(StepCounters, unc_count) == offsetof(StepCounters, n_manifolds) + 4,
              "n_manifolds | unc_count[0] are one aligned 64-bit word");
constexpr size_t kCountersStepResetBytes = offsetof(StepCounters, max_extent_bits);
constexpr size_t kCountersExtentResetBytes = offsetof(StepCounters, sticky_overflow);

// raise overflow bits: this step's word (the solver kernels of the step look at it) and the sticky one
__device__ __forceinline__ void flag_overflow(StepCounters* ctr, uint32_t bits) {
    atomicOr(&ctr->overflow, bits);
    atomicOr(&ctr->sticky_overflow, bits);
}

// per-stage device timing with HIP events on the world's stream (phys_profile_enable)
struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs: [2k] start, [2k+1] stop
    std::vector<uint32_t> stage;     // stage of pair k
    size_t used = 0;                 // pairs in flight
    double ms[PHYS_STAGE_COUNT] = {};
    uint64_t launches[PHYS_STAGE_COUNT] = {};
    uint64_t steps = 0;
    void begin(hipStream_t s, uint32_t st) {
        if (2 * used + 2 > ev.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            ev.push_back(a); ev.push_back(b); stage.push_back(0);
        }
        stage[used] = st;
        (void)hipEventRecord(ev[2 * used], s);
    }
    void end(hipStream_t s) { (void)hipEventRecord(ev[2 * used + 1], s); ++used; }
    void collect(hipStream_t s) {  // the stream must be idle or is synchronised here
        if (!used) return;
        (void)hipStreamSynchronize(s);
        for (size_t k = 0; k < used; ++k) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]) == hipSuccess) { ms[stage[k]] += t; launches[stage[k]] += 1; }
        }
        used = 0;
    }
    void reset() { for (auto& m : ms) m = 0; for (auto& l : launches) l = 0; steps = 0; used = 0; }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); stage.clear(); used = 0; }
};

struct ProfScope {
    Profiler& p; hipStream_t s;
    ProfScope(Profiler& p_, hipStream_t s_, uint32_t st) : p(p_), s(s_) { if (p.on) p.begin(s, st); }
    ~ProfScope() { if (p.on) p.end(s); }
};

// launch-size hints taken from an EARLIER step's counters (asynchronous read-back); never needed for
// correctness
struct StepHint {
    bool valid = false;
    uint32_t n_manifolds = 0, n_colors = 0, n_pairs = 0, max_region = 0, n_used_buckets = 0, n_contacts = 0;
    uint32_t n_active = 0;         // owned bodies with a manifold (0 = unknown)
    uint32_t color_rounds = 0;     // max over the recent INCREMENTAL updates
    uint32_t full_rounds = 0;      // rounds of the last full re-colouring (0 = unknown)
    uint32_t recent_rounds[8] = {};
    uint32_t recent_pos = 0;
    uint32_t n_new = 0xFFFFFFFFu;  // most manifolds without a kept colour in one of the recent incremental updates (~0: unknown)
    uint32_t recent_new[8] = {};
    uint32_t color_count[kMaxColors] = {};
};

// split of the broad phase's bucket table over the three axes (kernels.hpp: grid_bucket)
struct GridShape {
    uint32_t mx = 7, my = 7, mz = 7;  // per-axis masks: cells per axis - 1 (each >= 3)
    uint32_t sx = 1, sy = 1;          // bits of the brick coordinates along x and y (= axis bits - 2)
};

// worlds alive per device in this process (abi.hip): two of them step on two streams, i.e. beside each other
int worlds_on_device(int device);

struct Constraint {
    uint32_t kind;  // 0 fix point, 1 fix orientation
    uint32_t body;
    float target[3];
};

}  // namespace phys

struct phys_world {
    phys_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t n = 0;        // body slots the kernels run over = n_owned + max_ghosts
    uint64_t n_owned = 0;  // bodies of phys_set_bodies: every host-facing size and index check
    uint64_t max_ghosts = 0;
    float slab_lo = -3.0e38f, slab_hi = 3.0e38f, slab_reach = 0.0f;  // phys_set_slab
    phys::DevBuf<uint32_t> halo_block_counts;  // per-workgroup counts of the two ordered compactions (pack / unpack)
    uint64_t steps = 0;
    bool forces_dirty = false;      // force / torque arrays hold non-zero accumulators
    bool singular_inertia = false;  // some body's inertia tensor has det == 0 (reference panics in step)
    bool all_diag_inertia = true;
    bool uniform_inertia = true;  // all diagonal AND identical for every body
    bool aabbs_valid = false;
    bool grid_valid = false;  // bucket grid + AABBs of the last broad phase are on the device (halo entry points)

    // body SoA
    phys::DevBuf<float> pos, rot, vel /* 8n: v.xyz inv_mass w.xyz mass */, force, torque, inv_inertia, inv_inertia_diag /* 4n, valid when all_diag_inertia */, half_extent, aabb;
    phys::DevBuf<uint32_t> shape;
    phys::DevBuf<uint32_t> global_id;
    // what the narrow phase needs of a body, in ONE 64-byte line: {pos.xyz, shape} {rot ijkw} {half extent xyz, lowest y of
    // the fattened AABB}; written once per update by k_step_velocity_aabb (which has it all in registers), gathered twice
    // per candidate pair - instead of four 4-to-16-byte gathers per body out of four arrays (C5: 3M pairs per update)
    phys::DevBuf<float> geo;  // 16 floats per body (48 bytes used)

    // constraints (A3-A7)
    std::vector<phys::Constraint> constraints;
    phys::DevBuf<phys::Constraint> d_constraints;
    bool constraints_dirty = false;
    bool have_lambda = false;  // previous_solution.is_some()
    phys::DevBuf<float> cg_x, cg_r, cg_p, cg_ap, cg_rhs, cg_c, cg_scratch;
    phys::DevBuf<float> cg_jl;         // J^T lambda of entity 0 (6 floats), added by the step kernel behind gravity (quirk Q3)
    phys::DevBuf<uint32_t> cg_status;  // [0] converged flag, [1] iterations, [2] previous_solution.is_some()
    phys::DevBuf<uint32_t> cg_cols;    // col_id | col_ptr | col_rows | row_cidx (constraints.hip)
    uint32_t cg_n_cols = 0;
    uint32_t last_cg_iterations = 0;
    int32_t last_cg_converged = 1;

    // collision pipeline (A10-A12)
    uint64_t max_pairs = 0, max_manifolds = 0;
    uint32_t grid_table_size = 0;  // hashed-grid buckets (power of two)
    phys::GridShape grid_shape{};  // its split over the three axes (kernels.hpp GridShape; set by grid_plan)
    // counters, bucket_count and color_state are windows into ONE allocation (step_zero) laid out
    // [bucket counts | colouring state | StepCounters], so one memset per step zeroes all three (up to, not
    // including, StepCounters::max_extent_bits at the very end)
    phys::DevBuf<uint8_t> step_zero;
    size_t step_zero_reset_bytes = 0, step_zero_full_bytes = 0;
    phys::DevBuf<phys::StepCounters> counters;
    phys::DevBuf<uint32_t> bucket_of;    // n
    phys::DevBuf<uint32_t> bucket_count, bucket_start, bucket_cursor;  // table
    phys::DevBuf<uint32_t> slot_ids;     // 4 per bucket: the slot grid of small scenes (no scan, no scatter)
    phys::DevBuf<float> slot_box;        // the AABB next to every id slot (6 floats)
    phys::DevBuf<uint32_t> grid_ovf;     // n: bodies beyond the fourth of their bucket
    bool sorted_grid_valid = false;      // bucket_start / sorted_ids / sorted_box describe the last broad phase
    phys::DevBuf<uint32_t> sorted_ids;   // n: body ids grouped by bucket
    phys::DevBuf<float> sorted_box;      // 6n: AABBs in bucket order (streamed by the pair kernel)
    phys::DevBuf<uint32_t> scan_block_sums;
    phys::DevBuf<uint32_t> pairs;        // 2 * max_pairs
    // manifolds, geometry stage (storage order = emission order, arbitrary)
    phys::DevBuf<uint32_t> man_a, man_b, man_color;  // SoA: what the colouring rounds and the row sort scan again and again
    // geometry of a manifold as ONE 128-byte record = one cache line (32 floats: {a, b, count, -} {normal, -} 4 x {point,
    // depth}, 32 bytes spare): k_rows_build reads it through the row permutation, and seven scattered 4-to-64-byte
    // accesses per row had cost it 896 bytes of line fetches for 88 useful ones
    phys::DevBuf<float> man_geo;
    // warm starting (contact_solve.h): the geometry records of the PREVIOUS update (the two buffers swap every update), the
    // accumulated impulses every manifold's solve ended with (12 floats = three float4 per manifold: {pn, pt0, pt1} x 4
    // points, packed; this update's / the previous one's), and per manifold of this update the index of the same pair's
    // manifold in the previous update (from the colour table's value word; ~0: none)
    phys::DevBuf<float> man_geo_prev, man_imp, man_imp_prev;
    phys::DevBuf<uint32_t> man_prev;
    bool warm = false;  // warm starting is on for this world (not PHYS_FLAG_NO_WARM_START, manifold indices fit the table's 26 bits)
    phys::DevBuf<uint64_t> man_prio;
    // persistent colouring: two hash tables (this update's / the previous update's), key -> colour
    phys::DevBuf<uint32_t> unc_list;  // 2 x max_manifolds: ids of the manifolds uncoloured at the start of a round (ping-pong)
    phys::DevBuf<uint64_t> ctab;  // persistent colour table: 2 words per slot {key, stamp << 32 | colour} (kernels.hpp)
    uint32_t ctab_mask = 0;     // capacity - 1 (power of two >= 1.5 * max_manifolds)
    bool ctab_valid = false;    // a table of the previous update exists
    uint64_t color_epoch = 0;   // updates with collisions since phys_set_bodies
    bool ctab_job_pending = false;  // launch_coloring prepared a table build for launch_solver's k_rows_build
    uint32_t ctab_job_stamp = 0;    // the stamp its entries get
    bool ctab_job_all = false;      // table rebuild: every manifold of the update is inserted, not only the new ones
    phys::DevBuf<uint32_t> color_block_hist;  // [colour][workgroup] histogram / offsets of the colour sort
    // colouring state
    phys::DevBuf<unsigned long long> color_state;  // 4n: used masks | three rotating per-body priority buffers
    // solver rows, colour-major, plane-major arrays of 16-byte elements (layout: solver.hip)
    phys::DevBuf<uint32_t> row_src;  // row -> manifold (the colour sort)
    // ONE allocation of 16 planes of cap float4 each; the arrays below are windows into it (plane 0 hdr, 1 n, 2-3 tb,
    // 4-11 pt, 12-15 acc), so a kernel can address plane p of row d as all[p * cap + d] without choosing a pointer
    phys::DevBuf<float> row_all;
    phys::DevBuf<uint32_t> row_hdr;  // 4 per row: body a, body b, point count, update tickets
    phys::DevBuf<float> row_n;       // 4 per row
    phys::DevBuf<float> row_pt;      // 8 planes of float4
    phys::DevBuf<float> row_tb;      // 2 planes of float4
    phys::DevBuf<float> row_acc;     // 4 planes of float4 {pn, pt0, pt1, tag}
    // single-launch dataflow solver (k_solve_flow): in-flight body velocities travel between workgroups as
    // 16-byte granules {x, y, z, tag} (the tag says WHICH update of that body the data is: the data is its own
    // ready flag)
    phys::DevBuf<float> flow_vel;    // 8 per body: {v.xyz, tag} {w.xyz, tag}; null = per-colour launches only
    // cluster solver (cluster.hip): spatial clusters fixed at phys_set_bodies, rows sorted by (cluster, colour) per step
    uint32_t cluster_count = 0, cluster_slots = 0;  // 0 clusters: not available for this scene (dynamic: set per update)
    bool cluster_dynamic = false;           // clusters are remade every update from the bodies that have manifolds (cluster.hip)
    int cluster_cus = 0;                    // CUs of the device (dynamic planning)
    uint32_t cluster_age = 0;               // cluster steps since the homes were dealt out (dynamic: remade every kClusterDynamicPeriod)
    bool cluster_homes_valid = false;
    uint64_t cluster_cap_limit = 0;         // PHYS_DEBUG_CLUSTER_CAP: fewer homes than the LDS would hold (tests)
    phys::DevBuf<uint32_t> active_flag, active_rank;  // n + 4 each: flag / exclusive rank in the broad phase's bucket order
    bool cluster_step = false;                      // this update's rows are in (cluster, colour) order
    phys::DevBuf<uint32_t> cluster_slot;   // body -> cluster * slots + slot
    phys::DevBuf<uint32_t> cluster_body;   // cluster * slots + slot -> body (0xFFFFFFFF: empty)
    phys::DevBuf<uint32_t> body_shared;    // 2 per body: 64-bit mask of the colours in which ANOTHER cluster's row updates it
    bool flow_wide = false;          // this update: the four-lane dataflow kernel may use three workgroups per CU (exclusive GPU)
    bool seg_count_dirty = false;    // the (cluster, colour) counters were left non-zero by the last cluster step (three-launch scan)
    uint32_t seg_count_bins = 0;     // ... which used this many of them
    phys::DevBuf<uint32_t> seg_count, seg_start;  // rows per (cluster, colour) - kept behind body_shared, seg_count itself is unused - and their exclusive scan
    phys::DevBuf<uint32_t> man_rank;       // manifold -> arrival rank inside its segment
    uint32_t flow_epoch = 0;         // solves since the buffers were cleared (upper half of every tag)
    // multi-GPU halo
    phys::DevBuf<uint32_t> cross_pairs;
    uint64_t max_cross_pairs = 0;

    phys::StepHint hint;
    static constexpr int kSnapRing = 4;
    phys::StepCounters* h_snap[kSnapRing] = {};  // pinned snapshots of the counters
    hipEvent_t snap_event[kSnapRing] = {};
    bool snap_pending[kSnapRing] = {};
    bool snap_full[kSnapRing] = {};
    bool snap_tag_full = false;
    uint32_t snap_next = 0;
    phys::Profiler prof;
    uint32_t host_sticky_overflow = 0;  // overflow bits seen in counter snapshots (poll_snapshots), until phys_sync
    phys_stats stats{};
    // pinned host mirror of the counters for read-back
    phys::StepCounters* h_counters = nullptr;
};
