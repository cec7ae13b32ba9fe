// comm.hip — the one exchange step of a sharded world behind the C ABI, and the slab partition
// arithmetic. Host code only.
//
// RCCL is loaded at run time (dlopen): libphysics_hip.so itself has no link dependency on it, so single-GPU hosts
// never map the ~500 MB library, and a process that already carries an RCCL (PyTorch's bundled one) shares it - it
// must be the one built against the HIP runtime of the process. Search order: $PHYS_RCCL_PATH, "librccl.so.1" (the
// loader finds an already mapped copy by its SONAME first), /opt/rocm/lib/librccl.so.1.
// One communicator rank per world = per GPU; every collective is enqueued on the world's own stream between its
// pack and unpack kernels, so a step's exchange needs no host synchronisation. Fixed-size blocks, no count exchange,
// no ragged second collective. Two topologies:
//   all-gather (default)  every rank receives every rank's block: right for any partition, but n - 1 blocks travel to
//                         each rank and every rank scans n blocks (8 ranks of C5: 7 x 6 MB per rank and step);
//   neighbours            phys_comm_set_neighbours: ranks are slabs ordered along x by rank, none thinner than the reach,
//                         so only ranks r - 1 and r + 1 can hold bodies near this rank's faces: one grouped
//                         ncclSend / ncclRecv pair per neighbour over the direct xGMI link, two blocks to scan.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace phys;

namespace {

// the handful of RCCL entry points used, with the types of rccl.h (ncclResult_t / ncclDataType_t are ints)
typedef void* nccl_comm_t;
struct nccl_unique_id { char internal[PHYS_COMM_ID_BYTES]; };
constexpr int kNcclUint8 = 1;

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* p = getenv("PHYS_RCCL_PATH")) names.push_back(p);
        names.push_back("librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const auto& n : names) {
            api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            api.error += n + ": " + (dlerror() ? dlerror() : "?") + "; ";
        }
        if (!api.handle) return;
#define PHYS_RCCL_SYM(field, name)                                              \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name)); \
    if (!api.field) { api.error += std::string("missing symbol ") + name + "; "; }
        PHYS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        PHYS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        PHYS_RCCL_SYM(CommInitAll, "ncclCommInitAll");
        PHYS_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        PHYS_RCCL_SYM(AllGather, "ncclAllGather");
        PHYS_RCCL_SYM(Send, "ncclSend");
        PHYS_RCCL_SYM(Recv, "ncclRecv");
        PHYS_RCCL_SYM(GroupStart, "ncclGroupStart");
        PHYS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        PHYS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef PHYS_RCCL_SYM
        if (!api.error.empty() && api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.AllGather && api.Send && api.Recv &&
            api.GroupStart && api.GroupEnd && api.GetErrorString)
            api.error.clear();  // an earlier candidate failed to load, a later one is complete
    });
    return api;
}

int32_t rccl_ready() {
    RcclApi& api = rccl();
    if (!api.handle || !api.AllGather || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd) {
        set_error("RCCL is not available (librccl.so.1): " + api.error);
        return PHYS_ERR_UNSUPPORTED;
    }
    return PHYS_OK;
}

int32_t rccl_fail(const char* what, int rc) {
    const char* txt = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
    set_error(std::string(what) + ": " + txt);
    return PHYS_ERR_HIP;
}

}  // namespace

struct phys_comm {
    nccl_comm_t comm = nullptr;
    int32_t rank = 0, n_ranks = 1;
    int device = 0;
    uint64_t cap = 0;        // records per rank and step
    void* send = nullptr;    // 2 x cap records of PHYS_HALO_BODY_RECORD_BYTES (the 32-byte AABB records use the front of a block);
                             // all-gather: block 0 only; neighbours: block 0 for rank - 1 (low face), block 1 for rank + 1
    void* recv = nullptr;    // max(n_ranks, 2) x cap records
    bool neighbours = false; // block 0 of recv: from rank - 1, block 1: from rank + 1 (all 0xFF where there is none)
};

static int32_t comm_buffers(phys_comm* c) {
    PHYS_HIP_TRY(hipSetDevice(c->device));
    PHYS_HIP_TRY(hipMalloc(&c->send, 2 * c->cap * PHYS_HALO_BODY_RECORD_BYTES));
    PHYS_HIP_TRY(hipMalloc(&c->recv, (size_t)std::max(c->n_ranks, 2) * c->cap * PHYS_HALO_BODY_RECORD_BYTES));
    return PHYS_OK;
}

// pack -> [all-gather] -> unpack of one world; `gather` is skipped when the caller groups the collectives itself
static int32_t exchange_front(phys_world* w, phys_comm* c, size_t* bytes_per_rank) {
    const bool ghosts = w->max_ghosts > 0;
    const size_t rec = ghosts ? (size_t)PHYS_HALO_BODY_RECORD_BYTES : (size_t)32;
    *bytes_per_rank = c->cap * rec;
    if (!c->neighbours) {
        PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->n_halo_low, 0, 4, w->stream));
        if (ghosts) return halo_pack_bodies(w, c->send, c->cap);
        return halo_pack(w, w->slab_lo, w->slab_hi, w->slab_reach, c->send, c->cap, nullptr);
    }
    // one block per face: what rank - 1 may need lies within reach of the low face, what rank + 1 may need of the high one
    const float far = 3.0e38f;
    char* send = static_cast<char*>(c->send);
    const size_t block = c->cap * rec;
    int32_t rc = PHYS_OK;
    if (c->rank > 0)
        rc = ghosts ? halo_pack_bodies_faces(w, send, c->cap, w->slab_lo, far)
                    : halo_pack(w, w->slab_lo, far, w->slab_reach, send, c->cap, nullptr);
    // the count of the low-face block survives the packing of the high-face one (phys_stats.n_halo_records = both)
    if (c->rank > 0 && rc == PHYS_OK)
        PHYS_HIP_TRY(hipMemcpyAsync(&w->counters.p->n_halo_low, &w->counters.p->n_halo, 4, hipMemcpyDeviceToDevice, w->stream));
    else
        PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->n_halo_low, 0, 4, w->stream));
    if (c->rank + 1 >= c->n_ranks) PHYS_HIP_TRY(hipMemsetAsync(&w->counters.p->n_halo, 0, 4, w->stream));  // no high face: its count is 0
    if (rc == PHYS_OK && c->rank + 1 < c->n_ranks)
        rc = ghosts ? halo_pack_bodies_faces(w, send + block, c->cap, -far, w->slab_hi)
                    : halo_pack(w, -far, w->slab_hi, w->slab_reach, send + block, c->cap, nullptr);
    return rc;
}
static int32_t exchange_back(phys_world* w, phys_comm* c) {
    // all-gather: n blocks, this rank's own one skipped; neighbours: the two received blocks, nothing to skip
    const uint64_t total = (uint64_t)(c->neighbours ? 2 : c->n_ranks) * c->cap;
    const uint64_t own = c->neighbours ? 0 : (uint64_t)c->rank * c->cap, own_count = c->neighbours ? 0 : c->cap;
    if (w->max_ghosts > 0) return halo_unpack_ghosts(w, c->recv, total, own, own_count);
    return halo_pairs(w, c->recv, total, own, own_count, nullptr);
}
// the collective(s) of one rank, to be called between ncclGroupStart / ncclGroupEnd when there are several per thread
static int exchange_wire(phys_world* w, phys_comm* c, size_t bytes) {
    if (!c->neighbours) return rccl().AllGather(c->send, c->recv, bytes, kNcclUint8, c->comm, w->stream);
    char* recv = static_cast<char*>(c->recv);
    const size_t block = bytes;  // blocks are packed at the size of the records in use (96-byte bodies or 32-byte AABBs)
    int nr = 0;
    const char* send = static_cast<const char*>(c->send);
    if (c->rank > 0) {
        nr = rccl().Send(send, bytes, kNcclUint8, c->rank - 1, c->comm, w->stream);
        if (nr == 0) nr = rccl().Recv(recv, bytes, kNcclUint8, c->rank - 1, c->comm, w->stream);
    }
    if (nr == 0 && c->rank + 1 < c->n_ranks) {
        nr = rccl().Send(send + block, bytes, kNcclUint8, c->rank + 1, c->comm, w->stream);
        if (nr == 0) nr = rccl().Recv(recv + block, bytes, kNcclUint8, c->rank + 1, c->comm, w->stream);
    }
    return nr;
}

extern "C" {

int32_t phys_comm_unique_id(uint8_t id_out[PHYS_COMM_ID_BYTES]) {
    if (!id_out) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    const int32_t ok = rccl_ready();
    if (ok != PHYS_OK) return ok;
    nccl_unique_id id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, PHYS_COMM_ID_BYTES);
    return PHYS_OK;
}

int32_t phys_comm_create(phys_world* w, const uint8_t id[PHYS_COMM_ID_BYTES], int32_t rank, int32_t n_ranks, uint64_t capacity,
                         phys_comm** out) {
    if (!w || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || capacity == 0) {
        set_error("invalid argument");
        return PHYS_ERR_INVALID_ARG;
    }
    const int32_t ok = rccl_ready();
    if (ok != PHYS_OK) return ok;
    PHYS_HIP_TRY(hipSetDevice(w->device));
    phys_comm* c = new phys_comm();
    c->rank = rank; c->n_ranks = n_ranks; c->device = w->device; c->cap = capacity;
    nccl_unique_id uid;
    std::memcpy(uid.internal, id, PHYS_COMM_ID_BYTES);
    const int rc = rccl().CommInitRank(&c->comm, n_ranks, uid, rank);
    if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    const int32_t rb = comm_buffers(c);
    if (rb != PHYS_OK) { phys_comm_destroy(c); return rb; }
    *out = c;
    return PHYS_OK;
}

int32_t phys_comm_create_local(phys_world** worlds, int32_t n, uint64_t capacity, phys_comm** comms_out) {
    if (!worlds || !comms_out || n < 1 || capacity == 0) { set_error("invalid argument"); return PHYS_ERR_INVALID_ARG; }
    const int32_t ok = rccl_ready();
    if (ok != PHYS_OK) return ok;
    std::vector<int> devs(n);
    for (int32_t k = 0; k < n; ++k) {
        if (!worlds[k]) { set_error("null world"); return PHYS_ERR_INVALID_ARG; }
        devs[k] = worlds[k]->device;
        for (int32_t j = 0; j < k; ++j)
            if (devs[j] == devs[k]) { set_error("phys_comm_create_local needs one DIFFERENT device per world (RCCL: one rank per GPU)"); return PHYS_ERR_INVALID_ARG; }
    }
    std::vector<nccl_comm_t> raw(n, nullptr);
    const int rc = rccl().CommInitAll(raw.data(), n, devs.data());
    if (rc != 0) return rccl_fail("ncclCommInitAll", rc);
    for (int32_t k = 0; k < n; ++k) {
        phys_comm* c = new phys_comm();
        c->comm = raw[k]; c->rank = k; c->n_ranks = n; c->device = devs[k]; c->cap = capacity;
        comms_out[k] = c;
        const int32_t rb = comm_buffers(c);
        if (rb != PHYS_OK) return rb;
    }
    return PHYS_OK;
}

int32_t phys_comm_destroy(phys_comm* c) {
    if (!c) return PHYS_OK;
    (void)hipSetDevice(c->device);
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    delete c;
    return PHYS_OK;
}

int32_t phys_halo_exchange(phys_world* w, phys_comm* c) {
    if (!w || !c) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    if (c->device != w->device) { set_error("communicator belongs to another device"); return PHYS_ERR_INVALID_ARG; }
    PHYS_HIP_TRY(hipSetDevice(w->device));
    size_t bytes = 0;
    int32_t rc = exchange_front(w, c, &bytes);
    if (rc != PHYS_OK) return rc;
    int nr = 0;
    if (c->neighbours) {  // up to two send / receive pairs: one group, or they would wait for each other
        nr = rccl().GroupStart();
        if (nr != 0) return rccl_fail("ncclGroupStart", nr);
        nr = exchange_wire(w, c, bytes);
        const int ne = rccl().GroupEnd();
        if (nr == 0) nr = ne;
    } else {
        nr = exchange_wire(w, c, bytes);
    }
    if (nr != 0) return rccl_fail(c->neighbours ? "ncclSend / ncclRecv" : "ncclAllGather", nr);
    return exchange_back(w, c);
}

int32_t phys_comm_set_neighbours(phys_comm* c, int32_t enable) {
    if (!c) { set_error("null argument"); return PHYS_ERR_INVALID_ARG; }
    PHYS_HIP_TRY(hipSetDevice(c->device));
    c->neighbours = enable != 0;
    // a block nobody sends into (first / last rank, or one rank alone) reads as "no records" for ever
    if (c->neighbours) {
        PHYS_HIP_TRY(hipMemset(c->recv, 0xFF, (size_t)2 * c->cap * PHYS_HALO_BODY_RECORD_BYTES));
        PHYS_HIP_TRY(hipDeviceSynchronize());  // set-up call: the fill is in place before any world's stream reads the blocks
    }
    return PHYS_OK;
}

// one process, n worlds on n devices: the n collectives of a step must be issued as ONE group
int32_t phys_halo_exchange_all(phys_world** worlds, phys_comm** comms, int32_t n) {
    if (!worlds || !comms || n < 1) { set_error("invalid argument"); return PHYS_ERR_INVALID_ARG; }
    std::vector<size_t> bytes(n, 0);
    for (int32_t k = 0; k < n; ++k) {
        PHYS_HIP_TRY(hipSetDevice(worlds[k]->device));
        const int32_t rc = exchange_front(worlds[k], comms[k], &bytes[k]);
        if (rc != PHYS_OK) return rc;
    }
    int nr = rccl().GroupStart();
    if (nr != 0) return rccl_fail("ncclGroupStart", nr);
    for (int32_t k = 0; k < n; ++k) {
        nr = exchange_wire(worlds[k], comms[k], bytes[k]);
        if (nr != 0) { (void)rccl().GroupEnd(); return rccl_fail("halo exchange", nr); }
    }
    nr = rccl().GroupEnd();
    if (nr != 0) return rccl_fail("ncclGroupEnd", nr);
    for (int32_t k = 0; k < n; ++k) {
        PHYS_HIP_TRY(hipSetDevice(worlds[k]->device));
        const int32_t rc = exchange_back(worlds[k], comms[k]);
        if (rc != PHYS_OK) return rc;
    }
    return PHYS_OK;
}

// ---- slab partition: host arithmetic only -----------------------------------------------------------------------
int32_t phys_slab_histogram(const float* pos, uint64_t n, float x_min, float x_max, uint32_t bins, uint64_t* hist) {
    if ((n && !pos) || !hist || bins == 0 || !(x_min < x_max)) { set_error("invalid argument"); return PHYS_ERR_INVALID_ARG; }
    const double scale = (double)bins / ((double)x_max - (double)x_min);
    for (uint64_t i = 0; i < n; ++i) {
        const double t = ((double)pos[3 * i] - (double)x_min) * scale;
        const int64_t b = t < 0.0 ? 0 : (t >= (double)bins ? (int64_t)bins - 1 : (int64_t)t);  // outliers: first / last bin
        hist[b] += 1;
    }
    return PHYS_OK;
}

int32_t phys_slab_cuts(const uint64_t* hist, uint32_t bins, float x_min, float x_max, int32_t n_ranks, float* cuts_out) {
    if (!hist || !cuts_out || bins == 0 || n_ranks < 1 || !(x_min < x_max)) { set_error("invalid argument"); return PHYS_ERR_INVALID_ARG; }
    uint64_t total = 0;
    for (uint32_t b = 0; b < bins; ++b) total += hist[b];
    const double width = ((double)x_max - (double)x_min) / (double)bins;
    cuts_out[0] = x_min;
    cuts_out[n_ranks] = x_max;
    uint64_t before = 0;  // bodies in the bins in front of bin b
    uint32_t b = 0;
    for (int32_t r = 1; r < n_ranks; ++r) {
        // plane r leaves r / n_ranks of the bodies on its left: walk to the bin that crosses that count and
        // interpolate inside it (bodies taken as uniformly spread over their bin)
        const double want = (double)total * (double)r / (double)n_ranks;
        while (b < bins && (double)(before + hist[b]) < want) { before += hist[b]; ++b; }
        double x;
        if (b >= bins) x = x_max;
        else {
            const double inside = hist[b] ? (want - (double)before) / (double)hist[b] : 0.0;
            x = (double)x_min + ((double)b + inside) * width;
        }
        cuts_out[r] = (float)x;
        if (cuts_out[r] < cuts_out[r - 1]) cuts_out[r] = cuts_out[r - 1];  // monotone whatever the rounding
    }
    return PHYS_OK;
}

int32_t phys_slab_owners(const float* pos, uint64_t n, const float* cuts, int32_t n_ranks, int32_t* owner_out) {
    if ((n && (!pos || !owner_out)) || !cuts || n_ranks < 1) { set_error("invalid argument"); return PHYS_ERR_INVALID_ARG; }
    for (uint64_t i = 0; i < n; ++i) {
        const float x = pos[3 * i];
        // the rank r with cuts[r] <= x < cuts[r + 1]; beyond the outer planes: first / last rank
        const float* it = std::upper_bound(cuts + 1, cuts + n_ranks, x);
        owner_out[i] = (int32_t)(it - (cuts + 1));
    }
    return PHYS_OK;
}

}  // extern "C"
