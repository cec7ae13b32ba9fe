"""Sharded broad phase over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" in the CPU tests).
Bodies are partitioned into slabs along x; each rank simulates the bodies it owns (narrow phase and
solver stay per-GPU on owned bodies, as BASELINE.json's north_star prescribes) and, once per step,
exchanges the AABBs of the bodies near its slab faces:

    pack (HIP kernel)  ->  all_gather of fixed-size record blocks (RCCL)  ->  cross pairs (HIP kernel)

A record is 32 bytes {lo xyz, hi xyz, global id, pad}; unused slots carry global id 0xFFFFFFFF. The
message is small (C2-shaped slab: ~1.3k records = 42 KB per rank), i.e. latency-bound on xGMI, which is why
it is ONE collective of a fixed shape instead of a count exchange followed by a ragged gather.
Cross pairs are emitted by the rank owning the body with the smaller global id, so the union over ranks
of (local pairs, cross pairs) equals the single-world pair set (tests/test_sharding_gloo.py).

The object handed to HaloExchange only needs halo_pack / halo_pairs / get_cross_pairs, so the host logic
here is exercised on CPU with a stand-in world defined in the tests.
"""
import ctypes as C

import numpy as np

from . import _abi, scenes

RECORD_FLOATS = 8  # 32 bytes: the AABB record of the broad-phase-only exchange
BODY_RECORD_FLOATS = 24  # 96 bytes: the full-state record of the ghost exchange (PHYS_HALO_BODY_RECORD_BYTES)


# ---- slab partition: thin wrappers over the host arithmetic of the library (phys_slab_*, no device involved) ------
def slab_histogram(pos, x_min, x_max, bins):
    lib = _abi.load_library()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    hist = np.zeros(bins, np.uint64)
    rc = lib.phys_slab_histogram(pos.ctypes.data_as(_abi.f32p), len(pos), x_min, x_max, bins, hist.ctypes.data_as(_abi.u64p))
    if rc != 0:
        raise RuntimeError(lib.phys_last_error().decode())
    return hist


def slab_cuts(hist, x_min, x_max, n_ranks):
    lib = _abi.load_library()
    hist = np.ascontiguousarray(hist, np.uint64)
    cuts = np.zeros(n_ranks + 1, np.float32)
    rc = lib.phys_slab_cuts(hist.ctypes.data_as(_abi.u64p), len(hist), x_min, x_max, n_ranks, cuts.ctypes.data_as(_abi.f32p))
    if rc != 0:
        raise RuntimeError(lib.phys_last_error().decode())
    return cuts


def slab_owners(pos, cuts):
    lib = _abi.load_library()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    cuts = np.ascontiguousarray(cuts, np.float32)
    owner = np.zeros(len(pos), np.int32)
    rc = lib.phys_slab_owners(pos.ctypes.data_as(_abi.f32p), len(pos), cuts.ctypes.data_as(_abi.f32p), len(cuts) - 1,
                              owner.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError(lib.phys_last_error().decode())
    return owner


def slab_bounds(scene_nx, spacing, rank):
    width = scene_nx * spacing
    x0 = rank * width
    return x0, x0 - 0.5 * width, x0 + 0.5 * width


def rank_scene(workload, rank, world_size, shape=None):
    """Weak scaling: every rank owns one copy of the workload's lattice, the copies laid side by side
    along x at the lattice spacing (so the slab faces carry real boundary pairs). Returns
    (scene, x_lo, x_hi, global_ids)."""
    shapes = {"c1": (4, 4, 4, 2.5), "c2": (25, 16, 25, 2.5), "c3": (50, 40, 50, 2.5), "c4": (100, 100, 100, 2.2),
              "c5": (16, 1000, 16, 2.0), "t1m": (100, 100, 100, 2.5)}
    nx, ny, nz, spacing = shape if shape is not None else shapes[workload]
    x0, x_lo, x_hi = slab_bounds(nx, spacing, rank)
    if workload in ("c1", "c2", "t1m"):
        sc = scenes.falling_cubes(nx, ny, nz, f"{workload.upper()}_slab{rank}of{world_size}", spacing=spacing, x0=x0)
    elif workload == "c4":
        sc = scenes.c4(nx, ny, nz, x0=x0)
    else:
        sc = scenes.SCENES[workload]()
        sc.pos = sc.pos.copy()
        sc.pos[:, 0] += np.float32(x0)
    # decorrelate the jitter between ranks deterministically
    if workload in ("c1", "c2", "t1m", "c4"):
        jit = {"c4": 0.3}.get(workload, 0.05)
        sc.pos = scenes.lattice(nx, ny, nz, spacing, 2.0, jit, seed=12345 + 7919 * rank, x0=x0)
    gids = (np.arange(sc.n, dtype=np.uint64) + np.uint64(rank) * np.uint64(sc.n)).astype(np.uint32)
    return sc, float(x_lo), float(x_hi), gids


def strong_rank_scene(workload, rank, world_size, bins=4096, scene=None):
    """STRONG scaling (BASELINE.json config 4: "1M bodies ... sharded across 8 x MI355X"): ONE scene cut into world_size
    equal-count x-slabs - histogram of body x -> prefix sums -> cut planes (phys_slab_*); every rank
    derives the same planes from the same scene and keeps the bodies of its slab under their ids of the whole scene.
    Returns (scene of this rank, x_lo, x_hi, global ids, bodies of the whole scene)."""
    full = scene if scene is not None else scenes.SCENES[workload]()
    x = full.pos[:, 0]
    x_min, x_max = float(x.min()) - 1.0e-3, float(x.max()) + 1.0e-3
    cuts = slab_cuts(slab_histogram(full.pos, x_min, x_max, bins), x_min, x_max, world_size)
    mine = np.nonzero(slab_owners(full.pos, cuts) == rank)[0]
    sc = scenes.Scene(f"{full.name}_slab{rank}of{world_size}", np.ascontiguousarray(full.pos[mine]),
                      None if full.shape_type is None else np.ascontiguousarray(full.shape_type[mine]),
                      None if full.half_extent is None else np.ascontiguousarray(full.half_extent[mine]),
                      full.flags, full.solver_iterations, **full.cfg_overrides)
    far = 1.0e30  # the outer faces of the first and the last slab: nobody lives beyond them
    x_lo = -far if rank == 0 else float(cuts[rank])
    x_hi = far if rank == world_size - 1 else float(cuts[rank + 1])
    return sc, x_lo, x_hi, mine.astype(np.uint32), full.n


def static_reach(half_extent, margin):
    """Upper bound of any fattened AABB edge: a rotated box is at most 2*|h| wide."""
    h = np.asarray(half_extent, np.float64).reshape(-1, 3)
    return float(2.0 * np.sqrt((h * h).sum(axis=1)).max() + 2.0 * margin) * 1.001


class HaloExchange:
    """Per-step boundary exchange of one rank. `dist` is torch.distributed (already initialised);
    `device` is the torch device the record buffers live on ("cuda:k" for RCCL, "cpu" for gloo)."""

    def __init__(self, dist, rank, world_size, device, cap, pinned_host=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.rank = rank
        self.world_size = world_size
        self.cap = int(cap)
        self.device = device
        kw = dict(dtype=torch.float32, device=device)
        if pinned_host:  # host memory the GPU can address (rehearsal of the N > 1 flow over gloo on one GPU)
            kw = dict(dtype=torch.float32, device="cpu", pin_memory=True)
            self.device = "cpu"
        self.send = torch.empty((self.cap, RECORD_FLOATS), **kw)
        self.recv = torch.empty((world_size, self.cap, RECORD_FLOATS), **kw)
        self.x_lo = self.x_hi = 0.0
        self.reach = 0.0
        self.last_cross_pairs = 0
        self.last_halo_records = 0
        self._ext_stream = None

    def attach(self, world, x_lo, x_hi, global_ids, half_extent, margin):
        self.x_lo, self.x_hi = float(x_lo), float(x_hi)
        self._ext_stream = None  # a new world launches on a new stream
        world.set_global_ids(global_ids)
        # reach must cover the largest AABB of ANY rank: one all-reduce at set-up time
        r = self.torch.tensor([static_reach(half_extent, margin)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(r, op=self.dist.ReduceOp.MAX)
        self.reach = float(r.item())

    def exchange(self, world):
        """Call after world.update(): packs this rank's boundary AABBs, all-gathers them, and finds the
        cross pairs this rank owns.

        Device buffers (RCCL): everything - pack kernel, all-gather, cross-pair kernel - is ENQUEUED on the
        world's own HIP stream (wrapped as a torch ExternalStream so the collective is ordered on it); no
        host synchronisation, returns None (phys_get_stats().n_cross_pairs has the count).
        Host buffers (gloo, CPU tests / one-GPU rehearsal): synchronous, returns the cross-pair count."""
        own_first = self.rank * self.cap
        if self.send.is_cuda:
            if self._ext_stream is None:
                view = world.device_view()
                self._ext_stream = self.torch.cuda.ExternalStream(view.stream, device=self.send.device)
            world.halo_pack(self.x_lo, self.x_hi, self.reach, self.send.data_ptr(), self.cap, wait=False)
            with self.torch.cuda.stream(self._ext_stream):
                # ONE collective of fixed shape (latency-bound on xGMI: no count exchange, no ragged gather)
                self.dist.all_gather_into_tensor(self.recv.view(-1, RECORD_FLOATS), self.send)
            world.halo_pairs(self.recv.data_ptr(), self.world_size * self.cap, own_first, self.cap, wait=False)
            self.last_cross_pairs = None
            return None
        n = world.halo_pack(self.x_lo, self.x_hi, self.reach, self.send.data_ptr(), self.cap)
        self.last_halo_records = n
        self.dist.all_gather_into_tensor(self.recv.view(-1, RECORD_FLOATS), self.send)
        self.last_cross_pairs = world.halo_pairs(self.recv.data_ptr(), self.world_size * self.cap, own_first, self.cap)
        return self.last_cross_pairs


class _BenchHalo:
    """bench.py glue: the exchange object + the attach arguments of one rank's scene. `before_update` says where the
    exchange belongs in a step: ghost bodies must be in place BEFORE the update that collides with them; the AABB-only
    exchange of the broad-phase-only workload (C4) uses the grid the update has just built."""

    def __init__(self, halo, x_lo, x_hi, gids, before_update):
        self.halo, self.x_lo, self.x_hi, self.gids, self.before_update = halo, x_lo, x_hi, gids, before_update
        self.last_cross_pairs = 0

    def attach(self, world, scene):
        cfg = scene.config()
        self.halo.attach(world, self.x_lo, self.x_hi, self.gids, scene.half_extent, cfg.contact_margin)

    def exchange(self, world):
        self.last_cross_pairs = self.halo.exchange(world)
        return self.last_cross_pairs


def make_rank_scene(workload, rank, world_size, dist, local_rank, pinned_host=False, strong=False):
    """strong: the workload's ONE scene cut into equal-count slabs (the broad-phase-only configuration C4 by default);
    otherwise weak scaling: a workload-shaped slab per rank, side by side along x."""
    if strong:
        sc, x_lo, x_hi, gids, _ = strong_rank_scene(workload, rank, world_size)
    else:
        sc, x_lo, x_hi, gids = rank_scene(workload, rank, world_size)
    # records per slab face and step: the bodies that start within reach of it, with 1.5x headroom for what the pile does
    # later (a block is sent whole: capacity is bandwidth), never fewer than 4096
    reach = static_reach(sc.half_extent, sc.config().contact_margin)
    x = sc.pos[:, 0]
    near = max(int(np.count_nonzero(x < x_lo + reach)), int(np.count_nonzero(x > x_hi - reach)))
    cap = max(4096, near + near // 2)
    if sc.flags & scenes.FLAG_BROADPHASE_ONLY:
        if pinned_host:  # rehearsal of N ranks on one GPU: gloo over pinned host buffers around the pack / pairs kernels
            halo = HaloExchange(dist, rank, world_size, f"cuda:{local_rank}", cap, pinned_host=True)
        else:
            # behind the C ABI (phys_halo_exchange: pack AABB records -> RCCL -> cross pairs, on the world's stream); the
            # slabs are ordered by rank and far wider than the reach: one grouped ncclSend / ncclRecv pair per face
            halo = GhostExchange(dist, rank, world_size, cap, transport="rccl", neighbours=True)
        return sc, _BenchHalo(halo, x_lo, x_hi, gids, before_update=False)
    # full step: neighbours' boundary bodies become ghosts of this rank (contacts across the cut planes)
    sc.cfg_overrides["max_ghosts"] = 2 * cap
    # the rank scenes are slabs side by side along x in rank order, each far wider than the reach: neighbours only, one
    # block per face (the host transport of the rehearsal gathers one block holding both faces)
    halo = GhostExchange(dist, rank, world_size, 2 * cap if pinned_host else cap, transport="host" if pinned_host else "rccl",
                         neighbours=True)
    return sc, _BenchHalo(halo, x_lo, x_hi, gids, before_update=True)


class GhostExchange:
    """Per-step exchange of a rank whose world keeps ghost bodies (phys_config.max_ghosts > 0): call exchange(world)
    BEFORE world.update(). Afterwards the neighbours' boundary bodies sit in this world's ghost slots as kinematic
    bodies, and the ordinary broad phase / narrow phase / solver of the update collide the owned bodies with them.

    transport "rccl": everything happens behind the C ABI (phys_halo_exchange: pack kernel, ncclAllGather, unpack
    kernel, all enqueued on the world's stream); torch.distributed only carries the 128-byte communicator id once.
    transport "host": the same pack / unpack entry points around a torch.distributed all_gather on pinned host
    buffers (gloo): the CPU tests with a stand-in world, and the rehearsal of N ranks on a one-GPU box."""

    def __init__(self, dist, rank, world_size, cap, transport="rccl", neighbours=False):
        """neighbours (rccl transport): the ranks are x-slabs ordered by rank and none is thinner than the reach, so a rank
        exchanges with ranks r - 1 and r + 1 only (phys_comm_set_neighbours) - two point-to-point messages over direct
        xGMI links instead of an all-gather of every rank's block."""
        import torch
        self.torch, self.dist, self.rank, self.world_size, self.cap = torch, dist, rank, world_size, int(cap)
        self.transport = transport
        self.neighbours = bool(neighbours)
        self.comm = None
        self.x_lo = self.x_hi = self.reach = 0.0
        if transport == "host":
            pin = torch.cuda.is_available()
            self.send = torch.empty((self.cap, BODY_RECORD_FLOATS), dtype=torch.float32, pin_memory=pin)
            self.recv = torch.empty((world_size, self.cap, BODY_RECORD_FLOATS), dtype=torch.float32, pin_memory=pin)

    def attach(self, world, x_lo, x_hi, global_ids, half_extent, margin):
        """One collective at set-up: the reach must cover the largest body of ANY rank."""
        self.x_lo, self.x_hi = float(x_lo), float(x_hi)
        device = "cpu" if self.transport == "host" else f"cuda:{self.torch.cuda.current_device()}"
        r = self.torch.tensor([static_reach(half_extent, margin)], dtype=self.torch.float64, device=device)
        self.dist.all_reduce(r, op=self.dist.ReduceOp.MAX)
        self.reach = float(r.item())
        world.set_global_ids(global_ids)
        world.set_slab(self.x_lo, self.x_hi, self.reach)
        if self.transport == "rccl":
            from .world import Comm
            if self.comm is not None:
                self.comm.close()
            ids = [Comm.unique_id() if self.rank == 0 else None]
            self.dist.broadcast_object_list(ids, src=0)
            self.comm = Comm(world, ids[0], self.rank, self.world_size, self.cap, neighbours=self.neighbours)

    def move_slab(self, world, x_lo, x_hi):
        self.x_lo, self.x_hi = float(x_lo), float(x_hi)
        world.set_slab(self.x_lo, self.x_hi, self.reach)

    def exchange(self, world):
        if self.transport == "rccl":
            world.halo_exchange(self.comm)
            return
        world.halo_pack_bodies(self.send.data_ptr(), self.cap)
        world.sync()
        self.dist.all_gather_into_tensor(self.recv.view(-1, BODY_RECORD_FLOATS), self.send)
        world.halo_unpack_ghosts(self.recv.data_ptr(), self.world_size * self.cap, self.rank * self.cap, self.cap)
        world.sync()


class SlabSharder:
    """Equal-count x-slabs over the ranks, re-cut on demand (histogram of body x -> prefix sums ->
    cut planes; every k steps), with the hand-over of every body that changed owner. The partition arithmetic is the
    library's (phys_slab_*); the transport of histograms and body records is torch.distributed (a Rust host would use
    its own). The hand-over goes through the host - download the rank's state, exchange the leavers, upload with
    phys_set_bodies - which is what a re-cut every few dozen steps can afford (PCIe: 124 B per body both ways) and
    keeps the device-side body set static between re-cuts.

    `state` is a dict of per-body arrays of THIS rank: pos, rot, lin_vel, ang_vel, mass, inertia (n x 9), shape_type,
    half_extent, gid. make_world(state) -> world builds / refills the rank's world from such a dict."""

    FIELDS = (("pos", 3, np.float32), ("rot", 4, np.float32), ("lin_vel", 3, np.float32), ("ang_vel", 3, np.float32),
              ("mass", 1, np.float32), ("inertia", 9, np.float32), ("shape_type", 1, np.uint32), ("half_extent", 3, np.float32),
              ("gid", 1, np.uint32))

    def __init__(self, dist, rank, world_size, bins=4096):
        import torch
        self.torch, self.dist, self.rank, self.world_size, self.bins = torch, dist, rank, world_size, bins
        self.cuts = None

    def compute_cuts(self, pos):
        """Collective: global x-range (all-reduce min / max), summed histogram, then every rank derives the same planes."""
        t = self.torch
        x = np.asarray(pos, np.float32).reshape(-1, 3)[:, 0]
        lo = t.tensor([float(x.min()) if len(x) else 3.0e38], dtype=t.float64)
        hi = t.tensor([float(x.max()) if len(x) else -3.0e38], dtype=t.float64)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        x_min, x_max = float(lo.item()) - 1.0e-3, float(hi.item()) + 1.0e-3
        hist = t.from_numpy(slab_histogram(pos, x_min, x_max, self.bins).astype(np.int64))
        self.dist.all_reduce(hist)
        self.cuts = slab_cuts(hist.numpy().astype(np.uint64), x_min, x_max, self.world_size)
        return self.cuts

    def my_slab(self):
        lo = -3.0e38 if self.rank == 0 else float(self.cuts[self.rank])
        hi = 3.0e38 if self.rank == self.world_size - 1 else float(self.cuts[self.rank + 1])
        return lo, hi

    def migrate(self, state):
        """Collective: every body goes to the rank whose slab holds its x. Returns this rank's new state (bodies in
        global-id order, so the result does not depend on who sent what when)."""
        owner = slab_owners(state["pos"], self.cuts)
        n = len(owner)
        width = sum(w for _, w, _ in self.FIELDS) + 1
        rec = np.zeros((n, width), np.float64)  # exact for f32 values and u32 ids
        c = 0
        for name, w, _ in self.FIELDS:
            rec[:, c:c + w] = np.asarray(state[name]).reshape(n, w)
            c += w
        rec[:, c] = owner
        gathered = [None] * self.world_size
        self.dist.all_gather_object(gathered, rec[owner != self.rank])
        mine = [rec[owner == self.rank]] + [g[g[:, -1] == self.rank] for r, g in enumerate(gathered) if r != self.rank and len(g)]
        allrec = np.concatenate(mine) if mine else rec[:0]
        gid_col = sum(w for _, w, _ in self.FIELDS[:-1])
        allrec = allrec[np.argsort(allrec[:, gid_col], kind="stable")]
        out, c = {}, 0
        for name, w, dt in self.FIELDS:
            a = allrec[:, c:c + w].astype(dt)
            out[name] = a.reshape(-1) if w == 1 else a
            c += w
        return out
