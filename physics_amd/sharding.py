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
import numpy as np

from . import scenes

RECORD_FLOATS = 8  # 32 bytes


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
    """bench.py glue: HaloExchange + the attach arguments of one rank's scene."""

    def __init__(self, halo, x_lo, x_hi, gids):
        self.halo, self.x_lo, self.x_hi, self.gids = halo, x_lo, x_hi, gids
        self.last_cross_pairs = 0

    def attach(self, world, scene):
        cfg = scene.config()
        self.halo.attach(world, self.x_lo, self.x_hi, self.gids, scene.half_extent, cfg.contact_margin)

    def exchange(self, world):
        self.last_cross_pairs = self.halo.exchange(world)
        return self.last_cross_pairs


def make_rank_scene(workload, rank, world_size, dist, local_rank, pinned_host=False):
    sc, x_lo, x_hi, gids = rank_scene(workload, rank, world_size)
    # boundary layers: two lattice layers per face is generous; 4x headroom
    ny_nz = sc.n // {"c1": 4, "c2": 25, "c3": 50, "c4": 100, "c5": 16, "t1m": 100}[workload]
    cap = max(4096, 16 * ny_nz)
    halo = HaloExchange(dist, rank, world_size, f"cuda:{local_rank}", cap, pinned_host=pinned_host)
    return sc, _BenchHalo(halo, x_lo, x_hi, gids)
