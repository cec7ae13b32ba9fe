"""CPU: slab partition and re-cut of the sharded world.

Part 1 (no process group, no device): the partition arithmetic of the C ABI - phys_slab_histogram / phys_slab_cuts /
phys_slab_owners - on skewed body distributions: equal counts within one bin's worth of bodies, monotone planes,
every body exactly one owner, outliers to the end ranks.
Part 2 (gloo, world_size 2 and 3): physics_amd.sharding.SlabSharder (collective cuts + hand-over of the bodies that
changed owner) and GhostExchange (host transport) with a numpy stand-in for the two ghost entry points of the world:
after a re-cut the union of the ranks' bodies is the original set, each body sits on the rank its x belongs to, and
every rank's ghosts are exactly the other ranks' bodies within `reach` of its slab."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from physics_amd import sharding


# ---------------------------------------------------------------- part 1: the partition arithmetic
@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_equal_count_cuts_on_skewed_distributions(n_ranks):
    rng = np.random.default_rng(n_ranks)
    for case in range(3):
        n = 50_000
        x = {0: rng.uniform(-100, 100, n), 1: rng.exponential(20.0, n) - 30.0,
             2: np.concatenate([rng.normal(-80, 2, n // 2), rng.normal(90, 10, n - n // 2)])}[case]
        pos = np.zeros((n, 3), np.float32)
        pos[:, 0] = x
        x_min, x_max, bins = float(pos[:, 0].min()) - 1e-3, float(pos[:, 0].max()) + 1e-3, 4096
        hist = sharding.slab_histogram(pos, x_min, x_max, bins)
        assert hist.sum() == n
        cuts = sharding.slab_cuts(hist, x_min, x_max, n_ranks)
        assert len(cuts) == n_ranks + 1 and cuts[0] == np.float32(x_min) and cuts[-1] == np.float32(x_max)
        assert (np.diff(cuts) >= 0).all()
        owner = sharding.slab_owners(pos, cuts)
        assert owner.min() >= 0 and owner.max() <= n_ranks - 1
        for r in range(n_ranks):  # the rule of phys_slab_owners: cuts[r] <= x < cuts[r + 1]
            inside = pos[owner == r, 0]
            if r > 0:
                assert (inside >= cuts[r]).all()
            if r < n_ranks - 1:
                assert (inside < cuts[r + 1]).all()
        counts = np.bincount(owner, minlength=n_ranks)
        # equal counts up to the bodies of the bins the planes were interpolated in
        assert np.abs(counts - n / n_ranks).max() <= max(2 * hist.max(), 8), counts


def test_histograms_add_up_and_outliers_go_to_the_end_ranks():
    rng = np.random.default_rng(0)
    pos = np.zeros((1000, 3), np.float32)
    pos[:, 0] = rng.uniform(0, 10, 1000)
    h = sharding.slab_histogram(pos[:400], 0.0, 10.0, 64) + sharding.slab_histogram(pos[400:], 0.0, 10.0, 64)
    assert np.array_equal(h, sharding.slab_histogram(pos, 0.0, 10.0, 64))  # ranks sum their histograms
    cuts = sharding.slab_cuts(h, 0.0, 10.0, 4)
    far = np.array([[-50.0, 0, 0], [70.0, 0, 0], [cuts[2], 0, 0]], np.float32)
    assert sharding.slab_owners(far, cuts).tolist() == [0, 3, 2]


# ---------------------------------------------------------------- part 2: re-cut + ghost exchange over gloo
class StandInWorld:
    """phys_set_slab / phys_halo_pack_bodies / phys_halo_unpack_ghosts of include/physics_hip.h restated on numpy."""

    def __init__(self, state):
        self.s = state
        self.ghosts = np.zeros((0, sharding.BODY_RECORD_FLOATS), np.float32)

    def set_global_ids(self, gids):
        assert np.array_equal(np.asarray(gids, np.uint32), self.s["gid"])

    def set_slab(self, lo, hi, reach):
        self.lo, self.hi, self.reach = np.float32(lo), np.float32(hi), np.float32(reach)

    def sync(self):
        pass

    @staticmethod
    def _view(ptr, rows):
        buf = (ctypes.c_float * (rows * sharding.BODY_RECORD_FLOATS)).from_address(ptr)
        return np.ctypeslib.as_array(buf).reshape(rows, sharding.BODY_RECORD_FLOATS)

    def halo_pack_bodies(self, ptr, cap):
        out = self._view(ptr, cap)
        out.view(np.uint32)[:] = 0xFFFFFFFF
        x = self.s["pos"][:, 0]
        idx = np.nonzero((x < self.lo + self.reach) | (x > self.hi - self.reach))[0]  # index order: the ordered compaction
        assert len(idx) <= cap
        k = len(idx)
        out[:k] = 0
        out[:k, 0:3] = self.s["pos"][idx]
        out[:k, 3:7] = self.s["rot"][idx]
        out[:k, 7:10] = self.s["lin_vel"][idx]
        out[:k, 10:13] = self.s["ang_vel"][idx]
        out[:k, 13:16] = self.s["half_extent"][idx]
        out[:k, 16] = self.s["shape_type"][idx].view(np.float32)
        out[:k, 17] = self.s["gid"][idx].view(np.float32)

    def halo_unpack_ghosts(self, ptr, n_records, skip_first=0, skip_count=0):
        rec = self._view(ptr, n_records).copy()
        gid = rec[:, 17].copy().view(np.uint32)
        keep = gid != 0xFFFFFFFF
        keep[skip_first:skip_first + skip_count] = False
        keep &= (rec[:, 0] >= self.lo - self.reach) & (rec[:, 0] <= self.hi + self.reach)
        self.ghosts = rec[keep]


def _initial_state(rank, world_size):
    """Every rank starts with an arbitrary third of a skewed cloud: the first re-cut has to move most bodies."""
    rng = np.random.default_rng(99)
    n = 3000
    pos = np.zeros((n, 3), np.float32)
    pos[:, 0] = np.concatenate([rng.normal(-20, 3, n // 3), rng.uniform(-10, 40, n - n // 3)])
    pos[:, 1:] = rng.uniform(0, 10, (n, 2))
    mine = np.arange(n) % world_size == rank
    k = int(mine.sum())
    return dict(pos=pos[mine], rot=np.tile(np.array([0, 0, 0, 1], np.float32), (k, 1)), lin_vel=rng.normal(size=(n, 3)).astype(np.float32)[mine],
                ang_vel=np.zeros((k, 3), np.float32), mass=np.ones(k, np.float32),
                inertia=np.tile(np.eye(3, dtype=np.float32).reshape(9), (k, 1)), shape_type=np.full(k, 2, np.uint32),
                half_extent=np.ones((k, 3), np.float32), gid=np.arange(n, dtype=np.uint32)[mine]), pos


def _worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        state, _ = _initial_state(rank, world_size)
        sh = sharding.SlabSharder(dist, rank, world_size, bins=512)
        cuts = sh.compute_cuts(state["pos"])
        state = sh.migrate(state)
        lo, hi = sh.my_slab()
        assert ((state["pos"][:, 0] >= lo) & (state["pos"][:, 0] < hi)).all()
        assert (np.diff(state["gid"].astype(np.int64)) > 0).all()  # global-id order: independent of arrival order
        # the bodies move, the cut planes follow: second re-cut after a shift of the cloud
        state["pos"][:, 0] += np.where(state["pos"][:, 0] > 0, 15.0, 0.0).astype(np.float32)
        cuts2 = sh.compute_cuts(state["pos"])
        state = sh.migrate(state)
        lo, hi = sh.my_slab()
        assert ((state["pos"][:, 0] >= lo) & (state["pos"][:, 0] < hi)).all()
        # ghost exchange on the new partition
        w = StandInWorld(state)
        gx = sharding.GhostExchange(dist, rank, world_size, cap=2048, transport="host")
        gx.attach(w, lo, hi, state["gid"], state["half_extent"], 0.02)
        gx.exchange(w)
        np.save(os.path.join(out_dir, f"state_{rank}.npy"), np.concatenate([state["gid"][:, None].astype(np.float64), state["pos"].astype(np.float64), state["lin_vel"].astype(np.float64)], 1))
        np.save(os.path.join(out_dir, f"ghosts_{rank}.npy"), w.ghosts)
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([lo, hi, gx.reach] + cuts.tolist() + cuts2.tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world_size", [2, 3])
def test_recut_hands_bodies_over_and_ghosts_follow_the_new_planes(world_size, tmp_path):
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    states = [np.load(tmp_path / f"state_{r}.npy") for r in range(world_size)]
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world_size)]
    allrec = np.concatenate(states)
    # nobody lost, nobody duplicated, velocities travelled with their bodies
    assert sorted(allrec[:, 0].astype(int).tolist()) == list(range(3000))
    _, pos0 = _initial_state(0, world_size)
    rng = np.random.default_rng(99)
    counts = [len(s) for s in states]
    assert max(counts) - min(counts) <= 3000 * 0.05, counts  # equal counts (512 bins, skewed cloud)
    for r in range(world_size):
        assert np.array_equal(metas[r][3:], metas[0][3:])  # every rank derived the same planes, both times
        lo, hi, reach = metas[r][:3]
        others = np.concatenate([states[q] for q in range(world_size) if q != r])
        want = others[(others[:, 1] >= np.float32(lo) - np.float32(reach)) & (others[:, 1] <= np.float32(hi) + np.float32(reach))]
        # ... of which only boundary bodies of their owners were sent; with reach >= body size both filters agree on
        # every body that can touch a body of rank r
        ghosts = np.load(tmp_path / f"ghosts_{r}.npy")
        got = set(ghosts[:, 17].copy().view(np.uint32).tolist())
        near = set(want[:, 0].astype(int).tolist())
        assert got <= near
        touching = others[(others[:, 1] >= lo - 2.1) & (others[:, 1] <= hi + 2.1)]  # unit boxes: centres within 2 + margin
        assert set(touching[:, 0].astype(int).tolist()) <= got, "a body that can touch rank r's slab is missing from its ghosts"
