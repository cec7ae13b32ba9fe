"""CPU, world_size 2 and 3 over gloo: the N > 1 host path of the sharded broad phase.
physics_amd.sharding.HaloExchange runs unchanged; the per-rank world is a stand-in built on the CPU
oracle (test infrastructure) that implements the two halo entry points by brute force. Property checked:
union over ranks of (local pairs, cross pairs), in global ids, == the single-world pair set."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from physics_amd import FLAG_COLLISIONS, default_config, sharding

SHAPE = (5, 4, 6, 2.03)  # lattice gap 0.03 < 2 * margin: neighbours overlap, across slab faces too


class OracleHaloWorld:
    """halo_pack / halo_pairs of include/physics_hip.h restated on numpy over the oracle's AABBs."""

    def __init__(self, scene):
        from oracle import binding as ob
        self.o = ob.OracleWorld(default_config(flags=FLAG_COLLISIONS, gravity_offset=(0, 0, 0)), trig=ob.TRIG_DET)
        scene.populate(self.o)
        self.gids = None
        self.cross = np.zeros((0, 2), np.uint32)

    def set_global_ids(self, gids):
        self.gids = np.asarray(gids, np.uint32)

    def local_pairs_global(self):
        p = self.o.broadphase()
        return np.sort(self.gids[p], axis=1) if len(p) else np.zeros((0, 2), np.uint32)

    @staticmethod
    def _view(ptr, rows):
        buf = (ctypes.c_float * (rows * 8)).from_address(ptr)
        return np.ctypeslib.as_array(buf).reshape(rows, 8)

    def halo_pack(self, x_lo, x_hi, reach, ptr, cap, wait=True):
        a = self.o.get_aabbs()
        take = (a[:, 0] < np.float32(x_lo) + np.float32(reach)) | (a[:, 3] > np.float32(x_hi) - np.float32(reach))
        idx = np.nonzero(take)[0]
        assert len(idx) <= cap
        out = self._view(ptr, cap)
        out.view(np.uint32)[:] = 0xFFFFFFFF  # phys_halo_pack blanks the whole buffer first
        out[:len(idx), :6] = a[idx]
        out[:len(idx), 6] = self.gids[idx].view(np.float32)
        out[:len(idx), 7] = 0
        return len(idx)

    def halo_pairs(self, ptr, n_remote, skip_first=0, skip_count=0, wait=True):
        rec = self._view(ptr, n_remote)
        rg = rec[:, 6].copy().view(np.uint32)
        rg[skip_first:skip_first + skip_count] = 0xFFFFFFFF  # own block of the gathered buffer
        a = self.o.get_aabbs()
        found = []
        for k in np.nonzero(rg != 0xFFFFFFFF)[0]:
            ov = (rec[k, :3] <= a[:, 3:]).all(1) & (a[:, :3] <= rec[k, 3:6]).all(1) & (self.gids < rg[k])
            found += [(int(j), int(rg[k])) for j in np.nonzero(ov)[0]]
        self.cross = np.array(sorted(found), np.uint32).reshape(-1, 2)
        return len(found)


def _worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        sc, x_lo, x_hi, gids = sharding.rank_scene("c1", rank, world_size, shape=SHAPE)
        w = OracleHaloWorld(sc)
        halo = sharding.HaloExchange(dist, rank, world_size, "cpu", cap=256)
        halo.attach(w, x_lo, x_hi, gids, sc.half_extent, 0.02)
        n_cross = halo.exchange(w)
        assert n_cross == len(w.cross)
        local = w.local_pairs_global()
        cross = np.stack([gids[w.cross[:, 0]], w.cross[:, 1]], 1) if len(w.cross) else np.zeros((0, 2), np.uint32)
        assert (cross[:, 0] < cross[:, 1]).all()  # ownership rule: the smaller global id is the local body
        np.save(os.path.join(out_dir, f"pairs_{rank}.npy"), np.concatenate([local, cross]).astype(np.uint32))
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([halo.last_halo_records, n_cross, halo.reach]))
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
def test_union_of_rank_pairs_equals_single_world(world_size, tmp_path):
    from oracle import binding as ob
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    got = np.concatenate([np.load(tmp_path / f"pairs_{r}.npy") for r in range(world_size)])
    got = np.array(sorted(map(tuple, got.tolist())), np.uint32)
    # single world holding every rank's bodies in global-id order
    parts = [sharding.rank_scene("c1", r, world_size, shape=SHAPE) for r in range(world_size)]
    pos = np.concatenate([p[0].pos for p in parts])
    st = np.concatenate([p[0].shape_type for p in parts])
    he = np.concatenate([p[0].half_extent for p in parts])
    o = ob.OracleWorld(default_config(flags=FLAG_COLLISIONS), trig=ob.TRIG_DET)
    o.set_bodies(pos, shape_type=st, half_extent=he)
    want = o.broadphase()
    assert len(np.unique(got, axis=0)) == len(got), "a pair was emitted twice"
    assert np.array_equal(got, want)
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world_size)]
    assert sum(m[1] for m in metas) > 0, "no cross-rank pair in the test scene"
    assert all(m[0] > 0 for m in metas)


def test_rank_scene_layout():
    a = sharding.rank_scene("c2", 0, 4)
    b = sharding.rank_scene("c2", 1, 4)
    assert a[0].n == b[0].n == 10_000
    assert a[2] == b[1]  # slabs abut
    assert a[3][-1] + 1 == b[3][0]  # global ids are contiguous across ranks
    assert a[0].pos[:, 0].max() < a[2] + 0.06 and b[0].pos[:, 0].min() > b[1] - 0.06
    from physics_amd import scenes
    assert np.array_equal(sharding.rank_scene("c2", 0, 1)[0].pos, scenes.c2().pos)


# ---- strong scaling: ONE scene cut into equal-count slabs (BASELINE config 4) -----------------------------------------
def _strong_scene():
    from physics_amd import scenes
    return scenes.c4(14, 6, 6)  # 504 bodies, spacing 2.2, jitter 0.3: dense AABB overlaps, across the cut planes too


def _strong_worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        sc, x_lo, x_hi, gids, n_full = sharding.strong_rank_scene("c4", rank, world_size, scene=_strong_scene())
        w = OracleHaloWorld(sc)
        halo = sharding.HaloExchange(dist, rank, world_size, "cpu", cap=512)
        halo.attach(w, x_lo, x_hi, gids, sc.half_extent, 0.02)
        halo.exchange(w)
        local = w.local_pairs_global()
        cross = np.stack([gids[w.cross[:, 0]], w.cross[:, 1]], 1) if len(w.cross) else np.zeros((0, 2), np.uint32)
        np.save(os.path.join(out_dir, f"pairs_{rank}.npy"), np.concatenate([local, cross]).astype(np.uint32))
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([sc.n, len(cross), n_full, x_lo, x_hi]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_strong_scaling_cut_of_one_scene(world_size, tmp_path):
    """sharding.strong_rank_scene: histogram of body x -> prefix sums -> equal-count cut planes (phys_slab_*), every rank
    keeps its slab of the ONE scene under the scene's own ids. Slabs are disjoint, cover the scene, hold n / ranks bodies
    each (+- the bodies of one histogram bin), and the union of (local pairs, cross pairs) is the single-world pair set."""
    from oracle import binding as ob
    mp.spawn(_strong_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    full = _strong_scene()
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world_size)]
    assert sum(int(m[0]) for m in metas) == full.n and all(int(m[2]) == full.n for m in metas)
    assert all(abs(int(m[0]) - full.n / world_size) <= 0.08 * full.n for m in metas), [m[0] for m in metas]
    assert all(metas[r][4] == metas[r + 1][3] for r in range(world_size - 1))  # the slabs abut
    got = np.concatenate([np.load(tmp_path / f"pairs_{r}.npy") for r in range(world_size)])
    got = np.array(sorted(map(tuple, got.tolist())), np.uint32)
    o = ob.OracleWorld(default_config(flags=FLAG_COLLISIONS), trig=ob.TRIG_DET)
    full.populate(o)
    want = o.broadphase()
    assert len(np.unique(got, axis=0)) == len(got), "a pair was emitted twice"
    assert np.array_equal(got, want)
    assert sum(m[1] for m in metas) > 0, "no cross-rank pair in the test scene"
