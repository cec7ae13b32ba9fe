"""GPU: the two halo entry points of the sharded broad phase through the C ABI.
Two worlds on one GPU play two ranks; the "all-gather" is a plain exchange of the packed record
buffers (torch tensors: device memory plumbing). Property: union of local + cross pairs in global ids
== the single-world pair set of the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SHAPE = (6, 5, 7, 2.03)


def _setup(world_size=2):
    import physics_amd
    import torch
    from physics_amd import sharding
    parts = [sharding.rank_scene("c1", r, world_size, shape=SHAPE) for r in range(world_size)]
    worlds = []
    for sc, x_lo, x_hi, gids in parts:
        w = physics_amd.World(physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_offset=(0, 0, 0)))
        sc.populate(w)
        w.set_global_ids(gids)
        worlds.append(w)
    return parts, worlds, torch


def _single_world_pairs(parts):
    import physics_amd
    from oracle import binding as ob
    o = ob.OracleWorld(physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS), trig=ob.TRIG_DET)
    o.set_bodies(np.concatenate([p[0].pos for p in parts]), shape_type=np.concatenate([p[0].shape_type for p in parts]),
                 half_extent=np.concatenate([p[0].half_extent for p in parts]))
    return o.broadphase()


def test_pack_exchange_pairs_union_matches_single_world():
    from physics_amd import sharding
    parts, worlds, torch = _setup(2)
    cap = 512
    reach = sharding.static_reach(parts[0][0].half_extent, 0.02)
    local, bufs, counts = [], [], []
    for (sc, x_lo, x_hi, gids), w in zip(parts, worlds):
        p = w.broadphase()  # builds AABBs + grid on the device
        local.append(np.sort(gids[p], axis=1))
        buf = torch.zeros((cap, 8), dtype=torch.int32, device="cuda")  # the pack call blanks it itself
        torch.cuda.synchronize()
        counts.append(w.halo_pack(x_lo, x_hi, reach, buf.data_ptr(), cap))
        bufs.append(buf)
    assert all(0 < c < cap for c in counts)
    # records: gid column valid exactly for the first `count` rows
    for buf, c in zip(bufs, counts):
        g = buf[:, 6].cpu().numpy().view(np.uint32)
        assert (g[:c] != 0xFFFFFFFF).all() and (g[c:] == 0xFFFFFFFF).all()
    cross = []
    for r, w in enumerate(worlds):
        n = w.halo_pairs(bufs[1 - r].data_ptr(), cap)
        cp = w.get_cross_pairs()
        assert len(cp) == n
        gids = parts[r][3]
        cross.append(np.stack([gids[cp[:, 0]], cp[:, 1]], 1) if n else np.zeros((0, 2), np.uint32))
        assert (cross[-1][:, 0] < cross[-1][:, 1]).all()
    got = np.concatenate(local + cross)
    got = np.array(sorted(map(tuple, got.tolist())), np.uint32)
    assert sum(len(c) for c in cross) > 0
    assert np.array_equal(got, _single_world_pairs(parts))


def test_halo_after_update_uses_the_step_grid():
    import physics_amd
    import torch
    from physics_amd import scenes, sharding
    sc, x_lo, x_hi, gids = sharding.rank_scene("c1", 0, 2, shape=(4, 4, 4, 2.03))
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.set_global_ids(gids)
    buf = torch.full((256, 8), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(physics_amd.PhysError):
        w.halo_pack(x_lo, x_hi, 0.0, buf.data_ptr(), 256)  # no grid yet
    w.update(scenes.DT_NANOS)
    n = w.halo_pack(x_lo, x_hi, 0.0, buf.data_ptr(), 256)  # reach <= 0: this rank's own cell size
    assert 0 < n <= 64
    # fed its OWN records back (HaloExchange never does: it blanks the own slot), the kernel must find exactly
    # the local pairs whose higher-id body is in the halo set
    n_back = w.halo_pairs(buf.data_ptr(), 256)
    halo_ids = set(buf[:n, 6].cpu().numpy().view(np.uint32).tolist())
    local = w.broadphase()
    assert n_back == sum(1 for a, b in local.tolist() if int(gids[b]) in halo_ids)


def test_async_exchange_on_the_worlds_stream_with_rccl_world_size_1():
    """The N > 1 production path of physics_amd.sharding.HaloExchange (pack kernel -> RCCL all-gather ->
    cross-pair kernel, all enqueued on the world's own stream through torch.cuda.ExternalStream, no host
    synchronisation) exercised with a one-rank RCCL group: the gathered buffer is this rank's own block, which
    phys_halo_pairs must skip."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    import physics_amd
    from physics_amd import scenes, sharding
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc, x_lo, x_hi, gids = sharding.rank_scene("c1", 0, 1, shape=(6, 5, 7, 2.03))
        w = physics_amd.World(sc.config())
        sc.populate(w)
        halo = sharding.HaloExchange(dist, 0, 1, "cuda:0", cap=512)
        halo.attach(w, x_lo, x_hi, gids, sc.half_extent, 0.02)
        for _ in range(5):
            w.update(scenes.DT_NANOS)
            assert halo.exchange(w) is None  # asynchronous form
        w.sync()
        st = w.get_stats()
        assert 0 < st.n_halo_records <= 512
        assert st.n_cross_pairs == 0  # the only block in the gathered buffer is our own
        # the records really went through the collective on the world's stream
        got = halo.recv[0].cpu().numpy().view(np.uint32)[:, 6]
        assert (got != 0xFFFFFFFF).sum() == st.n_halo_records
        w.close()
    finally:
        dist.destroy_process_group()
