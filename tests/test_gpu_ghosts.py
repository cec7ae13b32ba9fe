"""GPU: sharded worlds with contacts across the cut plane (SURVEY §8 rows E + N4) through the C ABI.

Two worlds on ONE GPU stand for two ranks (RCCL wants one rank per GPU, so the all-gather between them is a device
copy made by the test; the RCCL entry points themselves are driven with a one-rank communicator at the end). Each step:
phys_halo_pack_bodies on both -> "all-gather" -> phys_halo_unpack_ghosts on both -> phys_update on both.
Checked: boundary bodies arrive as ghosts with the right global ids; manifolds against ghosts exist; bodies owned by
different ranks do not pass through each other (they do without the exchange: the control); an impact across the plane
conserves momentum and follows the single-world trajectory (ghosts are dynamic bodies with their owner's mass: both
sides solve the two-body contact); a sharded run repeats bit for bit (ordered compaction, no atomics)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 16_666_667
REC = 96  # PHYS_HALO_BODY_RECORD_BYTES


def _make(pos, vel, gids, x_lo, x_hi, ground, cap, gravity=(0.0, -9.81, 0.0), mass=None):
    import physics_amd
    flags = physics_amd.FLAG_COLLISIONS | (physics_amd.FLAG_GROUND_PLANE if ground else 0)
    cfg = physics_amd.default_config(flags=flags, gravity_offset=(0, 0, 0), gravity_force=gravity, max_ghosts=2 * cap)
    w = physics_amd.World(cfg)
    n = len(pos)
    w.set_bodies(pos, lin_vel=vel, mass=mass, shape_type=np.full(n, physics_amd.SHAPE_BOX, np.uint32), half_extent=np.ones((n, 3), np.float32))
    w.set_global_ids(gids)
    w.set_slab(x_lo, x_hi, 4.0)  # reach: cube diagonal 3.47 + margins
    return w


class TwoRanks:
    """Left world owns x < 0, right world x >= 0; exchange() plays the all-gather with device copies."""

    def __init__(self, pos, vel, ground=True, exchange=True, cap=1024, gravity=(0.0, -9.81, 0.0), neighbours=False, mass=None):
        import torch
        self.torch = torch
        left = pos[:, 0] < 0
        self.idx = [np.nonzero(left)[0], np.nonzero(~left)[0]]
        ms = [None if mass is None else np.asarray(mass, np.float32)[ix] for ix in self.idx]
        self.worlds = [_make(pos[self.idx[0]], vel[self.idx[0]], self.idx[0].astype(np.uint32), -1.0e6, 0.0, ground, cap, gravity, ms[0]),
                       _make(pos[self.idx[1]], vel[self.idx[1]], self.idx[1].astype(np.uint32), 0.0, 1.0e6, ground, cap, gravity, ms[1])]
        self.cap = cap
        self.send = [torch.empty(cap * REC, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.do_exchange = exchange
        self.neighbours = neighbours  # the blocks a neighbour exchange (phys_comm_set_neighbours) moves: one per face
        self.n_total = len(pos)

    def step(self, k=1):
        torch = self.torch
        for _ in range(k):
            if self.do_exchange and self.neighbours:
                # rank 0 sends its HIGH-face block to rank 1, rank 1 its LOW-face block to rank 0; a rank scans two
                # received blocks {from r - 1, from r + 1}, the missing one empty (0xFF), and skips nothing
                self.worlds[0].halo_pack_bodies_face(self.send[0].data_ptr(), self.cap, +1)
                self.worlds[1].halo_pack_bodies_face(self.send[1].data_ptr(), self.cap, -1)
                for w in self.worlds:
                    w.sync()
                empty = torch.full_like(self.send[0], 0xFF)
                recv = [torch.cat([empty, self.send[1]]), torch.cat([self.send[0], empty])]
                torch.cuda.synchronize()
                for w, blocks in zip(self.worlds, recv):
                    w.halo_unpack_ghosts(blocks.data_ptr(), 2 * self.cap, 0, 0)
                for w in self.worlds:
                    w.sync()
            elif self.do_exchange:
                for w, buf in zip(self.worlds, self.send):
                    w.halo_pack_bodies(buf.data_ptr(), self.cap)
                for w in self.worlds:
                    w.sync()
                gathered = torch.cat(self.send)  # rank-major blocks, as ncclAllGather lays them out
                torch.cuda.synchronize()
                for r, w in enumerate(self.worlds):
                    w.halo_unpack_ghosts(gathered.data_ptr(), 2 * self.cap, r * self.cap, self.cap)
                for w in self.worlds:
                    w.sync()  # `gathered` may be freed after this
            for w in self.worlds:
                w.update(DT)
        for w in self.worlds:
            w.sync()

    def positions(self):
        out = np.zeros((self.n_total, 3), np.float32)
        for ids, w in zip(self.idx, self.worlds):
            out[ids] = w.get_transforms()[0]
        return out

    def state(self):
        return [a for w in self.worlds for a in w.get_transforms() + w.get_velocities()]

    def close(self):
        for w in self.worlds:
            w.close()


def test_head_on_across_the_plane_stops_with_ghosts_and_passes_through_without():
    pos = np.array([[-3.0, 5.0, 0.0], [3.0, 5.0, 0.0]], np.float32)
    vel = np.array([[4.0, 0.0, 0.0], [-4.0, 0.0, 0.0]], np.float32)
    for exchange in (True, False):
        t = TwoRanks(pos, vel, ground=False, exchange=exchange, gravity=(0.0, 0.0, 0.0))
        t.step(1)
        if exchange:
            for r, w in enumerate(t.worlds):
                st = w.get_stats()
                assert st.n_ghosts == 1 and st.n_bodies == 1
                gid = w.get_global_ids()
                assert gid[0] == r and gid[1] == 1 - r, gid[:3]  # own body, then the neighbour's as the first ghost
        t.step(119)
        x = t.positions()[:, 0]
        if exchange:
            assert x[0] < x[1] - 1.9, f"cubes of different ranks ended up inside each other: {x}"
            lin = [w.get_velocities()[0][0] for w in t.worlds]
            # equal masses, no restitution: the two-body impulse stops both (a kinematic ghost - round 2 - sent each back
            # with the other's velocity); what is left is the push-out of the contact slop
            assert abs(lin[0][0]) < 0.2 and abs(lin[1][0]) < 0.2, lin
        else:
            assert x[0] > x[1], "control: without the exchange the two ranks do not see each other"
        t.close()


@pytest.mark.parametrize("masses", [(1.0, 1.0), (1.0, 3.0), (5.0, 0.5)])
def test_impact_across_the_plane_conserves_momentum_and_equals_the_single_world(masses):
    """SURVEY N4 / VERDICT r2 item 7b: a contact across the cut plane carries a SHARED impulse. Two cubes of different
    mass, owned by different ranks, meet head-on (with a glancing offset, so friction and torque take part). Every rank
    solves the contact against a dynamic ghost of the other's cube, from the same state, and keeps its own cube's half:
    total linear momentum after the impact equals the momentum before to 1e-4 (relative to the momentum in play), and
    both cubes follow the trajectory of ONE world holding both (positions to 1e-3, velocities to 1e-3: the two sides
    evaluate the same manifold with the roles of A and B exchanged, nothing more)."""
    import physics_amd
    pos = np.array([[-2.5, 5.0, 0.0], [2.5, 5.3, 0.2]], np.float32)
    vel = np.array([[3.0, 0.0, 0.0], [-2.0, 0.0, 0.0]], np.float32)
    mass = np.array(masses, np.float32)
    t = TwoRanks(pos, vel, ground=False, gravity=(0.0, 0.0, 0.0), mass=mass)
    cfg = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS, gravity_offset=(0, 0, 0), gravity_force=(0, 0, 0))
    one = physics_amd.World(cfg)
    one.set_bodies(pos, lin_vel=vel, mass=mass, shape_type=np.full(2, physics_amd.SHAPE_BOX, np.uint32), half_extent=np.ones((2, 3), np.float32))
    p_before = (mass[:, None] * vel).sum(0)
    touched = False
    for step in range(90):
        t.step(1)
        one.update(DT)
        one.sync()
        touched = touched or any(w.get_stats().n_manifolds > 0 for w in t.worlds)
        lin = np.stack([w.get_velocities()[0][0] for w in t.worlds])
        p_now = (mass[:, None] * lin).sum(0)
        scale = float((mass[:, None] * np.abs(vel)).sum())
        assert np.abs(p_now - p_before).max() <= 1.0e-4 * scale, f"step {step}: momentum {p_now} vs {p_before}"
        ref_pos, ref_lin = one.get_transforms()[0], one.get_velocities()[0]
        assert np.abs(t.positions() - ref_pos).max() < 1.0e-3, f"step {step}"
        assert np.abs(lin - ref_lin).max() < 1.0e-3, f"step {step}: {lin} vs {ref_lin}"
    assert touched, "the cubes never met"
    assert abs(lin[0][0] - lin[1][0]) < 0.3, "no restitution: after the impact they move together"
    one.close()
    t.close()


def _two_piles():
    from physics_amd import scenes
    pos = scenes.lattice(8, 3, 4, 2.5, 2.0, 0.05)
    vel = np.zeros_like(pos)
    vel[:, 0] = np.where(pos[:, 0] < 0, 1.5, -1.5)  # the two halves drift into each other while they fall
    return pos, vel


def _min_cross_distance(t):
    p = t.positions().astype(np.float64)
    a, b = p[t.idx[0]], p[t.idx[1]]
    d = np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2)
    return float(d.min())


def test_piles_meeting_at_the_plane_do_not_interpenetrate_and_runs_repeat_bit_for_bit():
    pos, vel = _two_piles()
    runs = []
    for _ in range(2):
        t = TwoRanks(pos, vel)
        ghost_contacts = 0
        for _chunk in range(24):  # the halves meet, push each other back and settle: contacts across the plane come and go
            t.step(10)
            st = [w.get_stats() for w in t.worlds]
            assert all(s.overflow == 0 for s in st)
            ids = t.worlds[0].get_manifolds()[0]
            n_owned = st[0].n_bodies
            ghost_contacts += int(((ids[:, 1] >= n_owned) & (ids[:, 1] != 0xFFFFFFFF)).sum())
        assert sum(s.n_ghosts for s in st) > 0, "nobody near the plane?"
        assert ghost_contacts > 0, "no manifold against a ghost on rank 0 at any of the sampled steps"
        # two unit cubes touch at a centre distance of 2 (face to face) or more: 1.8 allows slop + solver softness
        assert _min_cross_distance(t) > 1.8
        runs.append(t.state())
        t.close()
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b), "the sharded run is not reproducible"
    control = TwoRanks(pos, vel, exchange=False)
    control.step(240)
    assert _min_cross_distance(control) < 1.5, "control: the halves should have slid into each other"
    control.close()


def test_neighbour_blocks_give_the_same_run_as_the_all_gather():
    """phys_comm_set_neighbours sends one block per slab face to the rank beyond it instead of gathering every rank's
    block everywhere. With two ranks both move the same bodies, so the sharded run must repeat the all-gather run bit
    for bit (the ghosts arrive in the same order: ordered compaction of one block)."""
    pos, vel = _two_piles()
    runs = []
    for neighbours in (False, True):
        t = TwoRanks(pos, vel, neighbours=neighbours)
        t.step(120)
        assert sum(w.get_stats().n_ghosts for w in t.worlds) > 0
        runs.append(t.state())
        t.close()
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b), "per-face blocks changed the sharded run"


def test_cluster_solver_with_ghosts_equals_the_per_colour_kernels():
    """Two 16 x 130 x 16 towers side by side, one per rank, in contact across the plane: dense enough for the cluster
    solver (cluster.hip), which gives ghost bodies no home in any cluster (for every row they are 'another cluster's
    body'). The same sharded run with the per-colour kernels forced (PHYS_FLAG_SOLVER_PER_COLOR) must give the same
    bits, and the cluster solver must really have run."""
    import physics_amd
    import torch
    from physics_amd import scenes
    sc = scenes.c5(16, 130, 16)
    width = float(sc.pos[:, 0].max() - sc.pos[:, 0].min()) + 2.0  # lattice spacing of the tower: 2.0
    results = []
    for flags_extra in (0, physics_amd.FLAG_SOLVER_PER_COLOR):
        worlds, sends = [], []
        cap = 8192
        for r in range(2):
            cfg = sc.config(flags=sc.flags | (flags_extra or physics_amd.FLAG_SOLVER_CLUSTER), max_ghosts=2 * cap)
            w = physics_amd.World(cfg)
            pos = sc.pos.copy()
            pos[:, 0] += np.float32(r * width)
            w.set_bodies(pos, shape_type=sc.shape_type, half_extent=sc.half_extent)
            w.set_global_ids((np.arange(sc.n) + r * sc.n).astype(np.uint32))
            lo = float(sc.pos[:, 0].min()) - 1.0 + r * width
            w.set_slab(lo if r else -1.0e6, lo + width if r == 0 else 1.0e6, 4.0)
            worlds.append(w)
            sends.append(torch.empty(cap * REC, dtype=torch.uint8, device="cuda"))
        empty = torch.full_like(sends[0], 0xFF)
        ran_cluster = False
        for step in range(14):
            worlds[0].halo_pack_bodies_face(sends[0].data_ptr(), cap, +1)
            worlds[1].halo_pack_bodies_face(sends[1].data_ptr(), cap, -1)
            for w in worlds:
                w.sync()
            recv = [torch.cat([empty, sends[1]]), torch.cat([sends[0], empty])]
            torch.cuda.synchronize()
            for w, blocks in zip(worlds, recv):
                w.halo_unpack_ghosts(blocks.data_ptr(), 2 * cap, 0, 0)
            for w in worlds:
                w.sync()
            if step == 10:
                worlds[0].profile_enable(True)
            for w in worlds:
                w.update(DT)
        for w in worlds:
            w.sync()
        prof, _ = worlds[0].profile_get()
        ran_cluster = "solve_cluster" in prof
        st = [w.get_stats() for w in worlds]
        assert all(s.overflow == 0 for s in st) and all(s.n_ghosts > 0 for s in st)
        ids = worlds[0].get_manifolds()[0]
        assert ((ids[:, 1] >= st[0].n_bodies) & (ids[:, 1] != 0xFFFFFFFF)).any(), "no contact against a ghost"
        assert ran_cluster == (flags_extra == 0), f"solver path: {sorted(prof)}"
        results.append([a for w in worlds for a in w.get_transforms() + w.get_velocities()])
        for w in worlds:
            w.close()
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b), "cluster solver with ghosts differs from the per-colour kernels"


def test_ghost_capacity_overflow_is_reported():
    import physics_amd
    pos, vel = _two_piles()
    t = TwoRanks(pos, vel, cap=1024)
    # shrink the ghost room of rank 0 below its share of boundary bodies
    t.worlds[0].close()
    cfg_small = physics_amd.default_config(flags=physics_amd.FLAG_COLLISIONS | physics_amd.FLAG_GROUND_PLANE,
                                           gravity_offset=(0, 0, 0), max_ghosts=2)
    w = physics_amd.World(cfg_small)
    ids = t.idx[0]
    w.set_bodies(pos[ids], lin_vel=vel[ids], shape_type=np.full(len(ids), physics_amd.SHAPE_BOX, np.uint32),
                 half_extent=np.ones((len(ids), 3), np.float32))
    w.set_global_ids(ids.astype(np.uint32))
    w.set_slab(-1.0e6, 0.0, 4.0)
    t.worlds[0] = w
    with pytest.raises(physics_amd.PhysError) as e:
        t.step(1)  # the unpack of rank 0 finds more boundary bodies of rank 1 than it has ghost slots
    assert e.value.code == -5
    t.close()


def test_rccl_exchange_behind_the_c_abi_with_one_rank():
    """phys_comm_* / phys_halo_exchange: librccl is loaded by the library, ncclCommInitRank + ncclAllGather run on the
    world's own stream. With one rank the gathered buffer is the rank's own block, which unpack skips: no ghosts, but
    the boundary bodies were packed and the collective completed."""
    import physics_amd
    pos, vel = _two_piles()
    w = _make(pos, vel, np.arange(len(pos), dtype=np.uint32), -2.0, 2.0, True, 256)
    for neighbours in (False, True):  # all-gather / grouped ncclSend + ncclRecv per neighbour (none with one rank)
        comm = physics_amd.Comm(w, physics_amd.Comm.unique_id(), 0, 1, 256, neighbours=neighbours)
        for _ in range(40):  # long enough for the lowest layer to land
            w.halo_exchange(comm)
            w.update(DT)
        w.sync()
        st = w.get_stats()
        assert st.n_ghosts == 0 and st.overflow == 0
        comm.close()
    assert st.n_manifolds > 0
    w.close()


def test_rccl_exchange_of_aabb_records_with_one_rank():
    """The broad-phase-only form of the exchange (no ghost slots: 32-byte AABB records, phys_halo_pack / phys_halo_pairs
    around the collective) through phys_halo_exchange, in both topologies, with a one-rank communicator: the collective
    completes on the world's stream, nothing comes back (a rank has no cross pairs with itself)."""
    import physics_amd
    from physics_amd import scenes
    sc = scenes.c4(20, 20, 20)
    w = physics_amd.World(sc.config())
    sc.populate(w)
    w.set_global_ids(np.arange(sc.n, dtype=np.uint32))
    w.set_slab(float(sc.pos[:, 0].min()) - 1.0, float(sc.pos[:, 0].max()) + 1.0, 3.0)  # the outermost layers are boundary bodies
    for neighbours in (False, True):
        comm = physics_amd.Comm(w, physics_amd.Comm.unique_id(), 0, 1, 8192, neighbours=neighbours)
        for _ in range(3):
            w.update(DT)
            w.halo_exchange(comm)
        w.sync()
        st = w.get_stats()
        assert st.overflow == 0 and st.n_cross_pairs == 0 and st.n_pairs > 0
        comm.close()
    w.close()
