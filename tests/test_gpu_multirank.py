"""GPU test of the DEVICE side of the multi-rank path on ONE GPU: R DeviceEngine contexts (2, 4, 8 sub-boxes) live in
one process on cuda:0; the transport is replaced by direct device-buffer hand-over (export_pack -> import), so the
leaving/halo packing kernels, receiver-frame shifts, import + re-bin and the inner/boundary force split are checked
against the single-domain result.  (The torch.distributed protocol itself is covered under gloo in
tests/test_decomp_cpu.py; real RCCL needs one GPU per rank and is exercised by bench.py --gpus N.)"""
import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu

decomp = load_pkg("decomp")
engine_mod = load_pkg("engine")
inp = load_pkg("inp")


class InProcessCluster:
    def __init__(self, world, comps, rc, L, ids, r, v, grid=None, kernel=0, cid=None, q=None, D=None, skin=None):
        import torch
        self.torch = torch
        self.world = world
        self.overlap_order = True  # inner pass before the halo phase (False: the older halo-first order, still valid)
        self.dcs = [decomp.CartesianDecomposition(world, k, L, grid) for k in range(world)]
        self.eng = []
        for dc in self.dcs:
            lo, hi = dc.bounding_box()
            e = engine_mod.DeviceEngine(0)
            e.set_components(comps, rc)
            e.set_option("force_kernel", kernel)
            if skin:
                e.set_verlet(skin, force=True)
            e.set_domain(L, lo, hi, dc.rank, dc.neighbor_table())
            m = np.all((r >= lo) & (r < hi), axis=1)
            e.upload(ids[m], np.zeros(m.sum(), np.int32) if cid is None else cid[m], r[m], v[m],
                     None if q is None else q[m], None if D is None else D[m])
            self.eng.append(e)

    def exchange(self, kind):
        torch = self.torch
        w = decomp.RECORD_DOUBLES[kind]
        counts = [e.export_counts(kind) for e in self.eng]
        bufs = []
        plan = []  # (source context, destination, directions, records): a position refresh repeats exactly these messages
        for k, e in enumerate(self.eng):
            nbr = self.dcs[k].neighbor_table()
            if k % 2 == 0:  # one message per direction (ls1hip_export_pack)
                for d in range(27):
                    c = int(counts[k][d])
                    if c and nbr[d] != k:
                        assert nbr[d] >= 0
                        t = torch.empty(c * w, dtype=torch.float64, device="cuda:0")
                        e.export_pack(kind, d, t.data_ptr(), c)
                        bufs.append((int(nbr[d]), t, c))
                        plan.append((k, int(nbr[d]), [d], c))
            else:  # one message per peer (ls1hip_export_pack_dirs), as decomp.HaloExchanger sends them
                for peer in sorted({int(p) for p in nbr if p >= 0 and p != k}):
                    dirs = [d for d in range(27) if d != 13 and nbr[d] == peer and counts[k][d]]
                    c = int(sum(int(counts[k][d]) for d in dirs))
                    if c:
                        t = torch.empty(c * w, dtype=torch.float64, device="cuda:0")
                        e.export_pack_dirs(kind, dirs, t.data_ptr(), c)
                        bufs.append((peer, t, c))
                        plan.append((k, peer, dirs, c))
        torch.cuda.synchronize()
        for dest, t, c in bufs:
            self.eng[dest].import_records(kind, t.data_ptr(), c)
        for e in self.eng:
            e.import_done(kind)
        if kind == decomp.HALO:
            self.halo_plan = plan

    def refresh(self):
        """list mode: current positions for the halo copies of the last build, through the messages of that build"""
        torch = self.torch
        for e in self.eng:
            e.halo_refresh()
        bufs = []
        for k, dest, dirs, c in self.halo_plan:
            t = torch.empty(c * 3, dtype=torch.float64, device="cuda:0")
            if len(dirs) == 1:
                self.eng[k].export_pack(decomp.REFRESH, dirs[0], t.data_ptr(), c)
            else:
                self.eng[k].export_pack_dirs(decomp.REFRESH, dirs, t.data_ptr(), c)
            bufs.append((dest, t, c))
        torch.cuda.synchronize()
        for dest, t, c in bufs:
            self.eng[dest].import_records(decomp.REFRESH, t.data_ptr(), c)
        for e in self.eng:
            e.import_done(decomp.REFRESH)

    def run_lists(self, dt, nsteps):
        """decomp.DistributedSimulation.run_lists with the transport replaced by device-buffer hand-over"""
        advanced, builds, tot, kin = False, 0, None, None
        for s in range(nsteps):
            last = s == nsteps - 1
            if advanced:
                rebuild = any([e.verlet_poll() for e in self.eng])  # every context is polled (no short circuit)
            else:
                for e in self.eng:
                    e.kick_drift(dt)
                rebuild = True
            advanced = not last
            fdt = dt if advanced else 0.0
            if rebuild:
                builds += 1
                for e in self.eng:
                    e.rebin()
                self.exchange(decomp.LEAVING)
                for e in self.eng:
                    e.halo()
                self.exchange(decomp.HALO)
                for e in self.eng:
                    e.verlet_build()
                res = [e.forces_list(0, fdt, want_macro=last) for e in self.eng]
            else:
                for e in self.eng:
                    e.forces_list(1, fdt)
                self.refresh()
                res = [e.forces_list(2, fdt, want_macro=last) for e in self.eng]
            if last:
                tot = np.sum(np.array(res), axis=0)
                kin = np.sum(np.array([e.kick(0.5 * dt)[:2] for e in self.eng]), axis=0)
        return tot, kin, builds

    def forces(self, split=True):
        for e in self.eng:
            e.rebin()
        self.exchange(decomp.LEAVING)
        if split and self.overlap_order:
            # the distributed loop's order: inner pass first, then the halo phase on the engines' second streams
            for e in self.eng:
                e.forces(1, want_macro=False)
            for e in self.eng:
                e.halo()
        else:
            for e in self.eng:
                e.halo()
            if split:
                for e in self.eng:
                    e.forces(1, want_macro=False)
        self.exchange(decomp.HALO)
        tot = np.zeros(2)
        for e in self.eng:
            tot += np.array(e.forces(2 if split else 0))
        return tot

    def fused_forces(self, dt):
        """re-bin, exchange, inner pass, halo exchange, boundary pass with the integration fused into the force passes"""
        for e in self.eng:
            e.rebin()
        self.exchange(decomp.LEAVING)
        for e in self.eng:
            e.forces_kick_drift(1, dt)
        for e in self.eng:
            e.halo()
        self.exchange(decomp.HALO)
        for e in self.eng:
            e.forces_kick_drift(2, dt)

    def step(self, dt):
        for e in self.eng:
            e.kick_drift(dt)
        tot = self.forces()
        kin = np.zeros(2)
        for e in self.eng:
            k = e.kick(0.5 * dt)
            kin += np.array(k[:2])
        return tot, kin

    def gather(self):
        st = [e.download_state() for e in self.eng]
        fr = [e.download_forces() for e in self.eng]
        ids = np.concatenate([s["ids"] for s in st])
        o = np.argsort(ids)
        return dict(ids=ids[o], r=np.concatenate([s["r"] for s in st])[o], v=np.concatenate([s["v"] for s in st])[o],
                    q=np.concatenate([s["q"] for s in st])[o], D=np.concatenate([s["D"] for s in st])[o],
                    F=np.concatenate([f["F"] for f in fr])[o], M=np.concatenate([f["M"] for f in fr])[o])


def _liquid(n, seed=11, rho=0.785302672):
    rng = np.random.default_rng(seed)
    N = 2 * n ** 3
    L = (N / rho) ** (1 / 3)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = (np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.1 * (rng.random((N, 3)) - 0.5)) % L
    v = rng.normal(0, 1.0, (N, 3)) * 3.0  # hot: many molecules cross sub-box faces within a few steps
    v -= v.mean(0)
    return np.array([L] * 3), r, v


@pytest.mark.parametrize("world,grid", [(2, None), (4, None), (8, None), (2, (1, 1, 2))])
def test_subboxes_equal_single_domain(world, grid):
    L, r, v = _liquid(16)  # 8192 atoms, L = 21.85: 2x2x2 sub-boxes of 10.9 = 4 cells each
    rc, dt, nsteps = 2.5, 0.004, 5
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    single = InProcessCluster(1, comps, rc, L, ids, r, v)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v, grid)
    multi.overlap_order = world != 4  # world 4 keeps the halo-first call order covered
    t1 = single.forces(split=False)
    tm = multi.forces()
    a, b = single.gather(), multi.gather()
    assert np.array_equal(a["ids"], b["ids"])
    Fmax = np.max(np.abs(a["F"]))
    assert np.max(np.abs(a["F"] - b["F"])) < 1e-12 * Fmax
    assert np.allclose(t1, tm, rtol=1e-12)
    for _ in range(nsteps):
        t1, k1 = single.step(dt)
        tm, km = multi.step(dt)
    a, b = single.gather(), multi.gather()
    assert np.array_equal(a["ids"], b["ids"])  # nobody lost or duplicated while migrating
    dr = a["r"] - b["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-11
    assert np.max(np.abs(a["v"] - b["v"])) < 1e-11 * np.max(np.abs(a["v"]))
    assert np.allclose(t1, tm, rtol=1e-11)
    assert np.allclose(k1, km, rtol=1e-12)
    moved = sum(e.count()[0] for e in multi.eng)
    assert moved == len(ids)


@pytest.mark.parametrize("world", [2, 8])
def test_fused_integration_across_subboxes(world):
    """Reduced-memory mode on a decomposed domain: sub-boxes running fused inner/boundary passes with migration and halo
    exchange in between == the single domain running the ordinary loop."""
    L, r, v = _liquid(16)
    rc, dt, nsteps = 2.5, 0.004, 5
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    single = InProcessCluster(1, comps, rc, L, ids, r, v)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v)
    single.forces(split=False)
    multi.forces()
    for _ in range(nsteps):
        single.step(dt)
    for e in multi.eng:
        e.kick_drift(dt)
    for s in range(nsteps):
        if s + 1 < nsteps:
            multi.fused_forces(dt)
        else:
            multi.forces()
            for e in multi.eng:
                e.kick(0.5 * dt)
    a, b = single.gather(), multi.gather()
    assert np.array_equal(a["ids"], b["ids"])
    dr = a["r"] - b["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-11
    assert np.max(np.abs(a["v"] - b["v"])) < 1e-11 * np.max(np.abs(a["v"]))
    assert np.max(np.abs(a["F"] - b["F"])) < 1e-10 * np.max(np.abs(a["F"]))


def test_loopback_transport_equals_local_images():
    """The distributed step loop on ONE rank with its periodic images routed through the real transport (RCCL
    send/recv to the own rank, all 26 directions exported, packed, transferred, imported on the second stream while the
    inner-cell pass runs) == the same loop with local images == the in-engine loop: positions, velocities bitwise,
    U_pot to rounding.  Covers the stream discipline of the overlapped exchange on a single GPU."""
    import os
    import torch
    import torch.distributed as dist
    import socket
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        L, r, v = _liquid(24)  # 27 648 atoms, 12 cells per dimension
        rc, dt, nsteps = 2.5, 0.004, 6
        ids = np.arange(1, len(r) + 1, dtype=np.uint64)
        comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
        out = {}
        for mode in ("engine", "local", "loopback"):
            e = engine_mod.DeviceEngine(0)
            e.set_components(comps, rc)
            if mode == "engine":
                e.set_domain(L)
                e.upload(ids, np.zeros(len(ids), np.int32), r, v)
                e.rebin(); e.halo(); e.forces(0)
                res = e.run(dt, nsteps)
            else:
                dc = decomp.CartesianDecomposition(1, 0, L, loopback=(mode == "loopback"))
                lo, hi = dc.bounding_box()
                e.set_domain(L, lo, hi, 0, dc.neighbor_table())
                e.upload(ids, np.zeros(len(ids), np.int32), r, v)
                sim = decomp.DistributedSimulation(dc, e, dist, torch.device("cuda", 0))
                sim.n_global = len(ids)
                sim.initial_forces()
                res = sim.run(dt, nsteps)
            st = e.download_state()
            o = np.argsort(st["ids"])
            out[mode] = (st["r"][o], st["v"][o], res["upot"], res["summv2"])
            e.close()
        for mode in ("local", "loopback"):
            assert np.array_equal(out["engine"][0], out[mode][0]), mode
            assert np.array_equal(out["engine"][1], out[mode][1]), mode
            assert abs(out["engine"][2] - out[mode][2]) <= 1e-12 * abs(out["engine"][2]), mode
            assert abs(out["engine"][3] - out[mode][3]) <= 1e-12 * abs(out["engine"][3]), mode
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_multisite_subboxes_equal_single_domain(world):
    """BASELINE configs[4] flavour on a decomposed domain: the five-component LJ + charge + dipole + quadrupole set
    (orientations and angular momenta travel in the leaving records, orientations in the halo records): sub-boxes ==
    single domain for forces, torques and a short rotational trajectory with migration."""
    from golden_io import input_path
    ps0 = inp.read_inp(input_path("VectorizationMultiComponentMultiPotentials.inp"))
    comps = ps0.components
    rng = np.random.default_rng(17)
    n = 10
    N = 2 * n ** 3
    rho = 250.0 / 134.266123 ** 3
    Lx = (N / rho) ** (1.0 / 3.0)
    a = Lx / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = (np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.2 * a * rng.uniform(-0.5, 0.5, (N, 3))) % Lx
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
    cid = (np.arange(N) % 5).astype(np.int32)
    v = rng.normal(0, 1.0, (N, 3))  # fast enough that some molecules cross sub-box faces within a few steps
    D = rng.normal(0, 1e-3, (N, 3))
    L = np.array([Lx] * 3)
    rc, dt, nsteps = 35.0, 0.5, 4
    ids = np.arange(1, N + 1, dtype=np.uint64)
    single = InProcessCluster(1, comps, rc, L, ids, r, v, cid=cid, q=q, D=D)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v, cid=cid, q=q, D=D)
    t1 = single.forces(split=False)
    tm = multi.forces()
    a1, b1 = single.gather(), multi.gather()
    assert np.array_equal(a1["ids"], b1["ids"])
    for k in ("F", "M"):
        assert np.max(np.abs(a1[k] - b1[k])) < 1e-11 * np.max(np.abs(a1[k])), k
    assert np.allclose(t1, tm, rtol=1e-11)
    # trajectory: components 0 and 1 of the fixture are massless multipoles (static force tests only), so the moving
    # system uses components 2..4 (LJ / charge / dipole / quadrupole sites with mass and moments of inertia)
    cid = (2 + np.arange(N) % 3).astype(np.int32)
    single = InProcessCluster(1, comps, rc, L, ids, r, v, cid=cid, q=q, D=D)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v, cid=cid, q=q, D=D)
    single.forces(split=False)
    multi.forces()
    for _ in range(nsteps):
        single.step(dt)
        multi.step(dt)
    a1, b1 = single.gather(), multi.gather()
    assert np.array_equal(a1["ids"], b1["ids"])
    moved = np.max(np.abs(a1["r"] - r))
    assert moved > 1.0  # the molecules really travelled
    dr = a1["r"] - b1["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9
    for k in ("v", "q", "D"):
        assert np.max(np.abs(a1[k] - b1[k])) < 1e-9 * max(np.max(np.abs(a1[k])), 1e-300), k


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_decompositions_equal_single_domain(seed):
    """Randomised: non-cubic boxes, rank grids (2,1,1) ... (2,2,2) in any orientation, both call orders, forces and a few
    hot steps (migration across faces, edges and corners) against the single domain."""
    rng = np.random.default_rng(3000 + seed)
    rc = 2.5
    grid = tuple(int(x) for x in rng.permutation([(2, 1, 1), (2, 2, 1), (2, 2, 2), (4, 1, 1), (4, 2, 1)][seed % 5]))
    world = int(np.prod(grid))
    ncell = np.array([int(rng.integers(3, 6)) * g for g in grid])  # >= 3 cells per sub-box and dimension
    L = ncell * rc * rng.uniform(1.01, 1.12, 3)
    rho = 0.6
    N = int(rho * np.prod(L))
    m = np.ceil((N / np.prod(L)) ** (1 / 3) * L).astype(int)
    g = np.stack(np.meshgrid(*[np.arange(k) for k in m], indexing="ij"), -1).reshape(-1, 3)
    sel = rng.permutation(len(g))[:N]
    N = len(sel)
    a = L / m
    r = ((g[sel] + 0.5) * a + rng.uniform(-0.2, 0.2, (N, 3)) * a) % L
    v = rng.normal(0, 3.0, (N, 3)); v -= v.mean(0)
    ids = np.arange(1, N + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    single = InProcessCluster(1, comps, rc, L, ids, r, v)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v, grid)
    multi.overlap_order = bool(seed % 2)
    t1 = single.forces(split=False)
    tm = multi.forces()
    a1, b1 = single.gather(), multi.gather()
    assert np.array_equal(a1["ids"], b1["ids"])
    assert np.max(np.abs(a1["F"] - b1["F"])) < 1e-12 * np.max(np.abs(a1["F"])), (grid, ncell.tolist())
    assert np.allclose(t1, tm, rtol=1e-12)
    for _ in range(4):
        single.step(0.004)
        multi.step(0.004)
    a1, b1 = single.gather(), multi.gather()
    assert np.array_equal(a1["ids"], b1["ids"])
    dr = a1["r"] - b1["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10
    assert np.max(np.abs(a1["v"] - b1["v"])) < 1e-10 * np.max(np.abs(a1["v"]))


def test_leaver_through_a_periodic_face_lands_inside_the_receiver():
    """a molecule that leaves through the global low face at x = -tiny arrives on the far rank at x + L, which
    rounds to exactly L = the receiver's bmax — outside [bmin, bmax).  The import clamps it onto the box (the rounding rule
    of the local wrap, DomainDecompBase.cpp:206-219); without it the next halo generation reported a lost molecule.  Same at
    x = L - tiny leaving upwards (arrives at exactly 0 - handled by the lower clamp)."""
    L, r, v = _liquid(12)
    rc, dt = 2.5, 0.004
    v = v * 0.0
    r = r.copy()
    # two molecules a hair inside the global faces, moving out by less than the rounding granularity of L
    r[0] = [1e-17, 0.3 * L[1], 0.4 * L[2]]
    v[0] = [-2e-17 / dt, 0.0, 0.0]
    r[1] = [np.nextafter(L[0], 0.0), 0.6 * L[1], 0.2 * L[2]]
    v[1] = [1e-3, 0.0, 0.0]
    keep = np.ones(len(r), bool)  # drop neighbours that sit on top of the two probes
    for p in (0, 1):
        d = r - r[p]
        d -= L * np.round(d / L)
        keep &= (np.linalg.norm(d, axis=1) > 0.8) | (np.arange(len(r)) == p)
    r, v = r[keep], v[keep]
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    single = InProcessCluster(1, comps, rc, L, ids, r, v)
    multi = InProcessCluster(2, comps, rc, L, ids, r, v, (2, 1, 1))
    single.forces(split=False)
    multi.forces()
    for _ in range(3):
        t1, _k = single.step(dt)
        tm, _k = multi.step(dt)  # raised LS1HIP_ELOST before the clamp
    a, b = single.gather(), multi.gather()
    assert np.array_equal(a["ids"], b["ids"])
    dr = a["r"] - b["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-11
    assert np.max(np.abs(a["F"] - b["F"])) < 1e-11 * np.max(np.abs(a["F"]))
    assert np.allclose(t1, tm, rtol=1e-11)
    for e in single.eng + multi.eng:
        e.close()


@pytest.mark.parametrize("world,grid", [(1, None), (2, None), (8, None), (4, (1, 2, 2))])
def test_list_mode_across_subboxes_equals_single_domain_loop(world, grid):
    """The decomposed loop in LIST MODE (no migration / re-binning / halo regeneration between rebuilds, halo copies fed
    by position-only refresh messages, inner-brick pass before the refresh, collective rebuild decision) against the
    single domain running the ordinary per-step loop: same trajectory over several list lifetimes, hot enough that
    molecules cross sub-box faces and periodic faces while the lists are alive."""
    L, r, v = _liquid(16)
    rc, dt, nsteps = 2.5, 0.002, 30
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    single = InProcessCluster(1, comps, rc, L, ids, r, v)
    single.forces(split=False)
    for _ in range(nsteps):
        t1, k1 = single.step(dt)
    multi = InProcessCluster(world, comps, rc, L, ids, r, v, grid, skin=0.3)
    multi.forces()
    tm, km, builds = multi.run_lists(dt, nsteps)
    assert 2 <= builds <= nsteps // 2, builds
    a, b = single.gather(), multi.gather()
    assert np.array_equal(a["ids"], b["ids"])  # nobody lost or duplicated
    dr = a["r"] - b["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10
    assert np.max(np.abs(a["v"] - b["v"])) < 1e-10 * np.max(np.abs(a["v"]))
    assert np.max(np.abs(a["F"] - b["F"])) < 1e-9 * np.max(np.abs(a["F"]))
    assert np.allclose(t1, tm, rtol=1e-10) and np.allclose(k1, km, rtol=1e-10)
    for e in single.eng + multi.eng:
        e.close()
