"""CPU tests of the multi-GPU path: decomposition geometry, and the exchange protocol / overlapped step loop of
ls1-mardyn_amd/decomp.py driven under gloo with world_size 2 and 4 over a numpy stand-in engine, compared with the
oracle's single-domain result (reference semantics: DomainDecomposition.cpp:112-123, DomainDecompBase.cpp:174-348)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, load_pkg
from oracle.oracle import Oracle

decomp = load_pkg("decomp")
inp = load_pkg("inp")


def test_dims_create():
    assert decomp.dims_create(1) == (1, 1, 1)
    assert decomp.dims_create(2) == (2, 1, 1)
    assert decomp.dims_create(4) == (2, 2, 1)
    assert decomp.dims_create(8) == (2, 2, 2)
    assert decomp.dims_create(6) == (3, 2, 1)


def test_bounding_boxes_tile_the_global_box():
    L = np.array([10.0, 7.0, 9.0])
    for world in (2, 4, 8):
        vol = 0.0
        for r in range(world):
            dc = decomp.CartesianDecomposition(world, r, L)
            lo, hi = dc.bounding_box()
            vol += np.prod(hi - lo)
            for d in range(3):
                if dc.coords[d] == dc.grid[d] - 1:
                    assert hi[d] == L[d]  # exact: the device recognises global faces by equality
                if dc.coords[d] == 0:
                    assert lo[d] == 0.0
        assert abs(vol - np.prod(L)) < 1e-9


def test_neighbor_table_2x2x2_has_seven_peers():
    for r in range(8):
        dc = decomp.CartesianDecomposition(8, r, [1, 1, 1])
        t = dc.neighbor_table()
        assert t[13] == r
        assert len(dc.peers()) == 7
        # symmetric: if s is my neighbour in direction d, I am its neighbour in direction -d
        for d in range(27):
            other = decomp.CartesianDecomposition(8, int(t[d]), [1, 1, 1]).neighbor_table()
            assert other[26 - d] == r
    dc = decomp.CartesianDecomposition(2, 0, [1, 1, 1])
    t = dc.neighbor_table()
    assert t[12] == 1 and t[14] == 1 and t[10] == 0 and t[4] == 0  # +-x -> peer, y/z -> periodic self
    open_dc = decomp.CartesianDecomposition(2, 0, [1, 1, 1], periodic=(False, True, True))
    assert open_dc.neighbor_table()[12] == -1 and open_dc.neighbor_table()[14] == 1


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _liquid(n, seed=5, rho=0.6, rc=2.5):
    rng = np.random.default_rng(seed)
    N = 2 * n ** 3
    L = (N / rho) ** (1 / 3)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.15 * (rng.random((N, 3)) - 0.5)
    r %= L
    v = rng.normal(0, 1.2, (N, 3))
    v -= v.mean(0)
    return np.array([L] * 3), r, v


@pytest.mark.parametrize("world,grid", [(2, "2x1x1"), (4, "2x2x1"), (2, "1x1x2"), (1, "1x1x1+loopback"),
                                        (2, "2x1x1+loopback"), (2, "2x1x1+lists"), (4, "2x2x1+lists"),
                                        (2, "1x1x2+lists+loopback")])
def test_gloo_world_matches_single_domain_oracle(world, grid):
    """'+loopback': the periodic images a rank would create locally travel through the transport instead (messages to
    the own rank), the rehearsal mode of bench.py --decomp --loopback."""
    loopback = "+loopback" in grid
    lists = "+lists" in grid  # list mode: halo copies refreshed by position-only messages, collective rebuild decision
    grid = grid.split("+")[0]
    L, r, v = _liquid(5)  # 250 atoms, L ~ 7.5 (>= 2 rc per sub-box edge is NOT required: images handle it)
    rc, dt, nsteps = 1.8, 0.004, (12 if lists else 4)
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    orc = Oracle(comps.flat(), rc)
    cid = np.zeros(len(r), np.int32); q = np.tile([1.0, 0, 0, 0], (len(r), 1)); D = np.zeros_like(r)
    ro, vo = r.copy(), v.copy()
    o0 = orc.forces(ro, q, cid, L, True)
    F, M = o0["F"].copy(), o0["M"].copy()
    for _ in range(nsteps):
        o = orc.step(dt, cid, ro, vo, q, D, F, M, L, True)
    with tempfile.TemporaryDirectory() as td:
        inp_path = os.path.join(td, "in.npz")
        np.savez(inp_path, L=L, r=r, v=v, ids=ids, rc=rc)
        out_path = os.path.join(td, "out.npz")
        env = dict(os.environ, LS1_TEST_INPUT=inp_path, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1",
                   LS1_TEST_LOOPBACK="1" if loopback else "0", LS1_TEST_SKIN="0.08" if lists else "")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "tests", "decomp_worker.py"), out_path, str(nsteps), repr(dt), grid]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        z = np.load(out_path)
        # initial forces of all ranks == oracle single domain
        oi = np.argsort(z["ids0"])
        assert np.array_equal(z["ids0"][oi], ids)
        assert np.max(np.abs(z["F0"][oi] - o0["F"])) < 1e-10 * np.max(np.abs(o0["F"]))
        assert abs(z["upot0"] - o0["upot"]) < 1e-10 * abs(o0["upot"])
        assert abs(z["virial0"] - o0["virial"]) < 1e-10 * abs(o0["virial"])
        # after nsteps with migration between ranks: same trajectory, nobody lost or duplicated
        of = np.argsort(z["ids"])
        assert np.array_equal(z["ids"][of], ids)
        dr = z["r"][of] - ro
        dr -= L * np.round(dr / L)
        assert np.max(np.abs(dr)) < 1e-10
        assert np.max(np.abs(z["v"][of] - vo)) < 1e-10
        assert abs(z["upot"] - o["upot"]) < 1e-9 * abs(o["upot"])
        assert abs(z["summv2"] - o["summv2"]) < 1e-10 * o["summv2"]
        assert int(z["n"]) == len(ids)


@pytest.mark.parametrize("where", ["export", "import"])
def test_engine_error_on_one_rank_fails_every_rank_without_hanging(where):
    """error handling must be collective.  One rank's engine fails (capacity overflow / lost molecule are raised
    by export_counts / import_done); every rank must leave with DecompositionError instead of blocking in the next
    all_gather / irecv."""
    L, r, v = _liquid(5)
    ids = np.arange(1, len(r) + 1, dtype=np.uint64)
    with tempfile.TemporaryDirectory() as td:
        inp_path = os.path.join(td, "in.npz")
        np.savez(inp_path, L=L, r=r, v=v, ids=ids, rc=1.8)
        env = dict(os.environ, LS1_TEST_INPUT=inp_path, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", LS1_TEST_LOOPBACK="0",
                   LS1_TEST_FAIL=f"1:5:{where}")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "tests", "decomp_worker.py"), os.path.join(td, "out.npz"), "4", "0.004", "2x1x1"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)  # a hang would hit the timeout
        assert res.returncode != 0
        assert res.stderr.count("DecompositionError") >= 2, res.stderr[-3000:]
        assert "injected engine failure" in res.stderr


@pytest.mark.parametrize("world,loopback", [(1, 0), (1, 1), (2, 0), (4, 0), (8, 0), (8, 1), (12, 0), (27, 0)])
def test_cpp_rank_grid_matches_the_python_decomposition(world, loopback):
    """CartDecomp of host/DomainDecompRccl.hpp (the C++ / RCCL decomposed loop) == decomp.CartesianDecomposition: grid,
    coordinates, bounding boxes (bitwise), the 27-entry neighbour tables incl. loopback aliases, peer sets."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "hostcpp", "decomp_rccl_main")
    if not os.path.exists(exe):
        pytest.skip("tests/hostcpp/decomp_rccl_main not built")
    out = subprocess.run([exe, "--geometry", str(world), str(loopback)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == world
    L = np.array([10., 12., 14.])
    for ln in lines:
        left, right = ln.split("|")
        t = left.split()
        r = int(t[0])
        dc = decomp.CartesianDecomposition(world, r, L, loopback=bool(loopback))
        assert tuple(int(x) for x in t[1:4]) == dc.grid and tuple(int(x) for x in t[4:7]) == dc.coords
        lo, hi = dc.bounding_box()
        assert np.array_equal(np.array([float(x) for x in t[7:10]]), lo) and np.array_equal(np.array([float(x) for x in t[10:13]]), hi)
        assert [int(x) for x in t[13:40]] == [int(x) for x in dc.neighbor_table()]
        assert [int(x) for x in right.split()] == dc.peers()
