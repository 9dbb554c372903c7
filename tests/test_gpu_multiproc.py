"""Multi-PROCESS run of the decomposed loop on the GPU: 2 / 4 ranks (one process each, launched by torch.distributed.run as
bench.py --gpus N is), every rank with its own DeviceEngine context on cuda:0, decomp.HaloExchanger /
DistributedSimulation moving the messages through gloo (staged through the host — RCCL refuses two ranks on one GPU).
Compared with the ORACLE on the single domain (oracle/ls1_oracle.c, pinned to the reference's goldens): initial forces,
U_pot, virial; trajectory, U_pot and the kinetic sum after several steps with migration across rank and periodic faces.
This is the closest a one-GPU box gets to bench.py --gpus N: real process boundaries, real collectives, real engines."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inp = load_pkg("inp")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _liquid(n, seed=23, rho=0.785302672):
    rng = np.random.default_rng(seed)
    N = 2 * n ** 3
    L = (N / rho) ** (1 / 3)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = (np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.1 * (rng.random((N, 3)) - 0.5)) % L
    v = rng.normal(0, 1.0, (N, 3)) * 2.5  # hot: molecules cross rank faces within a few steps
    v -= v.mean(0)
    return np.array([L] * 3), r, v


@pytest.mark.parametrize("world,grid,mode", [(2, "2x1x1", "lists"), (4, "2x2x1", "lists"), (2, "1x1x2", "fused"),
                                             (4, "1x2x2", "unfused")])
def test_ranks_in_separate_processes_match_the_oracle(world, grid, mode):
    from oracle.oracle import Oracle
    L, r, v = _liquid(16)  # 8192 atoms, L = 21.85
    rc, dt = 2.5, 0.002
    nsteps = 24 if mode == "lists" else 6
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
                   LS1_TEST_SKIN="0.3" if mode == "lists" else "", LS1_TEST_FUSE="0" if mode == "unfused" else "1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "tests", "decomp_gpu_worker.py"), out_path, str(nsteps), repr(dt), grid]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        z = np.load(out_path)
    oi = np.argsort(z["ids0"])
    assert np.array_equal(z["ids0"][oi], ids)
    assert np.max(np.abs(z["F0"][oi] - o0["F"])) < 1e-11 * np.max(np.abs(o0["F"]))
    assert abs(z["upot0"] - o0["upot"]) < 1e-11 * abs(o0["upot"])
    assert abs(z["virial0"] - o0["virial"]) < 1e-11 * abs(o0["virial"])
    of = np.argsort(z["ids"])
    assert np.array_equal(z["ids"][of], ids)  # nobody lost or duplicated while migrating between processes
    dr = z["r"][of] - ro
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10
    assert np.max(np.abs(z["v"][of] - vo)) < 1e-10 * np.max(np.abs(vo))
    assert np.max(np.abs(z["F"][of] - F)) < 1e-9 * np.max(np.abs(F))
    assert abs(z["upot"] - o["upot"]) < 1e-10 * abs(o["upot"])
    assert abs(z["virial"] - o["virial"]) < 1e-9 * abs(o["virial"])
    assert abs(z["summv2"] - o["summv2"]) < 1e-11 * o["summv2"]
    assert int(z["n"]) == len(ids)


def test_bench_py_launched_as_the_driver_launches_it_with_two_ranks():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`
    — the driver's own command line for N > 1 — on the one GPU of the test box (LS1_BENCH_BACKEND=gloo: both ranks on cuda:0, messages
    staged through the host; the rates are meaningless).  Rank 0 must print ONE JSON line with the contract's fields, the start-up
    check of the decomposed path against the single domain, the communicator's world size and every rank's molecule
    count, strong scaling (the ranks' molecules add up to the global box)."""
    import json
    n = 48  # 2 * 48^3 = 221 184 molecules split 2 x 1 x 1
    env = dict(os.environ, LS1_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
           "--n-per-dim", str(n), "--melt", "0", "--no-live-pmc", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "particle-updates/s"
    assert d["rccl_world_size"] == 2
    per = d["config"]["molecules_per_gpu_by_rank"]
    assert len(per) == 2 and sum(per) == 2 * n ** 3 and min(per) > 0
    sc = d["multi_gpu_self_check"]
    assert sc["ids_match"] and sc["owned_molecules_summed_over_ranks"] == sc["molecules"]
    assert sc["forces_max_rel_over_ranks"] < 1e-11 and sc["upot_rel"] < 1e-11
    assert abs(d["value"] - 2 * n ** 3 * 8 / (d["ms_per_step"] * 8e-3)) <= 1e-6 * d["value"]
