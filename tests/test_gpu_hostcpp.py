"""The C++17 host layer (ls1-mardyn_amd/host/ls1hip_host.hpp: LinkedCells / VectorizedCellProcessor / Leapfrog /
DomainDecompBase / simulate with the reference's names, over the C ABI) driven by tests/hostcpp/host_sim.cpp against the
golden trajectories of the REAL reference — the same assertions as test_gpu_parity.py::test_trajectory_matches_reference
makes for the Python mirror."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import load_pkg
from golden_io import input_path, manifest, read_golden, rel_max, sorted_phase_space

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
MAN = manifest()
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "hostcpp", "host_sim")
CASES = [k for k, c in MAN.items() if c["steps"] > 0 and not c["nvt"]] + ["ethan", "multi_periodic"]


def write_case(path, ps, st, rc, dt, nsteps):
    f = ps.components.flat()
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    with open(path, "wb") as fh:
        fh.write(b"LS1CASE1")
        fh.write(struct.pack("<iidd3dd", int(f["ncomp"]), int(nsteps), float(rc), float(dt), *[float(x) for x in ps.length],
                             float(f["eps_rf"])))
        for k in ("nlj", "nc", "nd", "nq"):
            fh.write(np.ascontiguousarray(f[k], dtype=np.int32).tobytes())
        for k in ("lj", "ch", "dp", "qp", "mass", "I", "mix"):
            a = np.ascontiguousarray(f[k], dtype=np.float64).reshape(-1)
            fh.write(struct.pack("<Q", a.size))
            fh.write(a.tobytes())
        n = len(st["ids"])
        fh.write(struct.pack("<Q", n))
        fh.write(np.ascontiguousarray(st["ids"], dtype=np.uint64).tobytes())
        fh.write(np.ascontiguousarray(st["cid"], dtype=np.int32).tobytes())
        for a in (st["r"], st["v"], q, st["D"]):
            fh.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())


def read_result(path):
    with open(path, "rb") as fh:
        assert fh.read(8) == b"LS1RES01"
        n, = struct.unpack("<Q", fh.read(8))
        upot, virial, summv2, sumIw2 = struct.unpack("<4d", fh.read(32))
        ids = np.frombuffer(fh.read(8 * n), dtype=np.uint64)
        out = dict(ids=ids, upot=upot, virial=virial, summv2=summv2, sumIw2=sumIw2)
        for k, w in (("r", 3), ("v", 3), ("q", 4), ("D", 3), ("F", 3), ("M", 3)):
            out[k] = np.frombuffer(fh.read(8 * w * n), dtype=np.float64).reshape(n, w)
    return out


@pytest.mark.parametrize("name", CASES)
def test_cpp_host_layer_matches_reference(name, tmp_path):
    assert os.path.exists(EXE), "tests/hostcpp/host_sim not built (python -c 'import __graft_entry__ as g; g.build()')"
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    cf, rf = str(tmp_path / "case.bin"), str(tmp_path / "res.bin")
    write_case(cf, ps, st, case["rc"], case["dt"] if case["steps"] else 0.0, case["steps"])
    res = subprocess.run([EXE, cf, rf], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    out = read_result(rf)
    o = np.argsort(out["ids"], kind="stable")
    rec = g["recs"]
    assert np.array_equal(out["ids"][o], rec["id"])
    tol = 1e-9 if case["steps"] else 1e-10
    assert rel_max(out["F"][o], rec["F"]) < 10 * tol
    assert rel_max(out["M"][o], rec["M"]) < 10 * tol or np.max(np.abs(rec["M"])) == 0
    assert abs(out["upot"] - g["upot"]) <= tol * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= 10 * tol * abs(g["virial"])
    if case["steps"]:
        L = ps.length
        dr = out["r"][o] - rec["r"]
        dr -= L * np.round(dr / L)
        assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
        assert rel_max(out["v"][o], rec["v"]) < 1e-9
        assert rel_max(out["q"][o], rec["q"]) < 1e-9
        assert abs(out["summv2"] - g["summv2"]) <= 1e-9 * abs(g["summv2"])
