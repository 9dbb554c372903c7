"""CPU tests: the oracle (oracle/ls1_oracle.c) against golden vectors produced by the REAL reference
(tests/golden/*.bin <- oracle/_ref/refdump) and against the reference tests' analytic known answers
(/root/reference/src/particleContainer/adapter/tests/VectorizedCellProcessorTest.cpp:45-134)."""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import FORCE_FLOOR, input_path, manifest, read_golden, rel_max, sorted_phase_space
from oracle.oracle import Oracle

inp = load_pkg("inp")
MAN = manifest()
FORCE_CASES = [k for k, c in MAN.items() if c["steps"] == 0]
STEP_CASES = [k for k, c in MAN.items() if c["steps"] > 0]
TOL = 1e-11  # reference's own VCP-vs-legacy tolerance is 1e-12 abs (LJ) / 1e-11 (electrostatics)


def _setup(case):
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    orc = Oracle(ps.components.flat(), case["rc"])
    return ps, st, orc


def test_known_answer_U0():
    # F = (+-24, +-24, 0), U = 0, virial = 96 (VectorizedCellProcessorTest.cpp:56-59,88-91)
    case = MAN["U0"]
    ps, st, orc = _setup(case)
    out = orc.forces(st["r"], st["q"], st["cid"], ps.length, periodic=False)
    exp = np.array([[-24, -24, 0], [24, -24, 0], [-24, 24, 0], [24, 24, 0]], dtype=float)
    assert np.allclose(out["F"], exp, atol=1e-12)
    assert abs(out["upot"]) < 1e-12
    assert abs(out["virial"] - 96.0) < 1e-10


def test_known_answer_F0():
    # r = 2^(1/6): F = 0, U = -4 (VectorizedCellProcessorTest.cpp:97-134)
    case = MAN["F0"]
    ps, st, orc = _setup(case)
    out = orc.forces(st["r"], st["q"], st["cid"], ps.length, periodic=False)
    assert np.max(np.abs(out["F"])) < 1e-6
    assert abs(out["upot"] + 4.0) < 1e-10


@pytest.mark.parametrize("name", FORCE_CASES)
def test_forces_match_reference(name):
    case = MAN[name]
    g = read_golden(name)
    ps, st, orc = _setup(case)
    assert np.array_equal(st["ids"], g["recs"]["id"])
    assert np.array_equal(st["cid"], g["recs"]["cid"].astype(np.int32))
    # the reader reproduces the reference's state exactly (positions are not touched by a force evaluation)
    assert np.array_equal(st["r"], g["recs"]["r"])
    out = orc.forces(st["r"], st["q"], st["cid"], ps.length, periodic=bool(case["periodic"]))
    for key in ("F", "M", "Vi"):
        assert rel_max(out[key], g["recs"][key], FORCE_FLOOR.get(name, 0.0)) < TOL, key
    scale = max(abs(g["upot"]), 1e-300)
    assert abs(out["upot"] - g["upot"]) <= TOL * max(scale, 1.0) or abs(out["upot"] - g["upot"]) / scale < 1e-10
    vs = max(abs(g["virial"]), 1e-300)
    assert abs(out["virial"] - g["virial"]) <= TOL * max(vs, 1.0) or abs(out["virial"] - g["virial"]) / vs < 1e-10


@pytest.mark.parametrize("name", STEP_CASES)
def test_trajectory_matches_reference(name):
    """Leapfrog + wrap + halo + forces for several steps (Simulation.cpp:995-1099 order) vs the reference."""
    case = MAN[name]
    g = read_golden(name)
    ps, st, orc = _setup(case)
    L = ps.length
    r, v, q, D, cid = st["r"], st["v"], st["q"], st["D"], st["cid"]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = orc.forces(r, q, cid, L, periodic=True)
    F, M = out["F"].copy(), out["M"].copy()
    T = ps.temperature if case["nvt"] else None
    several = (ps.thermostat_T, ps.comp_thermostat) if case["nvt"] and ps.comp_thermostat else None
    for _ in range(case["steps"]):
        out = orc.step(case["dt"], cid, r, v, q, D, F, M, L, periodic=True, target_T=T, thermostats=several)
    if case["nvt"] and several is None:
        # the golden kinetic sums are taken after the final velocity scaling
        out["summv2"] *= out["beta_trans"] ** 2
        out["sumIw2"] *= out["beta_rot"] ** 2
    rec = g["recs"]
    # positions modulo the box (a molecule sitting exactly on a face may be represented on either side)
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(v, rec["v"]) < 1e-9
    assert rel_max(q, rec["q"]) < 1e-9
    if np.max(np.abs(rec["D"])) > 0:
        assert rel_max(D, rec["D"]) < 1e-9
    assert rel_max(out["F"], rec["F"]) < 1e-8
    assert abs(out["upot"] - g["upot"]) / abs(g["upot"]) < 1e-9
    assert abs(out["virial"] - g["virial"]) / abs(g["virial"]) < 1e-8
    assert abs(out["summv2"] - g["summv2"]) / abs(g["summv2"]) < 1e-9
    if g["sumIw2"] != 0:
        assert abs(out["sumIw2"] - g["sumIw2"]) / abs(g["sumIw2"]) < 1e-9


@pytest.mark.parametrize("name", [k for k in FORCE_CASES if not MAN[k]["legacy"]])
def test_homogeneous_long_range_correction(name):
    """Homogeneous LRC constants vs the reference (longRange/Homogeneous.cpp)."""
    case = MAN[name]
    g = read_golden(name)
    if g["lrc"] is None:
        pytest.skip("fixture without LRC trailer")
    ps, st, orc = _setup(case)
    u, v = orc.lrc_homogeneous(st["cid"], ps.length)
    # the harness obtains the corrections as (global - local) differences: compare on the scale of those sums
    assert abs(u - g["lrc"][0]) <= 1e-12 * max(abs(g["lrc"][0]), abs(g["upot"]))
    assert abs(v - g["lrc"][1]) <= 1e-12 * max(abs(g["lrc"][1]), abs(g["virial"]))
