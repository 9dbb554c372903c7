"""Rigid-body trajectories of ASYMMETRIC tops with multipole torques on the device, against the REAL reference (round 4).

Rounds 1-3 pinned the rotational half of the Leapfrog (FullMolecule::upd_preF / upd_postF, /root/reference/src/molecules/
FullMolecule.cpp:334-389; Quaternion::rotate / rotateinv / differentiate, Quaternion.cpp:43-98) on linear rotors only (ethane: I = (a, a,
0)).  The golden trajectories used here come from oracle/_ref/refdump (the reference's own objects) on
  * periodic water (the reference's VectorizationWater.inp: three non-zero moments, one LJ centre + three charges), NVE and — with
    thermal velocities, see tests/golden/make_golden.py — under the global velocity-scaling thermostat,
  * the integrable five-component LJ + charge + dipole + quadrupole set of BASELINE configs[4] (synth.mixed5_components: three
    asymmetric tops carrying dipoles / quadrupoles, one linear rotor, water), NVE and NVT,
  * a two-component mixture with one thermostat per component assigned by the legacy .inp header (component-wise branch of
    VelocityScalingThermostat::apply, Simulation.cpp:1112-1126, Leapfrog.cpp:84-112, Domain.cpp:204-240).
Each runs through the per-step kernels (piecewise loop: test_gpu_parity.py picks the new cases up by itself), through ls1hip_run with
the multi-site neighbour lists (here), and through the unmodified reference driver with the device container (here).
Tolerances: trajectories 1e-9, forces after the last step 1e-8 (as every trajectory test of this suite)."""
import os
import re
import shutil

import numpy as np
import pytest

from conftest import ROOT, load_pkg
from golden_io import input_path, manifest, read_golden, rel_max, sorted_phase_space

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
capi = load_pkg("capi")
engine_mod = load_pkg("engine")
MAN = manifest()


def _engine(ps, st, rc, skin, **opts):
    e = engine_mod.DeviceEngine(0)
    e.set_components(ps.components, rc)
    for k, v in opts.items():
        e.set_option(k, v)
    if skin:
        e.set_verlet(skin)
    e.set_domain(ps.length)
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    e.upload(st["ids"], st["cid"], st["r"], st["v"], q, st["D"])
    return e


def _sorted(e):
    st = e.download_state()
    o = np.argsort(st["ids"], kind="stable")
    f = e.download_forces()
    return {"ids": st["ids"][o], "cid": st["cid"][o], "r": st["r"][o], "v": st["v"][o], "q": st["q"][o], "D": st["D"][o],
            "F": f["F"][o], "M": f["M"][o]}


def _check_trajectory(s, g, L):
    rec = g["recs"]
    assert np.array_equal(s["ids"], rec["id"])
    dr = s["r"] - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(s["v"], rec["v"]) < 1e-9
    assert rel_max(s["q"], rec["q"]) < 1e-9
    assert np.max(np.abs(rec["D"])) > 0 and rel_max(s["D"], rec["D"]) < 1e-9
    assert rel_max(s["F"], rec["F"]) < 1e-8
    assert np.max(np.abs(rec["M"])) > 0 and rel_max(s["M"], rec["M"]) < 1e-8


# skin as a fraction of r_c: small enough that the cell grid keeps at least three cells per dimension (water: L = 37, r_c = 12)
@pytest.mark.parametrize("name,skin_frac", [("water_rc12_steps5", 0.02), ("waterT_250_nvt5", 0.02), ("mixed5_1024_steps5", 0.05),
                                            ("mixed5_1024_nvt5", 0.05)])
def test_asymmetric_rotors_through_the_list_loop(name, skin_frac):
    """ls1hip_run with per-wave pair-stream lists: positions, velocities, quaternions, angular momenta, forces, torques, globals."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    assert sum(1 for c in ps.components.components if np.all(c.I > 0)) >= 1  # at least one asymmetric top in the set
    st = sorted_phase_space(ps)
    e = _engine(ps, st, case["rc"], skin_frac * case["rc"])
    assert e.get_option("verlet_lists") == 1
    if case["nvt"]:
        e.set_thermostat(True, ps.temperature)
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(case["dt"], case["steps"])
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
    assert e.get_option("verlet_steps") == case["steps"] and 1 <= e.get_option("verlet_builds") <= case["steps"]
    _check_trajectory(_sorted(e), g, ps.length)
    assert abs(out["upot"] - g["upot"]) <= 1e-9 * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= 1e-8 * abs(g["virial"])
    if not case["nvt"]:  # (golden sums of a thermostatted run are taken after the last scaling, ls1hip_run reports them before)
        assert abs(out["summv2"] - g["summv2"]) <= 1e-9 * abs(g["summv2"])
        assert abs(out["sumIw2"] - g["sumIw2"]) <= 1e-9 * abs(g["sumIw2"])
    e.close()


@pytest.mark.parametrize("name", ["water_rc12_steps5", "mixed5_1024_steps5"])
def test_asymmetric_rotors_device_loop_per_step_kernels(name):
    """the same through ls1hip_run WITHOUT lists (search every step: brick kernel + separate rigid-body integrator passes)"""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = _engine(ps, st, case["rc"], 0.0)
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(case["dt"], case["steps"])
    assert e.get_option("last_force_kernel") != capi.FK_NEIGHBOUR_LIST
    _check_trajectory(_sorted(e), g, ps.length)
    assert abs(out["upot"] - g["upot"]) <= 1e-9 * abs(g["upot"])
    assert abs(out["sumIw2"] - g["sumIw2"]) <= 1e-9 * abs(g["sumIw2"])
    e.close()


def test_mixed5_forces_list_pass_and_brick_kernel_against_golden():
    """one evaluation of the integrable five-component set (all ten site-type combinations, reaction field on, full LJ mixing)"""
    case = MAN["mixed5_1024"]
    g = read_golden("mixed5_1024")
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    for skin in (0.0, 2.0):
        e = _engine(ps, st, case["rc"], skin)
        if skin:
            assert e.update() is True
            u, w = e.forces_list(0, 0.0, want_macro=True)
            assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
        else:
            e.rebin(); e.halo()
            u, w = e.forces(0)
        s = _sorted(e)
        assert rel_max(s["F"], g["recs"]["F"]) < 1e-10 and rel_max(s["M"], g["recs"]["M"]) < 1e-10
        assert abs(u - g["upot"]) <= 1e-10 * abs(g["upot"]) and abs(w - g["virial"]) <= 1e-10 * abs(g["virial"])
        e.close()


def _betas(sums, T):
    """Domain::calculateGlobalValues for one thermostat (Domain.cpp:225-240)"""
    mv2, iw2, n, rd = sums
    bt = (3.0 * n * T / mv2) ** 0.4
    br = 1.0 if (iw2 == 0.0 or rd == 0) else (rd * T / iw2) ** 0.4
    return bt, br


@pytest.mark.parametrize("skin", [0.0, 0.15])
def test_component_wise_thermostats_against_reference_golden(skin):
    """Two components, two thermostats (legacy .inp header): the piecewise loop a driver with several thermostats runs —
    ls1hip_kick, ls1hip_kinetic_sums_by_component, per-thermostat betas on the host (as Domain::calculateGlobalValues),
    ls1hip_scale_kick_drift_components — against the reference's component-wise branch, per-step kernels and list mode."""
    name = "twotherm_1024_nvt8"
    case = MAN[name]
    assert case["componentwise"]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    assert ps.comp_thermostat == {0: 1, 1: 2} and ps.thermostat_T == {1: 0.8, 2: 1.1}
    st = sorted_phase_space(ps)
    ncomp = len(ps.components.components)
    e = _engine(ps, st, case["rc"], skin)
    dt = case["dt"]
    if skin:
        assert e.update() is True
        e.forces_list(0, 0.0, want_macro=False)
    else:
        e.rebin(); e.halo(); e.forces(0)
    bt = np.ones(ncomp); br = np.ones(ncomp)
    for step in range(case["steps"]):
        e.scale_kick_drift_components(bt, br, dt)  # (step 0: factors 1 = the plain pre-force kick + drift)
        if skin:
            e.update()
            u, w = e.forces_list(0, 0.0, want_macro=True)
        else:
            e.rebin(); e.halo()
            u, w = e.forces(0)
        e.kick(0.5 * dt, want_sums=False)
        s = e.kinetic_sums_by_component(ncomp)
        # thermostat id -> sums over its components (Leapfrog.cpp:84-112), then one pair of betas per id
        for th, T in ps.thermostat_T.items():
            comps = [c for c, t in ps.comp_thermostat.items() if t == th]
            tot = (sum(s["summv2"][c] for c in comps), sum(s["sumIw2"][c] for c in comps), sum(int(s["n"][c]) for c in comps),
                   sum(int(s["rot_dof"][c]) for c in comps))
            b = _betas(tot, T)
            for c in comps:
                bt[c], br[c] = b
    assert np.all(bt > 0.9) and np.all(br > 0.9)  # (below 0.9 the reference switches to its explosion heuristics, Domain.cpp:255)
    out = _sorted(e)
    # the golden state is the one after the last scaling
    out["v"] = out["v"] * bt[out["cid"]][:, None]
    out["D"] = out["D"] * br[out["cid"]][:, None]
    rec = g["recs"]
    L = ps.length
    dr = out["r"] - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(out["v"], rec["v"]) < 1e-9
    assert rel_max(out["q"], rec["q"]) < 1e-9
    assert rel_max(out["D"], rec["D"]) < 1e-9
    assert rel_max(out["F"], rec["F"]) < 1e-8
    assert abs(u - g["upot"]) <= 1e-9 * abs(g["upot"]) and abs(w - g["virial"]) <= 1e-8 * abs(g["virial"])
    if skin:
        assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
    e.close()


def test_rotational_degrees_of_freedom_follow_the_reference_count():
    """Component::getRotationalDegreesOfFreedom counts the moments of the SITE masses (Component.cpp:140-167); an I line that turns a
    zero site moment into a non-zero one changes the motion but not that count — ls1hip_set_rot_dof carries it."""
    c0 = inp.make_component(lj=[(0, 0, 0, 1.0, 1.0, 1.0, 0, 0)], dipoles=[(0, 0, 0, 0, 0, 1, 0.5)], I_file=(0.3, 0.4, 0.5))
    c1 = inp.make_component(lj=[(0, 0, -0.3, 0.5, 1.0, 1.0, 0, 0), (0, 0, 0.3, 0.5, 1.0, 1.0, 0, 0)])
    assert (c0.rot_dof, c1.rot_dof) == (0, 2) and np.all(c0.I > 0)
    cs = inp.ComponentSet([c0, c1], np.ones((1, 2)), 1e10)
    rng = np.random.default_rng(3)
    n, L = 400, 9.0
    r = rng.uniform(0, L, (n, 3))
    cid = (np.arange(n) % 2).astype(np.int32)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    e = engine_mod.DeviceEngine(0)
    e.set_components(cs, 2.5)
    e.set_domain([L, L, L])
    e.upload(np.arange(1, n + 1, dtype=np.uint64), cid, r, rng.normal(size=(n, 3)), q, rng.normal(size=(n, 3)) * 0.1)
    e.rebin(); e.halo(); e.forces(0)
    _, iw2, nn, rd = e.kick(0.0)
    assert nn == n and rd == (n // 2) * 2 and iw2 > 0  # component 0 rotates (its I w^2 is counted) but adds no degree of freedom
    s = e.kinetic_sums_by_component(2)
    assert list(s["rot_dof"]) == [0, n] and s["sumIw2"][0] > 0
    e.close()


HIPB = os.path.join(ROOT, "oracle", "_ref", "MarDyn_hipB")
REF = os.path.join(ROOT, "oracle", "_ref", "MarDyn")


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPB)), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("skin", ["default", "0"])
def test_reference_driver_five_component_set_on_the_device_container(tmp_path, skin):
    """The unmodified driver (XML config, NVT as every shipped example) on the integrable five-component set: MarDyn_hipB (device
    container + integrator; multi-site neighbour lists by default, search every step with LS1HIP_SKIN=0) prints the per-step
    T / U_pot / p of the unmodified MarDyn and leaves the same final checkpoint (positions, velocities, orientations, angular momenta)."""
    from test_gpu_seam_a import HEAD, _run
    from test_gpu_seam_b import _restart_records
    src = input_path("synthetic_mixed5_8.inp")
    ps = inp.read_inp(src)
    steps = 12
    cfg = HEAD.format(dt=0.1, steps=steps, temp=repr(float(ps.temperature)), L=repr(float(ps.length[0])), rc=35.0,
                      components=inp.components_xml(ps.components), phasespace='<file type="ASCII">m5.inp</file>')
    out = {}
    env_skin = os.environ.get("LS1HIP_SKIN")
    try:
        if skin != "default":
            os.environ["LS1HIP_SKIN"] = skin
        for tag, binary in (("ref", REF), ("hipB", HIPB)):
            d = tmp_path / tag
            d.mkdir()
            shutil.copy(src, d / "m5.inp")
            (d / "config.xml").write_text(cfg)
            rows, log = _run(binary, "config.xml", str(d), steps, final_checkpoint=1)
            out[tag] = (rows, log, d)
    finally:
        if env_skin is None:
            os.environ.pop("LS1HIP_SKIN", None)
        else:
            os.environ["LS1HIP_SKIN"] = env_skin
    ref, hip = out["ref"][0], out["hipB"][0]
    assert "LinkedCellsHip: device-resident container" in out["hipB"][1]
    m = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+) list force evaluations", out["hipB"][1])
    assert m and m.group(1) == ("on" if skin == "default" else "off")
    n = min(len(ref), len(hip))
    assert n >= steps
    assert np.allclose(hip[:n], ref[:n], rtol=2e-5, atol=1e-12), (ref[:n], hip[:n])
    fr = [f for f in os.listdir(out["ref"][2]) if f.endswith(".restart.dat")]
    fh = [f for f in os.listdir(out["hipB"][2]) if f.endswith(".restart.dat")]
    a, b = _restart_records(out["ref"][2] / fr[0]), _restart_records(out["hipB"][2] / fh[0])
    assert a.keys() == b.keys() and len(a) == len(ps.ids)
    ids = sorted(a)
    A, B = np.array([a[i] for i in ids]), np.array([b[i] for i in ids])
    Lbox = float(ps.length[0])
    dr = A[:, :3] - B[:, :3]
    dr -= Lbox * np.round(dr / Lbox)
    assert np.max(np.abs(dr)) < 1e-7 * Lbox
    for lo, hi, what in ((3, 6, "v"), (6, 10, "q"), (10, 13, "D")):
        assert np.max(np.abs(A[:, lo:hi] - B[:, lo:hi])) < 1e-6 * np.max(np.abs(A[:, lo:hi])), what
