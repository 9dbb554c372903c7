"""Neighbour lists for MULTI-SITE component sets (kernels_force_mslist.hip: per-wave, component-pair-sorted pair streams, reused
until the device-side displacement bound says otherwise) against the REAL reference and the pinned oracle.

Reference behaviour matched: VectorizedCellProcessor::_calculatePairs with all ten site-type combinations
(VectorizedCellProcessor.cpp:796-2732, bodies :173-794), centre-of-mass cutoff masks (:967-968,1013-1024), calcFM
(FullMolecule.cpp:526-629), Leapfrog incl. rotation (FullMolecule.cpp:334-389); list reuse as in AutoPasContainer.cpp:281-346.
Tolerances: one evaluation 1e-10 (north_star), trajectories 1e-9."""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import input_path, manifest, read_golden, rel_max, sorted_phase_space
from oracle.oracle import Oracle
from test_gpu_fullsize import check_replicas, replicate, tile

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
capi = load_pkg("capi")
engine_mod = load_pkg("engine")
MAN = manifest()
TOL = 1e-10

# every multi-site / electrostatic golden case of one evaluation (open clusters and periodic boxes)
MULTISITE_FORCE_CASES = [k for k, c in MAN.items() if c["steps"] == 0 and not c["legacy"]
                         and not k.startswith(("bcc1clj", "U0", "F0", "lj1clj"))]


def _engine(ps, st, rc, skin, periodic=True, **opts):
    e = engine_mod.DeviceEngine(0)
    e.set_components(ps.components, rc)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_verlet(skin)
    e.set_domain(ps.length, periodic=periodic)
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    e.upload(st["ids"], st["cid"], st["r"], st["v"], q, st["D"])
    return e


def _sorted(e):
    st = e.download_state()
    o = np.argsort(st["ids"], kind="stable")
    f = e.download_forces()
    return {"ids": st["ids"][o], "r": st["r"][o], "v": st["v"][o], "q": st["q"][o], "D": st["D"][o], "F": f["F"][o], "M": f["M"][o]}


@pytest.mark.parametrize("name", MULTISITE_FORCE_CASES)
def test_multisite_list_pass_against_reference_golden(name):
    """One list build + one list force pass on every multi-site golden case of the real reference."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = _engine(ps, st, case["rc"], 0.07 * case["rc"], periodic=bool(case["periodic"]))
    assert e.get_option("verlet_lists") == 1
    assert e.update() is True
    u, w = e.forces_list(0, 0.0, want_macro=True)
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST and e.get_option("verlet_builds") == 1
    out = _sorted(e)
    rec = g["recs"]
    assert np.array_equal(out["ids"], rec["id"])
    assert rel_max(out["F"], rec["F"]) < TOL
    assert rel_max(out["M"], rec["M"]) < TOL
    assert abs(u - g["upot"]) <= TOL * max(abs(g["upot"]), 1e-300) or abs(u - g["upot"]) < 1e-12
    assert abs(w - g["virial"]) <= TOL * max(abs(g["virial"]), 1e-300) or abs(w - g["virial"]) < 1e-9
    e.close()


@pytest.mark.parametrize("name", ["ethan_steps5", "ethan_nvt5", "lj_steps3"])
def test_multisite_list_loop_against_reference_trajectory(name):
    """ls1hip_run with multi-site neighbour lists (NVE and with the velocity-scaling thermostat) against the golden
    trajectories of the real reference: positions, velocities, orientations, angular momenta, forces, globals."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = _engine(ps, st, case["rc"], 0.05 * case["rc"])
    if case["nvt"]:
        e.set_thermostat(True, ps.temperature)
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(case["dt"], case["steps"])
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
    assert e.get_option("verlet_steps") == case["steps"] and 1 <= e.get_option("verlet_builds") <= case["steps"]
    s = _sorted(e)
    rec = g["recs"]
    L = ps.length
    dr = s["r"] - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(s["v"], rec["v"]) < 1e-9
    assert rel_max(s["q"], rec["q"]) < 1e-9
    if np.max(np.abs(rec["D"])) > 0:
        assert rel_max(s["D"], rec["D"]) < 1e-9
    assert rel_max(s["F"], rec["F"]) < 1e-8
    if np.max(np.abs(rec["M"])) > 0:
        assert rel_max(s["M"], rec["M"]) < 1e-8
    assert abs(out["upot"] - g["upot"]) <= 1e-9 * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= 1e-8 * abs(g["virial"])
    if not case["nvt"]:  # (the golden sums of a thermostatted run are taken after the last scaling, ls1hip_run reports them before)
        assert abs(out["summv2"] - g["summv2"]) <= 1e-9 * abs(g["summv2"])
        if g["sumIw2"] != 0:
            assert abs(out["sumIw2"] - g["sumIw2"]) <= 1e-9 * abs(g["sumIw2"])
    e.close()


def _mixed_box(n, seed=11):
    """the five-component LJ + charge + dipole + quadrupole set on a jittered bcc lattice (configs[4] recipe)"""
    ps0 = inp.read_inp(input_path("VectorizationMultiComponentMultiPotentials.inp"))
    comps = ps0.components
    ncomp = len(comps.components)
    rng = np.random.default_rng(seed)
    n0 = 2 * n ** 3
    rho = 250.0 / 134.266123 ** 3
    L = (n0 / rho) ** (1.0 / 3.0)
    a = L / n
    gpts = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = np.concatenate([gpts + 0.25 * a, gpts + 0.75 * a])
    r = (r + 0.2 * a * rng.uniform(-0.5, 0.5, r.shape)) % L
    q = rng.normal(size=(n0, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    cid = (np.arange(n0) % ncomp).astype(np.int32)
    return comps, np.array([L, L, L]), np.arange(1, n0 + 1, dtype=np.uint64), cid, r, q


def test_mixed_component_set_list_pass_against_oracle_and_reproducible():
    """Five components (25 ordered component pairs sorted inside every wave's pair block) against the pinned oracle; two
    builds + evaluations give bitwise the same forces (the LDS accumulation of a wave is reproducible)."""
    comps, length, ids, cid, r, q = _mixed_box(12)
    rc = 35.0
    ref = Oracle(comps.flat(), rc).forces(r, q, cid, length, True)
    outs = []
    for _ in range(2):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, rc)
        e.set_verlet(2.5)
        e.set_domain(length)
        e.upload(ids, cid, r, np.zeros_like(r), q, np.zeros_like(r))
        e.update()
        u, w = e.forces_list(0, 0.0, want_macro=True)
        assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
        s = _sorted(e)
        outs.append((s["F"], s["M"], u, w))
        e.close()
    F, M, u, w = outs[0]
    assert rel_max(F, ref["F"]) < TOL and rel_max(M, ref["M"]) < TOL
    assert abs(u - ref["upot"]) <= TOL * abs(ref["upot"]) and abs(w - ref["virial"]) <= TOL * abs(ref["virial"])
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]


def _mixture_in_the_ethane_box():
    """three components that can all be integrated, in the reference's equilibrated ethane box: ethane (2CLJ), a single LJ
    centre, and a polar dumbbell (two LJ centres + a point dipole on the axis: torques from every site type pair)"""
    ps = inp.read_inp(input_path(MAN["ethan"]["input"]))
    st = sorted_phase_space(ps)
    eth = ps.components.components[0]
    eps, sig = eth.lj[0][4], eth.lj[0][5]
    rc = MAN["ethan"]["rc"]
    one = inp.make_component(lj=[(0, 0, 0, 0.03, 1.2 * eps, 0.9 * sig, rc, 0)])
    polar = inp.make_component(lj=[(0, 0, -1.5, 0.02, 0.8 * eps, 0.8 * sig, rc, 0), (0, 0, 1.5, 0.02, 0.8 * eps, 0.8 * sig, rc, 0)],
                               dipoles=[(0, 0, 0.4, 0, 0, 1, 0.6)])
    comps = inp.ComponentSet([eth, one, polar], np.array([[1.0, 1.0], [0.95, 1.02], [1.05, 0.98]]), 1e10)
    cid = (np.arange(len(st["ids"])) % 3).astype(np.int32)
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    return comps, ps.length, rc, st["ids"], cid, st["r"], st["v"], q, st["D"]


@pytest.mark.parametrize("which", ["ethane", "mixture", "linear_pair", "lj_rotors"])
def test_multisite_list_loop_equals_per_step_kernels_over_many_rebuilds(which):
    """Several list lifetimes (hot start: a rebuild every few steps) — same trajectory as the search-every-step kernels;
    "mixture": nine ordered component pairs sorted inside every wave's pair block, LJ and dipole sites; "linear_pair": two linear
    LJ-only components (the axis form of the orientation together with the slot map of several components); "lj_rotors": LJ-only
    with a non-linear three-centre frame (the general LJ-only instantiation, two filter trips per iteration, slot map)."""
    comps, length, rc, ids, cid, r, v, q, D = _mixture_in_the_ethane_box()
    if which == "ethane":
        comps = inp.ComponentSet([comps.components[0]], np.zeros((0, 2)), 1e10)
        cid = np.zeros_like(cid)
    elif which in ("linear_pair", "lj_rotors"):
        eth = comps.components[0]
        eps, sig = eth.lj[0][4], eth.lj[0][5]
        short = inp.make_component(lj=[(0, 0, -1.2, 0.012, 1.1 * eps, 0.85 * sig, rc, 0), (0, 0, 1.2, 0.012, 1.1 * eps, 0.85 * sig, rc, 0)])
        frame = inp.make_component(lj=[(0, 0.3, 0, 0.012, 0.9 * eps, 0.95 * sig, rc, 0), (1.2, -0.9, 0, 0.004, 0.3 * eps, 0.5 * sig, rc, 0),
                                       (-1.2, -0.9, 0, 0.004, 0.3 * eps, 0.5 * sig, rc, 0)])
        if which == "linear_pair":
            comps = inp.ComponentSet([eth, short], np.array([[0.97, 1.03]]), 1e10)
            cid = (np.arange(len(cid)) % 2).astype(np.int32)
        else:
            comps = inp.ComponentSet([eth, short, frame], np.array([[0.97, 1.03], [1.04, 0.99], [0.95, 1.01]]), 1e10)
            cid = (np.arange(len(cid)) % 3).astype(np.int32)
    v = v * 3.0  # hot: the fastest molecules cross skin / 2 = 2 within two or three steps
    dt, steps, skin = 0.5, 40, 4.0
    res = {}
    for mode, sk in (("step", None), ("list", skin)):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, rc)
        e.set_verlet(sk)
        e.set_domain(length)
        e.upload(ids, cid, r, v, q, D)
        e.rebin(); e.halo(); e.forces(0)
        out = e.run(dt, steps)
        res[mode] = (_sorted(e), out, e.run_log())
        if sk:
            assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
            assert 3 <= e.get_option("verlet_builds") <= steps // 2, e.get_option("verlet_builds")
        e.close()
    a, b = res["step"], res["list"]
    L = length
    dr = a[0]["r"] - b[0]["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10 * np.max(L)
    for k in ("v", "q", "D"):
        if np.max(np.abs(a[0][k])) > 0:
            assert rel_max(b[0][k], a[0][k]) < 1e-10, k
    assert rel_max(b[0]["F"], a[0]["F"]) < 1e-9 and rel_max(b[0]["M"], a[0]["M"]) < 1e-9
    for k in ("upot", "virial", "summv2", "sumIw2"):
        assert abs(a[1][k] - b[1][k]) <= 1e-10 * abs(a[1][k]), k
    assert np.allclose(a[2][:, :4], b[2][:, :4], rtol=1e-10, atol=0, equal_nan=True)  # (NaN = not computed in that step, in both loops)


@pytest.mark.parametrize("which", ["ethane", "polar"])
def test_fused_rigid_body_list_pass_equals_the_separate_integrator_bitwise(which):
    """ls1hip_run over several list lifetimes with the list pass integrating its own molecules (fuse_integration = 1: the
    epilogue of k_force_ms_list, leapfrog_body.hpp) and with the separate kick + kick + drift pass (0): the same bits — a run
    may switch between the two at any step (the last step of a run is always unfused)."""
    comps, length, rc, ids, cid, r, v, q, D = _mixture_in_the_ethane_box()
    # one component each (the fused pass serves single-component rigid sets): the LJ-only linear instantiation with groups of 64,
    # and the general one (two LJ centres + a dipole: torques from the multipole bodies, groups of 128)
    comps = inp.ComponentSet([comps.components[0 if which == "ethane" else 2]], np.zeros((0, 2)), 1e10)
    cid = np.zeros_like(cid)
    v = v * 3.0
    res = {}
    for fuse in (1, 0):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, rc)
        e.set_verlet(4.0)
        e.set_domain(length)
        e.set_option("fuse_integration", fuse)
        e.upload(ids, cid, r, v, q, D)
        e.rebin(); e.halo(); e.forces(0)
        out = e.run(0.5, 17)
        out2 = e.run(0.5, 13)  # (a second run: starts from the unfused last step of the first)
        assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
        assert e.get_option("verlet_builds") >= 3
        res[fuse] = (_sorted(e), out, out2)
        e.close()
    a, b = res[1], res[0]
    for k in ("r", "v", "q", "D", "F", "M"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for o in (1, 2):
        for k in ("upot", "virial", "summv2", "sumIw2"):
            assert a[o][k] == b[o][k], k


def test_piecewise_fused_rigid_body_loop_equals_ls1hip_run():
    """The calls a driver makes itself — ls1hip_kick_drift, then per step ls1hip_update + ls1hip_forces_list(dt > 0) (the list pass
    integrates the rigid bodies), the last step ls1hip_update + ls1hip_forces_list_kick — leave the state ls1hip_run leaves, bit for
    bit (single-component rigid set; INTEGRATION.md, 'rigid multi-site sets under lists')."""
    comps, length, rc, ids, cid, r, v, q, D = _mixture_in_the_ethane_box()
    comps = inp.ComponentSet([comps.components[0]], np.zeros((0, 2)), 1e10)
    cid = np.zeros_like(cid)
    v = v * 3.0
    dt, steps = 0.5, 19
    res = []
    for piecewise in (True, False):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, rc)
        e.set_verlet(4.0)
        e.set_domain(length)
        e.upload(ids, cid, r, v, q, D)
        e.rebin(); e.halo(); e.forces(0)
        if piecewise:
            assert e.get_option("can_fuse_rigid_lists") == 1
            e.kick_drift(dt)
            for s in range(steps - 1):
                e.update()
                e.forces_list(0, dt)
            e.update()
            e.forces_list_kick(0.5 * dt)
            kin = e.kinetic_sums()
        else:
            out = e.run(dt, steps)
            kin = (out["summv2"], out["sumIw2"])
        assert e.get_option("verlet_builds") >= 3
        res.append((_sorted(e), kin))
        e.close()
    a, b = res
    for k in ("r", "v", "q", "D", "F", "M"):
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1][0] == b[1][0] and a[1][1] == b[1][1]


def test_config3_ethane_10m_replicated_through_the_list_pass():
    """configs[3] at full size (the reference's ethane box replicated 10^3 = 9 826 000 molecules) through the list build and
    the list force pass: every replica reproduces the golden forces / torques of the real reference, U_pot and virial are
    k^3 times the small box's (periodic replication, tests/test_gpu_fullsize.py)."""
    case = MAN["ethan"]
    g = read_golden("ethan")
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    n0, k = len(st["ids"]), 10
    e = engine_mod.DeviceEngine(0)
    e.set_components(ps.components, case["rc"])
    e.set_verlet(2.0)
    e.set_domain(ps.length * k)
    n = n0 * k ** 3
    e.upload(np.arange(1, n + 1, dtype=np.uint64), tile(st["cid"], k), replicate(ps.length, st["r"], k), np.zeros((n, 3)),
             tile(st["q"], k), np.zeros((n, 3)))
    e.update()
    u, w = e.forces_list(0, 0.0, want_macro=True)
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
    stt = e.download_state()
    o = np.argsort(stt["ids"], kind="stable")
    f = e.download_forces()
    big = dict(F=f["F"][o], M=f["M"][o], Vi=np.zeros((n, 3)), upot=u, virial=w)
    small = dict(F=g["recs"]["F"], M=g["recs"]["M"], Vi=np.zeros((n0, 3)), upot=g["upot"], virial=g["virial"])
    check_replicas(big, small, n0, k, np.random.default_rng(5))
    e.close()


def test_config4_mixed_10m_replicated_through_the_list_pass():
    """configs[4] at full size through the round-4 list kernels (slot map of the five components, cutoff filter + queue): the box
    of test_mixed_component_set_list_pass_against_oracle (2*12^3 molecules, checked against the pinned oracle here once more)
    replicated 14^3 times = 9 483 264 molecules — every replica feels the small box's forces and torques, U_pot and virial are
    k^3 times the small box's (tests/test_gpu_fullsize.py: periodic replication)."""
    comps, length, ids, cid, r, q = _mixed_box(12)
    rc = 35.0
    ref = Oracle(comps.flat(), rc).forces(r, q, cid, length, True)
    n0, k = len(ids), 14
    n = n0 * k ** 3
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, rc)
    e.set_verlet(2.5)
    e.set_domain(length * k)
    e.upload(np.arange(1, n + 1, dtype=np.uint64), tile(cid, k), replicate(length, r, k), np.zeros((n, 3)), tile(q, k), np.zeros((n, 3)))
    e.update()
    u, w = e.forces_list(0, 0.0, want_macro=True)
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST
    stt = e.download_state()
    o = np.argsort(stt["ids"], kind="stable")
    f = e.download_forces()
    big = dict(F=f["F"][o], M=f["M"][o], Vi=np.zeros((n, 3)), upot=u, virial=w)
    small = dict(F=ref["F"], M=ref["M"], Vi=np.zeros((n0, 3)), upot=ref["upot"], virial=ref["virial"])
    check_replicas(big, small, n0, k, np.random.default_rng(7))
    e.close()


def test_component_wise_thermostat_entry_points():
    """Several thermostats (Domain::severalThermostats; integrators/Leapfrog.cpp:84-104, thermostats/VelocityScalingThermostat.cpp:45-69):
    ls1hip_kinetic_sums_by_component == the sums of the downloaded state per component (sum m v^2, sum I w^2 through rotateinv and
    I^-1 as FullMolecule::upd_postF, N, rotational DOF), and ls1hip_scale_kick_drift_components == scaling every molecule by its
    component's factors followed by ls1hip_kick_drift (bitwise: same arithmetic, one pass).  (The XML driver of this reference version
    cannot assign thermostats to components — Simulation.cpp:484-487 always finds thermostat 0 — so the LeapfrogHip branch that
    calls these is exercised here, at the C ABI.)"""
    comps, length, rc, ids, cid, r, v, q, D = _mixture_in_the_ethane_box()
    rng = np.random.default_rng(4)
    D = rng.normal(0, 0.01, r.shape)
    ncomp = len(comps.components)
    res = []
    for fused in (True, False):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, rc)
        e.set_domain(length)
        e.upload(ids, cid, r, v, q, D)
        e.rebin(); e.halo(); e.forces(0)
        e.kick(0.25)
        if fused:
            s = e.kinetic_sums_by_component(ncomp)
            st = e.download_state()
            o = np.argsort(st["ids"], kind="stable")
            vv, DD, qq, cc = st["v"][o], st["D"][o], st["q"][o], st["cid"][o]
            for k, c in enumerate(comps.components):
                m = cc == k
                assert s["n"][k] == m.sum() and s["rot_dof"][k] == m.sum() * c.rot_dof
                mv2 = c.mass * (vv[m] ** 2).sum()
                assert abs(s["summv2"][k] - mv2) <= 1e-12 * mv2
                # w = I^-1 R^T D  (Quaternion::rotateinv), sum I w^2
                w0, x, y, z = qq[m].T
                R = np.array([[w0*w0+x*x-y*y-z*z, 2*(x*y-w0*z), 2*(w0*y+x*z)], [2*(w0*z+x*y), w0*w0-x*x+y*y-z*z, 2*(y*z-w0*x)],
                              [2*(x*z-w0*y), 2*(w0*x+y*z), w0*w0-x*x-y*y+z*z]])
                Db = np.einsum("jin,nj->ni", R, DD[m])
                Iw2 = sum((Db[:, d] ** 2 / c.I[d]).sum() for d in range(3) if c.I[d] > 0)
                assert abs(s["sumIw2"][k] - Iw2) <= 1e-11 * max(Iw2, 1e-300) + 1e-300
            e.scale_kick_drift_components([0.97, 1.02, 1.05], [1.01, 1.0, 0.96], 0.5)
        else:
            # the same by hand: scale per component on the host, then the plain kick + drift
            st = e.download_state()
            f = e.download_forces()
            bt = np.array([0.97, 1.02, 1.05])[st["cid"]][:, None]
            br = np.array([1.01, 1.0, 0.96])[st["cid"]][:, None]
            e2 = engine_mod.DeviceEngine(0)
            e2.set_components(comps, rc)
            e2.set_domain(length)
            e2.upload(st["ids"], st["cid"], st["r"], st["v"] * bt, st["q"], st["D"] * br)
            e2.rebin(); e2.halo(); e2.forces(0)
            e2.kick_drift(0.5)
            e.close()
            e = e2
        st = e.download_state()
        o = np.argsort(st["ids"], kind="stable")
        res.append({k: st[k][o] for k in ("r", "v", "q", "D")})
        e.close()
    for k in ("r", "v", "q", "D"):
        assert rel_max(res[0][k], res[1][k]) < 1e-13, k


def test_folded_post_force_kick_of_multisite_sets():
    """ls1hip_forces_list_kick serves the single-centre LJ list pass and the pair-stream pass of ONE rigid component (round 4: the
    epilogue does upd_postF and the kinetic sums — same state and sums as ls1hip_forces_list + ls1hip_kick, leapfrog_body.hpp); a set of
    several components answers with an error (and says so in the read-only option), the caller falls back to the two calls — as
    LinkedCellsHip does."""
    comps, length, rc, ids, cid, r, v, q, D = _mixture_in_the_ethane_box()
    # several components: refused
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, rc)
    e.set_verlet(4.0)
    e.set_domain(length)
    e.upload(ids, cid, r, v, q, D)
    assert e.update() is True
    assert e.get_option("list_kick_available") == 0
    with pytest.raises(capi.Ls1HipError):
        e.forces_list_kick(0.25)
    e.forces_list(0, 0.0)  # the context is still usable
    e.kick(0.25)
    e.close()
    # one rigid component (ethane: LJ-only, groups of 64; the polar dumbbell: multipole body, groups of 128): folded = separate
    for which in (0, 2):
        one = inp.ComponentSet([comps.components[which]], np.zeros((0, 2)), 1e10)
        res = []
        for folded in (True, False):
            e = engine_mod.DeviceEngine(0)
            e.set_components(one, rc)
            e.set_verlet(4.0)
            e.set_domain(length)
            e.upload(ids, np.zeros_like(cid), r, v, q, D)
            assert e.update() is True
            assert e.get_option("list_kick_available") == 1
            if folded:
                e.forces_list_kick(0.25)
                sums = e.kinetic_sums()
            else:
                e.forces_list(0, 0.0)
                sums = e.kick(0.25)
            res.append((_sorted(e), sums))
            e.close()
        a, b = res
        for k in ("v", "D", "F", "M"):
            assert np.array_equal(a[0][k], b[0][k]), (which, k)
        # (sum m v^2, sum I w^2, N, rotational DOF): the sums are reduced over groups instead of blocks of 256 — equal to rounding
        assert a[1][2:] == b[1][2:] and a[1][2] == len(ids)
        assert abs(a[1][0] - b[1][0]) <= 1e-12 * b[1][0] and abs(a[1][1] - b[1][1]) <= 1e-12 * b[1][1]
