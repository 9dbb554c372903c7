"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  (1) golden vectors produced by the REAL reference VectorizedCellProcessor (tests/golden/*.bin),
  (2) the oracle (oracle/ls1_oracle.c) on the same inputs,
  (3) size-independent properties at larger N (Newton's third law, permutation invariance, inner/outer split).
Tolerance: north_star demands <= 1e-10 relative (FP64); metric max|a-b| / max|b| per quantity.
"""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import FORCE_FLOOR, input_path, manifest, read_golden, rel_max, sorted_phase_space
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
mirror = load_pkg("mirror")
capi = load_pkg("capi")
MAN = manifest()
FORCE_CASES = [k for k, c in MAN.items() if c["steps"] == 0 and not c["legacy"]]
STEP_CASES = [k for k, c in MAN.items() if c["steps"] > 0 and not c["componentwise"]]  # (component-wise thermostats: test_gpu_rotors.py)
TOL = 1e-10


def make_container(ps, rc, periodic, **kw):
    c = mirror.LinkedCells(np.zeros(3), ps.length, rc, components=ps.components, periodic=periodic, **kw)
    return c


def by_id(d, ids):
    o = np.argsort(ids, kind="stable")
    return {k: (v[o] if v is not None else None) for k, v in d.items()}


def run_forces(ps, st, rc, periodic, kernel=0, vi=True, cic=1, split=2):
    cont = make_container(ps, rc, periodic, cellsInCutoffRadius=cic)
    cont.engine.set_option("force_kernel", kernel)
    cont.engine.set_option("lj_split", split)
    cont.engine.set_option("compute_vi", 1 if vi else 0)
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, rc, rc)
    dd = mirror.DomainDecompBase()
    cont.addParticles(st["ids"], st["cid"], st["r"], st["v"], st["q"], st["D"])
    cont.update()
    dd.balanceAndExchange(0.0, False, cont, dom)
    cont.updateMoleculeCaches()
    cont.traverseCells(cp)
    mol = cont.molecules()
    frc = cont.forces(with_vi=vi)
    out = by_id(frc, mol["ids"])
    out["ids"] = np.sort(mol["ids"])
    out["upot"] = dom.getLocalUpot()
    out["virial"] = dom.getLocalVirial()
    out["container"] = cont
    out["kernel_family"] = cont.engine.get_option("last_force_kernel")
    if kernel == capi.FK_LDS_LIST and not vi:
        assert out["kernel_family"] == 2, "the LJ fast path was requested but the generic kernel ran"
    return out


@pytest.mark.parametrize("name", FORCE_CASES)
def test_forces_match_reference_golden(name):
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    out = run_forces(ps, st, case["rc"], bool(case["periodic"]))
    rec = g["recs"]
    assert np.array_equal(out["ids"], rec["id"])
    fl = FORCE_FLOOR.get(name, 0.0)
    assert rel_max(out["F"], rec["F"], fl) < TOL
    assert rel_max(out["M"], rec["M"]) < TOL
    assert rel_max(out["Vi"], rec["Vi"], fl) < TOL
    assert abs(out["upot"] - g["upot"]) <= TOL * max(abs(g["upot"]), 1e-300) or abs(out["upot"] - g["upot"]) < 1e-12
    assert abs(out["virial"] - g["virial"]) <= TOL * max(abs(g["virial"]), 1e-300) or abs(out["virial"] - g["virial"]) < 1e-9


@pytest.mark.parametrize("name", ["lj_periodic", "multi_periodic", "bcc1clj_3456", "ethan"])
def test_forces_match_oracle(name):
    case = MAN[name]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    orc = Oracle(ps.components.flat(), case["rc"])
    ref = orc.forces(st["r"], st["q"], st["cid"], ps.length, True)
    out = run_forces(ps, st, case["rc"], True)
    for k in ("F", "M", "Vi"):
        assert rel_max(out[k], ref[k]) < TOL, k
    assert abs(out["upot"] - ref["upot"]) <= TOL * abs(ref["upot"])
    assert abs(out["virial"] - ref["virial"]) <= TOL * abs(ref["virial"])


def test_lj_parameter_table_matches_oracle():
    ps = inp.read_inp(input_path(MAN["multi"]["input"]))
    cont = make_container(ps, 35.0, False)
    e, s, sh = cont.engine.lj_table()
    oe, os_, osh = Oracle(ps.components.flat(), 35.0).lj_table()
    assert np.array_equal(e, oe) and np.array_equal(s, os_) and np.array_equal(sh, osh)


@pytest.mark.parametrize("name", ["bcc1clj_3456", "lj_periodic"])
def test_cells_in_cutoff_2_gives_same_forces(name):
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    out = run_forces(ps, st, case["rc"], True, cic=2)
    assert rel_max(out["F"], g["recs"]["F"]) < TOL
    assert abs(out["upot"] - g["upot"]) <= TOL * abs(g["upot"])


@pytest.mark.parametrize("name", STEP_CASES)
def test_trajectory_matches_reference(name):
    """Leapfrog + rebin + halo + forces for several steps through the mirrored Simulation loop."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    cont = make_container(ps, case["rc"], True)
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, case["rc"], case["rc"])
    dd = mirror.DomainDecompBase()
    integ = mirror.Leapfrog(case["dt"])
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    cont.addParticles(st["ids"], st["cid"], st["r"], st["v"], q, st["D"])
    thermo = None
    if case["nvt"]:
        dom.setGlobalTemperature(ps.temperature)
        thermo = mirror.VelocityScalingThermostat()
    mirror.simulate(cont, dd, cp, integ, dom, case["steps"], thermostat=thermo)
    mol = cont.molecules()
    o = np.argsort(mol["ids"], kind="stable")
    rec = g["recs"]
    L = ps.length
    dr = mol["r"][o] - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(mol["v"][o], rec["v"]) < 1e-9
    assert rel_max(mol["q"][o], rec["q"]) < 1e-9
    if np.max(np.abs(rec["D"])) > 0:
        assert rel_max(mol["D"][o], rec["D"]) < 1e-9
    F = cont.forces()["F"][o]
    assert rel_max(F, rec["F"]) < 1e-8
    assert abs(dom.getLocalUpot() - g["upot"]) / abs(g["upot"]) < 1e-9
    assert abs(dom.getLocalVirial() - g["virial"]) / abs(g["virial"]) < 1e-8
    bt2 = dom.getGlobalBetaTrans() ** 2 if case["nvt"] else 1.0  # golden sums are taken after the last scaling
    br2 = dom.getGlobalBetaRot() ** 2 if case["nvt"] else 1.0
    assert abs(dom.getLocalSummv2() * bt2 - g["summv2"]) / abs(g["summv2"]) < 1e-9
    if g["sumIw2"] != 0:
        assert abs(dom.getLocalSumIw2() * br2 - g["sumIw2"]) / abs(g["sumIw2"]) < 1e-9


@pytest.mark.parametrize("name", [k for k, c in MAN.items() if c["nvt"] and not c["componentwise"]])
def test_device_nvt_loop_matches_reference(name):
    """ls1hip_run with the on-device global velocity-scaling thermostat (no host round trips) vs the reference."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    cont = make_container(ps, case["rc"], True)
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, case["rc"], case["rc"])
    q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
    cont.addParticles(st["ids"], st["cid"], st["r"], st["v"], q, st["D"])
    mirror.simulate(cont, mirror.DomainDecompBase(), cp, mirror.Leapfrog(case["dt"]), dom, 0)
    cont.engine.set_thermostat(True, ps.temperature)
    out = cont.engine.run(case["dt"], case["steps"])
    mol = cont.molecules()
    o = np.argsort(mol["ids"], kind="stable")
    rec = g["recs"]
    assert rel_max(mol["v"][o], rec["v"]) < 1e-9
    dr = mol["r"][o] - rec["r"]
    dr -= ps.length * np.round(dr / ps.length)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(ps.length)
    if np.max(np.abs(rec["D"])) > 0:
        assert rel_max(mol["D"][o], rec["D"]) < 1e-9
    assert abs(out["upot"] - g["upot"]) / abs(g["upot"]) < 1e-9


def test_device_loop_equals_piecewise_calls():
    """ls1hip_run (no host round trips) == the piecewise mirrored loop, bit for bit (deterministic order)."""
    case = MAN["bcc1clj_3456_steps10"]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    res = []
    for mode in (0, 1):
        cont = make_container(ps, case["rc"], True)
        dom = mirror.Domain(ps.length)
        cp = mirror.VectorizedCellProcessor(dom, case["rc"], case["rc"])
        dd = mirror.DomainDecompBase()
        integ = mirror.Leapfrog(case["dt"])
        cont.addParticles(st["ids"], st["cid"], st["r"], st["v"], st["q"], st["D"])
        if mode == 0:
            mirror.simulate(cont, dd, cp, integ, dom, 10)
            upot = dom.getLocalUpot()
        else:
            mirror.simulate(cont, dd, cp, integ, dom, 0)
            upot = cont.engine.run(case["dt"], 10)["upot"]
        mol = cont.molecules()
        o = np.argsort(mol["ids"])
        res.append((mol["r"][o], mol["v"][o], upot))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("cic,split", [(1, 0), (1, 2), (1, 6), (2, 0)])
def test_fused_force_integration_is_bitwise_the_unfused_loop(cic, split):
    """SURVEY 8f-4 (reduced-memory mode): ls1hip_run with the force pass doing kick + kick + drift between steps
    (ls1hip_forces_kick_drift) == the same run with separate integrator passes, bit for bit: positions, velocities,
    forces of the last step, U_pot, virial, kinetic sum.  Also the piecewise fused calls incl. the inner/boundary
    split and the state rules (no F after a fused pass, download_state sees the advanced positions)."""
    L, r, v = _bcc(24, seed=3)
    N = len(r)
    comp = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    engine_mod = load_pkg("engine")
    dt, nsteps = 0.002, 7
    out = {}
    for mode in ("unfused", "fused", "piecewise"):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comp, 2.5)
        e.set_option("cells_in_cutoff", cic)
        e.set_option("lj_split", split)
        e.set_option("fuse_integration", 0 if mode == "unfused" else 1)
        e.set_option("overlap_halo", {"unfused": 0, "fused": 1 + (cic + split) % 2, "piecewise": 0}[mode])  # all halo modes
        e.set_domain([L, L, L])
        assert e.get_option("can_fuse_integration") == 1
        e.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v)
        e.rebin(); e.halo(); e.forces(0)
        if mode != "piecewise":
            res = e.run(dt, nsteps)
        else:
            e.kick_drift(dt)
            for s in range(nsteps):
                e.rebin(); e.halo()
                if s + 1 < nsteps:
                    if s % 2:
                        e.forces_kick_drift(0, dt)
                    else:  # overlap split
                        e.forces_kick_drift(1, dt)
                        with pytest.raises(capi.Ls1HipError):
                            e.forces(2)  # a fused inner pass must be completed by a fused boundary pass
                        e.forces_kick_drift(2, dt)
                    with pytest.raises(capi.Ls1HipError):
                        e.download_forces()  # F was consumed
                    with pytest.raises(capi.Ls1HipError):
                        e.kick_drift(dt)  # already advanced
                    if s == 2:
                        st_mid = e.download_state()  # materialises the advanced positions; the loop must go on unharmed
                        assert np.all(np.isfinite(st_mid["r"]))
                else:
                    u, w = e.forces(0)
            k = e.kick(0.5 * dt)
            res = dict(upot=u, virial=w, summv2=k[0])
        st = e.download_state()
        o = np.argsort(st["ids"])
        out[mode] = (st["r"][o], st["v"][o], e.download_forces()["F"][o], res["upot"], res["virial"], res["summv2"])
        e.close()
    for mode in ("fused", "piecewise"):
        for a, b in zip(out["unfused"][:3], out[mode][:3]):
            assert np.array_equal(a, b), mode
        # macroscopic sums: the which=1/2 split adds partial sums in a different order (rounding only)
        tol = 1e-13  # split passes add the partial sums in a different order (rounding only)
        for a, b in zip(out["unfused"][3:], out[mode][3:]):
            assert abs(a - b) <= tol * abs(a), mode


def test_inner_outer_split_equals_full_traversal():
    """traversePartialInnermostCells + traverseNonInnermostCells == traverseCells (LinkedCellsTest.cpp:239-280)."""
    case = MAN["bcc1clj_16000"]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    full = run_forces(ps, st, case["rc"], True, vi=False)
    cont = full["container"]
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, case["rc"], case["rc"])
    cont.traversePartialInnermostCells(cp, 0, 1)
    cont.traverseNonInnermostCells(cp)
    mol = cont.molecules()
    F = cont.forces()["F"][np.argsort(mol["ids"])]
    assert np.array_equal(F, full["F"])
    assert abs(dom.getLocalUpot() - full["upot"]) <= 1e-13 * abs(full["upot"])
    assert abs(dom.getLocalVirial() - full["virial"]) <= 1e-13 * abs(full["virial"])


def _bcc(n_per_dim, seed=0, rho=0.785302672, jitter=0.1):
    rng = np.random.default_rng(seed)
    n = n_per_dim
    N = 2 * n ** 3
    L = (N / rho) ** (1.0 / 3.0)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = np.concatenate([g + 0.25 * a, g + 0.75 * a])
    r = (r + jitter * rng.uniform(-0.5, 0.5, r.shape)) % L
    v = rng.normal(0, 1, r.shape)
    v -= v.mean(0)
    return L, r, v


def test_large_box_properties():
    """N = 2*50^3 = 250 000 (too big for the scalar oracle in seconds): size-independent properties —
    sum F = 0 (Newton 3 through the periodic images), results invariant under permutation of the input order,
    virial/upot consistent between cells_in_cutoff 1 and 2."""
    L, r, v = _bcc(50)
    N = len(r)
    comp = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    ps = inp.PhaseSpace(comp, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32),
                        r, v, np.tile([1., 0, 0, 0], (N, 1)), np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    a = run_forces(ps, st, 2.5, True, vi=False)
    Fmax = np.max(np.abs(a["F"]))
    assert np.max(np.abs(a["F"].sum(0))) < 1e-9 * Fmax * np.sqrt(N)
    perm = np.random.default_rng(3).permutation(N)
    st2 = {k: v_[perm] for k, v_ in st.items()}
    b = run_forces(ps, st2, 2.5, True, vi=False)
    assert np.array_equal(a["F"], b["F"])  # canonical in-cell order => bitwise identical
    assert a["upot"] == b["upot"]
    c = run_forces(ps, st, 2.5, True, vi=False, cic=2)
    assert rel_max(c["F"], a["F"]) < 1e-12
    assert abs(c["upot"] - a["upot"]) < 1e-12 * abs(a["upot"])
    assert abs(c["virial"] - a["virial"]) < 1e-11 * abs(a["virial"])


def test_full_size_properties_bench_workload():
    """BASELINE configs[1] at FULL size (N = 2*171^3 = 10 000 422, the bench.py workload), through the C ABI:
    size-independent properties instead of the oracle —
      * sum F = 0 (Newton 3 through the periodic images) for the shipped MFMA kernel,
      * MFMA kernel == list kernel == generic kernel on the same state (forces 1e-13, U_pot / virial 1e-12),
      * inner + boundary traversal == full traversal bit for bit,
      * 5 NVE steps: total momentum conserved, energy fluctuation |dE|/|E| < 2e-5 (leapfrog, dt = 0.002, jittered-lattice start),
      * every molecule survives re-binning (ids are a permutation of the input)."""
    engine_mod = load_pkg("engine")
    L, r, v = _bcc(171, seed=1)
    N = len(r)
    assert N == 10000422
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    eng = engine_mod.DeviceEngine(0)
    eng.set_components(comps, 2.5)
    eng.set_domain([L, L, L])
    eng.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v)
    eng.rebin(); eng.halo()
    res = {}
    for name, fk, split in (("mfma", capi.FK_LDS_LIST, 4), ("list", capi.FK_LDS_LIST, 2), ("generic", capi.FK_GENERIC, 0)):
        eng.set_option("force_kernel", fk)
        eng.set_option("lj_split", split)
        u, w = eng.forces(0)
        res[name] = (eng.download_forces()["F"], u, w)
    F, u, w = res["mfma"]
    Fmax = np.max(np.abs(F))
    assert np.max(np.abs(F.sum(0))) < 1e-9 * Fmax * np.sqrt(N)
    for other in ("list", "generic"):
        Fo, uo, wo = res[other]
        assert np.max(np.abs(F - Fo)) < 1e-13 * Fmax
        assert abs(u - uo) < 1e-12 * abs(uo) and abs(w - wo) < 1e-12 * abs(wo)
    del res
    eng.set_option("force_kernel", capi.FK_AUTO)
    eng.set_option("lj_split", 0)
    eng.forces(1, want_macro=False)
    u2, w2 = eng.forces(2)
    assert np.array_equal(eng.download_forces()["F"], F)
    assert abs(u2 - u) <= 1e-13 * abs(u)
    # short NVE run from this state
    ekin0 = 0.5 * float((v * v).sum())
    out = eng.run(0.002, 5)
    st = eng.download_state()
    assert np.array_equal(np.sort(st["ids"]), np.arange(1, N + 1, dtype=np.uint64))
    p = st["v"].sum(0)
    assert np.max(np.abs(p)) < 1e-9 * np.sqrt(N)
    e0 = ekin0 + u
    e1 = 0.5 * out["summv2"] + out["upot"]
    assert abs(e1 - e0) < 2e-5 * abs(e0)
    eng.close()


def test_error_conventions():
    ps = inp.read_inp(input_path(MAN["U0"]["input"]))
    st = sorted_phase_space(ps)
    cont = make_container(ps, 1.1, True)
    with pytest.raises(capi.Ls1HipError):  # forces before rebin
        cont.engine.forces(0)
    bad = st["r"].copy()
    bad[0, 0] = -1.0
    with pytest.raises(capi.Ls1HipError):  # molecule outside the bounding box
        cont.addParticles(st["ids"], st["cid"], bad, st["v"])
    with pytest.raises(capi.Ls1HipError):  # region too small for the cutoff (reference: exit(1))
        mirror.LinkedCells(np.zeros(3), ps.length, 5.0, components=ps.components)
    # empty container: every step is a no-op, not a crash
    cont.addParticles(np.zeros(0, np.uint64), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros((0, 3)))
    cont.update()
    cont.engine.halo()
    assert cont.engine.forces(0) == (0.0, 0.0)


def test_seam_a_soa_forces_matches_reference():
    """CellProcessor-level seam: flat cell-major molecule list incl. halo copies, as the reference's LinkedCells
    holds it after balanceAndExchange; compare with golden (periodic LJ cluster)."""
    case = MAN["lj_periodic"]
    g = read_golden("lj_periodic")
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    rc = case["rc"]
    L = ps.length
    # build the reference's cell structure on the host (plain numpy): cells = floor(L/rc)+2, halo copies within rc
    box = np.floor(L / np.float32(rc)).astype(int)
    dims = box + 2
    clen = L / box
    r, q, cid = [st["r"]], [st["q"]], [st["cid"]]
    src = [np.arange(len(st["r"]))]
    for d in range(3):
        allr = np.concatenate(r); allq = np.concatenate(q); allc = np.concatenate(cid); alls = np.concatenate(src)
        for sign, lo, hi in ((+1, 0.0, rc), (-1, L[d] - rc, L[d])):
            m = (allr[:, d] >= lo) & (allr[:, d] < hi)
            rr = allr[m].copy()
            rr[:, d] += sign * L[d]
            r.append(rr); q.append(allq[m]); cid.append(allc[m]); src.append(alls[m])
    r = np.concatenate(r); q = np.concatenate(q); cid = np.concatenate(cid); src = np.concatenate(src)
    ci = np.floor(r / clen).astype(int) + 1
    ci = np.clip(ci, 0, dims - 1)
    lin = (ci[:, 2] * dims[1] + ci[:, 1]) * dims[0] + ci[:, 0]
    order = np.argsort(lin, kind="stable")
    ncells = int(np.prod(dims))
    cell_start = np.zeros(ncells + 1, dtype=np.uint32)
    np.add.at(cell_start, lin + 1, 1)
    cell_start = np.cumsum(cell_start).astype(np.uint32)
    eng = make_container(ps, rc, True).engine
    out = eng.soa_forces(dims, cell_start, r[order], q[order], cid[order])
    n = len(st["r"])
    real = src[order][np.isin(np.arange(len(order)), np.nonzero(order < n)[0])]
    F = np.zeros((n, 3)); M = np.zeros((n, 3))
    sel = order < n
    F[order[sel]] = out["F"][sel]
    M[order[sel]] = out["M"][sel]
    assert rel_max(F, g["recs"]["F"]) < TOL
    assert rel_max(M, g["recs"]["M"]) < TOL
    assert abs(out["upot"] - g["upot"]) <= TOL * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= TOL * abs(g["virial"])


# ---- the LDS-tiled single-centre LJ kernel (kernels_force_lj.hip) ------------------------------------------------
LJ1_CASES = ["U0", "F0", "U0_periodic", "lj1clj", "bcc1clj_3456", "bcc1clj_16000", "bcc1clj_8192"]


@pytest.mark.parametrize("split", [0, 1, 2, 4, 5, 6])
@pytest.mark.parametrize("cic", [1, 2])
@pytest.mark.parametrize("name", LJ1_CASES)
def test_lds_kernel_matches_reference_golden(name, cic, split):
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    if cic == 2 and np.min(ps.length) < 2 * case["rc"] * 0.5 * 2:
        pass  # tiny boxes still work with half-size cells (at least one cell per dimension)
    out = run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_LDS_LIST, vi=False, cic=cic, split=split)
    assert out["container"].engine.get_option("force_kernel") == capi.FK_LDS_LIST
    rec = g["recs"]
    fl = FORCE_FLOOR.get(name, 0.0)
    assert rel_max(out["F"], rec["F"], fl) < TOL
    assert abs(out["upot"] - g["upot"]) <= TOL * max(abs(g["upot"]), 1e-300) or abs(out["upot"] - g["upot"]) < 1e-12
    assert abs(out["virial"] - g["virial"]) <= TOL * max(abs(g["virial"]), 1e-300) or abs(out["virial"] - g["virial"]) < 1e-9


@pytest.mark.parametrize("n", [24, 50])
def test_lds_kernel_equals_generic_kernel_large_box(n):
    """N = 250 000 (27 cells per dimension) and N = 27 648 (13 cells per dimension: the last brick row of the 4-cell
    pencil bricks holds ONE cell row — the partial-brick case in which a candidate tile that overruns its range would
    reach the next plane's neighbour rows): LDS kernels vs the generic kernel (same inputs): forces to 1e-13, sums to
    1e-12; also the inner/boundary split and run-to-run bitwise reproducibility of the LDS kernels."""
    L, r, v = _bcc(n, seed=7)
    N = len(r)
    comp = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    ps = inp.PhaseSpace(comp, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32),
                        r, v, np.tile([1., 0, 0, 0], (N, 1)), np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, 2.5, True, kernel=capi.FK_GENERIC, vi=False)
    for cic, split in ((1, 1), (1, 2), (1, 4), (1, 5), (1, 6), (1, 0), (2, 1), (2, 2)):
        lds = run_forces(ps, st, 2.5, True, kernel=capi.FK_LDS_LIST, vi=False, cic=cic, split=split)
        assert rel_max(lds["F"], gen["F"]) < 1e-13
        assert abs(lds["upot"] - gen["upot"]) < 1e-12 * abs(gen["upot"])
        assert abs(lds["virial"] - gen["virial"]) < 1e-12 * abs(gen["virial"])
        again = run_forces(ps, st, 2.5, True, kernel=capi.FK_LDS_LIST, vi=False, cic=cic, split=split)
        assert np.array_equal(again["F"], lds["F"]) and again["upot"] == lds["upot"]
        cont = lds["container"]
        dom = mirror.Domain(ps.length)
        cp = mirror.VectorizedCellProcessor(dom, 2.5, 2.5)
        cont.traversePartialInnermostCells(cp, 0, 1)
        cont.traverseNonInnermostCells(cp)
        mol = cont.molecules()
        F = cont.forces()["F"][np.argsort(mol["ids"])]
        assert np.array_equal(F, lds["F"])
        assert abs(dom.getLocalUpot() - lds["upot"]) <= 1e-13 * abs(lds["upot"])


def test_lds_kernel_dense_cluster_fallbacks():
    """Pathological density: 3000 atoms in a few cells (list overflow + shell larger than the LDS staging area)
    must still match the generic kernel."""
    rng = np.random.default_rng(9)
    L = 12.0
    r = np.concatenate([rng.uniform(3.0, 6.0, (3000, 3)), rng.uniform(0, L, (500, 3))])
    N = len(r)
    comp = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1e-3, 0.3, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    ps = inp.PhaseSpace(comp, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32),
                        r, np.zeros((N, 3)), np.tile([1., 0, 0, 0], (N, 1)), np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, 2.5, True, kernel=capi.FK_GENERIC, vi=False)
    for split in (0, 1, 2, 4, 5, 6):
        lds = run_forces(ps, st, 2.5, True, kernel=capi.FK_LDS_LIST, vi=False, split=split)
        assert rel_max(lds["F"], gen["F"]) < 1e-12
        assert abs(lds["upot"] - gen["upot"]) < 1e-12 * abs(gen["upot"])


MULTISITE_CASES = [k for k in FORCE_CASES if k not in LJ1_CASES]


@pytest.mark.parametrize("name", MULTISITE_CASES)
def test_multisite_brick_kernel_is_bitwise_the_generic_kernel(name):
    """k_force_ms_brick (LDS-staged, default for multi-site component sets) visits the candidates in the order of
    k_force_generic (whatever lane a molecule is handed to): F, M, Vi bitwise equal, sums to rounding (different
    partial-sum grouping)."""
    case = MAN[name]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_GENERIC)
    brk = run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_MS_BRICK)
    assert gen["kernel_family"] == 1 and brk["kernel_family"] == 3  # generic vs multi-site brick kernel, really
    for k in ("F", "M", "Vi"):  # candidates are visited in the generic kernel's order: bitwise
        assert np.array_equal(gen[k], brk[k]), k
    assert abs(gen["upot"] - brk["upot"]) <= 1e-13 * max(abs(gen["upot"]), 1e-300) + 1e-300
    assert abs(gen["virial"] - brk["virial"]) <= 1e-13 * max(abs(gen["virial"]), 1e-300) + 1e-300


def _close_to_generic(gen, out, tol=1e-12, what=None):
    """site kernel vs generic kernel: different summation order, FMA, Newton-refined reciprocals -> 1e-12 of the largest
    component (forces, torques, per-molecule virials) and of the sums"""
    for k in ("F", "M", "Vi"):
        if k in gen and gen[k] is not None and np.max(np.abs(gen[k])) > 0:
            assert rel_max(out[k], gen[k]) < tol, (k, what)
    assert abs(gen["upot"] - out["upot"]) <= tol * max(abs(gen["upot"]), 1e-300) + 1e-300, what
    assert abs(gen["virial"] - out["virial"]) <= tol * max(abs(gen["virial"]), 1e-300) + 1e-300, what


@pytest.mark.parametrize("name", MULTISITE_CASES)
def test_multisite_site_kernel_matches_the_generic_kernel(name):
    """k_force_sites (LDS tables, cached own sites, FMA, v_rcp / v_rsq + Newton; LS1HIP_FK_MS_SITES, on request) against
    k_force_generic on the reference's fixtures: 1e-12; FK_AUTO runs the molecule-pair brick kernel."""
    case = MAN[name]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_GENERIC)
    sit = run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_MS_SITES)
    assert gen["kernel_family"] == 1 and sit["kernel_family"] == capi.FK_MS_SITES
    _close_to_generic(gen, sit, what=name)
    assert run_forces(ps, st, case["rc"], bool(case["periodic"]), kernel=capi.FK_AUTO)["kernel_family"] == capi.FK_MS_BRICK


def test_multisite_brick_kernel_fallbacks_and_split():
    """Ethane-like 2CLJ molecules: (a) liquid-like box against the oracle incl. the inner/boundary split,
    (b) a dense cluster (list overflow and a shell larger than the staging area) against the generic kernel."""
    ps0 = inp.read_inp(input_path(MAN["ethan"]["input"]))
    comps = ps0.components
    rng = np.random.default_rng(21)
    # (a) 6000 molecules at 3x the fixture density, random orientations
    N = 6000
    L = (N / (3 * 9826 / 571.607759 ** 3)) ** (1 / 3)
    rc = 32.1254
    r = rng.uniform(0, L, (N, 3))
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
    ps = inp.PhaseSpace(comps, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32),
                        r, np.zeros((N, 3)), q, np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, rc, True, kernel=capi.FK_GENERIC)
    brk = run_forces(ps, st, rc, True, kernel=capi.FK_MS_BRICK)
    for k in ("F", "M", "Vi"):
        assert np.array_equal(gen[k], brk[k]), k
    sit = run_forces(ps, st, rc, True, kernel=capi.FK_MS_SITES)
    assert sit["kernel_family"] == capi.FK_MS_SITES
    _close_to_generic(gen, sit, what="liquid-like ethane")
    for which, res in (("brick", brk), ("sites", sit)):
        _inner_boundary_split_equals_full(ps, rc, res)
    # (b) dense cluster: 1500 molecules inside one cutoff sphere (+ 300 spread out)
    N = 1800
    L = 6 * rc
    r = np.concatenate([rng.uniform(2.2 * rc, 3.0 * rc, (1500, 3)), rng.uniform(0, L, (300, 3))])
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
    ps = inp.PhaseSpace(comps, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32),
                        r, np.zeros((N, 3)), q, np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, rc, True, kernel=capi.FK_GENERIC, vi=False)
    brk = run_forces(ps, st, rc, True, kernel=capi.FK_MS_BRICK, vi=False)
    for k in ("F", "M"):
        assert np.array_equal(gen[k], brk[k]), k
    assert abs(gen["upot"] - brk["upot"]) <= 1e-12 * abs(gen["upot"])
    sit = run_forces(ps, st, rc, True, kernel=capi.FK_MS_SITES, vi=False)
    assert sit["kernel_family"] == capi.FK_MS_SITES  # bricks beyond the staging area evaluate from global memory
    _close_to_generic(gen, sit, tol=1e-11, what="dense cluster")


def _inner_boundary_split_equals_full(ps, rc, res):
    cont = res["container"]
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, rc, rc)
    cont.traversePartialInnermostCells(cp, 0, 1)
    cont.traverseNonInnermostCells(cp)
    mol = cont.molecules()
    frc = cont.forces()
    o = np.argsort(mol["ids"])
    assert np.array_equal(frc["F"][o], res["F"]) and np.array_equal(frc["M"][o], res["M"])
    assert abs(dom.getLocalUpot() - res["upot"]) <= 1e-13 * abs(res["upot"])


@pytest.mark.parametrize("name", [k for k in FORCE_CASES])
def test_homogeneous_long_range_correction(name):
    """ls1hip_long_range_homogeneous vs the reference's Homogeneous LRC (golden trailer)."""
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    cont = make_container(ps, case["rc"], bool(case["periodic"]))
    ncomp = len(ps.components.components)
    nmol = [int((ps.cid == k).sum()) for k in range(ncomp)]
    u, v = cont.engine.long_range_homogeneous(nmol, len(ps.ids) / float(np.prod(ps.length)))
    assert abs(u - g["lrc"][0]) <= 1e-12 * max(abs(g["lrc"][0]), abs(g["upot"]))
    assert abs(v - g["lrc"][1]) <= 1e-12 * max(abs(g["lrc"][1]), abs(g["virial"]))


@pytest.mark.parametrize("seed", list(range(24)))
def test_lj_fast_path_random_boxes(seed):
    """Randomised sweep of the 1CLJ fast path against the generic kernel: non-cubic boxes with 3..14 cells per
    dimension (every partial-brick combination of the 1x4x4 / 1x4x2 pencil bricks and of the 4x2x2 list bricks),
    densities 0.2..1.1, cutoff 1.8..3.2, random positions with a minimum distance, all lj_split variants and the fused /
    split passes.  Forces to 1e-12 of the largest force, U_pot / virial to 1e-11."""
    rng = np.random.default_rng(1000 + seed)
    rc = float(rng.uniform(1.8, 3.2))
    ncell = rng.integers(3, 15, 3)
    L = ncell * rc * rng.uniform(1.0, 1.18, 3)  # floor(L / rc) == ncell
    rho = float(rng.uniform(0.2, 1.1))
    N = int(rho * np.prod(L))
    # jittered simple-cubic start keeps a minimum distance at every density
    m = np.ceil((N / np.prod(L)) ** (1 / 3) * L).astype(int)
    g = np.stack(np.meshgrid(*[np.arange(k) for k in m], indexing="ij"), -1).reshape(-1, 3)
    sel = rng.permutation(len(g))[:N]
    a = L / m
    r = ((g[sel] + 0.5) * a + rng.uniform(-0.2, 0.2, (N, 3)) * a) % L
    comp = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1.3, 0.9, rc, int(seed % 2))])], np.zeros((0, 2)), 1e10)
    ps = inp.PhaseSpace(comp, L, np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, np.zeros((N, 3)),
                        np.tile([1., 0, 0, 0], (N, 1)), np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, rc, True, kernel=capi.FK_GENERIC, vi=False)
    fmax = np.max(np.abs(gen["F"]))
    for cic, split in ((1, 0), (1, 4), (1, 6), (1, 2), (2, 0)):
        out = run_forces(ps, st, rc, True, kernel=capi.FK_LDS_LIST, vi=False, cic=cic, split=split)
        assert np.max(np.abs(out["F"] - gen["F"])) < 1e-12 * fmax, (cic, split, ncell.tolist(), rho)
        assert abs(out["upot"] - gen["upot"]) <= 1e-11 * abs(gen["upot"]), (cic, split)
        assert abs(out["virial"] - gen["virial"]) <= 1e-11 * abs(gen["virial"]), (cic, split)
        if cic == 1 and split == 0:  # inner + boundary passes == full pass, bitwise
            cont = out["container"]
            dom = mirror.Domain(ps.length)
            cp = mirror.VectorizedCellProcessor(dom, rc, rc)
            cont.traversePartialInnermostCells(cp, 0, 1)
            cont.traverseNonInnermostCells(cp)
            mol = cont.molecules()
            assert np.array_equal(cont.forces()["F"][np.argsort(mol["ids"])], out["F"])


@pytest.mark.parametrize("seed", list(range(12)))
def test_multisite_brick_random_boxes(seed):
    """Randomised sweep of the multi-site brick kernel against the generic kernel (bitwise F, M, Vi): non-cubic boxes with
    3..12 cells per dimension, 0.5..14 molecules per cell (every brick shape from 8x4x4 to 2x2x2, and the generic
    fallback above that), the five-component multipole set (massless components included: statics only) or ethane,
    random orientations."""
    rng = np.random.default_rng(2000 + seed)
    if seed % 3 == 2:
        comps = inp.read_inp(input_path(MAN["ethan"]["input"])).components
        rc = 32.1254
    else:
        comps = inp.read_inp(input_path("VectorizationMultiComponentMultiPotentials.inp")).components
        rc = 35.0
    ncomp = len(comps.components)
    ncell = rng.integers(3, 13, 3)
    L = ncell * rc * rng.uniform(1.0, 1.08, 3)
    per_cell = float(np.exp(rng.uniform(np.log(0.5), np.log(14.0))))
    N = max(int(per_cell * np.prod(ncell)), 8)
    m = np.ceil((N / np.prod(L)) ** (1 / 3) * L).astype(int)
    g = np.stack(np.meshgrid(*[np.arange(k) for k in m], indexing="ij"), -1).reshape(-1, 3)
    sel = rng.permutation(len(g))[:N]
    N = len(sel)
    a = L / m
    r = ((g[sel] + 0.5) * a + rng.uniform(-0.25, 0.25, (N, 3)) * a) % L
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
    cid = rng.integers(0, ncomp, N).astype(np.int32)
    ps = inp.PhaseSpace(comps, L, np.arange(1, N + 1, dtype=np.uint64), cid, r, np.zeros((N, 3)), q, np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    gen = run_forces(ps, st, rc, True, kernel=capi.FK_GENERIC)
    brk = run_forces(ps, st, rc, True, kernel=capi.FK_MS_BRICK)
    assert gen["kernel_family"] == 1
    assert brk["kernel_family"] == (3 if per_cell * 64 * 1.08 <= 770 else brk["kernel_family"])
    for k in ("F", "M", "Vi"):
        assert np.array_equal(gen[k], brk[k]), (k, ncell.tolist(), per_cell)
    assert abs(gen["upot"] - brk["upot"]) <= 1e-12 * abs(gen["upot"]) + 1e-300
    assert abs(gen["virial"] - brk["virial"]) <= 1e-12 * abs(gen["virial"]) + 1e-300
    # the site kernel (every brick shape / lanes-per-molecule choice its host heuristic makes over this density range)
    sit = run_forces(ps, st, rc, True, kernel=capi.FK_MS_SITES)
    if per_cell < 6:
        assert sit["kernel_family"] == capi.FK_MS_SITES, per_cell
    _close_to_generic(gen, sit, tol=1e-11, what=(ncell.tolist(), per_cell, sit["kernel_family"]))


def test_dense_multisite_liquid_four_lanes_per_molecule():
    """Liquid-density ethane (2CLJ, r_c = 4.9 sigma: ~38 molecules per cell, ~150 neighbours): beyond the brick kernel's
    staging area, so the generic kernel runs, with four lanes per molecule (quarter lists).  Against the oracle, all cells
    and the inner / boundary split."""
    ps0 = inp.read_inp(input_path(MAN["ethan"]["input"]))
    comps, rc = ps0.components, 32.1254
    rng = np.random.default_rng(31)
    n = 13
    N = 2 * n ** 3
    rho = 20 * 9826 / 571.607759 ** 3
    L = (N / rho) ** (1 / 3)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = (np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.1 * a * rng.uniform(-0.5, 0.5, (N, 3))) % L
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
    ps = inp.PhaseSpace(comps, np.array([L, L, L]), np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r,
                        np.zeros((N, 3)), q, np.zeros((N, 3)))
    st = sorted_phase_space(ps)
    ref = Oracle(comps.flat(), rc).forces(st["r"], st["q"], st["cid"], ps.length, True)
    out = run_forces(ps, st, rc, True, kernel=capi.FK_AUTO)
    assert out["kernel_family"] == 1
    for k in ("F", "M", "Vi"):
        assert rel_max(out[k], ref[k]) < TOL, k
    assert abs(out["upot"] - ref["upot"]) <= TOL * abs(ref["upot"])
    assert abs(out["virial"] - ref["virial"]) <= TOL * abs(ref["virial"])
    cont = out["container"]
    dom = mirror.Domain(ps.length)
    cp = mirror.VectorizedCellProcessor(dom, rc, rc)
    cont.traversePartialInnermostCells(cp, 0, 1)
    cont.traverseNonInnermostCells(cp)
    mol = cont.molecules()
    o = np.argsort(mol["ids"])
    frc = cont.forces()
    assert np.array_equal(frc["F"][o], out["F"]) and np.array_equal(frc["M"][o], out["M"])
