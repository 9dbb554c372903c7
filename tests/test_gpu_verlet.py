"""GPU parity tests of the neighbour-list (Verlet) loop of ls1hip_run (ls1hip_set_verlet, kernels_force_verlet.hip):
against the REAL reference's golden trajectory, against the pinned oracle step by step, and against the per-step
kernels over several list lifetimes; plus the rebuild trigger, the list-overflow / unstaged fallbacks and the shifted
potential.  Tolerances: trajectories 1e-9 (as for the per-step kernels), forces of one evaluation 1e-12 of max|F|."""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import input_path, manifest, read_golden, rel_componentwise, rel_max, sorted_phase_space
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
capi = load_pkg("capi")
engine_mod = load_pkg("engine")
synth = load_pkg("synth")
MAN = manifest()


def _lj(rc=2.5, shift=0):
    return inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, shift)])], np.zeros((0, 2)), 1e10)


def _engine(comps, rc, L, ids, r, v, skin=None, **opts):
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, rc)
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_verlet(skin, force=True)
    e.set_domain(L)
    e.upload(ids, np.zeros(len(ids), np.int32), r, v)
    e.rebin(); e.halo(); e.forces(0)
    return e


def _state(e):
    st = e.download_state()
    o = np.argsort(st["ids"], kind="stable")
    return st["ids"][o], st["r"][o], st["v"][o], e.download_forces()["F"][o]


@pytest.mark.parametrize("skin", [0.3, 0.12])
def test_verlet_run_matches_reference_trajectory(skin):
    """10 Leapfrog steps of the REAL reference (bcc1clj_3456_steps10) with the list-reuse loop."""
    name = "bcc1clj_3456_steps10"
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = _engine(ps.components, case["rc"], ps.length, st["ids"], st["r"], st["v"], skin=skin)
    assert e.get_option("verlet_lists") == 1
    out = e.run(case["dt"], case["steps"])
    assert e.get_option("verlet_steps") == case["steps"] and 1 <= e.get_option("verlet_builds") < case["steps"]
    ids, r, v, F = _state(e)
    rec = g["recs"]
    L = ps.length
    assert r.min() >= 0 and np.all(r < L)  # reported wrapped although the lists outlive a periodic crossing
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(v, rec["v"]) < 1e-9
    assert rel_max(F, rec["F"]) < 1e-8 and rel_componentwise(F, rec["F"]) < 1e-7
    assert abs(out["upot"] - g["upot"]) / abs(g["upot"]) < 1e-9
    assert abs(out["virial"] - g["virial"]) / abs(g["virial"]) < 1e-8
    assert abs(out["summv2"] - g["summv2"]) / abs(g["summv2"]) < 1e-9
    # every step's globals against the pinned oracle
    log = e.run_log()
    orc = Oracle(ps.components.flat(), case["rc"])
    ro, vo, q, D, cid = st["r"].copy(), st["v"].copy(), st["q"].copy(), st["D"].copy(), st["cid"]
    o = orc.forces(ro, q, cid, ps.length, True)
    Fo, Mo = o["F"].copy(), o["M"].copy()
    for s in range(case["steps"]):
        o = orc.step(case["dt"], cid, ro, vo, q, D, Fo, Mo, ps.length, True)
        assert abs(log[s, 0] - o["upot"]) <= 1e-9 * abs(o["upot"]), s
        assert abs(log[s, 1] - o["virial"]) <= 1e-8 * abs(o["virial"]), s
        assert abs(log[s, 2] - o["summv2"]) <= 1e-9 * o["summv2"], s
    e.close()


@pytest.mark.parametrize("n,temp,dt,steps", [(24, 0.95, 0.002, 80), (20, 12.0, 0.002, 40)])
def test_verlet_loop_equals_per_step_kernels_over_many_rebuilds(n, temp, dt, steps):
    """Several list lifetimes (and, hot, a rebuild every few steps): same trajectory as the per-step MFMA kernel loop;
    the rebuild count follows the displacement bound (skin / 2 over dt * vmax), not a fixed interval."""
    L, ids, r, v = synth.bcc_box(n, temp=temp)
    res = {}
    for mode, skin in (("step", None), ("list", 0.3)):
        e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=skin)
        out = e.run(dt, steps)
        res[mode] = _state(e) + (out, e.run_log())
        if skin:
            builds = e.get_option("verlet_builds")
            vmax = np.sqrt((v * v).sum(1).max())
            lo = int(steps * dt * vmax * 0.5 / (0.5 * skin))  # ballistic estimate, halved: speeds relax
            assert max(1, lo) <= builds <= steps // 2 + 1, (builds, lo)
        e.close()
    a, b = res["step"], res["list"]
    assert np.array_equal(a[0], b[0])
    dr = a[1] - b[1]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10 * L
    assert rel_max(b[2], a[2]) < 1e-10
    assert rel_max(b[3], a[3]) < 1e-9
    for k in ("upot", "virial", "summv2"):
        assert abs(a[4][k] - b[4][k]) <= 1e-10 * abs(a[4][k]), k
    assert np.allclose(a[5][:, :3], b[5][:, :3], rtol=1e-10, atol=0)


def test_verlet_single_evaluation_forces_and_shifted_potential():
    """One list build + evaluation vs the generic kernel on the same state, unshifted and shifted LJ (the in-range pair
    count that carries the shift is only tallied when the shift is non-zero)."""
    L, ids, r, v = synth.bcc_box(16)
    for shift in (0, 1):
        comps = _lj(2.5, shift)
        ref = _engine(comps, 2.5, [L] * 3, ids, r, v, force_kernel=capi.FK_GENERIC)
        o0 = ref.run(1e-9, 1)
        F0 = _state(ref)[3]
        ref.close()
        e = _engine(comps, 2.5, [L] * 3, ids, r, v, skin=0.25)
        out = e.run(1e-9, 1)  # one (unfused) list build + evaluation
        assert e.get_option("verlet_builds") == 1
        F = _state(e)[3]
        assert rel_max(F, F0) < 1e-12
        assert abs(out["upot"] - o0["upot"]) <= 1e-11 * abs(o0["upot"]) and abs(out["virial"] - o0["virial"]) <= 1e-11 * abs(o0["virial"])
        e.close()


def test_verlet_dense_cluster_fallbacks():
    """Neighbourhoods longer than the list capacity (96 entries) and brick shells larger than the staging area fall back to
    direct evaluation with the same arithmetic: dense gas blob in a big box, against the generic kernel."""
    rng = np.random.default_rng(5)
    L = 40.0
    # jittered lattices (minimum distance 0.35): a blob of 6 molecules per unit volume (~390 neighbours inside rc = 2.5)
    # inside a dilute gas
    g = np.arange(14.0, 26.0, 0.55)
    blob = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.1, 0.1, (len(g) ** 3, 3))
    h = np.arange(1.0, L, 2.0)
    rest = np.stack(np.meshgrid(h, h, h, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (len(h) ** 3, 3))
    rest = rest[~np.all((rest > 13.0) & (rest < 27.0), axis=1)]
    r = np.concatenate([blob, rest])
    n = len(r)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    v = rng.normal(0, 0.3, (n, 3))
    # weakly interacting (eps 1e-3, sigma 0.3): the blob is dense in NEIGHBOURS, not in energy, so a few steps stay tame
    soft = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1e-3, 0.3, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    _lj = lambda: soft  # noqa: E731
    ref = _engine(_lj(), 2.5, [L] * 3, ids, r, v, force_kernel=capi.FK_GENERIC)
    o0 = ref.run(1e-9, 1)
    F0 = _state(ref)[3]
    ref.close()
    e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.3)
    out = e.run(1e-9, 1)
    F = _state(e)[3]
    assert rel_max(F, F0) < 1e-11
    assert abs(out["upot"] - o0["upot"]) <= 1e-10 * abs(o0["upot"]) and abs(out["virial"] - o0["virial"]) <= 1e-10 * abs(o0["virial"])
    # and a few fused steps through the fallbacks stay on the per-step kernels' trajectory
    e2 = _engine(_lj(), 2.5, [L] * 3, ids, r, v)
    a = e2.run(0.0005, 6)
    e3 = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.3)
    b = e3.run(0.0005, 6)
    assert rel_max(_state(e3)[2], _state(e2)[2]) < 1e-9
    assert abs(a["upot"] - b["upot"]) <= 1e-9 * abs(a["upot"])
    for x in (e, e2, e3):
        x.close()


def test_verlet_state_rules():
    """The list loop leaves the context where the piecewise entry points expect it, and they invalidate the lists."""
    L, ids, r, v = synth.bcc_box(12)
    e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.3)
    with pytest.raises(capi.Ls1HipError):
        e.set_verlet(0.2)  # after set_domain
    e.run(0.002, 5)
    b0 = e.get_option("verlet_builds")
    # piecewise step after a list run: plain entry points on the same context
    e.kick_drift(0.002); e.rebin(); e.halo()
    u, w = e.forces(0)
    e.kick(0.001)
    e.run(0.002, 5)  # starts with an unfused drift: rebuilds first
    assert e.get_option("verlet_builds") > b0
    # reference: per-step kernels for the same 11 steps
    e2 = _engine(_lj(), 2.5, [L] * 3, ids, r, v)
    e2.run(0.002, 5)
    e2.kick_drift(0.002); e2.rebin(); e2.halo(); e2.forces(0); e2.kick(0.001)
    e2.run(0.002, 5)
    a, b = _state(e), _state(e2)
    assert rel_max(a[2], b[2]) < 1e-10 and rel_max(a[3], b[3]) < 1e-9
    e.close(); e2.close()


@pytest.mark.parametrize("skin", [0.3, 0.1])
def test_verlet_nvt_run_matches_reference_trajectory(skin):
    """10 thermostatted Leapfrog steps of the REAL reference (bcc1clj_3456_nvt10) through the list loop: with the
    velocity-scaling thermostat the integrator passes stay separate kernels (kick, scale, kick+drift); the drift pass
    advances the displacement bound, ls1hip_update decides between a halo refresh and a rebuild."""
    name = "bcc1clj_3456_nvt10"
    case = MAN[name]
    g = read_golden(name)
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = _engine(ps.components, case["rc"], ps.length, st["ids"], st["r"], st["v"], skin=skin)
    e.set_thermostat(True, ps.temperature)
    out = e.run(case["dt"], case["steps"])
    assert e.get_option("verlet_steps") == case["steps"] and 1 <= e.get_option("verlet_builds") < case["steps"]
    ids, r, v, F = _state(e)
    rec = g["recs"]
    L = ps.length
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert r.min() >= 0 and np.all(r < L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(v, rec["v"]) < 1e-9
    assert rel_max(F, rec["F"]) < 1e-8 and rel_componentwise(F, rec["F"]) < 1e-7
    assert abs(out["upot"] - g["upot"]) / abs(g["upot"]) < 1e-9
    e.close()


@pytest.mark.parametrize("temp,steps", [(0.95, 60), (12.0, 30)])
def test_verlet_piecewise_unfused_loop_over_many_rebuilds(temp, steps):
    """The piecewise list-aware loop an adapter drives (ls1hip_update, ls1hip_forces_list with dt = 0, ls1hip_kick,
    ls1hip_kick_drift — what LinkedCellsHip / LeapfrogHip call) == the per-step search loop, over several list
    lifetimes; the rebuild decision comes from the bound the unfused drift kernel accumulates."""
    dt = 0.002
    L, ids, r, v = synth.bcc_box(20, temp=temp)
    a = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=None)
    b = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.25)
    rebuilt = 0
    for s in range(steps):
        a.kick_drift(dt); a.rebin(); a.halo(); ua = a.forces(0, want_macro=True); ka = a.kick(0.5 * dt)
        b.kick_drift(dt)
        rebuilt += b.update()
        ub = b.forces_list(0, 0.0, want_macro=True)
        kb = b.kick(0.5 * dt)
        assert abs(ua[0] - ub[0]) <= 1e-10 * abs(ua[0]) and abs(ua[1] - ub[1]) <= 1e-9 * abs(ua[1]), s
        assert abs(ka[0] - kb[0]) <= 1e-10 * ka[0], s
    vmax = np.sqrt((v * v).sum(1).max())
    assert max(1, int(steps * dt * vmax * 0.5 / 0.125)) <= rebuilt <= steps // 2 + 1, rebuilt
    assert b.get_option("verlet_builds") == rebuilt
    sa, sb = _state(a), _state(b)
    assert np.array_equal(sa[0], sb[0])
    dr = sa[1] - sb[1]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10 * L
    assert rel_max(sb[2], sa[2]) < 1e-10 and rel_max(sb[3], sa[3]) < 1e-9
    # a second update without a drift in between only refreshes; lists off: update == rebin + halo
    assert b.update() is False
    assert a.update() is True
    a.close(); b.close()


@pytest.mark.parametrize("precision,tol_f,tol_traj", [(1, 3e-5, 5e-5), (2, 5e-5, 1e-4)])
def test_single_precision_list_pass_against_fp64(precision, tol_f, tol_traj):
    """The reference's MARDYN_SPDP / MARDYN_SPSP build modes as a run-time option of the list force pass ("precision" = 1: FP32
    pair arithmetic, FP64 sums; 2: FP32 sums too; the molecule state and the integration stay FP64).  Parity is UNPINNED against
    the reference (no single-precision build of it here): the checks are against this library's FP64 kernels and, over 10 steps,
    over 20 steps, against its FP64 list loop, with single-precision tolerances."""
    # (a) one evaluation on a liquid box: forces, U_pot, virial against the FP64 list pass
    L, ids, r, v = synth.bcc_box(20, temp=0.95, rho=0.6)  # (cells of a small box are coarse: at rho 0.6 every brick's shell is staged)
    res = {}
    for prec in (0, precision):
        e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.25, precision=prec)
        e.update()
        assert e.get_option("precision_in_use") == prec
        res[prec] = (e.forces_list(0, 0.0, want_macro=True), _state(e)[3])
        e.close()
    (u0, w0), F0 = res[0]
    (u1, w1), F1 = res[precision]
    assert rel_max(F1, F0) < tol_f and abs(u1 - u0) <= tol_f * abs(u0) and abs(w1 - w0) <= 10 * tol_f * abs(w0)
    # (b) 20 fused steps (two list lifetimes) against the FP64 loop — whose trajectories the golden tests above pin to the real
    # reference.  (The reference's own golden boxes are too small for this mode: their coarse cells overflow the staging area,
    # which switches it off, see (c).)
    runs = {}
    for prec in (0, precision):
        e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.25, precision=prec)
        out = e.run(0.002, 20)
        assert e.get_option("precision_in_use") == prec and e.get_option("verlet_steps") == 20 and e.get_option("verlet_builds") >= 2
        runs[prec] = (out,) + _state(e)
        e.close()
    a, b = runs[0], runs[precision]
    assert np.array_equal(a[1], b[1])
    dr = a[2] - b[2]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < tol_traj * L
    assert rel_max(b[3], a[3]) < 20 * tol_traj and rel_max(b[4], a[4]) < 50 * tol_traj
    for k in ("upot", "virial", "summv2"):
        assert abs(a[0][k] - b[0][k]) <= 10 * tol_traj * abs(a[0][k]), k
    # (c) irregular bricks (the dense blob of test_verlet_dense_cluster_fallbacks: list overflow, shells beyond the staging
    # area) switch the mode off by themselves — the FP64 kernels run, with their FP64 results
    rng = np.random.default_rng(5)
    Lc = 40.0
    gl = np.arange(14.0, 26.0, 0.55)
    blob = np.stack(np.meshgrid(gl, gl, gl, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.1, 0.1, (len(gl) ** 3, 3))
    hl = np.arange(1.0, Lc, 2.0)
    rest = np.stack(np.meshgrid(hl, hl, hl, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (len(hl) ** 3, 3))
    rest = rest[~np.all((rest > 13.0) & (rest < 27.0), axis=1)]
    pts = np.concatenate([blob, rest])
    idc = np.arange(1, len(pts) + 1, dtype=np.uint64)
    soft = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1e-3, 0.3, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    res = {}
    for prec in (0, precision):
        e = _engine(soft, 2.5, [Lc] * 3, idc, pts, np.zeros_like(pts), skin=0.3, precision=prec)
        e.update()
        assert e.get_option("precision") == prec and e.get_option("precision_in_use") == 0
        e.forces_list(0, 0.0)
        res[prec] = _state(e)[3]
        e.close()
    assert np.array_equal(res[0], res[precision])


def test_local_rebuild_criterion_with_fast_outliers():
    """Rebuild decision by brick neighbourhoods (k_bound_local: sum of dt * (s1 + s2) / 2 over the two largest speeds of the 27
    bricks around a brick) instead of the global sum of dt * v_max.  One molecule at four times the thermal maximum
    makes the two criteria differ by almost a factor of two — and any listed pair the local one released too early would
    show as a force error of order 1e-2: the trajectory must still be that of the per-step kernels, with fewer builds than the
    global criterion needs.  A second run mixes fused and unfused drifts (runs of 7 steps: the first step of every run
    integrates separately) — the unfused ones must count for every brick."""
    n, dt, steps = 40, 0.002, 63
    L, ids, r, v = synth.bcc_box(n, temp=0.95)
    rng = np.random.default_rng(3)
    fast = rng.choice(len(ids), 1, replace=False)  # (two of them would share a 27-brick neighbourhood of this small box)
    d = rng.normal(size=(1, 3))
    v = v.copy()
    v[fast] = 16.0 * d / np.linalg.norm(d, axis=1)[:, None]
    res, builds = {}, {}
    for mode, skin, opts, chunk in (("step", None, {}, steps), ("local", 0.3, {}, steps), ("global", 0.3, {"local_rebuild": 0}, steps),
                                    ("mixed", 0.3, {}, 7)):
        e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=skin, **opts)
        for _ in range(steps // chunk):
            out = e.run(dt, chunk)
        res[mode] = _state(e) + (out,)
        if skin:
            builds[mode] = e.get_option("verlet_builds")
        e.close()
    assert builds["local"] < builds["global"], builds   # the criterion is active ...
    assert builds["local"] >= builds["global"] // 2, builds   # ... and still bounded by the pair argument (s2 > 0)
    a = res["step"]
    for mode in ("local", "global", "mixed"):
        b = res[mode]
        assert np.array_equal(a[0], b[0])
        dr = a[1] - b[1]
        dr -= L * np.round(dr / L)
        assert np.max(np.abs(dr)) < 1e-10 * L, mode
        assert rel_max(b[2], a[2]) < 1e-10, mode
        assert rel_max(b[3], a[3]) < 1e-9, mode
        for k in ("upot", "virial", "summv2"):
            assert abs(a[4][k] - b[4][k]) <= 1e-10 * abs(a[4][k]), (mode, k)


def test_local_rebuild_criterion_for_unfused_drifts_nvt():
    """NVT runs (and every piecewise driver: the reference's Simulation::simulate through LinkedCellsHip) drift in a separate
    kick + drift pass.  Round 3: the list force pass that does the post-force kick reports, per BRICK, the two largest bounds
    |v| + dt/2m |F| of the coming drift speed (|beta v + dt/2m F| <= max(beta, 1) times that) — the local criterion
    (k_bound_local) now also serves unfused drifts instead of the global dt * v_max.  Same set-up as above (one molecule at four times the thermal
    maximum), velocity-scaling thermostat on the device: trajectory of the per-step kernels, fewer builds than the global
    criterion."""
    n, dt, steps = 40, 0.002, 63
    L, ids, r, v = synth.bcc_box(n, temp=0.95)
    rng = np.random.default_rng(3)
    fast = rng.choice(len(ids), 1, replace=False)
    d = rng.normal(size=(1, 3))
    v = v.copy()
    v[fast] = 16.0 * d / np.linalg.norm(d, axis=1)[:, None]
    res, builds = {}, {}
    for mode, skin, opts in (("step", None, {}), ("local", 0.3, {}), ("global", 0.3, {"local_rebuild": 0})):
        e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=skin, **opts)
        e.set_thermostat(True, 0.95)
        out = e.run(dt, steps)
        res[mode] = _state(e) + (out,)
        if skin:
            builds[mode] = e.get_option("verlet_builds")
        e.close()
    assert builds["local"] < builds["global"], builds
    a = res["step"]
    for mode in ("local", "global"):
        b = res[mode]
        dr = a[1] - b[1]
        dr -= L * np.round(dr / L)
        assert np.max(np.abs(dr)) < 1e-10 * L, mode
        assert rel_max(b[2], a[2]) < 1e-10, mode
        assert rel_max(b[3], a[3]) < 1e-9, mode
        for k in ("upot", "virial"):
            assert abs(a[4][k] - b[4][k]) <= 1e-10 * abs(a[4][k]), (mode, k)


@pytest.mark.parametrize("precision,tol", [(0, 1e-11), (1, 3e-5), (2, 6e-5)])
def test_list_pass_with_a_crowded_cell_in_a_regular_brick(precision, tol):
    """A cell with more than 32 molecules inside a brick that is otherwise REGULAR (region fits the staging area, every list fits):
    the staging's second loop (molecules 33... of a cell) of the FP64 and of the single-precision list kernels.  Dilute soft gas
    (8 molecules per cell) + 27 extra molecules in one cell; against the generic per-step kernel."""
    rng = np.random.default_rng(11)
    L = 22.4  # 8 cells of 2.8 = rc + skin 0.3: 2 x 2 x 4 bricks
    h = np.arange(0.7, L, 1.4)
    gas = np.stack(np.meshgrid(h, h, h, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (len(h) ** 3, 3))
    c = 2.8 * 3 + 0.6 + 0.8 * np.arange(3)
    blob = np.stack(np.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.1, 0.1, (27, 3))
    r = np.concatenate([gas, blob]) % L
    n = len(r)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    v = np.zeros_like(r)
    soft = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1e-3, 0.3, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    in_cell = np.all((r >= 8.4) & (r < 11.2), axis=1).sum()
    assert in_cell > 32, in_cell
    ref = _engine(soft, 2.5, [L] * 3, ids, r, v, force_kernel=capi.FK_GENERIC)
    u0, w0 = ref.forces(0)
    F0 = _state(ref)[3]
    ref.close()
    e = engine_mod.DeviceEngine(0)
    e.set_components(soft, 2.5)
    e.set_option("precision", precision)
    e.set_verlet(0.3)  # not forced: the regions fit
    e.set_domain([L] * 3)
    e.upload(ids, np.zeros(n, np.int32), r, v)
    assert e.get_option("verlet_lists") == 1
    assert e.update() is True
    u, w = e.forces_list(0, 0.0, want_macro=True)
    assert e.get_option("verlet_irregular_bricks") == 0 and e.get_option("precision_in_use") == precision
    F = _state(e)[3]
    assert rel_max(F, F0) < tol
    assert abs(u - u0) <= 10 * tol * abs(u0) and abs(w - w0) <= 10 * tol * abs(w0)
    e.close()


def test_piecewise_list_loop_with_a_separate_velocity_scaling_between_pass_and_drift():
    """the documented piecewise sequence ls1hip_forces_list_kick -> ls1hip_scale_velocities(beta > 1) -> ls1hip_kick_drift.
    The per-brick drift-speed bounds the post-kick list pass leaves for the local rebuild criterion were formed BEFORE the scaling;
    a scaling in between must not leave them in force (they would under-count every brick's displacement by the factor beta and
    delay the rebuild).  A fast outlier makes the criterion bite; heating by 5 % per step for a stretch, then cooling; against the
    per-step search loop over many list lifetimes.  Also: ls1hip_scale_kick_drift_components refuses to drift a second time after
    a fused pass already advanced the positions (the guard ls1hip_kick_drift has)."""
    n, dt, steps = 24, 0.002, 70
    L, ids, r, v = synth.bcc_box(n, temp=0.95)
    rng = np.random.default_rng(5)
    d = rng.normal(size=3)
    v = v.copy()
    v[rng.integers(len(ids))] = 14.0 * d / np.linalg.norm(d)
    beta = lambda s: 1.05 if 10 <= s < 30 else (0.97 if 40 <= s < 55 else 1.0)  # noqa: E731
    a = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=None)
    b = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.3)
    assert b.get_option("verlet_lists") == 1
    rebuilt = 0
    for s in range(steps):
        a.kick_drift(dt); a.rebin(); a.halo(); ua = a.forces(0, want_macro=True); a.kick(0.5 * dt, want_sums=False)
        a.scale_velocities(beta(s))
        b.kick_drift(dt)
        rebuilt += b.update()
        if b.get_option("list_kick_available"):
            ub = b.forces_list_kick(0.5 * dt, want_macro=True)
        else:
            ub = b.forces_list(0, 0.0, want_macro=True); b.kick(0.5 * dt, want_sums=False)
        b.scale_velocities(beta(s))
        assert abs(ua[0] - ub[0]) <= 1e-10 * abs(ua[0]) and abs(ua[1] - ub[1]) <= 1e-9 * abs(ua[1]), s
    assert rebuilt >= 3
    sa, sb = _state(a), _state(b)
    dr = sa[1] - sb[1]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-10 * L
    assert rel_max(sb[2], sa[2]) < 1e-10 and rel_max(sb[3], sa[3]) < 1e-9
    a.close(); b.close()
    # the missing guard
    e = _engine(_lj(), 2.5, [L] * 3, ids, r, v, skin=0.3)
    e.update()
    e.forces_list(0, dt, want_macro=False)  # fused: positions already advanced by this pass
    with pytest.raises(capi.Ls1HipError):
        e.scale_kick_drift_components([1.0], [1.0], dt)
    e.close()
