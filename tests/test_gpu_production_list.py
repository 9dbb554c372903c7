"""The PRODUCTION configuration of the neighbour-list loop — the kernels bench.py times (k_lj_verlet_build +
k_force_lj_verlet on LDS-staged, fully listed bricks driven by per-brick records) — against the REAL reference.

VERDICT r2 (weak #1, #2): every earlier list test called set_verlet(skin, force=True) on boxes whose coarse cells put more
molecules into a brick region than the LDS staging area holds, i.e. they exercised the global-memory fallback.  Here the
lists are requested WITHOUT `force` on boxes whose regions fit (bcc1clj_16000: 16 molecules per cell at skin 0.2, region
~2 300 < 2 816), and every test asserts what ran: verlet_lists == 1, verlet_irregular_bricks == 0 (no brick left the
record-driven staged path), last_force_kernel == FK_NEIGHBOUR_LIST.

Reference behaviour matched: VectorizedCellProcessor::_loopBodyLJ (VectorizedCellProcessor.cpp:173-226), LinkedCells::update
(LinkedCells.cpp:243-356), Leapfrog (integrators/Leapfrog.cpp:35-150).  Tolerances: forces / U_pot / virial of one
evaluation 1e-10 (north_star), trajectories 1e-9, list loop vs per-step loop at N = 99 672 064: 1e-12 per step."""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import input_path, manifest, read_golden, rel_componentwise, rel_max, sorted_phase_space

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
capi = load_pkg("capi")
engine_mod = load_pkg("engine")
synth = load_pkg("synth")
MAN = manifest()


def _production_engine(name, skin=0.2, **opts):
    return _production_engine_case(MAN[name], skin, **opts)


def _production_engine_case(case, skin=0.2, **opts):
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    e = engine_mod.DeviceEngine(0)
    e.set_components(ps.components, case["rc"])
    for k, val in opts.items():
        e.set_option(k, val)
    e.set_verlet(skin)  # NOT forced: the engine itself must find that the brick regions fit
    e.set_domain(ps.length)
    e.upload(st["ids"], st["cid"], st["r"], st["v"])
    assert e.get_option("verlet_lists") == 1, "the engine switched the lists off: the regions of this box do not fit"
    return case, ps, st, e


def _assert_production_path(e):
    assert e.get_option("verlet_ready") == 1
    assert e.get_option("verlet_irregular_bricks") == 0
    assert e.get_option("last_force_kernel") == capi.FK_NEIGHBOUR_LIST


def _sorted(e):
    st = e.download_state()
    o = np.argsort(st["ids"], kind="stable")
    return st["ids"][o], st["r"][o], st["v"][o], e.download_forces()["F"][o]


@pytest.mark.parametrize("skin", [0.2, 0.12])
def test_production_list_pass_single_evaluation_against_reference_golden(skin):
    """bcc1clj_16000 (forces, U_pot, virial of the REAL reference): one list build + one plain list force pass."""
    name = "bcc1clj_16000"
    g = read_golden(name)
    case, ps, st, e = _production_engine(name, skin)
    assert e.update() is True  # re-bin + halo + list build
    u, w = e.forces_list(0, 0.0, want_macro=True)
    _assert_production_path(e)
    assert e.get_option("verlet_builds") == 1
    ids, r, v, F = _sorted(e)
    rec = g["recs"]
    assert np.array_equal(ids, rec["id"])
    assert rel_max(F, rec["F"]) < 1e-10
    assert rel_componentwise(F, rec["F"]) < 1e-8
    assert abs(u - g["upot"]) <= 1e-10 * abs(g["upot"])
    assert abs(w - g["virial"]) <= 1e-10 * abs(g["virial"])
    e.close()


@pytest.mark.parametrize("local", [1, 0])
def test_production_list_loop_against_reference_trajectory_20_steps(local):
    """bcc1clj_16000_steps20: 20 Leapfrog steps of the REAL reference through ls1hip_run's list loop (fused force +
    integration passes, device-side rebuild criterion — local and global form), >= 3 list lifetimes at skin 0.2."""
    name = "bcc1clj_16000_steps20"
    g = read_golden(name)
    case, ps, st, e = _production_engine(name, 0.2, local_rebuild=local)
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(case["dt"], case["steps"])
    _assert_production_path(e)
    assert e.get_option("verlet_steps") == case["steps"]
    assert 3 <= e.get_option("verlet_builds") < case["steps"] // 2
    ids, r, v, F = _sorted(e)
    rec = g["recs"]
    L = ps.length
    assert np.array_equal(ids, rec["id"])
    assert r.min() >= 0 and np.all(r < L)
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(v, rec["v"]) < 1e-9
    assert rel_max(F, rec["F"]) < 1e-8 and rel_componentwise(F, rec["F"]) < 1e-6
    assert abs(out["upot"] - g["upot"]) <= 1e-9 * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= 1e-8 * abs(g["virial"])
    assert abs(out["summv2"] - g["summv2"]) <= 1e-9 * abs(g["summv2"])
    e.close()


def test_production_list_piecewise_loop_against_reference_trajectory():
    """The same box with the piecewise list-aware loop an adapter drives (what LinkedCellsHip / LeapfrogHip call: unfused
    kicks, ls1hip_update, plain list force pass) against the 20-step golden trajectory."""
    name = "bcc1clj_16000_steps20"
    g = read_golden(name)
    case, ps, st, e = _production_engine(name, 0.2)
    dt = case["dt"]
    e.update()
    e.forces_list(0, 0.0)
    rebuilt = 0
    for _ in range(case["steps"]):
        e.kick_drift(dt)
        rebuilt += e.update()
        u, w = e.forces_list(0, 0.0, want_macro=True)
        _assert_production_path(e)
        kin = e.kick(0.5 * dt)
    assert 2 <= rebuilt < case["steps"] // 2 and e.get_option("verlet_builds") == rebuilt + 1  # (+ the initial build)
    ids, r, v, F = _sorted(e)
    rec = g["recs"]
    L = ps.length
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-9 * np.max(L)
    assert rel_max(v, rec["v"]) < 1e-9
    assert rel_max(F, rec["F"]) < 1e-8
    assert abs(u - g["upot"]) <= 1e-9 * abs(g["upot"]) and abs(w - g["virial"]) <= 1e-8 * abs(g["virial"])
    assert abs(kin[0] - g["summv2"]) <= 1e-9 * abs(g["summv2"])
    e.close()


def _headline_engine(torch, n, skin):
    N = 2 * n ** 3
    L = synth.box_length(n)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, 2.5)
    if skin:
        e.set_verlet(skin)
    e.set_domain([L] * 3)
    dev = torch.device("cuda", 0)
    e.upload_begin(N)
    psum = np.zeros(3)
    for ids_t, r_t, v_t in synth.bcc_chunks_device(torch, dev, n):
        torch.cuda.synchronize()
        psum += v_t.sum(0).cpu().numpy()
        e.upload_chunk_device(ids_t.numel(), ids_t.data_ptr(), 0, r_t.data_ptr(), v_t.data_ptr())
    del ids_t, r_t, v_t
    e.upload_end()
    torch.cuda.empty_cache()
    assert e.count()[0] == N
    return e, psum


@pytest.mark.parametrize("ensemble", ["nve", "nvt"])
def test_headline_box_1e8_list_loop_equals_per_step_loop(ensemble):
    """(nvt: velocity-scaling thermostat on the device — the list loop then takes the post-kick force pass + a separate scale / kick /
    drift pass, with the local rebuild criterion fed by the force pass's per-brick speed bounds.)
    N = 2*368^3 = 99 672 064 — the box and the loop bench.py times (set_verlet(0.2), not forced, fused list passes): every
    step's {U_pot, virial, sum m v^2} of the run log equals the per-step-kernel loop's (which the goldens pin to the reference
    and test_headline_box_1e8_properties to the generic kernel) to 1e-12 over 14 steps incl. a list rebuild; every brick runs
    the staged, record-driven path; ids stay a permutation; total momentum is conserved."""
    import torch

    n, steps, dt = 368, 14, 0.002
    N = 2 * n ** 3
    logs = {}
    for mode, skin in (("step", None), ("list", 0.2)):
        e, psum = _headline_engine(torch, n, skin)
        if ensemble == "nvt":
            e.set_thermostat(True, 0.95)
        e.rebin(); e.halo(); e.forces(0)
        e.run(dt, steps)
        logs[mode] = e.run_log()[:steps].copy()
        if skin:
            assert e.get_option("verlet_lists") == 1
            _assert_production_path(e)
            assert e.get_option("verlet_steps") == steps and e.get_option("verlet_builds") >= 2
            ids = e.download_ids()
            ids.sort()
            assert np.array_equal(ids, np.arange(1, N + 1, dtype=np.uint64))
            del ids
            p = e.download_velocities().sum(0)
            if ensemble == "nve":  # (the thermostat scales the momentum with the velocities)
                assert np.max(np.abs(p - psum)) < 1e-9 * np.sqrt(N)
        else:
            assert e.get_option("last_force_kernel") == capi.FK_LDS_LIST
        e.close()
        torch.cuda.empty_cache()
    a, b = logs["step"], logs["list"]
    assert np.all(np.isfinite(a[:, :3])) and np.all(np.isfinite(b[:, :3]))
    for col, what in ((0, "upot"), (1, "virial"), (2, "summv2")):
        err = np.max(np.abs(a[:, col] - b[:, col]) / np.abs(a[:, col]))
        assert err < (1e-12 if ensemble == "nve" else 1e-11), (what, err)


# ---- the reference's single-precision build modes ---------------------------------------------------------
SP = manifest(single_precision=True)


@pytest.mark.parametrize("name", [k for k, c in SP.items() if c["steps"] == 0])
def test_single_precision_list_pass_against_reference_sp_build(name):
    """"precision" = 1 (SPDP) / 2 (SPSP) of the list force pass against golden vectors of the REAL reference built with
    -DMARDYN_SPDP / -DMARDYN_SPSP (oracle/ref_build `make spdp spsp`; cmake/modules/options.cmake:13-15,
    vectorization/SIMD_TYPES.h:28-36, RealAccumVecSPDP.h), non-forced production configuration, regular bricks only.
    Tolerances are single-precision ones, set from the reference's own SP-vs-FP64 deviation on this box (forces 1.9e-5 of
    max|F|, U_pot 3e-8, virial 3e-8: the reference rounds ABSOLUTE coordinates to FP32, this pass region-relative ones, so it
    must lie at least as close to the FP64 golden as the reference's SP build does)."""
    case = SP[name]
    g, dp = read_golden(name), read_golden("bcc1clj_8192")
    _, ps, st, e = _production_engine_case(case, 0.2, precision=case["precision"])
    assert e.update() is True
    u, w = e.forces_list(0, 0.0, want_macro=True)
    _assert_production_path(e)
    assert e.get_option("precision_in_use") == case["precision"]
    ids, r, v, F = _sorted(e)
    assert np.array_equal(ids, g["recs"]["id"])
    assert rel_max(F, g["recs"]["F"]) < 5e-5
    assert abs(u - g["upot"]) <= 2e-7 * abs(g["upot"]) and abs(w - g["virial"]) <= 2e-7 * abs(g["virial"])
    ref_dev = rel_max(g["recs"]["F"], dp["recs"]["F"])
    assert rel_max(F, dp["recs"]["F"]) <= ref_dev
    assert abs(u - dp["upot"]) <= 1e-7 * abs(dp["upot"]) and abs(w - dp["virial"]) <= 1e-7 * abs(dp["virial"])
    e.close()


@pytest.mark.parametrize("name", [k for k, c in SP.items() if c["steps"] > 0])
def test_single_precision_list_loop_against_reference_sp_trajectory(name):
    """10 Leapfrog steps of the reference's SPDP / SPSP builds through ls1hip_run's list loop in the same precision mode."""
    case = SP[name]
    g = read_golden(name)
    _, ps, st, e = _production_engine_case(case, 0.2, precision=case["precision"])
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(case["dt"], case["steps"])
    _assert_production_path(e)
    assert e.get_option("precision_in_use") == case["precision"] and e.get_option("verlet_builds") >= 2
    ids, r, v, F = _sorted(e)
    rec = g["recs"]
    L = ps.length
    dr = r - rec["r"]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-7 * np.max(L)
    # (FP32 pair arithmetic on differently rounded coordinates — absolute in the reference, region-relative here — over 10 steps:
    # measured 5.6e-5 of max|v|; the reference's own SPDP and SPSP trajectories differ by 4e-7, its SP and FP64 forces by 2e-5)
    assert rel_max(v, rec["v"]) < 2e-4
    assert rel_max(F, rec["F"]) < 5e-4
    assert abs(out["upot"] - g["upot"]) <= 1e-6 * abs(g["upot"])
    assert abs(out["virial"] - g["virial"]) <= 5e-6 * abs(g["virial"])
    assert abs(out["summv2"] - g["summv2"]) <= 1e-6 * abs(g["summv2"])
    e.close()


def test_list_pass_with_folded_post_force_kick_equals_pass_plus_kick():
    """ls1hip_forces_list_kick (the piecewise form of what ls1hip_run does on unfused steps; used by LinkedCellsHip for the
    reference driver): forces, U_pot, virial of ls1hip_forces_list, velocities and sum m v^2 of ls1hip_kick
    (Leapfrog::transition2to3, Leapfrog.cpp:66-150) — and F against the reference's golden."""
    name = "bcc1clj_16000"
    g = read_golden(name)
    dt_half = 0.001
    out = []
    for folded in (False, True):
        case, ps, st, e = _production_engine(name, 0.2)
        assert e.update() is True
        assert e.get_option("list_kick_available") == 1
        if folded:
            u, w = e.forces_list_kick(dt_half, want_macro=True)
            kin = e.kinetic_sums()
        else:
            u, w = e.forces_list(0, 0.0, want_macro=True)
            kin = e.kick(dt_half)
        _assert_production_path(e)
        ids, r, v, F = _sorted(e)
        out.append((u, w, kin, v, F))
        e.close()
    (u0, w0, k0, v0, F0), (u1, w1, k1, v1, F1) = out
    assert rel_max(F1, g["recs"]["F"]) < 1e-10
    assert np.array_equal(F0, F1) and u0 == u1 and w0 == w1
    assert np.max(np.abs(v1 - v0)) <= 1e-15 * np.max(np.abs(v0))
    assert abs(k1[0] - k0[0]) <= 1e-13 * abs(k0[0])  # sum m v^2 (different summation order)
    assert k1[2] == k0[2] == 16000


def test_production_list_pass_truncated_shifted_potential():
    """Truncated AND shifted LJ (Component.cpp:105-118 shift6; the in-range pair count carries the shift, also across the lanes that
    share a molecule of a split leftover tile) on the production configuration: regular bricks of the 16 000-molecule box, against
    the generic per-step kernel."""
    case = MAN["bcc1clj_16000"]
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, case["rc"], 1)])], np.zeros((0, 2)), 1e10)
    res = {}
    for mode in ("generic", "lists"):
        e = engine_mod.DeviceEngine(0)
        e.set_components(comps, case["rc"])
        if mode == "generic":
            e.set_option("force_kernel", capi.FK_GENERIC)
        else:
            e.set_verlet(0.2)
        e.set_domain(ps.length)
        e.upload(st["ids"], st["cid"], st["r"], st["v"])
        if mode == "generic":
            e.rebin(); e.halo()
            u, w = e.forces(0)
        else:
            assert e.update() is True
            u, w = e.forces_list(0, 0.0, want_macro=True)
            _assert_production_path(e)
        res[mode] = (u, w, _sorted(e)[3])
        e.close()
    (u0, w0, F0), (u1, w1, F1) = res["generic"], res["lists"]
    assert rel_max(F1, F0) < 1e-11
    assert abs(u1 - u0) <= 1e-11 * abs(u0) and abs(w1 - w0) <= 1e-11 * abs(w0)


def test_library_under_test_is_the_regular_build():
    """the library these GPU tests load holds no timing-variant object (tools/ab_variant.sh builds; option "build_variant")"""
    e = engine_mod.DeviceEngine(0)
    assert e.get_option("build_variant") == 0 and b"variant" not in e.lib.ls1hip_version()
    e.close()
