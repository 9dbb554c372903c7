"""Full-size parity of BASELINE configs[3] and [4] (multi-site / electrostatic component sets at ~10^7 molecules).

The scalar oracle needs minutes at this size, so the size-independent property used is PERIODIC REPLICATION
(the reference builds its own large multi-site boxes the same way, io/ReplicaGenerator.cpp): a small periodic box that
IS checked — against the reference's golden vectors (ethane) or the oracle (mixed set) — is tiled k x k x k times.  The
big system has the small box's period, so every replica of a molecule must feel the force, torque and virial of the
original, U_pot and the virial are k^3 times the small box's.  Tolerance 1e-10 (north_star), metric max|a-b|/max|b|.
configs[1] (1CLJ, N = 10 000 422) is covered at full size in test_gpu_parity.py::test_full_size_properties_bench_workload.
"""
import numpy as np
import pytest

from conftest import load_pkg
from golden_io import input_path, manifest, read_golden, rel_max, sorted_phase_space
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
engine_mod = load_pkg("engine")
MAN = manifest()
TOL = 1e-10


def replicate(length, r, k):
    shifts = np.array([[i, j, l] for i in range(k) for j in range(k) for l in range(k)], dtype=float) * length
    return (r[None, :, :] + shifts[:, None, :]).reshape(-1, 3)


def tile(a, k):
    return np.tile(a, (k ** 3,) + (1,) * (a.ndim - 1))


def device_forces(components, rc, length, ids, cid, r, q, cic=1):
    eng = engine_mod.DeviceEngine(0)
    eng.set_components(components, rc)
    eng.set_option("cells_in_cutoff", cic)
    eng.set_option("compute_vi", 1)
    eng.set_domain(length)
    n = len(ids)
    eng.upload(ids, cid, r, np.zeros((n, 3)), q, np.zeros((n, 3)))
    eng.rebin(); eng.halo()
    u, w = eng.forces(0)
    st = eng.download_state()
    f = eng.download_forces(with_vi=True)
    o = np.argsort(st["ids"], kind="stable")
    assert np.array_equal(st["ids"][o], np.sort(ids))
    eng.close()
    return dict(F=f["F"][o], M=f["M"][o], Vi=f["Vi"][o], upot=u, virial=w)


def check_replicas(big, small, n0, k, rng):
    """replica 0, the last replica and 6 random ones against the small box, plus the extensive sums"""
    picks = sorted({0, k ** 3 - 1, *rng.integers(0, k ** 3, 6).tolist()})
    for key in ("F", "M", "Vi"):
        ref = small[key]
        if np.max(np.abs(ref)) == 0.0:
            assert np.max(np.abs(big[key])) == 0.0
            continue
        for p in picks:
            assert rel_max(big[key][p * n0:(p + 1) * n0], ref) < TOL, (key, p)
    assert abs(big["upot"] - k ** 3 * small["upot"]) <= TOL * abs(k ** 3 * small["upot"])
    assert abs(big["virial"] - k ** 3 * small["virial"]) <= TOL * abs(k ** 3 * small["virial"])
    # Newton 3 through the periodic images, all site types
    fmax = np.max(np.abs(big["F"]))
    assert np.max(np.abs(big["F"].sum(0))) < 1e-9 * fmax * np.sqrt(len(big["F"]))


def test_config3_ethane_2clj_10m_replicated():
    """configs[3]: the reference's equilibrated periodic ethane box (9 826 2CLJ molecules, r_c = 32.1254), pinned by the
    golden vectors of the real VectorizedCellProcessor, replicated 10^3 times = 9 826 000 molecules."""
    case = MAN["ethan"]
    g = read_golden("ethan")
    ps = inp.read_inp(input_path(case["input"]))
    st = sorted_phase_space(ps)  # sorted by id: row i <-> golden record i
    assert np.array_equal(st["ids"], g["recs"]["id"])
    n0, k = len(st["ids"]), 10
    small = dict(F=g["recs"]["F"], M=g["recs"]["M"], Vi=g["recs"]["Vi"], upot=g["upot"], virial=g["virial"])
    big = device_forces(ps.components, case["rc"], ps.length * k, np.arange(1, n0 * k ** 3 + 1, dtype=np.uint64),
                        tile(st["cid"], k), replicate(ps.length, st["r"], k), tile(st["q"], k))
    assert len(big["F"]) == 9826000
    check_replicas(big, small, n0, k, np.random.default_rng(5))


def test_config4_mixed_multipotential_10m_replicated():
    """configs[4]: the five components of VectorizationMultiComponentMultiPotentials.inp (LJ + charge + dipole +
    quadrupole sites) on a jittered bcc lattice at the fixture's number density, component = id mod 5, random unit
    quaternions, r_c = 35.  Small box 2*12^3 = 3 456 molecules checked against the oracle here, then
    replicated 14^3 times = 9 483 264 molecules."""
    ps0 = inp.read_inp(input_path("VectorizationMultiComponentMultiPotentials.inp"))
    comps = ps0.components
    ncomp = len(comps.components)
    assert ncomp == 5
    rc = 35.0
    rng = np.random.default_rng(11)
    n = 12
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
    length = np.array([L, L, L])
    ref = Oracle(comps.flat(), rc).forces(r, q, cid, length, True)
    small = device_forces(comps, rc, length, np.arange(1, n0 + 1, dtype=np.uint64), cid, r, q)
    for key in ("F", "M", "Vi"):
        assert rel_max(small[key], ref[key]) < TOL, key
    assert abs(small["upot"] - ref["upot"]) <= TOL * abs(ref["upot"])
    assert abs(small["virial"] - ref["virial"]) <= TOL * abs(ref["virial"])
    k = 14
    big = device_forces(comps, rc, length * k, np.arange(1, n0 * k ** 3 + 1, dtype=np.uint64), tile(cid, k),
                        replicate(length, r, k), tile(q, k))
    assert len(big["F"]) == 9483264
    check_replicas(big, ref, n0, k, np.random.default_rng(6))
