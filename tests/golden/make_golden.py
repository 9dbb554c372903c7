#!/usr/bin/env python3
"""Regenerates tests/golden/*.bin from the REAL reference (oracle/_ref/refdump, see oracle/ref_build/).

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
Inputs are the data files of the reference's own tests (/root/reference/test_input/*.inp, copied here as
fixtures) plus synthetic jittered-lattice boxes written by this script.  Output format: see refdump.cpp header.
"""
import gzip
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_INPUTS = "/root/reference/test_input"
REFDUMP = os.path.join(ROOT, "oracle", "_ref", "refdump")
INPUTS = os.path.join(HERE, "inputs")

# name, input file, cutoff, periodic, steps, dt, legacy
# cutoffs for the reference's own cases: VectorizedCellProcessorTest.cpp:61,109,150,350-388
CASES = [
    ("U0", "ForceCalculationTestU0.inp", 1.1, 0, 0, 0.0, 0),
    ("F0", "ForceCalculationTestF0.inp", 1.3, 0, 0, 0.0, 0),
    ("U0_periodic", "ForceCalculationTestU0.inp", 1.1, 1, 0, 0.0, 0),
    ("lj1clj", "VectorizationLennardJones1CLJ.inp", 35.0, 0, 0, 0.0, 0),
    ("lj", "VectorizationLennardJones.inp", 35.0, 0, 0, 0.0, 0),
    ("charge", "VectorizationCharge.inp", 35.0, 0, 0, 0.0, 0),
    ("chargedipole", "VectorizationChargeDipole.inp", 35.0, 0, 0, 0.0, 0),
    ("chargequadrupole", "VectorizationChargeQuadrupole.inp", 35.0, 0, 0, 0.0, 0),
    ("dipole", "VectorizationDipole.inp", 35.0, 0, 0, 0.0, 0),
    ("dipolequadrupole", "VectorizationDipoleQuadrupole.inp", 35.0, 0, 0, 0.0, 0),
    ("quadrupole", "VectorizationQuadrupole.inp", 35.0, 0, 0, 0.0, 0),
    ("water", "VectorizationWater.inp", 6.16, 0, 0, 0.0, 0),
    ("multi", "VectorizationMultiComponentMultiPotentials.inp", 35.0, 0, 0, 0.0, 0),
    ("multi50", "VectorizationMultiComponentMultiPotentials_50_molecules.inp", 35.0, 0, 0, 0.0, 0),
    ("multi_legacy", "VectorizationMultiComponentMultiPotentials.inp", 35.0, 0, 0, 0.0, 1),
    # the same clusters with the sequential periodic halo (exercises the halo/macroscopic rule for every body)
    ("lj_periodic", "VectorizationLennardJones.inp", 35.0, 1, 0, 0.0, 0),
    ("multi_periodic", "VectorizationMultiComponentMultiPotentials.inp", 35.0, 1, 0, 0.0, 0),
    ("water_periodic", "VectorizationWater.inp", 6.16, 1, 0, 0.0, 0),
    # the reference's water case has lattice spacing 7.4 > 6.16, i.e. no pair inside its cutoff; add a real one
    ("water_rc12", "VectorizationWater.inp", 12.0, 0, 0, 0.0, 0),
    ("water_rc12_periodic", "VectorizationWater.inp", 12.0, 1, 0, 0.0, 0),
    ("dipole_periodic", "VectorizationDipole.inp", 35.0, 1, 0, 0.0, 0),
    ("quadrupole_periodic", "VectorizationQuadrupole.inp", 35.0, 1, 0, 0.0, 0),
    # equilibrated periodic 2CLJ ethane, forces and a short trajectory
    ("ethan", "Ethan_equilibrated.inp", 32.1254, 1, 0, 0.0, 0),
    ("ethan_steps5", "Ethan_equilibrated.inp", 32.1254, 1, 5, 0.5, 0),
    ("lj_steps3", "VectorizationLennardJones.inp", 35.0, 1, 3, 2.0, 0),
    # synthetic jittered bcc 1CLJ liquid (BASELINE config[1] recipe, small N)
    ("bcc1clj_3456", "synthetic:bcc1clj:12", 2.5, 1, 0, 0.0, 0),
    ("bcc1clj_3456_steps10", "synthetic:bcc1clj:12", 2.5, 1, 10, 0.005, 0),
    ("bcc1clj_16000", "synthetic:bcc1clj:20", 2.5, 1, 0, 0.0, 0),
    # round 3: a trajectory on the box whose brick regions FIT the LDS staging area of the neighbour-list kernels (16 molecules
    # per cell at skin 0.2), i.e. the production configuration of the list loop; 20 steps span >= 3 list lifetimes
    ("bcc1clj_16000_steps20", "synthetic:bcc1clj:20", 2.5, 1, 20, 0.005, 0),
    # global velocity-scaling thermostat active (legacy flag value 2 = --nvt): SURVEY 8f-1
    ("bcc1clj_3456_nvt10", "synthetic:bcc1clj:12", 2.5, 1, 10, 0.005, 2),
    ("ethan_nvt5", "Ethan_equilibrated.inp", 32.1254, 1, 5, 0.5, 2),
    # round 3: the reference's single-precision build modes (legacy flag value 3 = refdump_spdp, built -DMARDYN_SPDP: FP32 pair
    # arithmetic, FP64 sums; 4 = refdump_spsp, -DMARDYN_SPSP: FP32 sums too; cmake/modules/options.cmake:13-15) on a box whose
    # brick regions fit the staging area of the list kernels (8 cells per dimension at skin 0.2, 16 molecules per cell)
    ("bcc1clj_8192_spdp", "synthetic:bcc1clj:16", 2.5, 1, 0, 0.0, 3),
    ("bcc1clj_8192_spsp", "synthetic:bcc1clj:16", 2.5, 1, 0, 0.0, 4),
    ("bcc1clj_8192_spdp_steps10", "synthetic:bcc1clj:16", 2.5, 1, 10, 0.005, 3),
    ("bcc1clj_8192_spsp_steps10", "synthetic:bcc1clj:16", 2.5, 1, 10, 0.005, 4),
    ("bcc1clj_8192", "synthetic:bcc1clj:16", 2.5, 1, 0, 0.0, 0),
    # round 4: rigid-body trajectories of ASYMMETRIC tops with multipole torques (FullMolecule.cpp:334-389 with three non-zero
    # moments; rounds 1-3 pinned linear rotors only).  Periodic water (the reference's VectorizationWater.inp, I = (0.0022,
    # 0.0041, 0.0063)) with a cutoff that holds pairs, NVE (the fixture's own, rotationally very hot state), and the same model with thermal
    # velocities under the global velocity-scaling thermostat
    ("water_rc12_steps5", "VectorizationWater.inp", 12.0, 1, 5, 0.02, 0),
    ("waterT_250_nvt5", "synthetic:water:5", 12.0, 1, 5, 0.02, 2),
    # the integrable form of the five-component LJ + charge + dipole + quadrupole set (BASELINE configs[4]; synth.mixed5_*:
    # the fixture's sites + a massive LJ core for the three components the fixture leaves without mass / core, asymmetric
    # moments through the I line, mixing block written) on a periodic jittered bcc lattice at the fixture's number density
    ("mixed5_1024", "synthetic:mixed5:8", 35.0, 1, 0, 0.0, 0),
    ("mixed5_1024_steps5", "synthetic:mixed5:8", 35.0, 1, 5, 0.2, 0),
    ("mixed5_1024_nvt5", "synthetic:mixed5:8", 35.0, 1, 5, 0.2, 2),
    # two components, two thermostats assigned by the legacy .inp header (ThermostatTemperature / ComponentThermostat,
    # ASCIIReader.cpp:104-124): the component-wise branch of VelocityScalingThermostat::apply (Simulation.cpp:1112-1126)
    ("twotherm_1024_nvt8", "synthetic:twotherm:8", 2.5, 1, 8, 0.004, 5),
]


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def bcc1clj(n_per_dim, rho=0.785302672, jitter=0.1, temp=0.95):
    """Jittered bcc lattice of 2*n^3 LJ atoms (sigma=eps=m=1), deterministic hash noise."""
    n = n_per_dim
    N = 2 * n ** 3
    L = (N / rho) ** (1.0 / 3.0)
    a = L / n
    idx = np.arange(n ** 3)
    ix, iy, iz = idx % n, (idx // n) % n, idx // (n * n)
    base = np.stack([ix, iy, iz], axis=1) * a
    r = np.concatenate([base + 0.25 * a, base + 0.75 * a], axis=0)
    u = np.array([[splitmix64(int(i) * 6 + k) / 2.0 ** 64 - 0.5 for k in range(6)] for i in range(N)])
    r = (r + jitter * u[:, :3]) % L
    # uniform-sum velocities with the requested temperature (deterministic, zero net momentum)
    v = u[:, 3:] * 2.0
    v -= v.mean(axis=0)
    v *= np.sqrt(3.0 * temp * N / (v * v).sum())
    return L, r, v


def _pkg(name):
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("ls1-mardyn_amd." + name)


def write_synth(path, spec):
    _, kind, n = spec.split(":")
    if kind == "mixed5":
        inp = _pkg("inp")
        inp.write_inp(path, _pkg("synth").mixed5_box(inp, int(n)))
        return
    if kind == "water":
        # the water model of the reference's VectorizationWater.inp (component block and density of that fixture: 250 molecules in
        # 37^3) with THERMAL velocities / angular momenta at the fixture's temperature: the fixture's own state has T_rot = 0.074
        # against T_trans = 1.7e-5 and a target of 9.4e-4, which sends the reference's thermostat into its explosion heuristics
        # (Domain.cpp:255-300: per-molecule clamps, outside the scope of the device thermostat)
        inp = _pkg("inp")
        fx = inp.read_inp(os.path.join(INPUTS, "VectorizationWater.inp"))
        inp.write_inp(path, _pkg("synth").thermal_box(inp, fx.components, int(n), len(fx.ids) / float(np.prod(fx.length)), fx.temperature))
        return
    if kind == "twotherm":
        # 1CLJ atoms (thermostat 1, T = 0.8) mixed with 2CLJ dumbbells (thermostat 2, T = 1.1), reduced units, rho* = 0.6
        inp = _pkg("inp")
        L, r, v = bcc1clj(int(n), rho=0.6)
        N = len(r)
        cs = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1.0, 1.0, 1.0, 0, 0)]),
                               inp.make_component(lj=[(0, 0, -0.25, 0.5, 0.6, 0.9, 0, 0), (0, 0, 0.25, 0.5, 0.6, 0.9, 0, 0)])],
                              np.array([[1.0, 1.0]]), 1e10)
        ids = np.arange(1, N + 1, dtype=np.uint64)
        cid = (((ids - 1) // 3) % 2).astype(np.int32)
        u = np.array([[splitmix64(int(i) * 16 + 9 + k) / 2.0 ** 64 - 0.5 for k in range(7)] for i in ids])
        q = u[:, :4] + 1e-3
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        D = u[:, 4:7] * 0.8 * (cid == 1)[:, None]
        D[:, 2] = 0.0
        ps = inp.PhaseSpace(cs, np.array([L, L, L]), ids, cid, r, v, q, D, 0.0, 0.95, {1: 0.8, 2: 1.1}, {0: 1, 1: 2})
        inp.write_inp(path, ps)
        return
    assert kind == "bcc1clj"
    L, r, v = bcc1clj(int(n))
    with open(path, "w") as fh:
        fh.write("mardyn trunk 20120726\ncurrentTime\t0\n")
        L = float(L)
        fh.write(f"Length\t{L!r} {L!r} {L!r}\nTemperature\t0.95\nNumberOfComponents\t1\n")
        fh.write("1\t0\t0\t0\t0\n0 0 0\t1\t1 1 2.5 0\n0 0 0\n1e+10\n")
        fh.write(f"NumberOfMolecules\t{len(r)}\nMoleculeFormat\tICRV\n")
        for i in range(len(r)):
            fh.write(f"{i + 1} 1 " + " ".join(repr(float(x)) for x in list(r[i]) + list(v[i])) + "\n")


def main():
    os.makedirs(INPUTS, exist_ok=True)
    if not os.path.exists(REFDUMP):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle", "ref_build")])
    for tgt in ("spdp", "spsp"):
        if not os.path.exists(REFDUMP + "_" + tgt):
            subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle", "ref_build"), tgt])
    only = set(sys.argv[1:])  # optional: regenerate the named cases only
    for name, inp, rc, periodic, steps, dt, legacy in CASES:
        if only and name not in only:
            continue
        if inp.startswith("synthetic:"):
            fname = inp.replace(":", "_") + ".inp"
            local = os.path.join(INPUTS, fname)
            if not os.path.exists(local) and not os.path.exists(local + ".gz"):
                write_synth(local, inp)
        else:
            fname = inp
            local = os.path.join(INPUTS, fname)
            if not os.path.exists(local) and not os.path.exists(local + ".gz"):
                shutil.copy(os.path.join(REF_INPUTS, inp), local)
        src = local
        tmp = None
        if not os.path.exists(local):
            tmp = "/tmp/_golden_" + fname
            with gzip.open(local + ".gz", "rb") as fi, open(tmp, "wb") as fo:
                shutil.copyfileobj(fi, fo)
            src = tmp
        out = os.path.join(HERE, name + ".bin")
        cmd = [REFDUMP + {3: "_spdp", 4: "_spsp"}.get(legacy, ""), src, repr(rc), str(periodic), out]
        if legacy == 1:
            cmd.append("--legacy")
        if legacy in (2, 5):  # (5: the header of the input assigns thermostats to components -> component-wise branch)
            cmd.append("--nvt")
        if steps:
            cmd += ["--steps", str(steps), "--dt", repr(dt)]
        env = dict(os.environ, OMP_NUM_THREADS="4")
        print(subprocess.check_output(cmd, env=env).decode().strip())
        if os.path.exists(local) and os.path.getsize(local) > 500_000:
            with open(local, "rb") as fi, gzip.open(local + ".gz", "wb", compresslevel=9) as fo:
                shutil.copyfileobj(fi, fo)
            os.remove(local)
        if tmp:
            os.remove(tmp)
    with open(os.path.join(HERE, "MANIFEST.txt"), "w") as fh:
        fh.write("# name input cutoff periodic steps dt legacy\n")
        for c in CASES:
            fn = c[1].replace(":", "_") + ".inp" if c[1].startswith("synthetic:") else c[1]
            fh.write(" ".join([c[0], fn] + [repr(x) for x in c[2:]]) + "\n")


if __name__ == "__main__":
    sys.exit(main())
