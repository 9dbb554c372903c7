#!/usr/bin/env python3
"""bench.py — particle-updates/s of the linked-cell pair-force hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full time step of the hot path (Leapfrog pre-force kick+drift -> re-bin -> halo -> pair forces ->
post-force kick, with the per-step global values U_pot / virial / sum m v^2) over the synthetic liquid BASELINE.json's
metric is quoted on: single-centre Lennard-Jones, rho*=0.785302672, rc=2.5 sigma, T*=0.95, dt=0.002, FP64,
N = 2*368^3 = 99 672 064 (the "10^8" box), which fits ONE MI355X; `--gpus N` splits that SAME box over N GPUs (strong
scaling, regular rank grid, RCCL ghost-cell halo).  `--n-per-dim 171` gives the 10^7 variant (configs[1]).  The start
configuration is generated in device memory and is resident in HBM before the timed region.  Prints ONE JSON line
(rank 0) with `roofline` (force kernel: algorithmic bytes / HIP-event kernel time vs 8 TB/s) and `cpu_baseline` (the
REAL reference binary oracle/_ref/MarDyn timed on the host cores on a bounded sample of the same workload; the oracle
restatement is used only if that binary is absent).
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RHO = 0.785302672
RC = 2.5
DT = 0.002
TEMP = 0.95
FORCE_BYTES_PER_MOLECULE = 48.0   # read r (24 B) + write F (24 B): SURVEY.md 8(d)
# fused force + integration pass (between steps, reduced-memory mode): read r, v (48 B) + write r', v' (48 B); F never
# reaches HBM and the 120 B integrator pass disappears (SURVEY.md 8(d) force 48 B + integrator 120 B -> 96 B)
FUSED_BYTES_PER_MOLECULE = 96.0
PMC_SUMMARY = "r2_pmc_summary.json"  # rocprofv3 PMC passes of this workload (tools/collect_profiles.sh)
STEP_BYTES_PER_MOLECULE = 292.0   # full step: force 48 + integrator 120 + re-bin 124
MS_FORCE_BYTES_PER_MOLECULE = 104.0  # multi-site force pass: read r 24 + q 32, write F 24 + M 24
# fused rigid-body list pass (single-component sets, between steps): read r q v D (104 B) + write r q v D (104 B); F and M never
# reach HBM and the 256 B integrator pass disappears
MS_FUSED_BYTES_PER_MOLECULE = 208.0
HBM_PEAK_GBS = 8000.0
BASELINE_METRIC = "particle-updates/sec (whole node), 10^8 LJ liquid Argon, rc=2.5\u03c3"  # BASELINE.json, verbatim


def lj_components(inp):
    return inp.ComponentSet([inp.make_component(lj=[(0., 0., 0., 1., 1., 1., RC, 0)])], np.zeros((0, 2)), 1e10)


MARDYN_XML = """<?xml version='1.0' encoding='UTF-8'?>
<mardyn version="20100525">
  <refunits type="SI"><length unit="nm">0.1</length><mass unit="u">1</mass><energy unit="K">1</energy></refunits>
  <simulation type="MD">
    <integrator type="Leapfrog"><timestep unit="reduced">{dt}</timestep></integrator>
    <run><currenttime>0</currenttime><production><steps>{steps}</steps></production></run>
    <ensemble type="NVT">
      <temperature unit="reduced">{temp}</temperature>
      <domain type="box"><lx>{L}</lx><ly>{L}</ly><lz>{L}</lz></domain>
      <components>
        <moleculetype id="1" name="1CLJ">
          <site type="LJ126" id="1"><coords><x>0.0</x><y>0.0</y><z>0.0</z></coords><mass>1.0</mass><sigma>1.0</sigma><epsilon>1.0</epsilon><shifted>0</shifted></site>
          <momentsofinertia rotaxes="xyz"><Ixx>0.0</Ixx><Iyy>0.0</Iyy><Izz>0.0</Izz></momentsofinertia>
        </moleculetype>
      </components>
      <phasespacepoint><generator name="CubicGridGenerator"><specification>density</specification><density>{rho}</density><binaryMixture>false</binaryMixture></generator></phasespacepoint>
    </ensemble>
    <algorithm>
      <parallelisation type="DomainDecomposition"></parallelisation>
      <datastructure type="LinkedCells"><cellsInCutoffRadius>1</cellsInCutoffRadius></datastructure>
      <cutoffs type="CenterOfMass"><radiusLJ unit="reduced">{rc}</radiusLJ></cutoffs>
      <electrostatic type="ReactionField"><epsilon>1.0e+10</epsilon></electrostatic>
    </algorithm>
    <output></output>
  </simulation>
</mardyn>
"""


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """(threads to use, description): the CPUs this process may really use — the cgroup CPU quota where one is set
    (cpu.max / cfs_quota), else the affinity mask counted in physical cores (one thread per core: the reference's
    vectorised kernels do not gain from SMT siblings)."""
    vis = len(os.sched_getaffinity(0))
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                quota = parse(txt)
            else:
                q = float(txt)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                quota = q / per if q > 0 else None
            break
        except (OSError, ValueError, IndexError):
            continue
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len(sib.replace("-", ",").split(",")))
    except OSError:
        pass
    phys = max(1, vis // smt)
    env = os.environ.get("LS1_BENCH_CPU_THREADS")
    if env:
        return max(1, int(env)), {"visible_cpus": vis, "cgroup_quota_cpus": quota, "physical_cores_visible": phys, "threads_from": "LS1_BENCH_CPU_THREADS"}
    if quota:
        n = max(1, min(phys, int(quota + 0.5)))
        src = "cgroup cpu quota"
    else:
        # no quota: the GPU pool gives one GPU's share of the host (16 CPUs per GPU: gpurun's sizing rule) — use it, not all
        # 128 cores of a box whose other seven GPUs belong to other jobs
        n = min(phys, int(os.environ.get("LS1_BENCH_CPU_SHARE", "16")))
        src = "one GPU's share of the host (16 CPUs per GPU on this pool; LS1_BENCH_CPU_SHARE overrides)"
    return n, {"visible_cpus": vis, "cgroup_quota_cpus": quota, "physical_cores_visible": phys, "threads_from": src}


def _run_reference(binary, cfg_text, steps, cores, budget_s, extra_files=None):
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, "config.xml")
        with open(cfg, "w") as fh:
            fh.write(cfg_text)
        for name, writer in (extra_files or {}).items():
            writer(os.path.join(td, name))
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
        try:
            out = subprocess.run([binary, cfg, "--steps", str(steps), "--final-checkpoint=0"], cwd=td, env=env,
                                 capture_output=True, text=True, timeout=budget_s).stdout
        except subprocess.TimeoutExpired:
            return None, None
        m = re.search(r"Simulation speed:\s*([0-9.eE+-]+)\s*Molecule-updates per second", out)
        nm = re.search(r"System initialised with\s*([0-9]+)\s*molecules", out) or re.search(r"[Nn]umber of molecules[^0-9]*([0-9]+)", out)
        ran = len(re.findall(r"Simstep = \d+", out)) - 1  # (the line of the initial state is not a step)
        if ran < steps:
            return None, None  # the driver did not run the requested steps: no baseline rather than a wrong one
        return (float(m.group(1)) if m else None), (nm.group(1) if nm else None)


def cpu_baseline(n_per_dim=171, steps=10, budget_s=150, workload="lj"):
    """Reference OpenMP CPU path on the host cores this job may use: the REAL reference binary, built twice by
    oracle/ref_build (AVX2 = the reference's portable vector mode, and -march=x86-64-v4 = its AVX-512 kernels, the
    VECTOR_INSTRUCTIONS=NATIVE class of an AVX-512 host); the faster of the two is reported, the other one next to it.
    Sample (bounded, ~10-30 s of CPU work): lj = configs[1] (N = 2*171^3 from the reference's own CubicGridGenerator);
    ethane = the reference's equilibrated ethane box replicated 6^3; mixed = the integrable five-component set on a 2*100^3
    lattice (both from a binary checkpoint written by inp.write_checkpoint, components through inp.components_xml)."""
    cores, share = cpu_share()
    inp = importlib.import_module("ls1-mardyn_amd.inp")
    builds = [("AVX2", os.path.join(ROOT, "oracle", "_ref", "MarDyn")), ("AVX-512 (-march=x86-64-v4)", os.path.join(ROOT, "oracle", "_ref", "MarDyn_avx512"))]
    extra = None
    if workload == "ethane":
        ps, comps_xml = ethane_fixture(inp)
        k = 6
        big = replicate_phase_space(inp, ps, k)
        L = float(big.length[0])
        cfg = MS_XML.format(dt=ETHANE_DT, steps=steps, temp=repr(float(ps.temperature)), L=repr(L), rc=ETHANE_RC,
                            components=inp.components_xml(ps.components, ["C2H6"]))
        extra = {"ms.header.xml": lambda path, big=big: inp.write_checkpoint(path[:-len(".header.xml")], big)}
        what = f"2CLJ ethane, the reference's Ethan_equilibrated box replicated {k}^3 = {len(big.ids)} molecules, rc={ETHANE_RC}"
    elif workload == "mixed":
        synth = importlib.import_module("ls1-mardyn_amd.synth")
        big = synth.mixed5_box(inp, MIXED_CPU_N)
        L = float(big.length[0])
        cfg = MS_XML.format(dt=MIXED_DT, steps=steps, temp=repr(float(big.temperature)), L=repr(L), rc=MIXED_RC,
                            components=inp.components_xml(big.components))
        extra = {"ms.header.xml": lambda path, big=big: inp.write_checkpoint(path[:-len(".header.xml")], big)}
        what = (f"five-component LJ + charge + dipole + quadrupole set (synth.mixed5_box: same recipe as the GPU workload), "
                f"2*{MIXED_CPU_N}^3 = {len(big.ids)} molecules, rc={MIXED_RC}, dt={MIXED_DT}")
    else:
        N = 2 * n_per_dim ** 3
        L = (N / RHO) ** (1.0 / 3.0)
        cfg = MARDYN_XML.format(dt=DT, steps=steps, temp=TEMP, L=repr(L), rho=RHO, rc=RC)
        what = f"1CLJ N={N} bcc rho*={RHO} rc={RC}"
    results = {}
    nmol = None
    for tag, binary in builds:
        if not os.path.exists(binary):
            continue
        # an AVX-512 build on a host without AVX-512 would die with SIGILL: ask the CPU first
        if "512" in tag:
            try:
                if "avx512f" not in open("/proc/cpuinfo").read():
                    continue
            except OSError:
                continue
        v, nm = _run_reference(binary, cfg, steps, cores, budget_s, extra)
        if v:
            results[tag] = v
            nmol = nmol or nm
    if results:
        best = max(results, key=results.get)
        return {"value": results[best], "unit": "particle-updates/s", "cores": cores, "kind": "reference", "cpu": cpu_model(),
                "cpu_share": share, "build": best, "all_builds": results,
                "sample": f"reference MarDyn ({best}, OpenMP c08, FP64) {what}{' (N=' + nmol + ')' if nmol else ''}, {steps} steps, {cores} threads"}
    # fallback: the oracle restatement (scalar, 1 core) — only when the reference binary did not travel
    from oracle.oracle import Oracle  # checker, used here only as the timed CPU baseline
    Lb, _ids, r, v = importlib.import_module("ls1-mardyn_amd.synth").bcc_box(20, rho=RHO, temp=TEMP)
    n = len(r)
    orc = Oracle(lj_components(inp).flat(), RC)
    cid = np.zeros(n, np.int32); q = np.tile([1., 0, 0, 0], (n, 1)); D = np.zeros((n, 3))
    out = orc.forces(r, q, cid, [Lb] * 3, True)
    F, M = out["F"].copy(), out["M"].copy()
    t0 = time.time()
    ksteps = 3
    for _ in range(ksteps):
        orc.step(DT, cid, r, v, q, D, F, M, np.array([Lb] * 3), True)
    dt = time.time() - t0
    return {"value": n * ksteps / dt, "unit": "particle-updates/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle/ls1_oracle.c scalar restatement, 1CLJ N={n}, {ksteps} steps, 1 thread"}


# ---- multi-site workloads (BASELINE.json configs[3] / configs[4]) ----------------------------------------------------------
ETHANE_RC = 32.1254   # cutoff of the reference's own ethane test (tests/golden MANIFEST, VectorizedCellProcessorTest)
ETHANE_DT = 0.5
MIXED_RC = 35.0
MIXED_DT = 0.0612     # ~2 fs in the reference's unit system (a0, 1000 u, E_h: time unit 32.7 fs; the shipped Argon example uses 0.0667516)
MIXED_CPU_N = 100     # cpu_baseline sample of the mixed workload: 2 * 100^3 = 2 000 000 molecules
MS_XML = """<?xml version='1.0' encoding='UTF-8'?>
<mardyn version="20100525">
  <refunits type="SI"><length unit="nm">0.1</length><mass unit="u">1</mass><energy unit="K">1</energy></refunits>
  <simulation type="MD">
    <integrator type="Leapfrog"><timestep unit="reduced">{dt}</timestep></integrator>
    <run><currenttime>0</currenttime><production><steps>{steps}</steps></production></run>
    <ensemble type="NVT">
      <temperature unit="reduced">{temp}</temperature>
      <domain type="box"><lx>{L}</lx><ly>{L}</ly><lz>{L}</lz></domain>
      <components>{components}</components>
      <phasespacepoint><file type="binary"><header>ms.header.xml</header><data>ms.dat</data></file></phasespacepoint>
    </ensemble>
    <algorithm>
      <parallelisation type="DomainDecomposition"></parallelisation>
      <datastructure type="LinkedCells"><cellsInCutoffRadius>1</cellsInCutoffRadius></datastructure>
      <cutoffs type="CenterOfMass"><radiusLJ unit="reduced">{rc}</radiusLJ></cutoffs>
      <electrostatic type="ReactionField"><epsilon>1.0e+10</epsilon></electrostatic>
    </algorithm>
    <output></output>
  </simulation>
</mardyn>
"""


def _fixture(name):
    """a reference test input kept as a fixture under tests/golden/inputs (plain or gzip)"""
    import gzip
    import shutil
    p = os.path.join(ROOT, "tests", "golden", "inputs", name)
    if os.path.exists(p):
        return p
    tmp = os.path.join(tempfile.gettempdir(), "ls1bench_" + name)
    if not os.path.exists(tmp):
        with gzip.open(p + ".gz", "rb") as fi, open(tmp + ".part", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        os.replace(tmp + ".part", tmp)
    return tmp


def ethane_fixture(inp):
    ps = inp.read_inp(_fixture("Ethan_equilibrated.inp"))
    return ps, None


def replicate_phase_space(inp, ps, k):
    """k x k x k periodic replication (what the reference's io/ReplicaGenerator.cpp does for its large multi-site boxes)"""
    n0 = len(ps.ids)
    shifts = np.array([[i, j, l] for i in range(k) for j in range(k) for l in range(k)], dtype=float) * ps.length
    r = (ps.r[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    t = lambda a: np.tile(a, (k ** 3,) + (1,) * (a.ndim - 1))  # noqa: E731
    q = ps.q / np.linalg.norm(ps.q, axis=1, keepdims=True)
    # time 0: the reference starts counting steps at round(time / dt) and --steps is an absolute step number (Simulation.cpp:911,1375)
    return inp.PhaseSpace(ps.components, ps.length * k, np.arange(1, n0 * k ** 3 + 1, dtype=np.uint64), t(ps.cid), r, t(ps.v), t(q),
                          t(ps.D), 0.0, ps.temperature)


def mixed_box(inp, n):
    """configs[4]: the five components of VectorizationMultiComponentMultiPotentials.inp in their INTEGRABLE form
    (synth.mixed5_components: every site of the fixture; the three components the fixture leaves without mass / repulsive core
    get a rigid three-centre LJ frame; mixing block written, reaction field on) on a jittered bcc lattice at the fixture's number
    density, component = (id - 1) mod 5, Maxwell velocities / angular momenta at the fixture's temperature."""
    ps = importlib.import_module("ls1-mardyn_amd.synth").mixed5_box(inp, n)
    return ps


def multi_gpu_self_check(comps, world, rank, local_rank, rehearse, torch, dist):
    """First action of an N > 1 run (ADVICE r1 / VERDICT r3 #9): a small box (2 * 24^3 molecules) split over the N ranks exactly as
    the timed box will be — every rank generates its sub-box, leaving molecules and halo copies travel through the same transport —
    against the SAME box held whole by this rank alone: forces of the molecules this rank owns (1e-12 of max|F|; not bitwise: the
    ranks' cell grids differ from the single domain's, so sums run in another order) and the global U_pot / virial.  A mismatch
    stops the run before anything is timed."""
    decomp = importlib.import_module("ls1-mardyn_amd.decomp")
    synth = importlib.import_module("ls1-mardyn_amd.synth")
    engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
    n = 24
    sim = decomp.build_strong_scaling_box(comps, RC, n, world, rank, local_rank, rho=RHO, temp=TEMP, stage_through_host=rehearse)
    macro = sim.initial_forces()
    g = sim.reduce_globals(macro, (0.0, 0.0, 0, 0))
    st = sim.engine.download_state()
    F = sim.engine.download_forces()["F"]
    sim.engine.close()
    L, ids, r, v = synth.bcc_box(n, rho=RHO, temp=TEMP)
    e1 = engine_mod.DeviceEngine(local_rank)
    e1.set_components(comps, RC)
    e1.set_domain([L, L, L])
    e1.upload(ids, np.zeros(len(ids), np.int32), r, v)
    e1.rebin(); e1.halo()
    u1, w1 = e1.forces(0)
    s1 = e1.download_state()
    F1 = e1.download_forces()["F"]
    e1.close()
    o1 = np.argsort(s1["ids"], kind="stable")
    pos = np.searchsorted(s1["ids"][o1], st["ids"])
    Fref = F1[o1][pos]
    ok_ids = bool(np.array_equal(s1["ids"][o1][pos], st["ids"]))
    df = float(np.max(np.abs(F - Fref)) / np.max(np.abs(F1))) if len(F) else 0.0
    du, dw = abs(g["upot"] - u1) / abs(u1), abs(g["virial"] - w1) / abs(w1)
    t = torch.tensor([df, du, dw, 0.0 if ok_ids else 1.0, float(len(F))], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    tm = t.clone()
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    ts = t.clone()
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    res = {"molecules": int(2 * n ** 3), "owned_molecules_summed_over_ranks": int(ts[4].item()), "forces_max_rel_over_ranks": float(tm[0].item()),
           "upot_rel": float(tm[1].item()), "virial_rel": float(tm[2].item()), "ids_match": tm[3].item() == 0.0,
           "what": "2*24^3 box split over the ranks (own sub-box generation, leaving + halo exchange through the run's transport) vs the same "
                   "box whole on every rank alone, before anything is timed"}
    if not (res["ids_match"] and res["owned_molecules_summed_over_ranks"] == res["molecules"] and res["forces_max_rel_over_ranks"] < 1e-11
            and res["upot_rel"] < 1e-11 and res["virial_rel"] < 1e-10):
        sys.exit(f"bench.py: the decomposed path disagrees with the single domain on the start-up check: {res}")
    return res


PMC_TRAFFIC = (("FETCH_SIZE",), ("WRITE_SIZE",))
PMC_COMPUTE = (("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE"),
               ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                "GRBM_GUI_ACTIVE"))


def live_pmc(argv_tail, groups, budget_s=300):
    """Hardware counters of the dominant force kernel measured LIVE: one short child run of this script per counter group
    under `rocprofv3 --pmc <group> --kernel-trace` (separate passes, no other trace domain — the recipe of
    MI355X_MICROARCH.md), per-launch means.  Returns ({counter: mean per launch}, kernel name) or (None, reason).  The
    children are separate processes started BEFORE they touch the GPU (no exec from a GPU-initialised process)."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "this run is itself being profiled (no nested profiler)"
    per = {}
    kernel = None
    for group in groups:
        td = tempfile.mkdtemp(prefix="ls1pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc"] + list(group) + ["--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable,
                                                   os.path.join(ROOT, "bench.py"), "--pmc-child"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=budget_s)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                return None, f"rocprofv3 pass {group[0]} failed (rc {res.returncode})"
            acc = {}
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                        if "k_force_" in name and "reduce" not in name:
                            acc.setdefault(name, {}).setdefault(r.get("Counter_Name"), []).append(float(r["Counter_Value"]))
            if not acc:
                return None, "no force kernel in the counter trace"
            # dominant force kernel = most launches (the list build / first-step kernels appear once or a few times)
            name = max(acc, key=lambda k: max(len(v) for v in acc[k].values()))
            kernel = kernel or name
            if name != kernel:
                return None, "the passes disagree on the dominant kernel"
            for cname, vals in acc[name].items():
                per[cname] = sum(vals) / len(vals)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 pass {group[0]} exceeded {budget_s} s"
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return per, kernel


def live_pmc_traffic(argv_tail, budget_s=300):
    """HBM traffic per launch with the guide's gfx950 corrections: both counters are reported in KB and FETCH_SIZE counts
    half of the bytes of 8 B/lane reads (x2; calibrated in profiles/ on a kernel of known traffic)."""
    per, kernel = live_pmc(argv_tail, PMC_TRAFFIC, budget_s)
    if per is None:
        return None, kernel
    return 2.0 * per["FETCH_SIZE"] * 1024.0 + per["WRITE_SIZE"] * 1024.0, kernel


def live_pmc_compute(argv_tail, avg_launch_s, budget_s=300):
    """VALU / LDS utilisation and the FP64 vector rate of the dominant force kernel from live counter passes (formulas of
    tools/pmc_summarize.py: SQ_* cycle counters in quad-cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
    per, kernel = live_pmc(argv_tail, PMC_COMPUTE, budget_s)
    if per is None:
        return None, kernel
    out = {"fp64_vector_peak_tflops": 78.6, "source": f"live: rocprofv3 --pmc child passes of this run, kernel {kernel}"}
    cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    flop = 64.0 * (2.0 * per.get("SQ_INSTS_VALU_FMA_F64", 0.0) + per.get("SQ_INSTS_VALU_ADD_F64", 0.0) + per.get("SQ_INSTS_VALU_MUL_F64", 0.0))
    out["fp64_flop_per_launch"] = flop
    out["fp64_tflops"] = flop / avg_launch_s / 1e12
    out["fp64_frac"] = out["fp64_tflops"] / 78.6
    if cyc > 0:
        out["kernel_cycles_under_profiler"] = cyc
        if "SQ_ACTIVE_INST_VALU" in per:
            out["valu_busy_frac_per_simd"] = per["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in per:
            out["lds_busy_frac_per_cu"] = per["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
            out["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(per["SQ_LDS_IDX_ACTIVE"], 1.0)
    if "SQ_INSTS_VALU" in per:
        out["valu_wave_insts_per_launch"] = per["SQ_INSTS_VALU"]
    return out, kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; mixed workload: 40)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10)")
    ap.add_argument("--n-per-dim", type=int, default=368,
                    help="bcc cells per dimension of the GLOBAL box (N = 2 n^3; 368 = the 10^8 box of the metric, 171 = configs[1])")
    ap.add_argument("--kernel", type=int, default=0, help="force kernel variant (LS1HIP_FK_*)")
    ap.add_argument("--cic", type=int, default=0, help="cells in cutoff (0 = engine default)")
    ap.add_argument("--split", type=int, default=0, help="lanes per molecule in the LDS LJ kernel (0 = engine default)")
    ap.add_argument("--overlap", type=int, default=-1, help="ls1hip_run halo mode: 0 single pass, 1 overlapped, 2 split sequential")
    ap.add_argument("--skin", type=float, default=0.2,
                    help="neighbour-list skin in sigma (list-reuse loop of ls1hip_run, single GPU); 0 = per-step search kernels")
    ap.add_argument("--precision", choices=("dp", "spdp", "spsp"), default="dp",
                    help="pair arithmetic of the list force pass: dp = FP64 (the metric's precision, default); spdp / spsp = the "
                         "reference's single-precision build modes (NOT the headline: dtype says so)")
    ap.add_argument("--nvt", action="store_true", help="velocity-scaling thermostat on the device (as every shipped example of the "
                                                        "reference; the headline metric is the NVE loop)")
    ap.add_argument("--no-fuse", action="store_true", help="separate integrator passes instead of the fused force pass")
    ap.add_argument("--decomp", action="store_true",
                    help="diagnostic: run the decomposed (multi-rank) step loop even with one rank")
    ap.add_argument("--loopback", action="store_true",
                    help="diagnostic (with --decomp): route the local periodic images through the RCCL transport "
                         "(send/recv to the own rank): the full multi-GPU exchange path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--local-rebuild", type=int, default=-1, choices=[-1, 0, 1],
                    help="diagnostic: 0 = rebuild the lists by the global displacement bound instead of the brick-neighbourhood one")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the two short rocprofv3 --pmc child passes that measure roofline.traffic live (the figure "
                         "of the committed profiles/ summary is reported instead)")
    ap.add_argument("--workload", choices=("lj", "ethane", "mixed"), default="lj",
                    help="lj = the metric's 1CLJ liquid (default); ethane = BASELINE configs[3]: the reference's equilibrated 2CLJ "
                         "ethane box replicated 10^3 = 9 826 000 molecules, full NVE step; mixed = configs[4]: the five-component LJ + "
                         "charge + dipole + quadrupole set in its integrable form (synth.mixed5_components) on the 171^3 bcc lattice "
                         "(10 000 422 molecules), full NVE step incl. rigid-body integration")
    ap.add_argument("--melt", type=int, default=-1,
                    help="untimed device steps BEFORE the warm-up steps (lj workload: default 200 — the lattice start melts, list "
                         "lengths spread; the timed window is then steady state)")
    ap.add_argument("--long-run", type=int, default=200,
                    help="lj workload, one GPU: steps of the secondary long window after the timed one (0 = none)")
    ap.add_argument("--no-align", action="store_true",
                    help="lj workload: do not place the timed window in the list lifetime (default: the melt phase is extended by a few "
                         "steps so that the K timed steps hold round(K * builds-per-step) list rebuilds, see config.timed_window)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    # The mixed set keeps the fixture's point multipoles (|mu| = 7.1 e a0 = 18 D, "modified to make the test more sensitive"): dipole
    # and charge sites sit 1-2 a0 off the molecules' centres and can meet inside the Lennard-Jones cores, so the liquid collapses
    # after ~6 time units (~100 steps at dt = 0.0612) WHATEVER the time step — the oracle and the reference binary show the same.
    # Its timed window therefore lies early (10 + 40 steps + 10 profiling steps = 3.7 time units from the lattice start).
    if args.steps is None:
        # mixed: the lists of this box live ~20 steps (skin 5 a0) and the first build is at step 0 — 10 warm-up + 40 timed steps hold
        # the rebuilds of steps ~20 and ~40: two per 40 steps, the steady-state density (`builds_in_timed_window`; a 20-step window
        # held none or one: +- 8 % on `value`)
        args.steps = 40 if args.workload == "mixed" else 50
    if args.warmup is None:
        args.warmup = 10
    if args.workload == "mixed" and args.steps + args.warmup > 60:
        sys.exit("bench.py --workload mixed: keep warm-up + timed steps <= 60 (the set's point multipoles collapse after ~100 steps, see --help)")
    if args.melt < 0:
        # every lj run melts, whatever the number of GPUs: the driver computes scaling efficiency from the per-N lines, which must
        # time the same physical state
        args.melt = 200 if args.workload == "lj" else 0
    if args.pmc_child:  # a counter pass: the same workload, a few steps, nothing else
        args.steps, args.warmup, args.no_cpu_baseline, args.no_live_pmc = 6, 2, True, True
    if args.workload != "lj":
        if args.gpus != 1 or args.decomp:
            sys.exit("bench.py --workload ethane|mixed runs on one GPU (multi-site lists serve single-rank domains)")
        if args.skin == 0.2:  # the default is in sigma of the LJ liquid; multi-site boxes are in atomic units
            args.skin = {"ethane": 6.0, "mixed": 5.0}[args.workload]  # (sweep: profiles/r4_ms_skin_sweep.txt — the cutoff filter of the force pass makes skin pairs cheap)

    # stdout carries ONE JSON line: libraries that print to file descriptor 1 (RCCL prints a version banner when the first
    # communicator is created) are sent to stderr for the duration of the run; the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    # rehearsal mode: several ranks on ONE GPU with a gloo transport staged through the host (RCCL needs a GPU per rank)
    rehearse = os.environ.get("LS1_BENCH_BACKEND", "nccl") == "gloo"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    inp = importlib.import_module("ls1-mardyn_amd.inp")
    comps = lj_components(inp)
    n = args.n_per_dim
    if world > 1 or args.decomp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        decomp = importlib.import_module("ls1-mardyn_amd.decomp")
        self_check = multi_gpu_self_check(comps, world, rank, local_rank, rehearse, torch, dist) if world > 1 and not args.pmc_child else None
        sim = decomp.build_strong_scaling_box(comps, RC, n, world, rank, local_rank, rho=RHO, temp=TEMP,
                                              cic=args.cic or None, kernel=args.kernel, stage_through_host=rehearse,
                                              loopback=args.loopback,
                                              skin=(args.skin if args.skin > 0 and not args.no_fuse and args.kernel != 1
                                                    and args.cic in (0, 1) else None))
        n_total = sim.n_global
        if args.loopback:  # rehearse the count exchange too (skipped otherwise when there is a single rank)
            sim.ex.force_count_exchange = True
    elif args.workload == "lj":
        engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
        synth = importlib.import_module("ls1-mardyn_amd.synth")
        L = synth.box_length(n, RHO)
        eng = engine_mod.DeviceEngine(local_rank)
        eng.set_components(comps, RC)
        if args.cic:
            eng.set_option("cells_in_cutoff", args.cic)
        eng.set_option("force_kernel", args.kernel)
        if args.skin > 0 and not args.no_fuse and args.kernel != 1 and args.cic in (0, 1):
            eng.set_verlet(args.skin)
        if args.split:
            eng.set_option("lj_split", args.split)
        eng.set_domain([L, L, L])
        N = 2 * n ** 3
        dev = torch.device("cuda", local_rank)
        eng.upload_begin(N)
        for ids_t, r_t, v_t in synth.bcc_chunks_device(torch, dev, n, rho=RHO, temp=TEMP):
            torch.cuda.synchronize()
            eng.upload_chunk_device(ids_t.numel(), ids_t.data_ptr(), 0, r_t.data_ptr(), v_t.data_ptr())
        del ids_t, r_t, v_t
        eng.upload_end()
        torch.cuda.empty_cache()
        assert eng.count()[0] == N
        eng.rebin(); eng.halo(); eng.forces(0)
        sim = None
        n_total = N
    else:
        # multi-site workloads: host-built phase space (fixtures of the reference's own tests), uploaded once
        engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
        eng = engine_mod.DeviceEngine(local_rank)
        if args.workload == "ethane":
            ps0, _ = ethane_fixture(inp)
            big = replicate_phase_space(inp, ps0, int(os.environ.get("LS1_BENCH_ETHANE_K", "10")))  # (K: diagnostics only)
            comps, rc_ms, L = big.components, ETHANE_RC, float(big.length[0])
            ids_h, cid_h, r_h, v_h, q_h, D_h = big.ids, big.cid, big.r, big.v, big.q, big.D
        else:
            big = mixed_box(inp, int(os.environ.get("LS1_BENCH_MIXED_N", "171")))  # (other sizes: diagnostics only)
            comps, rc_ms, L = big.components, MIXED_RC, float(big.length[0])
            big_T = float(big.temperature)
            ids_h, cid_h, r_h, v_h, q_h, D_h = big.ids, big.cid, big.r, big.v, big.q, big.D
        eng.set_components(comps, rc_ms)
        eng.set_option("force_kernel", args.kernel)
        if args.skin > 0 and args.kernel != 1:
            eng.set_verlet(args.skin)
        eng.set_domain([L, L, L])
        eng.upload(ids_h, cid_h, r_h, v_h, q_h, D_h)
        N = len(ids_h)
        del ids_h, cid_h, r_h, v_h, q_h, D_h, big
        eng.rebin(); eng.halo(); eng.forces(0)
        sim = None
        n_total = N
    if args.no_fuse:
        (sim.engine if sim is not None else eng).set_option("fuse_integration", 0)
    if args.overlap >= 0:
        (sim.engine if sim is not None else eng).set_option("overlap_halo", args.overlap)
    if args.nvt:
        (sim.engine if sim is not None else eng).set_thermostat(True, TEMP)
    if args.local_rebuild >= 0:
        (sim.engine if sim is not None else eng).set_option("local_rebuild", args.local_rebuild)
    if args.precision != "dp":
        (sim.engine if sim is not None else eng).set_option("precision", {"spdp": 1, "spsp": 2}[args.precision])

    step_dt = {"lj": DT, "ethane": ETHANE_DT, "mixed": MIXED_DT}[args.workload]

    def run(k):
        if sim is not None:
            return sim.run(DT, k, fuse=not args.no_fuse, lists=None if os.environ.get("LS1_BENCH_DECOMP_LISTS", "1") != "0" else False)
        return eng.run(step_dt, k)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    lattice = None
    melt_tail = None  # (list builds, steps) over the second half of the melt phase: the rebuild rate of the melted liquid
    if args.melt:
        if args.workload == "lj" and not args.pmc_child and args.melt >= 60:
            # secondary figure, for continuity with earlier rounds (which timed the lattice start): 10 + 30 of the melt steps are
            # timed as they go by — NOT the headline
            run(10)
            sync()
            t_l = time.perf_counter()
            run(30)
            sync()
            lattice = {"steps": 30, "ms_per_step": (time.perf_counter() - t_l) / 30 * 1e3,
                       "note": "steps 11-40 after the jittered-lattice start (what rounds 1-2 quoted); the headline `value` is the "
                               "steady state after the melt phase"}
            rest = args.melt - 40
            run(rest // 2)
            eb = sim.engine if sim is not None else eng
            mb0, ms0 = int(eb.get_option("verlet_builds")), int(eb.get_option("verlet_steps"))
            run(rest - rest // 2)
            melt_tail = (int(eb.get_option("verlet_builds")) - mb0, int(eb.get_option("verlet_steps")) - ms0)
        else:
            run(args.melt)
    e = sim.engine if sim is not None else eng
    window = None
    if args.workload == "lj" and world == 1 and sim is None and not args.pmc_child and not args.no_align and melt_tail \
            and melt_tail[0] >= 3 and e.get_option("verlet_lists"):
        # Place the K timed steps in the list lifetime.  A rebuild step costs ~3 ordinary steps and comes every P ~ 11 steps, so a
        # K-step window holds floor or ceil of K / P of them depending on where it starts (1 instead of the expected
        # 1.8 in the driver's 20 steps = +8 % on `value`).  The melt phase is therefore extended by a few untimed steps: run until
        # a rebuild has just happened, then on to the phase at which W warm-up steps + K timed steps hold round(K / P) rebuilds.
        P = melt_tail[1] / melt_tail[0]        # list lifetime in steps (second half of the melt phase)
        extra = 0
        b0 = int(e.get_option("verlet_builds"))
        for _ in range(40):                      # step until a rebuild has just happened (phase 0)
            run(1)
            extra += 1
            if int(e.get_option("verlet_builds")) > b0:
                break
        n_target = int(args.steps / P + 0.5)
        lo_phi, hi_phi = max(n_target * P - args.steps, 0.0), min((n_target + 1) * P - args.steps, P)
        phi = 0.5 * (lo_phi + hi_phi) if hi_phi > lo_phi else 0.0   # steps since the last rebuild at the START of the timed window
        # (the rebuild step just run is step 1 of the new lists' life: phase 1 now, phase 1 + pre + W when the timed window starts)
        pre = int(round(phi - args.warmup - 1)) % max(int(round(P)), 1)
        if pre:
            run(pre)
            extra += pre
        window = {"aligned": True, "list_lifetime_steps_estimate": P, "rebuilds_expected_in_K_steps": args.steps / P,
                  "rebuilds_targeted": n_target, "extra_untimed_steps_for_alignment": extra,
                  "note": "the melt phase was extended so that the K timed steps hold round(K / lifetime) list rebuilds; `value` is "
                          "still exactly the K timed steps; `steady_state_value` (a 200-step window) is the long-run rate"}
    run(args.warmup)
    e.timing_reset()
    e.timing_enable(2)  # the timed region carries the HIP-event pairs of the force launches only (roofline.avg_launch_ms)
    sync()
    builds_before = int(e.get_option("verlet_builds"))
    t0 = time.perf_counter()
    last = run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    builds_in_window = (int(e.get_option("verlet_builds")) - builds_before) if builds_before is not None else None
    e.timing_enable(0)
    force_ms, force_n = e.timing("force")
    # per-phase device times: a few extra steps with every phase timed, OUTSIDE the timed region
    e.timing_reset()
    e.timing_enable(1)
    nprof = max(2, min(10, args.steps))
    run(nprof)
    sync()
    e.timing_enable(0)
    long_run = None
    if args.workload in ("lj", "ethane") and world == 1 and not args.pmc_child and args.long_run > 0:  # (mixed: bounded window, see MIXED_DT)
        # secondary figure: a window long enough to average over the list lifetimes.  A rebuild step costs ~3 ordinary steps and comes
        # every ~11 steps, so a K-step window holds floor or ceil of K / 10.8 of them: +- 2 % on 50 steps, +- 5 % on the driver's 20
        b0 = int(e.get_option("verlet_builds"))
        sync()
        t_l = time.perf_counter()
        run(args.long_run)
        sync()
        dt_l = time.perf_counter() - t_l
        long_run = {"steps": args.long_run, "ms_per_step": dt_l / args.long_run * 1e3,
                    "list_builds": int(e.get_option("verlet_builds")) - b0,
                    "note": "run AFTER the timed window and the profiling steps; NOT the headline — the headline's K steps hold an "
                            "integer number of list rebuilds (see `builds_in_timed_window`), this window averages over them"}
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    world_seen, per_rank = world, None
    if world > 1:
        import torch.distributed as dist
        world_seen = dist.get_world_size()   # what the communicator saw (the driver checks it against --gpus)
        t = torch.zeros(world_seen, dtype=torch.int64, device="cpu" if rehearse else "cuda")
        t[dist.get_rank()] = int(e.count()[0])
        dist.all_reduce(t)
        per_rank = [int(x) for x in t.tolist()]
    integ_ms, _ = e.timing("integrate")
    rebin_ms, _ = e.timing("rebin")
    halo_ms, _ = e.timing("halo")
    build_ms, build_n = e.timing("build")
    n_local = e.count()[0]
    # HBM traffic of the force kernel from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), taken
    # offline on this exact workload and committed under profiles/ (PMC collection cannot run inside the timed loop)
    fused_on = bool(e.get_option("fuse_integration")) and bool(e.get_option("can_fuse_integration"))
    n_fused = (args.steps - 1) if fused_on else 0  # the last step of a run is unfused (F and kinetic sums are needed)
    if args.workload == "lj":
        alg_bytes_total = n_local * (FUSED_BYTES_PER_MOLECULE * n_fused + FORCE_BYTES_PER_MOLECULE * (args.steps - n_fused))
        # whole step per GPU: plain = force 48 + integrator 120 + re-bin 124 = 292 B; fused = 96 + 124 = 220 B per molecule
        step_bytes_total = n_local * ((FUSED_BYTES_PER_MOLECULE + 124.0) * n_fused + STEP_BYTES_PER_MOLECULE * (args.steps - n_fused))
    else:
        # multi-site force pass (48 B + 56 B for the orientation in and the torque out): read r 24 + q 32, write F 24 + M 24
        fused_on = bool(e.get_option("fuse_integration")) and bool(e.get_option("can_fuse_rigid_lists")) and e.get_option("verlet_builds") > 0
        n_fused = (args.steps - 1) if fused_on else 0
        alg_bytes_total = n_local * (MS_FUSED_BYTES_PER_MOLECULE * n_fused + MS_FORCE_BYTES_PER_MOLECULE * (args.steps - n_fused))
        # unfused: + rigid-body integrator pass: read r v q D F M (152 B), write r v q D (104 B)
        step_bytes_total = n_local * (MS_FUSED_BYTES_PER_MOLECULE * n_fused + (MS_FORCE_BYTES_PER_MOLECULE + 256.0) * (args.steps - n_fused))
    alg_bytes_per_launch = alg_bytes_total / max(force_n, 1)
    traffic = None
    pmc_extra = {}
    try:
        with open(os.path.join(ROOT, "profiles", PMC_SUMMARY)) as fh:
            js = json.load(fh)
        pm = js["force_kernel"]
        if args.workload == "lj" and world == 1 and js.get("molecules") == n_local and e.get_option("cells_in_cutoff") == 1 and \
                bool(js.get("neighbour_lists")) == (e.get_option("verlet_builds") > 0) and \
                e.get_option("force_kernel") in (0, 2) and e.get_option("lj_split") == 0 and \
                abs(pm["algorithmic_bytes_per_launch"] / alg_bytes_per_launch - 1.0) < 0.03:
            traffic = pm["traffic_bytes_per_launch"]
            pmc_extra = {k: pm[k] for k in ("fp64_flop_per_launch", "valu_busy_frac_per_simd", "mfma_busy_frac",
                                            "lds_busy_frac_per_cu", "l2_hit_rate") if k in pm}
    except Exception:
        traffic = None
    traffic_source = f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc passes of this workload, committed)" if traffic is not None else None
    live_compute = None
    if rank == 0 and world == 1 and not args.no_live_pmc and not args.decomp:
        tail = ["--n-per-dim", str(n), "--skin", str(args.skin), "--kernel", str(args.kernel), "--cic", str(args.cic),
                "--split", str(args.split), "--workload", args.workload, "--melt", str(min(args.melt, 40)),
                "--precision", args.precision] + (["--no-fuse"] if args.no_fuse else []) + \
               (["--nvt"] if args.nvt else [])
        live, what = live_pmc_traffic(tail)
        live_compute, _what2 = live_pmc_compute(tail, force_ms / 1e3 / max(force_n, 1))
        if live is not None:
            # the child runs 6 timed steps (5 fused + 1 plain launch): same kernel, per-launch mean dominated by the fused form
            traffic = live
            traffic_source = f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run, kernel {what}"
        elif traffic_source is not None:
            traffic_source += f" [live passes unavailable: {what}]"
    if rank == 0:
        avg_force_s = force_ms / 1e3 / max(force_n, 1)
        achieved = alg_bytes_total / (force_ms / 1e3) / 1e9
        value = n_total * args.steps / elapsed
        if args.workload == "lj":
            metric = BASELINE_METRIC if n == 368 else f"particle-updates/sec (whole node), N={n_total} LJ liquid Argon, rc=2.5\u03c3"
            wl = (f"1CLJ Lennard-Jones liquid, N={n_total} = 2*{n}^3 (global box, split over {world} GPU(s)), "
                  f"rho*={RHO}, rc={RC} sigma, dt={DT}, T*={TEMP}, NVE full time step (kick-drift, re-bin, halo, "
                  f"forces, kick) with per-step U_pot / virial / sum mv^2, FP64")
        elif args.workload == "ethane":
            metric = f"particle-updates/sec (whole node), N={n_total} 2CLJ ethane, rc={ETHANE_RC} (BASELINE configs[3])"
            wl = (f"2CLJ ethane (two LJ centres, rigid rotor): the reference's Ethan_equilibrated box (9 826 molecules, L=571.607759, "
                  f"rc={ETHANE_RC}) replicated 10^3 = {n_total} molecules, dt={ETHANE_DT}, NVE full time step (rigid-body kick-drift, "
                  f"re-bin + halo + list build on rebuild steps, site forces + torques, kick) with per-step U_pot / virial / "
                  f"sum mv^2 / sum Iw^2, FP64")
        else:
            metric = f"particle-updates/sec (whole node), N={n_total} five-component LJ+charge+dipole+quadrupole set, rc={MIXED_RC} (BASELINE configs[4])"
            wl = (f"the five components of VectorizationMultiComponentMultiPotentials.inp (LJ + charge + dipole + quadrupole sites) in "
                  f"their integrable form (synth.mixed5_components: every site of the fixture + a rigid three-centre LJ frame as mass "
                  f"carrier / repulsive core for the three components the fixture leaves without; mixing block xi = eta = 1, eps_RF = "
                  f"1e10) on a jittered bcc lattice at the fixture's number density, component = (id - 1) mod 5, N={n_total}, "
                  f"rc={MIXED_RC}, dt={MIXED_DT}, T={big_T}; NVE full time step (rigid-body kick-drift incl. asymmetric tops, "
                  f"re-bin + halo + list build on rebuild steps, site forces + torques of all ten site-type pairs, kick) with per-step "
                  f"U_pot / virial / sum mv^2 / sum Iw^2, FP64, lattice start (the timed window ends 2.1 time units after it: the "
                  f"fixture's 18-debye point dipoles sit off-centre and collapse the liquid after ~6 time units at any time step, in "
                  f"the reference as here)")
        out = {
            "metric": metric,
            "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": {"dp": "f64", "spdp": "f32 pair arithmetic, f64 sums and integration (SPDP mode, not the metric's precision)",
                                           "spsp": "f32 pair arithmetic and sums, f64 integration (SPSP mode, not the metric's precision)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": wl,
                       "untimed_steps_before_the_timed_window": {"melt": args.melt, "warmup": args.warmup},
                       "molecules_per_gpu": n_local, "decomposition": getattr(sim, "grid_desc", "single GPU, periodic images local"),
                       "force_kernel": e.get_option("force_kernel"), "cells_in_cutoff": e.get_option("cells_in_cutoff"),
                       "neighbour_lists": ({"skin": args.skin, "list_builds": e.get_option("verlet_builds"),
                                            "list_steps": e.get_option("verlet_steps"),
                                            "note": "lists, binning and halo slots reused until the device-side displacement "
                                                    "bound exceeds skin/2 (counts since start incl. warm-up and profiling steps)"}
                                           if e.get_option("verlet_lists") else None),
                       "ensemble": ("NVT: velocity-scaling thermostat on the device, post-force kick + kinetic sum inside the force "
                                    "pass, scaling folded into the kick + drift pass" if args.nvt else "NVE"),
                       "integration": ("fused into the force pass between steps (reduced-memory mode), last step separate"
                                       if fused_on else "separate integrator passes")},
            "roofline": {"bound": "hbm", "kernel": "pair-force traversal (k_force_*)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "avg_launch_ms": avg_force_s * 1e3, "launches": int(force_n),
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "full_step_frac": step_bytes_total / elapsed / 1e9 / HBM_PEAK_GBS},
            "device_ms_per_step": {"force": force_ms / args.steps, "integrate": integ_ms / nprof,
                                   "rebin": rebin_ms / nprof, "halo": halo_ms / nprof, "list_build": build_ms / nprof,
                                   "list_build_ms_per_build": (build_ms / build_n if build_n else None),
                                   "list_builds_per_step": (e.get_option("verlet_builds") / max(e.get_option("verlet_steps"), 1)
                                                            if e.get_option("verlet_lists") else 0.0),
                                   "note": "force: mean over the timed steps; the other phases: mean over the profiling steps "
                                           "after the timed window (rebuild steps included at their frequency)"},
            "last_step": {k: (float(v_) if not isinstance(v_, int) else v_) for k, v_ in last.items()} if isinstance(last, dict) else None,
        }
        if lattice is not None:
            lattice["value"] = n_total / (lattice["ms_per_step"] * 1e-3)
            out["lattice_start"] = lattice
        if builds_in_window is not None and out["config"]["neighbour_lists"]:
            out["config"]["neighbour_lists"]["builds_in_timed_window"] = builds_in_window
        if long_run:
            long_run["value"] = n_total / (long_run["ms_per_step"] * 1e-3)
            out["long_run"] = long_run
            # the number to quote: the rate over a window that averages over the list lifetimes
            out["steady_state_value"] = long_run["value"]
            out["value_over_steady_state"] = value / long_run["value"]
            rate = long_run["list_builds"] / long_run["steps"]
            tw = window or {"aligned": False}
            tw.update({"rebuilds_in_timed_window": builds_in_window, "rebuilds_expected_at_the_long_run_rate": args.steps * rate,
                       "long_run_rebuilds_per_step": rate})
            out["config"]["timed_window"] = tw
        if live_compute is not None:
            out["roofline"]["compute"] = live_compute
        elif pmc_extra:
            # the kernel is issue-bound, not HBM-bound: the FP64 vector rate actually sustained (PMC instruction counts
            # of this workload, profiles/, over the live launch time) next to the 78.6 TFLOP/s FP64 vector peak, and the
            # pipe utilisations of the same PMC passes
            comp = {"fp64_vector_peak_tflops": 78.6}
            if "fp64_flop_per_launch" in pmc_extra:
                comp["fp64_tflops"] = pmc_extra["fp64_flop_per_launch"] / avg_force_s / 1e12
                comp["fp64_frac"] = comp["fp64_tflops"] / 78.6
            comp.update({k: v for k, v in pmc_extra.items() if k != "fp64_flop_per_launch"})
            out["roofline"]["compute"] = comp
        if world > 1:
            out["rccl_world_size"] = world_seen
            out["multi_gpu_self_check"] = self_check
            out["config"]["molecules_per_gpu_by_rank"] = per_rank
            out["config"]["transport"] = "gloo staged through the host (rehearsal)" if rehearse else "RCCL (torch.distributed nccl backend)"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(workload=args.workload)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.decomp:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
