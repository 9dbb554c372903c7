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


def cpu_baseline(n_per_dim=171, steps=10, budget_s=150):
    """Reference OpenMP CPU path on the host cores: same liquid (bcc lattice start from the reference's own
    CubicGridGenerator), bounded sample N = 2*n^3, `steps` steps."""
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("LS1_BENCH_CPU_THREADS", "16")))
    N = 2 * n_per_dim ** 3
    L = (N / RHO) ** (1.0 / 3.0)
    binary = os.path.join(ROOT, "oracle", "_ref", "MarDyn")
    if os.path.exists(binary):
        with tempfile.TemporaryDirectory() as td:
            cfg = os.path.join(td, "config.xml")
            with open(cfg, "w") as fh:
                fh.write(MARDYN_XML.format(dt=DT, steps=steps, temp=TEMP, L=repr(L), rho=RHO, rc=RC))
            env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
            try:
                out = subprocess.run([binary, cfg, "--steps", str(steps), "--final-checkpoint=0"], cwd=td, env=env,
                                     capture_output=True, text=True, timeout=budget_s).stdout
            except subprocess.TimeoutExpired:
                out = ""
            m = re.search(r"Simulation speed:\s*([0-9.eE+-]+)\s*Molecule-updates per second", out)
            nm = re.search(r"[Nn]umber of molecules[^0-9]*([0-9]+)", out)
            if m:
                return {"value": float(m.group(1)), "unit": "particle-updates/s", "cores": cores, "kind": "reference",
                        "cpu": cpu_model(), "host_cpus_visible": len(os.sched_getaffinity(0)),
                        "sample": f"reference MarDyn (AVX2, OpenMP c08, FP64) 1CLJ N={nm.group(1) if nm else N} bcc rho*={RHO} rc={RC}, {steps} steps, {cores} threads"}
    # fallback: the oracle restatement (scalar, 1 core) — only when the reference binary did not travel
    from oracle.oracle import Oracle  # checker, used here only as the timed CPU baseline
    inp = importlib.import_module("ls1-mardyn_amd.inp")
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


def live_pmc_traffic(argv_tail, budget_s=240):
    """HBM traffic of the dominant force kernel measured LIVE: two short child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (separate passes, no other trace domain — the recipe of
    MI355X_MICROARCH.md), per-launch means, with the guide's gfx950 corrections: both counters are reported in KB and
    FETCH_SIZE counts half of the bytes of 8 B/lane reads (x2; calibrated in profiles/ on a kernel of known traffic).
    Returns (bytes per launch, kernel name) or (None, reason).  The children are separate processes started BEFORE they
    touch the GPU (no exec from a GPU-initialised process)."""
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
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="ls1pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=budget_s)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                return None, f"rocprofv3 pass {counter} failed (rc {res.returncode})"
            acc = {}
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") != counter:
                            continue
                        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                        if "k_force_" in name and "reduce" not in name:
                            acc.setdefault(name, []).append(float(r["Counter_Value"]))
            if not acc:
                return None, "no force kernel in the counter trace"
            # dominant force kernel = most launches (the list build / first-step kernels appear once or a few times)
            name = max(acc, key=lambda k: len(acc[k]))
            kernel = kernel or name
            if name != kernel:
                return None, "the two passes disagree on the dominant kernel"
            per[counter] = sum(acc[name]) / len(acc[name])
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 pass {counter} exceeded {budget_s} s"
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return 2.0 * per["FETCH_SIZE"] * 1024.0 + per["WRITE_SIZE"] * 1024.0, kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:  # a counter pass: the same workload, a few steps, nothing else
        args.steps, args.warmup, args.no_cpu_baseline, args.no_live_pmc = 6, 2, True, True

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
        sim = decomp.build_strong_scaling_box(comps, RC, n, world, rank, local_rank, rho=RHO, temp=TEMP,
                                              cic=args.cic or None, kernel=args.kernel, stage_through_host=rehearse,
                                              loopback=args.loopback,
                                              skin=(args.skin if args.skin > 0 and not args.no_fuse and args.kernel != 1
                                                    and args.cic in (0, 1) else None))
        n_total = sim.n_global
        if args.loopback:  # rehearse the count exchange too (skipped otherwise when there is a single rank)
            sim.ex.force_count_exchange = True
    else:
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

    def run(k):
        if sim is not None:
            return sim.run(DT, k, fuse=not args.no_fuse, lists=None if os.environ.get("LS1_BENCH_DECOMP_LISTS", "1") != "0" else False)
        return eng.run(DT, k)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    e = sim.engine if sim is not None else eng
    e.timing_reset()
    e.timing_enable(2)  # the timed region carries the HIP-event pairs of the force launches only (roofline.avg_launch_ms)
    sync()
    t0 = time.perf_counter()
    last = run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    e.timing_enable(0)
    force_ms, force_n = e.timing("force")
    # per-phase device times: a few extra steps with every phase timed, OUTSIDE the timed region
    e.timing_reset()
    e.timing_enable(1)
    nprof = max(2, min(10, args.steps))
    run(nprof)
    sync()
    e.timing_enable(0)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    integ_ms, _ = e.timing("integrate")
    rebin_ms, _ = e.timing("rebin")
    halo_ms, _ = e.timing("halo")
    n_local = e.count()[0]
    # HBM traffic of the force kernel from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), taken
    # offline on this exact workload and committed under profiles/ (PMC collection cannot run inside the timed loop)
    fused_on = bool(e.get_option("fuse_integration")) and bool(e.get_option("can_fuse_integration"))
    n_fused = (args.steps - 1) if fused_on else 0  # the last step of a run is unfused (F and kinetic sums are needed)
    alg_bytes_total = n_local * (FUSED_BYTES_PER_MOLECULE * n_fused + FORCE_BYTES_PER_MOLECULE * (args.steps - n_fused))
    alg_bytes_per_launch = alg_bytes_total / max(force_n, 1)
    # whole step per GPU: plain = force 48 + integrator 120 + re-bin 124 = 292 B; fused = 96 + 124 = 220 B per molecule
    step_bytes_total = n_local * ((FUSED_BYTES_PER_MOLECULE + 124.0) * n_fused + STEP_BYTES_PER_MOLECULE * (args.steps - n_fused))
    traffic = None
    pmc_extra = {}
    try:
        with open(os.path.join(ROOT, "profiles", PMC_SUMMARY)) as fh:
            js = json.load(fh)
        pm = js["force_kernel"]
        if world == 1 and js.get("molecules") == n_local and e.get_option("cells_in_cutoff") == 1 and \
                bool(js.get("neighbour_lists")) == (e.get_option("verlet_builds") > 0) and \
                e.get_option("force_kernel") in (0, 2) and e.get_option("lj_split") == 0 and \
                abs(pm["algorithmic_bytes_per_launch"] / alg_bytes_per_launch - 1.0) < 0.03:
            traffic = pm["traffic_bytes_per_launch"]
            pmc_extra = {k: pm[k] for k in ("fp64_flop_per_launch", "valu_busy_frac_per_simd", "mfma_busy_frac",
                                            "lds_busy_frac_per_cu", "l2_hit_rate") if k in pm}
    except Exception:
        traffic = None
    traffic_source = f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc passes of this workload, committed)" if traffic is not None else None
    if rank == 0 and world == 1 and not args.no_live_pmc and not args.decomp:
        tail = ["--n-per-dim", str(n), "--skin", str(args.skin), "--kernel", str(args.kernel), "--cic", str(args.cic),
                "--split", str(args.split)] + (["--no-fuse"] if args.no_fuse else [])
        live, what = live_pmc_traffic(tail)
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
        out = {
            "metric": BASELINE_METRIC if n == 368 else f"particle-updates/sec (whole node), N={n_total} LJ liquid Argon, rc=2.5\u03c3",
            "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": {"dp": "f64", "spdp": "f32 pair arithmetic, f64 sums and integration (SPDP mode, not the metric's precision)",
                                           "spsp": "f32 pair arithmetic and sums, f64 integration (SPSP mode, not the metric's precision)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"1CLJ Lennard-Jones liquid, N={n_total} = 2*{n}^3 (global box, split over {world} GPU(s)), "
                                   f"rho*={RHO}, rc={RC} sigma, dt={DT}, T*={TEMP}, NVE full time step (kick-drift, re-bin, halo, "
                                   f"forces, kick) with per-step U_pot / virial / sum mv^2, FP64",
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
                                   "rebin": rebin_ms / nprof, "halo": halo_ms / nprof},
            "last_step": {k: (float(v_) if not isinstance(v_, int) else v_) for k, v_ in last.items()} if isinstance(last, dict) else None,
        }
        if pmc_extra:
            # the kernel is issue-bound, not HBM-bound: the FP64 vector rate actually sustained (PMC instruction counts
            # of this workload, profiles/, over the live launch time) next to the 78.6 TFLOP/s FP64 vector peak, and the
            # pipe utilisations of the same PMC passes
            comp = {"fp64_vector_peak_tflops": 78.6}
            if "fp64_flop_per_launch" in pmc_extra:
                comp["fp64_tflops"] = pmc_extra["fp64_flop_per_launch"] / avg_force_s / 1e12
                comp["fp64_frac"] = comp["fp64_tflops"] / 78.6
            comp.update({k: v for k, v in pmc_extra.items() if k != "fp64_flop_per_launch"})
            out["roofline"]["compute"] = comp
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.decomp:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
