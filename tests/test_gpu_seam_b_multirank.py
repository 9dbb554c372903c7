"""Multi-rank seam B: the decomposed path entered from the UNMODIFIED reference driver.

oracle/_ref/MarDyn_hipB constructs `DomainDecompHip : public DomainDecompBase` at Simulation.cpp:1356 (mapped by
ls1-mardyn_amd/host/seam_b_register.h) next to the device container and integrator.  Here 2 and 4 MarDyn_hipB PROCESSES
(rank / world size from the environment, as any launcher exports them) share the one GPU of the test box through the
host-staged mailbox transport (RCCL refuses two ranks on one device; on a multi-GPU node the same class runs RCCL over xGMI):
each rank owns a sub-box of the regular rank grid (DomainDecomposition.cpp:114-123), reads the same phase-space file and keeps
its own molecules, exchanges leaving molecules and halo copies per step through the export / import entry points, and takes
part in every global reduction of the driver (Domain::calculateGlobalValues, Domain.cpp:151-181) through the collComm*
virtuals.  Rank 0 must print the per-step T / U_pot / p of the unmodified single-process reference binary, and the union of the
ranks' final checkpoints must hold the reference's molecules."""
import gzip
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from golden_io import GOLDEN
from test_gpu_seam_a import HEAD, LJ1, REF, _run
from test_gpu_seam_b import HIPB, _restart_records

pytestmark = pytest.mark.gpu


def _launch_ranks(world, cfg_dir_of_rank, steps, comm_dir, extra_env=None):
    procs = []
    for r in range(world):
        env = dict(os.environ, OMP_NUM_THREADS="2", RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LS1HIP_DEVICE="0",
                   LS1HIP_TRANSPORT="mailbox", LS1HIP_COMM_DIR=comm_dir)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([HIPB, "config.xml", "--steps", str(steps), "--final-checkpoint=1"], cwd=cfg_dir_of_rank[r],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()  # exactly the processes started here
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f"rank {r}: rc {rc}\n{o[-3000:]}\n{e[-2000:]}"
    return outs


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPB)), reason="oracle/_ref binaries not built")
# lists: the multi-rank seam in LIST MODE — the default since round 4 (between two rebuilds only positions travel, the ranks decide a
# rebuild together); skin 0.06 makes the 16-step run rebuild several times, the default skin (0.08 r_c) outlives it; "0" = search
# every step.  The 300-step runs see molecules change ranks (at list rebuilds only).  The final checkpoint is written from a
# snapshot in which a molecule that still awaits its migration on the device is handed to the rank whose box it lies in: every
# rank's file must hold molecules INSIDE ITS OWN BOX only, the union every molecule once.
# overlap: the inner pass queued ahead of the halo phase (default) vs one complete pass behind a blocking exchange (LS1HIP_OVERLAP=0).
@pytest.mark.parametrize("world,lists,steps,overlap", [(2, "0.06", 16, 1), (2, "0.06", 300, 1), (4, "0.06", 300, 1), (2, "default", 16, 1),
                                                       (2, "0", 16, 1), (2, "0", 16, 0), (2, "0.06", 40, 0), (4, "default", 37, 1)])
def test_reference_driver_decomposed_over_ranks(tmp_path, world, lists, steps, overlap):
    src = os.path.join(GOLDEN, "inputs", "synthetic_bcc1clj_20.inp.gz")  # 16 000 molecules, L = 27.3 sigma
    with gzip.open(src, "rb") as fi, open(tmp_path / "bcc.inp", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    L = None
    with open(tmp_path / "bcc.inp") as fh:
        for ln in fh:
            if ln.strip().startswith("Length"):
                L = float(ln.split()[1])
                break
    cfg = HEAD.format(dt=0.002, steps=steps, temp=0.95, L=repr(L), rc=2.5, components=LJ1,
                      phasespace='<file type="ASCII">bcc.inp</file>')
    # single-process reference
    dref = tmp_path / "ref"
    dref.mkdir()
    shutil.copy(tmp_path / "bcc.inp", dref / "bcc.inp")
    (dref / "config.xml").write_text(cfg)
    ref_rows, _ = _run(REF, "config.xml", str(dref), steps, final_checkpoint=1)
    # world ranks of the device build
    dirs = []
    for r in range(world):
        d = tmp_path / f"rank{r}"
        d.mkdir()
        shutil.copy(tmp_path / "bcc.inp", d / "bcc.inp")
        (d / "config.xml").write_text(cfg)
        dirs.append(str(d))
    comm = tempfile.mkdtemp(prefix="ls1hip_comm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        env = {"LS1HIP_OVERLAP": str(overlap)}
        if lists != "default":
            env["LS1HIP_SKIN"] = lists
        outs = _launch_ranks(world, dirs, steps, comm, env)
    finally:
        shutil.rmtree(comm, ignore_errors=True)
    log0 = outs[0][1]
    assert f"DomainDecompHip: rank 0 of {world}" in log0 and "transport mailbox" in log0
    m = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", log0)
    assert m, log0[-2000:]
    if lists == "0":
        assert m.group(1) == "off"
    else:
        builds, evals = int(m.group(2)), int(m.group(3))
        assert m.group(1) == "on" and evals == steps + 1 and 1 <= builds < evals, (builds, evals)
        if lists == "0.06":
            assert builds >= 3, builds
    rows = re.findall(r"Simstep = (\d+)\s+T = (\S+)\s+U_pot = (\S+)\s+p = (\S+)", log0)
    hip_rows = np.array([[float(x) for x in r[1:]] for r in rows])
    assert len(hip_rows) >= steps
    # the driver prints 6 significant digits
    assert np.allclose(hip_rows[:steps], ref_rows[:steps], rtol=2e-5, atol=1e-12), (ref_rows[:steps], hip_rows[:steps])
    # every rank reports the same global values
    for r in range(1, world):
        rr = re.findall(r"Simstep = (\d+)\s+T = (\S+)\s+U_pot = (\S+)\s+p = (\S+)", outs[r][1])
        assert [x[1:] for x in rr[:steps]] == [x[1:] for x in rows[:steps]], r
    # final checkpoints: the union of the ranks' molecules == the reference's, every molecule exactly once
    fr = [f for f in os.listdir(dref) if f.endswith(".restart.dat")]
    assert fr
    a = _restart_records(dref / fr[0])
    b = {}
    per_rank = []
    for d in dirs:
        f = [x for x in os.listdir(d) if x.endswith(".restart.dat")]
        assert f, d
        recs = _restart_records(os.path.join(d, f[0]))
        assert not (recs.keys() & b.keys()), "a molecule is owned by two ranks"
        b.update(recs)
        per_rank.append(len(recs))
    assert a.keys() == b.keys() and len(a) == 16000
    assert all(n > 0 for n in per_rank)
    # every molecule of a rank's file lies inside that rank's box of the regular grid (largest factor along x: 2 -> 2x1x1, 4 -> 2x2x1)
    grid = {2: (2, 1, 1), 4: (2, 2, 1)}[world]
    for r, d in enumerate(dirs):
        f = [x for x in os.listdir(d) if x.endswith(".restart.dat")]
        pos = np.array([v[:3] for v in _restart_records(os.path.join(d, f[0])).values()])
        c = (r % grid[0], (r // grid[0]) % grid[1], r // (grid[0] * grid[1]))
        for k in range(3):
            lo, hi = c[k] * L / grid[k], (c[k] + 1) * L / grid[k]
            assert pos[:, k].min() >= lo - 1e-12 and pos[:, k].max() < hi + 1e-12, (r, k, pos[:, k].min(), pos[:, k].max(), lo, hi)
    ids = sorted(a)
    A, B = np.array([a[i] for i in ids]), np.array([b[i] for i in ids])
    dr = A[:, :3] - B[:, :3]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-7 * L
    assert np.max(np.abs(A[:, 3:6] - B[:, 3:6])) < 1e-6 * np.max(np.abs(A[:, 3:6]))
    if steps >= 300:
        assert len(set(per_rank)) > 1, per_rank  # molecules have migrated
    print(f"[seam B, {world} ranks] molecules per rank {per_rank}")


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPB)), reason="oracle/_ref binaries not built")
def test_intermediate_checkpoints_of_a_decomposed_list_mode_run(tmp_path):
    """A reader of the host mirror BETWEEN two list rebuilds of a multi-rank run (CheckpointWriter with a write frequency): molecules
    that await their migration on the device (up to skin / 2 outside their owner's box) are handed to the right rank for the
    snapshot — the union of the ranks' intermediate checkpoints is the reference's checkpoint of that step, each molecule once and
    inside the box of the rank that wrote it.  (Round 3 stopped such a run with an explanation.)"""
    world, steps = 2, 240  # (the lattice start keeps 0.3 sigma clear of the rank boundaries: the first molecules arrive after ~100 steps)
    src = os.path.join(GOLDEN, "inputs", "synthetic_bcc1clj_20.inp.gz")
    with gzip.open(src, "rb") as fi, open(tmp_path / "bcc.inp", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    L = float(next(ln.split()[1] for ln in open(tmp_path / "bcc.inp") if ln.strip().startswith("Length")))
    plugin = ('<output><outputplugin name="CheckpointWriter"><type>ASCII</type><writefrequency>57</writefrequency>'
              '<outputprefix>cp</outputprefix></outputplugin></output>')
    # a hot liquid: molecules cross the rank boundary within a list lifetime
    cfg = HEAD.format(dt=0.002, steps=steps, temp=1.5, L=repr(L), rc=2.5, components=LJ1,
                      phasespace='<file type="ASCII">bcc.inp</file>').replace("<output></output>", plugin)
    dref = tmp_path / "ref"
    dref.mkdir()
    shutil.copy(tmp_path / "bcc.inp", dref / "bcc.inp")
    (dref / "config.xml").write_text(cfg)
    ref_rows, _ = _run(REF, "config.xml", str(dref), steps, final_checkpoint=0)
    dirs = []
    for r in range(world):
        d = tmp_path / f"rank{r}"
        d.mkdir()
        shutil.copy(tmp_path / "bcc.inp", d / "bcc.inp")
        (d / "config.xml").write_text(cfg)
        dirs.append(str(d))
    comm = tempfile.mkdtemp(prefix="ls1hip_comm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        procs = []
        for r in range(world):
            env = dict(os.environ, OMP_NUM_THREADS="2", RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LS1HIP_DEVICE="0",
                       LS1HIP_TRANSPORT="mailbox", LS1HIP_COMM_DIR=comm, LS1HIP_SKIN="0.2")
            procs.append(subprocess.Popen([HIPB, "config.xml", "--steps", str(steps), "--final-checkpoint=0"], cwd=dirs[r], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = []
        for p in procs:
            try:
                outs.append((p,) + p.communicate(timeout=600))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        if os.environ.get("LS1HIP_TEST_LOG_DIR"):  # (debugging aid: the ranks' full logs)
            for r, (p, o, e) in enumerate(outs):
                with open(os.path.join(os.environ["LS1HIP_TEST_LOG_DIR"], f"intermediate_rank{r}.log"), "w") as fh:
                    fh.write(o + "\n---- stderr\n" + e)
        for p, o, e in outs:
            assert p.returncode == 0, o[-3000:] + e[-2000:]
    finally:
        shutil.rmtree(comm, ignore_errors=True)
    m = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", outs[0][1])
    assert m and m.group(1) == "on" and int(m.group(2)) < int(m.group(3)) // 3  # reuse steps dominate: the writer meets them
    handed = [int(x) for p, o, e in outs for x in re.findall(r"snapshot at step \d+: (\d+) molecule\(s\) awaiting migration handed over", o)]
    assert handed and max(handed) > 0, ("no snapshot met a molecule awaiting its migration: the test does not exercise the hand-over",
                                         [ln for p, o, e in outs for ln in o.splitlines() if "snapshot" in ln])
    names = sorted(f for f in os.listdir(dref) if f.startswith("cp-") and f.endswith(".restart.dat"))
    assert len(names) >= 3
    strays_seen = 0
    for f in names:
        a = _restart_records(dref / f)
        b = {}
        for r, d in enumerate(dirs):
            recs = _restart_records(os.path.join(d, f))
            assert not (recs.keys() & b.keys()), (f, "a molecule is in two ranks' snapshots")
            pos = np.array([v[:3] for v in recs.values()])
            lo, hi = r * L / 2, (r + 1) * L / 2
            assert pos[:, 0].min() >= lo - 1e-12 and pos[:, 0].max() < hi + 1e-12, (f, r)
            b.update(recs)
        assert a.keys() == b.keys() and len(a) == 16000, (f, len(a), len(b))
        ids = sorted(a)
        A, B = np.array([a[i] for i in ids]), np.array([b[i] for i in ids])
        dr = A[:, :3] - B[:, :3]
        dr -= L * np.round(dr / L)
        assert np.max(np.abs(dr)) < 1e-6 * L, f
        assert np.max(np.abs(A[:, 3:6] - B[:, 3:6])) < 1e-5 * np.max(np.abs(A[:, 3:6])), f
    print(f"[seam B, 2 ranks, list mode] {len(names)} intermediate checkpoints equal the reference's")


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIPB)), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("lists", ["default", "0"])
def test_rccl_transport_of_the_driver_seam_in_loopback(tmp_path, lists):
    """VERDICT r3 #5d: the RCCL transport of THIS seam had never carried an exchange.  One MarDyn_hipB process with
    LS1HIP_LOOPBACK=1: the periodic images the rank would make locally are routed through RcclTransport instead — leaving
    molecules, halo copies and position refreshes of all 26 directions are exported, packed, sent (ncclSend / ncclRecv to the own
    rank inside one group, on the transport's stream), received and imported exactly as between GPUs, with the inner pass queued
    ahead of the halo phase.  Must reproduce the unmodified single-process reference."""
    steps = 40
    src = os.path.join(GOLDEN, "inputs", "synthetic_bcc1clj_20.inp.gz")
    with gzip.open(src, "rb") as fi, open(tmp_path / "bcc.inp", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    L = float(next(ln.split()[1] for ln in open(tmp_path / "bcc.inp") if ln.strip().startswith("Length")))
    cfg = HEAD.format(dt=0.002, steps=steps, temp=0.95, L=repr(L), rc=2.5, components=LJ1, phasespace='<file type="ASCII">bcc.inp</file>')
    out = {}
    for tag, binary, env in (("ref", REF, {}), ("hipB", HIPB, {"LS1HIP_LOOPBACK": "1", "LS1HIP_TRANSPORT": "rccl", "LS1HIP_SKIN": "0.06"})):
        if lists == "0" and tag == "hipB":
            env["LS1HIP_SKIN"] = "0"
        d = tmp_path / tag
        d.mkdir()
        shutil.copy(tmp_path / "bcc.inp", d / "bcc.inp")
        (d / "config.xml").write_text(cfg)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            rows, log = _run(binary, "config.xml", str(d), steps, final_checkpoint=1)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        out[tag] = (rows, log, d)
    log = out["hipB"][1]
    assert "transport rccl" in log and "LOOPBACK" in log
    m = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", log)
    assert m and m.group(1) == ("off" if lists == "0" else "on")
    if lists != "0":
        assert 3 <= int(m.group(2)) < int(m.group(3))
    assert np.allclose(out["hipB"][0][:steps], out["ref"][0][:steps], rtol=2e-5, atol=1e-12)
    fr = [f for f in os.listdir(out["ref"][2]) if f.endswith(".restart.dat")]
    fh = [f for f in os.listdir(out["hipB"][2]) if f.endswith(".restart.dat")]
    a, b = _restart_records(out["ref"][2] / fr[0]), _restart_records(out["hipB"][2] / fh[0])
    assert a.keys() == b.keys() and len(a) == 16000
    ids = sorted(a)
    A, B = np.array([a[i] for i in ids]), np.array([b[i] for i in ids])
    dr = A[:, :3] - B[:, :3]
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-7 * L
    assert np.max(np.abs(A[:, 3:6] - B[:, 3:6])) < 1e-6 * np.max(np.abs(A[:, 3:6]))
