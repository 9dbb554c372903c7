"""Readers for the committed golden fixtures (tests/golden/*.bin, format: oracle/ref_build/refdump.cpp)."""
import gzip
import os
import shutil
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

REC = np.dtype([("id", "<u8"), ("cid", "<u8"), ("r", "<f8", 3), ("v", "<f8", 3), ("q", "<f8", 4),
                ("D", "<f8", 3), ("F", "<f8", 3), ("M", "<f8", 3), ("Vi", "<f8", 3)])


def manifest(single_precision=False):
    """Golden cases of MANIFEST.txt.  Column `legacy`: 0 = VectorizedCellProcessor (FP64 build), 1 = LegacyCellProcessor,
    2 = with the velocity-scaling thermostat, 5 = the same with thermostats assigned to components by the header of the input
    (`componentwise`; the generic NVT tests drive one global thermostat and skip these), 3 / 4 = the reference built with -DMARDYN_SPDP / -DMARDYN_SPSP (returned only
    with single_precision=True: every other test iterates over the FP64 cases and their 1e-10 tolerances)."""
    out = {}
    with open(os.path.join(GOLDEN, "MANIFEST.txt")) as fh:
        for ln in fh:
            if ln.startswith("#") or not ln.strip():
                continue
            name, inp, rc, periodic, steps, dt, legacy = ln.split()
            prec = {3: 1, 4: 2}.get(int(legacy), 0)
            if bool(prec) != bool(single_precision):
                continue
            out[name] = dict(name=name, input=inp, rc=float(rc), periodic=int(periodic), steps=int(steps),
                             dt=float(dt), legacy=int(int(legacy) == 1), nvt=int(int(legacy) in (2, 5)),
                             componentwise=int(int(legacy) == 5), precision=prec)
    return out


def read_golden(name):
    path = os.path.join(GOLDEN, name + ".bin")
    with open(path, "rb") as fh:
        assert fh.read(8) == b"LS1GOLD1"
        n, steps = np.frombuffer(fh.read(16), dtype="<u8")
        hdr = np.frombuffer(fh.read(8 * 9), dtype="<f8")
        body = fh.read()
    lrc = None
    if body[-24:-16] == b"LS1LRC01":
        lrc = np.frombuffer(body[-16:], dtype="<f8").copy()
        body = body[:-24]
    recs = np.frombuffer(body, dtype=REC)
    assert len(recs) == n
    return dict(n=int(n), steps=int(steps), rc=hdr[0], dt=hdr[1], L=hdr[2:5].copy(), upot=hdr[5], virial=hdr[6],
                summv2=hdr[7], sumIw2=hdr[8], recs=recs, lrc=lrc)


def input_path(fname):
    """Path of a fixture input; .gz inputs are unpacked to a temp file."""
    p = os.path.join(GOLDEN, "inputs", fname)
    if os.path.exists(p):
        return p
    tmp = os.path.join(tempfile.gettempdir(), "ls1golden_" + fname)
    if not os.path.exists(tmp):
        with gzip.open(p + ".gz", "rb") as fi, open(tmp + ".part", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        os.replace(tmp + ".part", tmp)
    return tmp


def sorted_phase_space(ps):
    """Phase space arrays ordered by molecule id (the order of the golden records)."""
    o = np.argsort(ps.ids, kind="stable")
    return dict(ids=ps.ids[o], cid=ps.cid[o].astype(np.int32), r=ps.r[o].copy(), v=ps.v[o].copy(),
                q=ps.q[o].copy(), D=ps.D[o].copy())


# cases whose forces are pure cancellation residues (r = 2^(1/6): F = 0): compare on the natural force scale 24 eps/sigma
FORCE_FLOOR = {"F0": 24.0}


def rel_max(a, b, floor=0.0):
    """max|a-b| / max(max|b|, floor)  (the metric SURVEY §7 'hard parts' prescribes; avoids cancellation residues)."""
    den = max(float(np.max(np.abs(b))) if np.size(b) else 0.0, floor)
    if den == 0.0:
        return float(np.max(np.abs(a - b))) if np.size(a) else 0.0
    return float(np.max(np.abs(a - b)) / den)


def rel_componentwise(a, b, floor_frac=1e-3):
    """max over components with |b| > floor_frac * max|b| of |a-b| / |b|: catches a small force that is relatively wrong
    while the max-norm metric (rel_max) still passes.  Components below the floor are cancellation residues."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not b.size:
        return 0.0
    big = np.abs(b) > floor_frac * np.max(np.abs(b))
    if not np.any(big):
        return 0.0
    return float(np.max(np.abs(a - b)[big] / np.abs(b)[big]))
