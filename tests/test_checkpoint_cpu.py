"""CPU tests of the reference's binary checkpoint format against the reference's own fixture
(test_input/restart.test.{header.xml,dat}, used by its BinaryReader tests) — copied as data under tests/golden/inputs."""
import filecmp
import os

import numpy as np

from conftest import load_pkg
from golden_io import GOLDEN

inp = load_pkg("inp")
FIX = os.path.join(GOLDEN, "inputs", "restart.test")


def test_reads_the_reference_fixture():
    ps = inp.read_checkpoint(FIX)
    assert len(ps.ids) == 50 and np.allclose(ps.length, 134.266123)
    # the fixture is the binary twin of VectorizationMultiComponentMultiPotentials_50_molecules.inp
    txt = inp.read_inp(os.path.join(GOLDEN, "inputs", "VectorizationMultiComponentMultiPotentials_50_molecules.inp"))
    a, b = np.argsort(ps.ids), np.argsort(txt.ids)
    assert np.array_equal(ps.ids[a], txt.ids[b])
    assert np.array_equal(ps.cid[a], txt.cid[b])
    for k in ("r", "v", "q", "D"):
        assert np.allclose(getattr(ps, k)[a], getattr(txt, k)[b], rtol=1e-12, atol=0), k


def test_roundtrip_is_byte_exact(tmp_path):
    ps = inp.read_checkpoint(FIX)
    out = str(tmp_path / "cp")
    inp.write_checkpoint(out, ps)
    assert filecmp.cmp(out + ".dat", FIX + ".dat", shallow=False)
    again = inp.read_checkpoint(out)
    assert np.array_equal(again.r, ps.r) and again.time == ps.time and np.array_equal(again.length, ps.length)
    # header text layout of Domain::writeCheckpointHeaderXML (Domain.cpp:572-595)
    assert open(out + ".header.xml").read() == open(FIX + ".header.xml").read()


def test_short_record_formats(tmp_path):
    """ICRV (60 B) and IRV (56 B) records of io/BinaryReader.cpp:103-108,179-213: q = (1,0,0,0), D = 0, cid = 1 for IRV."""
    ps = inp.read_checkpoint(FIX)
    for fmt in ("ICRV", "IRV"):
        code, dt = inp.CHECKPOINT_FORMATS[fmt]
        assert dt.itemsize == {"ICRV": 60, "IRV": 56}[fmt]
        rec = np.zeros(len(ps.ids), dtype=dt)
        rec["id"], rec["r"], rec["v"] = ps.ids, ps.r, ps.v
        if fmt == "ICRV":
            rec["cid"] = ps.cid + 1
        pre = str(tmp_path / fmt)
        rec.tofile(pre + ".dat")
        hdr = open(FIX + ".header.xml").read().replace('type="ICRVQD"', f'type="{fmt}"')
        open(pre + ".header.xml", "w").write(hdr)
        back = inp.read_checkpoint(pre)
        assert np.array_equal(back.ids, ps.ids) and np.array_equal(back.r, ps.r) and np.array_equal(back.v, ps.v)
        assert np.array_equal(back.cid, ps.cid if fmt == "ICRV" else np.zeros_like(ps.cid))
        assert np.all(back.q == [1, 0, 0, 0]) and np.all(back.D == 0)
        assert inp.read_checkpoint_header(pre)["format_code"] == code


def test_synthetic_liquid_generators_agree():
    """bench.py's start configuration: the torch generator (run here on the CPU) reproduces the numpy generator, and
    sub-boxes of a rank grid partition the global box."""
    import torch

    synth = load_pkg("synth")
    L, ids, r, v = synth.bcc_box(8)
    assert len(ids) == 2 * 8 ** 3 == len(np.unique(ids)) and r.min() >= 0 and r.max() < L
    parts = list(synth.bcc_chunks_device(torch, torch.device("cpu"), 8, chunk=300))
    idt = torch.cat([p[0] for p in parts]).numpy().astype(np.uint64)
    a, b = np.argsort(ids), np.argsort(idt)
    assert np.array_equal(ids[a], idt[b])
    assert np.max(np.abs(r[a] - torch.cat([p[1] for p in parts]).numpy()[b])) < 1e-12
    assert np.max(np.abs(v[a] - torch.cat([p[2] for p in parts]).numpy()[b])) < 1e-12
    seen = []
    for cx in range(2):
        lo = np.array([cx * L / 2, 0, 0]); hi = np.array([L if cx else L / 2, L, L])
        for i_, r_, _ in synth.bcc_chunks(8, lo, hi, chunk=200):
            assert np.all((r_ >= lo) & (r_ < hi))
            seen.append(i_)
    assert len(np.unique(np.concatenate(seen))) == 2 * 8 ** 3 == sum(len(x) for x in seen)
