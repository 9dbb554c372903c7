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
