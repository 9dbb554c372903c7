"""Seam-A drop-in proof on the GPU box: the REAL reference driver (Simulation.cpp, LinkedCells, Leapfrog, thermostat,
Domain ... all unmodified object code) linked with OUR VectorizedCellProcessor translation unit
(ls1-mardyn_amd/host/VectorizedCellProcessorHip.cpp -> libls1hip -> HIP kernels) must print the same per-step
T / U_pot / p as the unmodified reference binary on the same config.  Both binaries are built in the build
container by oracle/ref_build/Makefile (targets all, hip) and travel in oracle/_ref/."""
import gzip
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from golden_io import GOLDEN

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref", "MarDyn")
HIP = os.path.join(ROOT, "oracle", "_ref", "MarDyn_hip")

HEAD = """<?xml version='1.0' encoding='UTF-8'?>
<mardyn version="20100525">
  <refunits type="SI"><length unit="nm">0.1</length><mass unit="u">1</mass><energy unit="K">1</energy></refunits>
  <simulation type="MD">
    <integrator type="Leapfrog"><timestep unit="reduced">{dt}</timestep></integrator>
    <run><currenttime>0</currenttime><production><steps>{steps}</steps></production></run>
    <ensemble type="NVT">
      <temperature unit="reduced">{temp}</temperature>
      <domain type="box"><lx>{L}</lx><ly>{L}</ly><lz>{L}</lz></domain>
      <components>{components}</components>
      <phasespacepoint>{phasespace}</phasespacepoint>
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
LJ1 = ('<moleculetype id="1" name="1CLJ"><site type="LJ126" id="1"><coords><x>0.0</x><y>0.0</y><z>0.0</z></coords>'
       '<mass>1.0</mass><sigma>1.0</sigma><epsilon>1.0</epsilon><shifted>0</shifted></site>'
       '<momentsofinertia rotaxes="xyz"><Ixx>0.0</Ixx><Iyy>0.0</Iyy><Izz>0.0</Izz></momentsofinertia></moleculetype>')
ETHANE = ('<moleculetype id="1" name="C2H6">'
          '<site type="LJ126" id="1"><coords><x>0.0</x><y>0.0</y><z>-2.2157048</z></coords><mass>0.0150347</mass>'
          '<sigma>6.6140441</sigma><epsilon>0.00042932536</epsilon><shifted>0</shifted></site>'
          '<site type="LJ126" id="2"><coords><x>0.0</x><y>0.0</y><z>2.2157048</z></coords><mass>0.0150347</mass>'
          '<sigma>6.6140441</sigma><epsilon>0.00042932536</epsilon><shifted>0</shifted></site>'
          '<momentsofinertia rotaxes="xyz"><Ixx>0.14762114</Ixx><Iyy>0.14762114</Iyy><Izz>0.0</Izz></momentsofinertia>'
          '</moleculetype>')


def _run(binary, cfg, cwd, steps, final_checkpoint=0):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    out = subprocess.run([binary, cfg, "--steps", str(steps), f"--final-checkpoint={final_checkpoint}"], cwd=cwd, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    rows = re.findall(r"Simstep = (\d+)\s+T = (\S+)\s+U_pot = (\S+)\s+p = (\S+)", out.stdout)
    assert len(rows) >= steps, out.stdout[-2000:]
    return np.array([[float(x) for x in r[1:]] for r in rows]), out.stdout


ARGON = ('<moleculetype id="1" name="Argon"><site type="LJ126" id="1"><coords><x>0.0</x><y>0.0</y><z>0.0</z></coords>'
         '<mass>0.039948</mass><sigma>6.4160007</sigma><epsilon>0.000369852537</epsilon><shifted>0</shifted></site>'
         '<momentsofinertia rotaxes="xyz"><Ixx>0.0</Ixx><Iyy>0.0</Iyy><Izz>0.0</Izz></momentsofinertia></moleculetype>')


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP)), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("case", ["1clj_generated", "ethane_inp", "argon_example"])
def test_reference_driver_with_hip_cell_processor(tmp_path, case):
    if case == "1clj_generated":
        N = 2 * 14 ** 3
        L = (N / 0.785302672) ** (1 / 3)
        cfg = HEAD.format(dt=0.002, steps=10, temp=0.95, L=repr(L), rc=2.5, components=LJ1,
                          phasespace='<generator name="CubicGridGenerator"><specification>density</specification>'
                                     '<density>0.785302672</density><binaryMixture>false</binaryMixture></generator>')
        steps = 10
    elif case == "argon_example":
        # BASELINE.json configs[0] (plumbing): the reference's shipped example examples/Argon/200K_18mol_l — its phase space
        # file (copied as a fixture), its component, box, time step, temperature and cutoff (r_c = 5.15 sigma, 3 cells per
        # dimension) from its config.xml, output plugins stripped as in BASELINE.md; NVT (velocity scaling) as shipped.
        with gzip.open(os.path.join(GOLDEN, "inputs", "Argon_200K_18mol_l.inp.gz"), "rb") as fi, \
                open(tmp_path / "Argon_200K_18mol_l.inp", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        cfg = HEAD.format(dt=0.0667516, steps=20, temp=0.000633363365, L="108.43455", rc=33.0702, components=ARGON,
                          phasespace='<file type="ASCII">Argon_200K_18mol_l.inp</file>')
        steps = 20
    else:
        with gzip.open(os.path.join(GOLDEN, "inputs", "Ethan_equilibrated.inp.gz"), "rb") as fi, \
                open(tmp_path / "ethan.inp", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        cfg = HEAD.format(dt=0.5, steps=5, temp=0.000855, L="571.607759", rc=32.1254, components=ETHANE,
                          phasespace='<file type="ASCII">ethan.inp</file>')
        steps = 5
    (tmp_path / "config.xml").write_text(cfg)
    ref, _ = _run(REF, "config.xml", str(tmp_path), steps)
    hip, log = _run(HIP, "config.xml", str(tmp_path), steps)
    assert "MI355X/HIP back end" in log
    n = min(len(ref), len(hip))
    # the driver prints 6 significant digits
    assert np.allclose(hip[:n], ref[:n], rtol=2e-5, atol=1e-12), (ref[:n], hip[:n])
