"""The C++17 / RCCL decomposed loop (ls1-mardyn_amd/host/DomainDecompRccl.hpp: CartDecomp, HaloExchangerRccl, DecomposedLoop —
the compiled-language counterpart of decomp.py) on the GPU box.

One GPU is available, so the transport is exercised as a LOOPBACK: the rank's periodic images are routed through
ncclSend / ncclRecv to the own rank instead of being created locally, i.e. all 26 directions are exported, packed, sent,
received and imported exactly as between 8 GPUs — leaving molecules, halo copies and, in list mode, the position refresh.
Reference: the engine's own single-domain loop (ls1hip_run), which the golden-trajectory tests pin to the real reference.
Cases: per-step kernels (fused), neighbour lists (several rebuilds), and no loopback (local images, world 1)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu

inp = load_pkg("inp")
engine_mod = load_pkg("engine")
synth = load_pkg("synth")
BIN = os.path.join(ROOT, "tests", "hostcpp", "decomp_rccl_main")


def _reference(L, ids, r, v, rc, dt, steps, skin):
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, rc)
    if skin:
        e.set_verlet(skin, force=True)
    e.set_domain([L] * 3)
    e.upload(ids, np.zeros(len(ids), np.int32), r, v)
    e.rebin(); e.halo()
    u0 = e.forces(0)
    out = e.run(dt, steps)
    st = e.download_state()
    F = e.download_forces()["F"]
    o = np.argsort(st["ids"], kind="stable")
    e.close()
    return u0, out, st["ids"][o], st["r"][o], st["v"][o], F[o]


@pytest.mark.skipif(not os.path.exists(BIN), reason="tests/hostcpp/decomp_rccl_main not built")
@pytest.mark.parametrize("skin,loopback,steps", [(0.0, 1, 12), (0.3, 1, 40), (0.3, 0, 25)])
def test_cpp_rccl_loop_equals_the_single_domain_loop(tmp_path, skin, loopback, steps):
    rc, dt = 2.5, 0.004
    L, ids, r, v = synth.bcc_box(16, temp=1.5)
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(b"LS1DCMP1")
        f.write(struct.pack("<6d", rc, dt, skin, L, L, L))
        f.write(struct.pack("<2i", steps, loopback))
        f.write(struct.pack("<Q", len(ids)))
        f.write(np.ascontiguousarray(ids, np.uint64).tobytes())
        f.write(np.ascontiguousarray(r, np.float64).tobytes())
        f.write(np.ascontiguousarray(v, np.float64).tobytes())
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LS1HIP_RCCL_ID_FILE=str(tmp_path / "id"))
    p = subprocess.run([BIN, str(case), str(tmp_path / "res")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    raw = open(str(tmp_path / "res") + ".0", "rb").read()
    assert raw[:8] == b"LS1DRES1"
    n = struct.unpack_from("<Q", raw, 8)[0]
    sums = np.frombuffer(raw, np.float64, 8, 16)
    off = 16 + 64
    oid = np.frombuffer(raw, np.uint64, n, off); off += 8 * n
    orr = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    ov = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    oF = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3)
    o = np.argsort(oid, kind="stable")
    u0, out, rid, rr, rv, rF = _reference(L, ids, r, v, rc, dt, steps, skin)
    assert n == len(ids) and np.array_equal(oid[o], rid) and int(sums[5]) == len(ids)
    # initial evaluation and the state after `steps` steps: the decomposed loop runs the same kernels on the same molecules;
    # with the transport in the loop the halo copies arrive in message order instead of generation order, so sums may differ
    # in the last bits
    assert abs(sums[0] - u0[0]) <= 1e-13 * abs(u0[0]) and abs(sums[1] - u0[1]) <= 1e-12 * abs(u0[1])
    dr = orr[o] - rr
    dr -= L * np.round(dr / L)
    assert np.max(np.abs(dr)) < 1e-11 * L
    assert np.max(np.abs(ov[o] - rv)) < 1e-11 * np.max(np.abs(rv))
    assert np.max(np.abs(oF[o] - rF)) < 1e-10 * np.max(np.abs(rF))
    assert abs(sums[2] - out["upot"]) <= 1e-11 * abs(out["upot"])
    assert abs(sums[3] - out["virial"]) <= 1e-10 * abs(out["virial"])
    assert abs(sums[4] - out["summv2"]) <= 1e-11 * out["summv2"]
    if skin:
        assert sums[7] >= steps and 2 <= sums[6] < steps  # list mode: several lifetimes, far fewer builds than steps
