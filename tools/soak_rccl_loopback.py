#!/usr/bin/env python3
"""Long run of the C++ / RCCL decomposed list loop (tests/hostcpp/decomp_rccl_main, one rank, loopback: every periodic image
travels through ncclSend / ncclRecv): many list lifetimes with migration through the transport.  Checks: no molecule lost
or duplicated, total energy conserved like the single-domain loop's.  usage: soak_rccl_loopback.py [n_per_dim] [steps]"""
import os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg  # noqa: E402
synth = load_pkg("synth"); inp = load_pkg("inp"); engine_mod = load_pkg("engine")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rc, dt, skin = 2.5, 0.002, 0.2
L, ids, r, v = synth.bcc_box(n, temp=0.95)
d = "/tmp/soak_rccl"; os.makedirs(d, exist_ok=True)
with open(d + "/case.bin", "wb") as f:
    f.write(b"LS1DCMP1"); f.write(struct.pack("<6d", rc, dt, skin, L, L, L)); f.write(struct.pack("<2i", steps, 1)); f.write(struct.pack("<Q", len(ids)))
    f.write(np.ascontiguousarray(ids, np.uint64).tobytes()); f.write(np.ascontiguousarray(r).tobytes()); f.write(np.ascontiguousarray(v).tobytes())
env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LS1HIP_RCCL_ID_FILE=d + "/id")
p = subprocess.run([os.path.join(ROOT, "tests/hostcpp/decomp_rccl_main"), d + "/case.bin", d + "/res"], env=env, capture_output=True, text=True, timeout=900)
assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
raw = open(d + "/res.0", "rb").read()
nn = struct.unpack_from("<Q", raw, 8)[0]
s = np.frombuffer(raw, np.float64, 8, 16)
oid = np.frombuffer(raw, np.uint64, nn, 16 + 64)
# the single-domain loop on the same start
comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
e = engine_mod.DeviceEngine(0); e.set_components(comps, rc); e.set_verlet(skin); e.set_domain([L] * 3)
e.upload(ids, np.zeros(len(ids), np.int32), r, v); e.rebin(); e.halo(); u0 = e.forces(0)[0]
out = e.run(dt, steps)
ek0 = 0.5 * float((v * v).sum())
E0 = ek0 + u0
print(f"N={len(ids)} steps={steps}: C++/RCCL loopback loop  E={0.5 * s[4] + s[2]:.6f}  (start {0.5 * float((v*v).sum()) + s[0]:.6f}), "
      f"{int(s[6])} list builds for {int(s[7])} list steps, ids intact: {bool(np.array_equal(np.sort(oid), np.sort(ids)))}")
print(f"                     single-domain loop     E={0.5 * out['summv2'] + out['upot']:.6f}  (start {E0:.6f}), {e.get_option('verlet_builds')} list builds")
