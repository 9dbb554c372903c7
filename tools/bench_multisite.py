#!/usr/bin/env python3
"""Times the generic multi-site force kernel on replicated fixtures (BASELINE configs[3]/[4] style, scaled down)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_io import input_path  # noqa: E402

inp = importlib.import_module("ls1-mardyn_amd.inp")
engine = importlib.import_module("ls1-mardyn_amd.engine")


def replicate(ps, k):
    L = ps.length
    n = len(ps.ids)
    shifts = np.array([[i, j, l] for i in range(k) for j in range(k) for l in range(k)], dtype=float) * L
    r = (ps.r[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    rep = lambda a: np.tile(a, (k ** 3,) + (1,) * (a.ndim - 1))  # noqa: E731
    return L * k, r, rep(ps.v), rep(ps.q), rep(ps.D), rep(ps.cid), np.arange(1, n * k ** 3 + 1, dtype=np.uint64)


def mixed_bcc(ps, n):
    """BASELINE configs[4] as SURVEY 8d-5 defines it: the fixture's five components on a jittered bcc lattice at the
    fixture's number density, component = id mod 5, random unit quaternions."""
    rng = np.random.default_rng(11)
    N = 2 * n ** 3
    rho = 250.0 / 134.266123 ** 3
    L = (N / rho) ** (1.0 / 3.0)
    a = L / n
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
    r = np.concatenate([g + 0.25 * a, g + 0.75 * a])
    r = (r + 0.2 * a * rng.uniform(-0.5, 0.5, r.shape)) % L
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    ncomp = len(ps.components.components)
    cid = (np.arange(N) % ncomp).astype(np.int32)
    if os.environ.get("LS1_MS_SLABS"):  # diagnostic: components in slabs along x -> (almost) uniform component per brick
        cid = np.minimum((r[:, 0] / L * ncomp).astype(np.int32), ncomp - 1)
    return np.array([L, L, L]), r, np.zeros((N, 3)), q, np.zeros((N, 3)), cid, np.arange(1, N + 1, dtype=np.uint64)


def main():
    name, rc, k = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    ps = inp.read_inp(input_path(name))
    if len(sys.argv) > 4 and sys.argv[4] == "bcc":
        L, r, v, q, D, cid, ids = mixed_bcc(ps, k)
    else:
        L, r, v, q, D, cid, ids = replicate(ps, k)
    e = engine.DeviceEngine(0)
    e.set_components(ps.components, rc)
    e.set_domain(L)
    if len(sys.argv) > 5:
        e.set_option("force_kernel", int(sys.argv[5]))
    e.upload(ids, cid, r, v, q, D)
    e.rebin(); e.halo(); e.forces(0)
    e.timing_enable(True); e.timing_reset()
    t0 = time.time()
    for _ in range(5):
        u = e.forces(0)
    dt = (time.time() - t0) / 5
    ms, nl = e.timing("force")
    print(f"{name} x{k}^3: N={len(ids)} force {ms/nl:.3f} ms/launch ({len(ids)/(ms/nl)*1e3:.3e} molecules/s) upot/N={u[0]/len(ids):.6g} wall {dt*1e3:.2f} ms")


if __name__ == "__main__":
    main()
