"""Synthetic phase spaces for benchmarks and full-size tests: the bcc start lattice of the reference's
CubicGridGenerator (two interleaved simple-cubic grids, ParticleCellBase::initCubicGrid,
/root/reference/src/particleContainer/ParticleCellBase.cpp:73-177; molecules per dimension from
CubicGridGeneratorInternal::determineMolsPerDimension, io/CubicGridGeneratorInternal.cpp:124-186) with the
deterministic per-molecule displacement SURVEY.md 8(d)-2 prescribes (0.1 sigma * (u - 1/2), u = splitmix64 of the
molecule id) and Maxwell velocities drawn from the same hash.

Every quantity is a pure function of the GLOBAL molecule id, so any rank can generate exactly its own sub-box of the
same global liquid (strong scaling) in bounded chunks, on the host (numpy) or directly in device memory (torch) —
a 10^8-molecule start configuration never exists as one host array.
"""
from __future__ import annotations

import numpy as np

RHO_DEFAULT = 0.785302672  # examples/Generators/cubic_grid_generator/config.xml:43
_M64 = (1 << 64) - 1
_G, _C1, _C2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def box_length(n_per_dim: int, rho: float = RHO_DEFAULT) -> float:
    return (2 * n_per_dim ** 3 / rho) ** (1.0 / 3.0)


# ---- numpy -------------------------------------------------------------------------------------------------------
def _splitmix_np(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(_G)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def _unit_np(h):
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _site_block_np(ix, iy, iz, n, L, jitter, temp):
    """molecules of the lattice cells (ix, iy, iz) (flat index arrays): both bcc sites each"""
    a = L / n
    cell = (iz.astype(np.uint64) * np.uint64(n) + iy.astype(np.uint64)) * np.uint64(n) + ix.astype(np.uint64)
    out = []
    for s, off in ((0, 0.25), (1, 0.75)):
        ids = cell * np.uint64(2) + np.uint64(s + 1)
        base = np.stack([(ix + off) * a, (iy + off) * a, (iz + off) * a], axis=1)
        with np.errstate(over="ignore"):
            u = np.stack([_unit_np(_splitmix_np(ids * np.uint64(16) + np.uint64(k))) for k in range(3)], axis=1)
            w = [_unit_np(_splitmix_np(ids * np.uint64(16) + np.uint64(4 + k))) for k in range(4)]
        r = base + jitter * (u - 0.5)
        r = np.mod(r, L)
        r[r >= L] = 0.0
        # Box-Muller: two pairs of uniforms -> three normals
        m0 = np.sqrt(-2.0 * np.log(1.0 - w[0]))
        m1 = np.sqrt(-2.0 * np.log(1.0 - w[2]))
        v = np.stack([m0 * np.cos(2 * np.pi * w[1]), m0 * np.sin(2 * np.pi * w[1]), m1 * np.cos(2 * np.pi * w[3])], axis=1)
        out.append((ids, r, v * np.sqrt(temp)))
    return out


def _index_ranges(n, L, lo, hi, jitter):
    """per dimension: lattice cell indices whose (displaced, wrapped) sites can fall into [lo, hi)"""
    a = L / n
    rng = []
    for d in range(3):
        if lo[d] <= 0.0 and hi[d] >= L:
            rng.append(np.arange(n, dtype=np.int64))
            continue
        i0 = int(np.floor((lo[d] - jitter) / a - 0.75)) - 1
        i1 = int(np.ceil((hi[d] + jitter) / a)) + 1
        if i1 - i0 >= n:
            rng.append(np.arange(n, dtype=np.int64))
        else:
            rng.append(np.mod(np.arange(i0, i1, dtype=np.int64), n))
    return rng


def bcc_chunks(n_per_dim: int, lo=None, hi=None, rho: float = RHO_DEFAULT, temp: float = 0.95, jitter: float = 0.1,
               chunk: int = 1 << 21):
    """Yield (ids uint64, r [m,3], v [m,3]) chunks of the global jittered bcc liquid restricted to the sub-box
    [lo, hi) (default: the whole box).  Every global molecule is produced by exactly one disjoint sub-box."""
    n = int(n_per_dim)
    L = box_length(n, rho)
    lo = np.zeros(3) if lo is None else np.asarray(lo, dtype=np.float64)
    hi = np.full(3, L) if hi is None else np.asarray(hi, dtype=np.float64)
    rx, ry, rz = _index_ranges(n, L, lo, hi, jitter)
    per_plane = len(rx) * len(ry) * 2
    planes = max(1, chunk // max(per_plane, 1))
    for z0 in range(0, len(rz), planes):
        zz = rz[z0:z0 + planes]
        iz, iy, ix = np.meshgrid(zz, ry, rx, indexing="ij")
        for ids, r, v in _site_block_np(ix.ravel(), iy.ravel(), iz.ravel(), n, L, jitter, temp):
            keep = np.all((r >= lo) & (r < hi), axis=1)
            if keep.all():
                yield ids, r, v
            elif keep.any():
                yield ids[keep], r[keep], v[keep]


def bcc_box(n_per_dim: int, **kw):
    """whole box in one piece (small sizes: tests, CPU comparisons)"""
    parts = list(bcc_chunks(n_per_dim, **kw))
    ids = np.concatenate([p[0] for p in parts])
    r = np.concatenate([p[1] for p in parts])
    v = np.concatenate([p[2] for p in parts])
    return box_length(n_per_dim, kw.get("rho", RHO_DEFAULT)), ids, r, v


# ---- torch (device-resident generation; same arithmetic in two's-complement int64) --------------------------------
def _s64(x):
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _splitmix_t(x):
    z = x + _s64(_G)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    return z ^ _lsr(z, 31)


def bcc_chunks_device(torch, device, n_per_dim: int, lo=None, hi=None, rho: float = RHO_DEFAULT, temp: float = 0.95,
                      jitter: float = 0.1, chunk: int = 1 << 22):
    """Same liquid as bcc_chunks, generated in device memory: yields (ids int64, r [m,3] f64, v [m,3] f64) tensors."""
    n = int(n_per_dim)
    L = box_length(n, rho)
    a = L / n
    lo_h = np.zeros(3) if lo is None else np.asarray(lo, dtype=np.float64)
    hi_h = np.full(3, L) if hi is None else np.asarray(hi, dtype=np.float64)
    rx, ry, rz = (torch.from_numpy(x).to(device) for x in _index_ranges(n, L, lo_h, hi_h, jitter))
    lo_t = torch.tensor(lo_h, dtype=torch.float64, device=device)
    hi_t = torch.tensor(hi_h, dtype=torch.float64, device=device)
    per_plane = len(rx) * len(ry) * 2
    planes = max(1, chunk // max(per_plane, 1))
    two_pi = 2.0 * np.pi
    for z0 in range(0, len(rz), planes):
        zz = rz[z0:z0 + planes]
        iz, iy, ix = torch.meshgrid(zz, ry, rx, indexing="ij")
        ix, iy, iz = ix.reshape(-1), iy.reshape(-1), iz.reshape(-1)
        cell = (iz * n + iy) * n + ix
        for s, off in ((0, 0.25), (1, 0.75)):
            ids = cell * 2 + (s + 1)
            base = torch.stack([(ix.double() + off) * a, (iy.double() + off) * a, (iz.double() + off) * a], dim=1)
            unit = lambda k: _lsr(_splitmix_t(ids * 16 + k), 11).double() * (1.0 / 9007199254740992.0)  # noqa: E731
            u = torch.stack([unit(0), unit(1), unit(2)], dim=1)
            r = base + jitter * (u - 0.5)
            r = torch.remainder(r, L)
            r[r >= L] = 0.0
            w0, w1, w2, w3 = unit(4), unit(5), unit(6), unit(7)
            m0 = torch.sqrt(-2.0 * torch.log(1.0 - w0))
            m1 = torch.sqrt(-2.0 * torch.log(1.0 - w2))
            v = torch.stack([m0 * torch.cos(two_pi * w1), m0 * torch.sin(two_pi * w1), m1 * torch.cos(two_pi * w3)], dim=1)
            v = v * float(np.sqrt(temp))
            keep = ((r >= lo_t) & (r < hi_t)).all(dim=1)
            if not bool(keep.all()):
                ids, r, v = ids[keep], r[keep], v[keep]
            if ids.numel():
                yield ids.contiguous(), r.contiguous(), v.contiguous()


# ---- the five-component LJ + charge + dipole + quadrupole set (BASELINE configs[4]) ----------------------------------------
# Sites of /root/reference/test_input/VectorizationMultiComponentMultiPotentials.inp:6-31.  That fixture is a static force test:
# its first two components consist of dipole / quadrupole sites only, which carry no mass in the reference (Component.cpp:
# 175-205 "massless"), so neither the reference nor anything else can integrate it (dt / 2m = inf); its third component is a
# 0.42 u point charge without a repulsive core, and it omits the mixing block.  The INTEGRABLE form used for trajectories and
# for the particle-updates/s bench keeps every site of the fixture and gives components 1-3 a rigid frame of three
# Lennard-Jones centres as mass carrier and repulsive core (a bent triatomic like the fixture's own water: centre of mass at
# the origin, axes = principal axes, three non-zero moments FROM THE SITE MASSES, so that the reference's count of rotational
# degrees of freedom, Component.cpp:140-167, is the physical one: asymmetric tops turned by dipole / quadrupole torques),
# writes the mixing block (xi = eta = 1) and eps_RF = 1e10.  Components 4 and 5 (linear LJ + charge + dipole + quadrupole
# rotor; water) are the fixture's.  LJ parameters of the frame: the fixture's own (component 4's centre) for the heavy site.
MIXED5_RC = 35.0
MIXED5_TEMPERATURE = 0.000855040543                      # the fixture's
MIXED5_NUMBER_DENSITY = 250.0 / 134.266123 ** 3          # the fixture's
_FRAME = [(0.0, 0.3, 0.0, 0.012, 0.00042, 6.7, 0.0, 0),      # x y z m eps sigma rc_shift do_shift
          (1.2, -0.9, 0.0, 0.002, 0.0001, 3.0, 0.0, 0),
          (-1.2, -0.9, 0.0, 0.002, 0.0001, 3.0, 0.0, 0)]


def mixed5_components(inp):
    """ComponentSet of the integrable five-component set (see above); `inp` = the ls1-mardyn_amd.inp module."""
    mk = inp.make_component
    comps = [
        mk(lj=_FRAME, dipoles=[(-2.0, 0, 0, 1, 0, 0, -7.1)]),
        # (axis normalised: the .inp reader keeps the vector as written — the fixture's (0, 1, 1) — while Component::readXML
        #  normalises it, Site.h:305-312; a unit vector means the same molecule through either reader)
        mk(lj=_FRAME, dipoles=[(-2.0, 0, 0, 0, np.sqrt(0.5), np.sqrt(0.5), -1.1)], quadrupoles=[(-2.0, 0, 0, 0, 0, 1, -1.3)]),
        mk(lj=_FRAME, charges=[(0, 0, 1.0, 0.00042, 0.5)], quadrupoles=[(-2.0, 0, 0, 0, 0, 1, -1.3)]),
        mk(lj=[(0, 0, -1.0, 0.02, 0.00042, 6.7, 0, 0)], charges=[(0, 0, 1.0, 0.00042, 0.5)], dipoles=[(0, 0, -2.0, 0, 0, 1, 0.9)],
           quadrupoles=[(-2.0, 0, 0, 0, 0, 1, -1.3)]),
        mk(lj=[(0, 0.123891518, 0, 0.016, 0.000246810271, 5.95953717, 0, 0)],
           charges=[(0, -0.159567514, 0, 0, -1.04), (1.43042933, -0.983266012, 0, 0.001008, 0.52),
                    (-1.43042933, -0.983266012, 0, 0.001008, 0.52)],
           I_file=(0.00219467882, 0.00412499417, 0.00631967299)),
    ]
    return inp.ComponentSet(comps, np.ones((10, 2)), 1e10)


def thermal_box(inp, cs, n_per_dim: int, number_density: float, temp: float, jitter_frac: float = 0.2):
    """PhaseSpace: 2 n^3 molecules of the component set `cs` on a jittered bcc lattice, component = (id - 1) mod ncomp,
    orientation / velocity / angular momentum drawn from splitmix64 hashes of the molecule id (Maxwell at `temp`:
    v ~ sqrt(T / m), D_k ~ sqrt(I_k T); D_k = 0 where I_k = 0)."""
    n = n_per_dim
    N = 2 * n ** 3
    L = (N / number_density) ** (1.0 / 3.0)
    a = L / n
    idx = np.arange(n ** 3, dtype=np.int64)
    ix, iy, iz = idx % n, (idx // n) % n, idx // (n * n)
    base = np.stack([ix, iy, iz], axis=1).astype(np.float64) * a
    r = np.concatenate([base + 0.25 * a, base + 0.75 * a], axis=0)
    ids = np.arange(1, N + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        u = [_unit_np(_splitmix_np(ids * np.uint64(32) + np.uint64(k))) for k in range(23)]
    r = np.mod(r + jitter_frac * a * (np.stack(u[0:3], axis=1) - 0.5), L)
    r[r >= L] = 0.0

    def normals(k0, cnt):  # Box-Muller, one normal per pair of uniforms
        out = []
        for k in range(cnt):
            m = np.sqrt(-2.0 * np.log(1.0 - u[k0 + 2 * k]))
            out.append(m * np.cos(2.0 * np.pi * u[k0 + 2 * k + 1]))
        return np.stack(out, axis=1)

    q = normals(3, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    cid = ((ids - np.uint64(1)) % np.uint64(len(cs.components))).astype(np.int32)
    mass = np.array([c.mass for c in cs.components])[cid]
    I = np.stack([c.I for c in cs.components])[cid]
    v = normals(11, 3) * np.sqrt(temp / mass)[:, None]
    v -= v.mean(axis=0)
    # angular momentum: Maxwell in the BODY frame (L_k ~ sqrt(I_k T)), turned into the lab frame the reference stores it in
    # (D = q.rotate(L_body), Quaternion.cpp:43-62; FullMolecule.cpp:341 takes w = I^-1 q.rotateinv(D))
    Lb = normals(17, 3) * np.sqrt(I * temp)
    w_, x_, y_, z_ = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    D = np.stack([(w_ * w_ + x_ * x_ - y_ * y_ - z_ * z_) * Lb[:, 0] + 2. * (x_ * y_ - w_ * z_) * Lb[:, 1] + 2. * (w_ * y_ + x_ * z_) * Lb[:, 2],
                  2. * (w_ * z_ + x_ * y_) * Lb[:, 0] + (w_ * w_ - x_ * x_ + y_ * y_ - z_ * z_) * Lb[:, 1] + 2. * (y_ * z_ - w_ * x_) * Lb[:, 2],
                  2. * (x_ * z_ - w_ * y_) * Lb[:, 0] + 2. * (w_ * x_ + y_ * z_) * Lb[:, 1] + (w_ * w_ - x_ * x_ - y_ * y_ + z_ * z_) * Lb[:, 2]],
                 axis=1)
    return inp.PhaseSpace(cs, np.array([L, L, L]), ids, cid, r, v, q, D, 0.0, temp)


def mixed5_box(inp, n_per_dim: int, jitter_frac: float = 0.2, temp: float = MIXED5_TEMPERATURE):
    """The integrable five-component set at the fixture's number density, see thermal_box."""
    return thermal_box(inp, mixed5_components(inp), n_per_dim, MIXED5_NUMBER_DENSITY, temp, jitter_frac)
