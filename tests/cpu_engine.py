"""Numpy stand-in for DeviceEngine used ONLY by the CPU (gloo) tests of the multi-rank exchange protocol.

It implements the engine surface decomp.HaloExchanger / DistributedSimulation need (export_counts, export_pack,
import_records, import_done, rebin, halo, forces, kick_drift, kick, count) for single-centre LJ with brute-force
numpy pair sums.  Record formats and the shift / neighbour rules follow include/ls1hip.h.  Test infrastructure —
the product never imports it.
"""
import ctypes

import numpy as np


class CpuEngine:
    def __init__(self, global_len, bmin, bmax, my_rank, nbr, rc, eps24=24.0, sig2=1.0, mass=1.0, skin=None):
        self.L = np.asarray(global_len, float); self.bmin = np.asarray(bmin, float); self.bmax = np.asarray(bmax, float)
        self.rank = my_rank; self.nbr = np.asarray(nbr); self.rc = rc
        self.eps24, self.sig2, self.mass = eps24, sig2, mass
        self.skin = skin  # list mode (ls1hip_set_verlet): halo shell of rc + skin, refreshable halo copies
        self.rc_halo = rc + (skin or 0.0)
        self.shift = np.zeros((27, 3))
        for sz in (-1, 0, 1):
            for sy in (-1, 0, 1):
                for sx in (-1, 0, 1):
                    d = (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1)
                    for k, s in enumerate((sx, sy, sz)):
                        if s < 0 and self.bmin[k] == 0.0:
                            self.shift[d, k] = self.L[k]
                        if s > 0 and self.bmax[k] == self.L[k]:
                            self.shift[d, k] = -self.L[k]
        self.exp = {0: [np.zeros((0, 15))] * 27, 1: [np.zeros((0, 9))] * 27, 2: [np.zeros((0, 3))] * 27}
        self.pending = []
        self.halo_r = np.zeros((0, 3))
        self.F = None
        self.macro = np.zeros(2)

    def upload(self, ids, r, v):
        self.ids = np.asarray(ids, np.uint64).copy(); self.r = np.array(r, float); self.v = np.array(v, float)
        self.F = np.zeros_like(self.r)

    def count(self):
        return len(self.ids), len(self.halo_r)

    def _dir(self, s):
        return (s[2] + 1) * 9 + (s[1] + 1) * 3 + (s[0] + 1)

    # --- step pieces
    def kick_drift(self, dt):
        self.v += 0.5 * dt / self.mass * self.F
        self.r += dt * self.v

    def kick_then_kick_drift(self, dt):
        self.v += 0.5 * dt / self.mass * self.F
        self.kick_drift(dt)

    def rebin(self):
        s = np.where(self.r < self.bmin, -1, np.where(self.r >= self.bmax, 1, 0))
        out = [[] for _ in range(27)]
        keep = np.ones(len(self.r), bool)
        for i in np.nonzero(np.any(s != 0, axis=1))[0]:
            d = self._dir(s[i]); dest = self.nbr[d]
            if dest == self.rank:
                self.r[i] += self.shift[d]
            else:
                assert dest >= 0
                rec = np.zeros(15)
                rec[0] = np.array([self.ids[i]], np.uint64).view(np.float64)[0]
                rec[1] = np.array([0], np.int64).view(np.float64)[0]
                rec[2:5] = self.r[i] + self.shift[d]; rec[5:8] = self.v[i]; rec[8] = 1.0
                out[d].append(rec); keep[i] = False
        self.ids, self.r, self.v = self.ids[keep], self.r[keep], self.v[keep]
        self.exp[0] = [np.array(o).reshape(-1, 15) for o in out]
        self.pending = []

    def halo(self):
        out = [[] for _ in range(27)]
        self.pending = []
        self.pending_src = []  # per pending block: (owned indices, shift) of a local image block, None for imported records
        self.exp_src = [np.zeros(0, int)] * 27
        lo = self.r < self.bmin + self.rc_halo
        hi = self.r >= self.bmax - self.rc_halo
        for sz in (-1, 0, 1):
            for sy in (-1, 0, 1):
                for sx in (-1, 0, 1):
                    if sx == sy == sz == 0:
                        continue
                    s = (sx, sy, sz); d = self._dir(s); dest = self.nbr[d]
                    if dest < 0:
                        continue
                    m = np.ones(len(self.r), bool)
                    for k in range(3):
                        if s[k] < 0: m &= lo[:, k]
                        if s[k] > 0: m &= hi[:, k]
                    rr = self.r[m] + self.shift[d]
                    if dest == self.rank:
                        self.pending.append(rr)
                        self.pending_src.append((np.nonzero(m)[0], self.shift[d].copy()))
                    else:
                        self.exp_src[d] = np.nonzero(m)[0]
                        rec = np.zeros((len(rr), 9))
                        rec[:, 0] = self.ids[m].view(np.float64); rec[:, 2:5] = rr; rec[:, 5] = 1.0
                        out[d] = rec
        self.exp[1] = [np.asarray(o).reshape(-1, 9) for o in out]

    def export_counts(self, kind):
        return np.array([len(x) for x in self.exp[kind]], dtype=np.uint64)

    @staticmethod
    def _view(ptr, n):
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    def export_pack(self, kind, d, ptr, cap):
        rec = self.exp[kind][d]
        assert cap >= len(rec)
        self._view(ptr, rec.size)[:] = rec.reshape(-1)

    def export_pack_dirs(self, kind, dirs, ptr, cap):
        w = {0: 15, 1: 9, 2: 3}[kind]
        off = 0
        for d in dirs:
            rec = self.exp[kind][d]
            assert cap - off >= len(rec)
            self.export_pack(kind, d, ptr + off * w * 8, cap - off)
            off += len(rec)

    def import_records(self, kind, ptr, n):
        w = {0: 15, 1: 9, 2: 3}[kind]
        rec = self._view(ptr, n * w).reshape(n, w).copy()
        if kind == 2:  # position refresh: the records of the build-time halo imports, same order
            for k, src in enumerate(self.pending_src):
                if src is None and self._refresh_cursor[k] is not None and len(rec):
                    m = len(self.pending[k])
                    if self._refresh_cursor[k] == 0:
                        self.pending[k] = rec[:m].copy()
                        rec = rec[m:]
                        self._refresh_cursor[k] = None
            assert len(rec) == 0
            return
        if kind == 0:
            self.ids = np.concatenate([self.ids, rec[:, 0].copy().view(np.uint64)])
            self.r = np.concatenate([self.r, rec[:, 2:5]]); self.v = np.concatenate([self.v, rec[:, 5:8]])
        else:
            self.pending.append(rec[:, 2:5])
            self.pending_src.append(None)

    def import_done(self, kind):
        if kind == 2:
            self.halo_r = np.concatenate(self.pending) if self.pending else np.zeros((0, 3))
            return
        if kind == 0:
            assert np.all(self.r >= self.bmin) and np.all(self.r < self.bmax)
            self.F = np.zeros_like(self.r)
        else:
            self.halo_r = np.concatenate(self.pending) if self.pending else np.zeros((0, 3))

    def finalize_local(self):  # single-rank convenience
        self.import_done(0); self.import_done(1)

    def forces(self, which=0, want_macro=True):
        if which == 1:
            return None  # the stand-in does all the work in the boundary call
        allr = np.concatenate([self.r, self.halo_r])
        n = len(self.r)
        F = np.zeros((n, 3)); u6 = 0.0; vir = 0.0
        rc2 = self.rc ** 2
        for i in range(n):
            d = self.r[i] - allr
            r2 = (d * d).sum(1)
            m = (r2 < rc2) & (r2 > 0)
            inv = 1.0 / r2[m]
            lj6 = (self.sig2 * inv) ** 3; lj12 = lj6 * lj6
            fac = self.eps24 * (2 * lj12 - lj6) * inv
            f = d[m] * fac[:, None]
            F[i] = f.sum(0)
            u6 += 0.5 * (self.eps24 * (lj12 - lj6)).sum()
            vir += 0.5 * (d[m] * f).sum()
        self.F = F
        self.macro = np.array([u6 / 6.0, vir])
        return (self.macro[0], self.macro[1]) if want_macro else None

    def can_fuse_integration(self):
        return True

    def forces_kick_drift(self, which, dt, want_macro=False):
        """forces(which) fused with kick + kick + drift of the next step (which=1 is a no-op in this stand-in)"""
        out = self.forces(which, want_macro)
        if which != 1:
            self.v += 0.5 * dt / self.mass * self.F
            self.kick_drift(dt)
        return out

    def kick(self, dt_half, want_sums=True):
        self.v += dt_half / self.mass * self.F
        return (float(self.mass * (self.v ** 2).sum()), 0.0, len(self.v), 0) if want_sums else None

    # --- list mode (ls1hip_verlet_build / ls1hip_halo_refresh / ls1hip_forces_list / ls1hip_verlet_poll)
    def can_verlet(self):
        return self.skin is not None

    def verlet_build(self):
        self.r0 = self.r.copy()  # (the stand-in keeps no lists: brute force over owned + halo copies)

    def halo_refresh(self):
        for k, src in enumerate(self.pending_src):
            if src is not None:
                self.pending[k] = self.r[src[0]] + src[1]
        self._refresh_cursor = [0 if src is None else None for src in self.pending_src]
        self.exp[2] = [self.r[self.exp_src[d]] + self.shift[d] if len(self.exp_src[d]) else np.zeros((0, 3)) for d in range(27)]
        if not any(src is None for src in self.pending_src):
            self.halo_r = np.concatenate(self.pending) if self.pending else np.zeros((0, 3))

    def forces_list(self, which, dt=0.0, want_macro=False):
        out = self.forces(which, want_macro)
        if which != 1 and dt > 0:
            self.v += dt / self.mass * self.F
            self.r += dt * self.v
        return out

    def verlet_poll(self):
        return bool(np.sqrt(((self.r - self.r0) ** 2).sum(1)).max() > 0.5 * self.skin) if len(self.r) else False
