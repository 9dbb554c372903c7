"""ctypes binding of oracle/ls1_oracle.c (the CPU restatement).  TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libls1oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ls1_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.ls1o_create.restype = C.c_void_p
        L.ls1o_create.argtypes = [C.c_int, _ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                  C.c_double, C.c_double, C.c_double]
        L.ls1o_destroy.argtypes = [C.c_void_p]
        L.ls1o_lj_table.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.ls1o_wrap.argtypes = [C.c_size_t, _dp, _dp]
        L.ls1o_forces.restype = C.c_int
        L.ls1o_forces.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _ip, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.ls1o_lrc_homogeneous.restype = C.c_int
        L.ls1o_lrc_homogeneous.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), _dp]
        L.ls1o_lrc_finish.argtypes = [C.c_void_p, _dp, C.c_double, C.c_ulong, _dp]
        L.ls1o_upd_preF.argtypes = [C.c_void_p, C.c_size_t, C.c_double, _ip, _dp, _dp, _dp, _dp, _dp, _dp]
        L.ls1o_upd_postF.argtypes = [C.c_void_p, C.c_size_t, C.c_double, _ip, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class Oracle:
    """One component set + cutoffs; evaluates forces / leapfrog on numpy arrays."""

    def __init__(self, flat: dict, rc: float, rc_lj: float | None = None):
        self.flat = flat
        self.rc = float(rc)
        self.rc_lj = float(rc if rc_lj is None else rc_lj)
        f = flat
        self._keep = [np.ascontiguousarray(f[k]) for k in ("nlj", "nc", "nd", "nq", "lj", "ch", "dp", "qp", "mass", "I", "mix")]
        k = self._keep
        pad = lambda a: a if a.size else np.zeros(1, dtype=a.dtype)  # noqa: E731
        self._keep = [pad(a) for a in k]
        k = self._keep
        self.h = lib().ls1o_create(int(f["ncomp"]), _i(k[0]), _i(k[1]), _i(k[2]), _i(k[3]), _d(k[4]), _d(k[5]),
                                   _d(k[6]), _d(k[7]), _d(k[8]), _d(k[9]), _d(k[10]),
                                   float(f["eps_rf"]), self.rc, self.rc_lj)
        self.ncenters = int(f["nlj"].sum())

    def __del__(self):
        try:
            if self.h:
                lib().ls1o_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def lj_table(self):
        n = max(self.ncenters, 1)
        e = np.zeros((n, n)); s = np.zeros((n, n)); sh = np.zeros((n, n))
        lib().ls1o_lj_table(self.h, _d(e), _d(s), _d(sh))
        return e, s, sh

    @staticmethod
    def wrap(r: np.ndarray, L) -> np.ndarray:
        r = np.ascontiguousarray(r, dtype=np.float64).copy()
        Ls = np.ascontiguousarray(L, dtype=np.float64)
        lib().ls1o_wrap(len(r), _d(r), _d(Ls))
        return r

    def forces(self, r, q, cid, L, periodic: bool):
        n = len(r)
        r = np.ascontiguousarray(r, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        cid = np.ascontiguousarray(cid, dtype=np.int32)
        Ls = np.ascontiguousarray(L, dtype=np.float64)
        F = np.zeros((n, 3)); M = np.zeros((n, 3)); Vi = np.zeros((n, 3)); out = np.zeros(4)
        rc = lib().ls1o_forces(self.h, n, _d(r), _d(q), _i(cid), int(bool(periodic)), _d(Ls), _d(F), _d(M), _d(Vi), _d(out))
        if rc != 0:
            raise RuntimeError(f"ls1o_forces failed: {rc}")
        return dict(F=F, M=M, Vi=Vi, upot=out[0], virial=out[1], upot_lj=out[2], upot_x=out[3])

    def lrc_homogeneous(self, cid, L):
        """Homogeneous long-range correction (UpotCorr, VirialCorr) for the molecule set `cid` in box L."""
        ncomp = int(self.flat["ncomp"])
        nmol = (C.c_ulong * ncomp)(*[int((np.asarray(cid) == k).sum()) for k in range(ncomp)])
        sums = np.zeros(3); out = np.zeros(2)
        if lib().ls1o_lrc_homogeneous(self.h, nmol, _d(sums)) != 0:
            raise RuntimeError("cutoff too small for the long-range correction")
        N = len(cid)
        lib().ls1o_lrc_finish(self.h, _d(sums), N / float(np.prod(L)), N, _d(out))
        return float(out[0]), float(out[1])

    def upd_preF(self, dt, cid, r, v, q, D, F, M):
        cid = np.ascontiguousarray(cid, dtype=np.int32)
        lib().ls1o_upd_preF(self.h, len(r), float(dt), _i(cid), _d(r), _d(v), _d(q), _d(D), _d(F), _d(M))

    def upd_postF(self, dt_half, cid, v, q, D, F, M):
        cid = np.ascontiguousarray(cid, dtype=np.int32)
        s = np.zeros(2)
        lib().ls1o_upd_postF(self.h, len(v), float(dt_half), _i(cid), _d(v), _d(q), _d(D), _d(F), _d(M), _d(s))
        return float(s[0]), float(s[1])

    @staticmethod
    def global_betas(summv2, sumIw2, n, rot_dof, target_T, tfactor=1.0):
        """Domain::calculateGlobalValues, thermostat 0 (/root/reference/src/Domain.cpp:225-240)."""
        Ti = tfactor * target_T
        if Ti > 0.0 and n > 0:
            bt = (3.0 * n * Ti / summv2) ** 0.4
            br = 1.0 if sumIw2 == 0.0 else (rot_dof * Ti / sumIw2) ** 0.4
            return bt, br
        return 1.0, 1.0

    def rot_dof(self, cid):
        """Sum of Component::getRotationalDegreesOfFreedom over the molecules (Leapfrog.cpp:126)."""
        if "rot_dof" in self.flat:  # Component::_rot_dof: from the site masses, before an I-line override (Component.cpp:140-167)
            per_comp = np.asarray(self.flat["rot_dof"])
        else:
            per_comp = (np.asarray(self.flat["I"]).reshape(-1, 3) != 0.0).sum(axis=1)
        return int(per_comp[np.asarray(cid)].sum())

    def step(self, dt, cid, r, v, q, D, F, M, L, periodic=True, target_T=None, thermostats=None):
        """One full time step in the reference's order (Simulation.cpp:995-1099): pre-force kick+drift,
        wrap + halo + forces, post-force kick.  Arrays are updated in place; returns the force dict.
        thermostats = (thermostat_T {id: T}, comp_thermostat {component: id}) selects the component-wise branch."""
        self.upd_preF(dt, cid, r, v, q, D, F, M)
        if periodic:
            r[:] = self.wrap(r, L)
        out = self.forces(r, q, cid, L, periodic)
        F[:] = out["F"]; M[:] = out["M"]
        if thermostats is not None and thermostats[1]:
            # Leapfrog::transition2to3, several thermostats (Leapfrog.cpp:84-112): the post-force kick with one set of sums per
            # thermostat id; Domain::calculateGlobalValues (Domain.cpp:204-240) turns each into its own pair of betas;
            # VelocityScalingThermostat::apply, component-wise branch (VelocityScalingThermostat.cpp:47-71, directed velocity 0)
            th_T, comp_th = thermostats
            cid = np.asarray(cid)
            th_of = np.array([comp_th.get(int(c), 0) for c in range(int(self.flat["ncomp"]))])[cid]
            out["summv2"] = out["sumIw2"] = 0.0
            out["betas"] = {}
            for th in sorted(set(th_of.tolist())):
                idx = np.nonzero(th_of == th)[0]
                vs, qs, Ds, Fs, Ms = (np.ascontiguousarray(a[idx]) for a in (v, q, D, F, M))
                mv2, iw2 = self.upd_postF(0.5 * dt, cid[idx], vs, qs, Ds, Fs, Ms)
                bt, br = self.global_betas(mv2, iw2, len(idx), self.rot_dof(cid[idx]), th_T.get(th, 0.0))
                v[idx] = vs * bt
                D[idx] = Ds * br
                out["betas"][th] = (bt, br)
                out["summv2"] += mv2 * bt * bt  # sums of the state the step leaves behind (after the scaling)
                out["sumIw2"] += iw2 * br * br
            return out
        out["summv2"], out["sumIw2"] = self.upd_postF(0.5 * dt, cid, v, q, D, F, M)
        if target_T is not None:
            # VelocityScalingThermostat::apply, global branch (thermostats/VelocityScalingThermostat.cpp:80-96)
            bt, br = self.global_betas(out["summv2"], out["sumIw2"], len(r), self.rot_dof(cid), target_T)
            v *= bt
            D *= br
            out["beta_trans"], out["beta_rot"] = bt, br
        return out
