"""DeviceEngine — thin Python owner of one ``ls1hip_ctx`` (one per process / GPU).

All compute happens in libls1hip.so (hand-written HIP for gfx950).  This class only marshals numpy arrays and
device pointers across the C ABI; it contains no physics and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .inp import ComponentSet


class DeviceEngine:
    def __init__(self, device: int = 0):
        self.lib = capi.load()
        self.ctx = C.c_void_p()
        rc = self.lib.ls1hip_create(int(device), C.byref(self.ctx))
        if rc != 0:
            msg = self.lib.ls1hip_last_error(None)
            raise capi.Ls1HipError(rc, msg.decode() if msg else "?")
        self.device = device
        self.rc = None
        self.has_rot = False

    # -- lifetime ---------------------------------------------------------------------------------------------------
    def close(self):
        if self.ctx:
            self.lib.ls1hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        capi.check(self.ctx, rc)

    # -- options / model --------------------------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        self._chk(self.lib.ls1hip_set_option(self.ctx, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_long()
        self._chk(self.lib.ls1hip_get_option(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def set_components(self, comps: ComponentSet, rc: float, rc_lj: float | None = None):
        f = comps.flat()
        pad = lambda a: a if a.size else np.zeros(1, dtype=a.dtype)  # noqa: E731
        keep = {k: pad(np.ascontiguousarray(f[k])) for k in ("nlj", "nc", "nd", "nq", "lj", "ch", "dp", "qp", "mass", "I", "mix")}
        self._chk(self.lib.ls1hip_set_components(
            self.ctx, int(f["ncomp"]), capi.iptr(keep["nlj"]), capi.iptr(keep["nc"]), capi.iptr(keep["nd"]),
            capi.iptr(keep["nq"]), capi.dptr(keep["lj"]), capi.dptr(keep["ch"]), capi.dptr(keep["dp"]),
            capi.dptr(keep["qp"]), capi.dptr(keep["mass"]), capi.dptr(keep["I"]), capi.dptr(keep["mix"]),
            float(f["eps_rf"]), float(rc), float(rc if rc_lj is None else rc_lj)))
        rd = np.ascontiguousarray(f["rot_dof"], dtype=np.int32)
        if not np.array_equal(rd, (np.asarray(f["I"]).reshape(-1, 3) != 0.0).sum(axis=1)):
            # an I line that overrides a zero site moment: the reference keeps counting the site-derived degrees of freedom
            self._chk(self.lib.ls1hip_set_rot_dof(self.ctx, int(f["ncomp"]), capi.iptr(rd)))
        self.rc = float(rc)
        self.has_rot = any(c.n_sites > 1 or len(c.dipoles) or len(c.quadrupoles) or
                           np.any(c.lj[:, :3] != 0) or np.any(c.charges[:, :3] != 0) for c in comps.components)

    def lj_table(self):
        n = C.c_int()
        self._chk(self.lib.ls1hip_get_lj_table(self.ctx, C.byref(n), None, None, None))
        m = max(n.value, 1)
        e = np.zeros((m, m)); s = np.zeros((m, m)); sh = np.zeros((m, m))
        self._chk(self.lib.ls1hip_get_lj_table(self.ctx, C.byref(n), capi.dptr(e), capi.dptr(s), capi.dptr(sh)))
        return e, s, sh

    def set_domain(self, global_len, box_min=None, box_max=None, my_rank: int = 0, neighbor_rank=None,
                   periodic: bool = True):
        gl = capi.f64(global_len, (3,))
        bmin = capi.f64(np.zeros(3) if box_min is None else box_min, (3,))
        bmax = capi.f64(gl if box_max is None else box_max, (3,))
        if neighbor_rank is None:
            neighbor_rank = np.full(27, my_rank if periodic else -1, dtype=np.int32)
            neighbor_rank[13] = my_rank
        nb = np.ascontiguousarray(neighbor_rank, dtype=np.int32)
        self._chk(self.lib.ls1hip_set_domain(self.ctx, capi.dptr(gl), capi.dptr(bmin), capi.dptr(bmax), int(my_rank),
                                             capi.iptr(nb)))

    def grid(self):
        dims = np.zeros(3, dtype=np.int32); cl = np.zeros(3); hw = C.c_int()
        self._chk(self.lib.ls1hip_get_grid(self.ctx, capi.iptr(dims), capi.dptr(cl), C.byref(hw)))
        return dims, cl, hw.value

    # -- molecules --------------------------------------------------------------------------------------------------
    def upload(self, ids, cid, r, v, q=None, D=None):
        n = len(ids)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        cid = np.ascontiguousarray(cid, dtype=np.int32)
        r = capi.f64(r, (n, 3)); v = capi.f64(v, (n, 3))
        q = capi.f64(q, (n, 4)) if q is not None else None
        D = capi.f64(D, (n, 3)) if D is not None else None
        self._chk(self.lib.ls1hip_upload(self.ctx, n, ids.ctypes.data_as(capi._u64p), cid.ctypes.data_as(capi._i32p),
                                         capi.dptr(r), capi.dptr(v), capi.dptr(q), capi.dptr(D)))

    def upload_begin(self, n_total: int):
        """Streaming upload: begin -> any number of upload_chunk / upload_records -> upload_end."""
        self._chk(self.lib.ls1hip_upload_begin(self.ctx, int(n_total)))

    def upload_chunk(self, ids, cid, r, v, q=None, D=None):
        n = len(ids)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        cid = np.ascontiguousarray(cid, dtype=np.int32) if cid is not None else None
        r = capi.f64(r, (n, 3)); v = capi.f64(v, (n, 3))
        q = capi.f64(q, (n, 4)) if q is not None else None
        D = capi.f64(D, (n, 3)) if D is not None else None
        self._chk(self.lib.ls1hip_upload_chunk(self.ctx, n, ids.ctypes.data_as(capi._u64p),
                                               cid.ctypes.data_as(capi._i32p) if cid is not None else None,
                                               capi.dptr(r), capi.dptr(v), capi.dptr(q), capi.dptr(D)))

    def upload_chunk_device(self, n: int, id_ptr: int, cid_ptr: int, r_ptr: int, v_ptr: int, q_ptr: int = 0, D_ptr: int = 0):
        """Chunk that already lives in device memory (AoS arrays: id u64/i64 [n], cid i32 [n] or 0, r [n,3], v [n,3], ...)."""
        vp = lambda p: C.c_void_p(p) if p else None  # noqa: E731
        self._chk(self.lib.ls1hip_upload_chunk_device(self.ctx, int(n), vp(id_ptr), vp(cid_ptr), vp(r_ptr), vp(v_ptr),
                                                      vp(q_ptr), vp(D_ptr)))

    def upload_records(self, records, fmt: int = capi.REC_ICRVQD):
        """records: bytes-like / uint8 array of packed checkpoint records (116 / 60 / 56 bytes each)."""
        buf = np.frombuffer(records, dtype=np.uint8) if not isinstance(records, np.ndarray) else records.view(np.uint8).reshape(-1)
        buf = np.ascontiguousarray(buf)
        rb = capi.REC_BYTES[fmt]
        if buf.size % rb:
            raise ValueError(f"record buffer of {buf.size} bytes is not a multiple of {rb}")
        self._chk(self.lib.ls1hip_upload_records(self.ctx, buf.size // rb, C.c_void_p(buf.ctypes.data), int(fmt)))

    def upload_end(self):
        self._chk(self.lib.ls1hip_upload_end(self.ctx))

    def download_records(self, first: int = 0, n: int | None = None):
        """Owned molecules [first, first+n) as packed ICRVQD records (uint8 array of n*116 bytes)."""
        if n is None:
            n = self.count()[0] - first
        out = np.zeros(n * 116, dtype=np.uint8)
        self._chk(self.lib.ls1hip_download_records(self.ctx, int(first), int(n), C.c_void_p(out.ctypes.data)))
        return out

    def count(self):
        a = C.c_size_t(); b = C.c_size_t()
        self._chk(self.lib.ls1hip_count(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def download_state(self):
        n, _ = self.count()
        ids = np.zeros(n, dtype=np.uint64); cid = np.zeros(n, dtype=np.int32)
        r = np.zeros((n, 3)); v = np.zeros((n, 3)); q = np.zeros((n, 4)); D = np.zeros((n, 3))
        self._chk(self.lib.ls1hip_download_state(self.ctx, n, ids.ctypes.data_as(capi._u64p),
                                                 cid.ctypes.data_as(capi._i32p), capi.dptr(r), capi.dptr(v),
                                                 capi.dptr(q), capi.dptr(D)))
        return dict(ids=ids, cid=cid, r=r, v=v, q=q, D=D)

    def download_ids(self):
        """Molecule ids only (device order) — the cheap survival check at sizes where a full download is 10 GB."""
        n, _ = self.count()
        ids = np.zeros(n, dtype=np.uint64)
        self._chk(self.lib.ls1hip_download_state(self.ctx, n, ids.ctypes.data_as(capi._u64p), None, None, None, None, None))
        return ids

    def download_velocities(self):
        n, _ = self.count()
        v = np.zeros((n, 3))
        self._chk(self.lib.ls1hip_download_state(self.ctx, n, None, None, None, capi.dptr(v), None, None))
        return v

    def download_forces(self, with_vi: bool = False):
        n, _ = self.count()
        F = np.zeros((n, 3)); M = np.zeros((n, 3)); Vi = np.zeros((n, 3)) if with_vi else None
        self._chk(self.lib.ls1hip_download_forces(self.ctx, n, capi.dptr(F), capi.dptr(M), capi.dptr(Vi)))
        return dict(F=F, M=M, Vi=Vi)

    # -- step pieces ------------------------------------------------------------------------------------------------
    def kick_drift(self, dt):
        self._chk(self.lib.ls1hip_kick_drift(self.ctx, float(dt)))

    def forces_kick_drift(self, which: int, dt: float, want_macro: bool = False):
        """Force pass fused with kick + kick + drift (reduced-memory mode); see ls1hip_forces_kick_drift."""
        if not want_macro:
            self._chk(self.lib.ls1hip_forces_kick_drift(self.ctx, int(which), float(dt), None, None))
            return None
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_forces_kick_drift(self.ctx, int(which), float(dt), C.byref(u), C.byref(w)))
        return u.value, w.value

    def can_fuse_integration(self) -> bool:
        return bool(self.get_option("can_fuse_integration"))

    def kinetic_sums(self):
        a = C.c_double(); b = C.c_double(); n = C.c_uint64(); rd = C.c_uint64()
        self._chk(self.lib.ls1hip_kinetic_sums(self.ctx, C.byref(a), C.byref(b), C.byref(n), C.byref(rd)))
        return a.value, b.value, n.value, rd.value

    def traversal_mark(self):
        self._chk(self.lib.ls1hip_traversal_mark(self.ctx))

    def traversal_sums(self):
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_traversal_sums(self.ctx, C.byref(u), C.byref(w)))
        return u.value, w.value

    def scale_kick_drift(self, beta_trans, beta_rot, dt):
        """scale_velocities + kick_drift in one pass (ls1hip_scale_kick_drift)"""
        self._chk(self.lib.ls1hip_scale_kick_drift(self.ctx, float(beta_trans), float(beta_rot), float(dt)))

    def kick_then_kick_drift(self, dt: float):
        """Post-force kick of step n fused with the pre-force kick + drift of step n+1 (one pass)."""
        self._chk(self.lib.ls1hip_kick_then_kick_drift(self.ctx, float(dt)))

    def rebin(self):
        self._chk(self.lib.ls1hip_rebin(self.ctx))

    def halo(self):
        self._chk(self.lib.ls1hip_halo(self.ctx))

    def forces(self, which: int = 0, want_macro: bool = True):
        if not want_macro:
            self._chk(self.lib.ls1hip_forces(self.ctx, int(which), None, None))
            return None
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_forces(self.ctx, int(which), C.byref(u), C.byref(w)))
        return u.value, w.value

    def kick(self, dt_half, want_sums: bool = True):
        if not want_sums:
            self._chk(self.lib.ls1hip_kick(self.ctx, float(dt_half), None, None, None, None))
            return None
        a = C.c_double(); b = C.c_double(); n = C.c_uint64(); rd = C.c_uint64()
        self._chk(self.lib.ls1hip_kick(self.ctx, float(dt_half), C.byref(a), C.byref(b), C.byref(n), C.byref(rd)))
        return a.value, b.value, n.value, rd.value

    def kinetic_sums_by_component(self, ncomp: int):
        """{summv2, sumIw2, n, rot_dof} arrays per component (component-wise thermostats)"""
        a = np.zeros(ncomp); b = np.zeros(ncomp); n = np.zeros(ncomp, np.uint64); rd = np.zeros(ncomp, np.uint64)
        self._chk(self.lib.ls1hip_kinetic_sums_by_component(self.ctx, int(ncomp), capi.dptr(a), capi.dptr(b),
                                                            n.ctypes.data_as(capi._u64p), rd.ctypes.data_as(capi._u64p)))
        return dict(summv2=a, sumIw2=b, n=n, rot_dof=rd)

    def scale_kick_drift_components(self, beta_trans, beta_rot, dt):
        bt = np.ascontiguousarray(beta_trans, dtype=np.float64); br = np.ascontiguousarray(beta_rot, dtype=np.float64)
        self._chk(self.lib.ls1hip_scale_kick_drift_components(self.ctx, len(bt), capi.dptr(bt), capi.dptr(br), float(dt)))

    def scale_velocities(self, beta_trans, beta_rot=1.0):
        self._chk(self.lib.ls1hip_scale_velocities(self.ctx, float(beta_trans), float(beta_rot)))

    def set_thermostat(self, enabled: bool, target_temperature: float = 0.0):
        self._chk(self.lib.ls1hip_set_thermostat(self.ctx, int(bool(enabled)), float(target_temperature)))

    def set_verlet(self, skin: float | None, force: bool = False):
        """Neighbour-list reuse in run() (skin > 0) or off (None / 0); call between set_components and set_domain.
        force: also when the brick regions exceed the LDS staging area (slow global-memory path; tests)."""
        on = bool(skin) and skin > 0
        self._chk(self.lib.ls1hip_set_verlet(self.ctx, (2 if force else 1) if on else 0, float(skin) if on else 0.0))

    # list mode, piecewise (multi-rank loops; see ls1hip.h)
    def can_verlet(self) -> bool:
        return bool(self.get_option("verlet_lists")) and self.can_fuse_integration()

    def verlet_build(self):
        self._chk(self.lib.ls1hip_verlet_build(self.ctx))

    def update(self) -> bool:
        """update + exchange + caches of a single-rank domain, list-aware; True if it re-binned (see ls1hip_update)"""
        n = C.c_int()
        self._chk(self.lib.ls1hip_update(self.ctx, C.byref(n)))
        return bool(n.value)

    def halo_refresh(self):
        self._chk(self.lib.ls1hip_halo_refresh(self.ctx))

    def forces_list(self, which: int, dt: float = 0.0, want_macro: bool = False):
        if not want_macro:
            self._chk(self.lib.ls1hip_forces_list(self.ctx, int(which), float(dt), None, None))
            return None
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_forces_list(self.ctx, int(which), float(dt), C.byref(u), C.byref(w)))
        return u.value, w.value

    def forces_list_kick(self, dt_half: float, want_macro: bool = False):
        """list traversal + the step's post-force kick in one pass (ls1hip_forces_list_kick); kinetic sums: kinetic_sums()"""
        if not want_macro:
            self._chk(self.lib.ls1hip_forces_list_kick(self.ctx, float(dt_half), None, None))
            return None
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_forces_list_kick(self.ctx, float(dt_half), C.byref(u), C.byref(w)))
        return u.value, w.value

    def verlet_poll(self) -> bool:
        n = C.c_int()
        self._chk(self.lib.ls1hip_verlet_poll(self.ctx, C.byref(n)))
        return bool(n.value)

    def long_range_homogeneous(self, n_per_component, global_rho):
        n = np.ascontiguousarray(n_per_component, dtype=np.uint64)
        u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_long_range_homogeneous(self.ctx, n.ctypes.data_as(capi._u64p), float(global_rho),
                                                         C.byref(u), C.byref(w)))
        return u.value, w.value

    def run(self, dt, nsteps):
        out = np.zeros(6)
        self._chk(self.lib.ls1hip_run(self.ctx, float(dt), int(nsteps), capi.dptr(out)))
        return dict(upot=out[0], virial=out[1], summv2=out[2], sumIw2=out[3], n=int(out[4]), rot_dof=int(out[5]))

    def run_log(self):
        """[nsteps, 6] per-step globals {upot, virial, summv2, sumIw2, N, rotDOF} of the last run (NaN = not computed)."""
        n = C.c_size_t()
        self._chk(self.lib.ls1hip_run_log(self.ctx, 0, None, C.byref(n)))
        out = np.zeros((n.value, 6))
        if n.value:
            self._chk(self.lib.ls1hip_run_log(self.ctx, n.value, capi.dptr(out), C.byref(n)))
        return out

    # -- multi-GPU plumbing -----------------------------------------------------------------------------------------
    def export_counts(self, kind: int):
        c = np.zeros(27, dtype=np.uint64)
        self._chk(self.lib.ls1hip_export_counts(self.ctx, int(kind), c.ctypes.data_as(capi._u64p)))
        return c

    def export_pack(self, kind: int, direction: int, dev_ptr: int, cap: int):
        self._chk(self.lib.ls1hip_export_pack(self.ctx, int(kind), int(direction), C.c_void_p(dev_ptr), int(cap)))

    def export_pack_dirs(self, kind: int, directions, dev_ptr: int, cap: int):
        """Records of all `directions` back to back (one message per peer, one synchronisation)."""
        d = (C.c_int * len(directions))(*[int(x) for x in directions])
        self._chk(self.lib.ls1hip_export_pack_dirs(self.ctx, int(kind), d, len(directions), C.c_void_p(dev_ptr), int(cap)))

    def import_records(self, kind: int, dev_ptr: int, n: int):
        """Asynchronous: the buffer must stay alive until import_done(kind) returns."""
        self._chk(self.lib.ls1hip_import(self.ctx, int(kind), C.c_void_p(dev_ptr), int(n)))

    def import_done(self, kind: int):
        self._chk(self.lib.ls1hip_import_done(self.ctx, int(kind)))

    # -- seam A -----------------------------------------------------------------------------------------------------
    def soa_forces(self, cell_dims, cell_start, r, q, cid):
        n = len(r)
        dims = np.ascontiguousarray(cell_dims, dtype=np.int32)
        cs = np.ascontiguousarray(cell_start, dtype=np.uint32)
        r = capi.f64(r, (n, 3)); q = capi.f64(q, (n, 4)) if q is not None else None
        cid = np.ascontiguousarray(cid, dtype=np.int32)
        F = np.zeros((n, 3)); M = np.zeros((n, 3)); Vi = np.zeros((n, 3)); u = C.c_double(); w = C.c_double()
        self._chk(self.lib.ls1hip_soa_forces(self.ctx, capi.iptr(dims), cs.ctypes.data_as(capi._u32p), n, capi.dptr(r),
                                             capi.dptr(q), cid.ctypes.data_as(capi._i32p), capi.dptr(F), capi.dptr(M),
                                             capi.dptr(Vi), C.byref(u), C.byref(w)))
        return dict(F=F, M=M, Vi=Vi, upot=u.value, virial=w.value)

    # -- measurement ------------------------------------------------------------------------------------------------
    def timing_enable(self, on):
        """0 / False = off, 1 / True = every phase, 2 = force passes only"""
        self._chk(self.lib.ls1hip_timing_enable(self.ctx, int(on)))

    def timing_reset(self):
        self._chk(self.lib.ls1hip_timing_reset(self.ctx))

    def timing(self, name: str):
        ms = C.c_double(); n = C.c_uint64()
        self._chk(self.lib.ls1hip_timing(self.ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def pair_stats(self):
        a = C.c_uint64(); b = C.c_uint64()
        self._chk(self.lib.ls1hip_pair_stats(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value
