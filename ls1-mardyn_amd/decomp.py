"""Multi-GPU spatial domain decomposition: regular 3-D grid of sub-boxes, one process per GPU, ghost-cell halo
exchange as point-to-point messages over ``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU box;
"gloo" in the CPU tests).

Mirrors the reference's regular-grid decomposition and direct full-shell exchange:
  DomainDecomposition (MPI_Dims_create / MPI_Cart_create, bounding boxes)   parallel/DomainDecomposition.cpp:19-41,112-123
  NeighbourCommunicationScheme (direct: LEAVING_ONLY, then HALO_COPIES)      parallel/NeighbourCommunicationScheme.cpp:115-136
  FullShell halo regions (26 directions)                                      parallel/ZonalMethods/ZonalMethod.cpp:85-128
  NonBlockingMPIMultiStepHandler (comm / inner-cell compute overlap)          parallel/NonBlockingMPIMultiStepHandler.cpp:30-97
  Domain::calculateGlobalValues (allreduce of U_pot, virial, kinetic sums)    Domain.cpp:151-176

Design for xGMI: on a 2x2x2 periodic grid every GPU has exactly 7 distinct peers (= its 7 xGMI links), so all 26
directions are merged per PEER into one message; every peer pair exchanges one send + one recv per phase
(batch_isend_irecv), there is no ring and no per-link serialisation.  The halo transfer runs on RCCL's stream
while the compute stream traverses the inner cells.  Scalars are reduced with one tiny all_reduce.

The exchanger is engine-agnostic (it needs export_counts / export_pack / import_records / import_done), so the
CPU test-suite drives it with a numpy stand-in engine under gloo.
"""
from __future__ import annotations

import os

import numpy as np

LEAVING, HALO, REFRESH = 0, 1, 2
RECORD_DOUBLES = {LEAVING: 15, HALO: 9, REFRESH: 3}


class DecompositionError(RuntimeError):
    """Raised on EVERY rank when any rank's engine reported an error during an exchange (capacity overflow, lost
    molecule ...): an error raised on one rank only would leave the others blocked in the next collective."""


def dims_create(world: int):
    """Balanced 3-D factorisation, largest factor first (what MPI_Dims_create returns for 1, 2, 4, 8, ...)."""
    dims = [1, 1, 1]
    n = world
    f = 2
    factors = []
    while n > 1:
        while n % f == 0:
            factors.append(f)
            n //= f
        f += 1
    for p in sorted(factors, reverse=True):
        dims[int(np.argmin(dims))] *= p
    return tuple(sorted(dims, reverse=True))


class CartesianDecomposition:
    """Geometry of the rank grid: coordinates, bounding boxes, the 27-entry neighbour table."""

    def __init__(self, world: int, rank: int, global_len, grid=None, periodic=(True, True, True), loopback=False):
        # loopback (rehearsal / diagnostics): periodic images that a rank would create locally (neighbour == itself) are
        # routed through the transport instead, as messages to the alias id world + rank (sent to and received from the
        # own rank).  Exercises the full export -> RCCL -> import path, on ONE GPU if need be.
        self.loopback = bool(loopback)
        self.world, self.rank = int(world), int(rank)
        self.grid = tuple(grid) if grid is not None else dims_create(world)
        assert int(np.prod(self.grid)) == world
        self.global_len = np.asarray(global_len, dtype=np.float64)
        self.periodic = tuple(periodic)
        self.coords = self.coords_of(rank)

    def coords_of(self, rank):
        gx, gy, gz = self.grid
        return (rank % gx, (rank // gx) % gy, rank // (gx * gy))

    def rank_of(self, c):
        gx, gy, gz = self.grid
        return (c[2] * gy + c[1]) * gx + c[0]

    def bounding_box(self, rank=None):
        """[c_d L/g_d, (c_d+1) L/g_d): DomainDecomposition::getBoundingBoxMin/Max (DomainDecomposition.cpp:114-123).
        The upper bound of the last rank is exactly L so that ls1hip_set_domain recognises the global faces."""
        c = self.coords if rank is None else self.coords_of(rank)
        lo = np.array([c[d] * self.global_len[d] / self.grid[d] for d in range(3)])
        hi = np.array([(c[d] + 1) * self.global_len[d] / self.grid[d] if c[d] + 1 < self.grid[d] else self.global_len[d]
                       for d in range(3)])
        return lo, hi

    def neighbor_table(self, rank=None):
        """neighbor_rank[27], index (sz+1)*9+(sy+1)*3+(sx+1); -1 = open boundary."""
        c = self.coords if rank is None else self.coords_of(rank)
        tab = np.full(27, -1, dtype=np.int32)
        for sz in (-1, 0, 1):
            for sy in (-1, 0, 1):
                for sx in (-1, 0, 1):
                    s = (sx, sy, sz)
                    n = []
                    ok = True
                    for d in range(3):
                        x = c[d] + s[d]
                        if x < 0 or x >= self.grid[d]:
                            if not self.periodic[d]:
                                ok = False
                                break
                            x %= self.grid[d]
                        n.append(x)
                    tab[(sz + 1) * 9 + (sy + 1) * 3 + (sx + 1)] = self.rank_of(n) if ok else -1
        if self.loopback:
            owner = self.rank if rank is None else int(rank)
            alias = np.where(tab == owner, self.world + owner, tab)
            alias[13] = owner
            tab = alias.astype(np.int32)
        return tab

    def real_rank(self, peer: int) -> int:
        """process that serves peer id `peer` (loopback aliases map to their owner)"""
        return int(peer) - self.world if int(peer) >= self.world else int(peer)

    def peers(self):
        me = self.rank
        return sorted({int(r) for r in self.neighbor_table() if r >= 0 and r != me})

    def describe(self):
        return f"{self.grid[0]}x{self.grid[1]}x{self.grid[2]} sub-boxes, full-shell halo, {len(self.peers())} peers/GPU"


class HaloExchanger:
    """Per-peer merged exchange of packed records (leaving molecules or halo copies)."""

    def __init__(self, decomp: CartesianDecomposition, engine, dist, device, group=None, stage_through_host=False):
        self.dc, self.engine, self.dist, self.device, self.group = decomp, engine, dist, device, group
        # stage_through_host: the engine works on `device` buffers but the process group moves CPU tensors (gloo);
        # used to rehearse the multi-process path when several ranks share one GPU (RCCL needs one GPU per rank)
        self.stage = bool(stage_through_host)
        self.nbr = decomp.neighbor_table()
        self.peers = decomp.peers()
        # directions of every rank that point at me, per source rank (to size the receive)
        self._incoming = {}
        for p in self.peers:
            t = decomp.neighbor_table(decomp.real_rank(p))
            # a message from peer p carries the directions of p's table that point at me (at my alias for a loopback)
            me = p if decomp.real_rank(p) == decomp.rank else decomp.rank
            self._incoming[p] = [d for d in range(27) if d != 13 and t[d] == me]
        self._outgoing = {p: [d for d in range(27) if d != 13 and self.nbr[d] == p] for p in self.peers}
        self._torch = __import__("torch")
        # Count exchange (27 record counts per rank, needed to size the receives).  It must not touch the HIP null stream:
        # that stream shares its hardware queue with the engine's main stream, so its small copies would queue behind
        # the 2.4 ms inner-cell kernel and serialise the halo exchange with it (rocprof trace).  Two transports:
        #   "nccl" (default): device all_gather issued from the dedicated high-priority side stream (~50 us);
        #   "gloo": host-side group (LS1_COUNTS_TRANSPORT=gloo) — no GPU work at all, but TCP round trips (measured
        #           0.36 ms for 2 ranks, 2.8 ms for 8 ranks on an 8-core host), kept as a fallback.
        self._comm_stream = None
        self.meta_group = None
        self.force_count_exchange = False  # diagnostics: run the count all_gather even with one rank
        self._deferred = None  # engine error caught on this rank, reported collectively with the next count exchange
        if not self.stage and decomp.world > 1 and os.environ.get("LS1_COUNTS_TRANSPORT", "nccl") == "gloo":
            self.meta_group = dist.new_group(backend="gloo")

    def _self_p2p_ok(self):
        """RCCL accepts grouped send/recv to the own rank (the loopback rehearsal relies on it); gloo does not."""
        if self.stage:
            return False
        try:
            return self.dist.get_backend(self.group) == "nccl"
        except Exception:
            return False

    def _side_stream(self):
        torch = self._torch
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.device, priority=-1)
        return self._comm_stream

    def _gather_counts(self, counts):
        """[world, len(counts)] int64 table of every rank's export counts (+ status column)"""
        torch, dist = self._torch, self.dist
        if self.dc.world == 1 and not self.force_count_exchange:
            return counts[None, :]
        if self.stage or self.meta_group is not None:
            mine = torch.from_numpy(counts)
            lst = [torch.empty_like(mine) for _ in range(self.dc.world)]
            dist.all_gather(lst, mine, group=self.group if self.meta_group is None else self.meta_group)
            return torch.stack(lst).numpy()
        if self.device.type != "cuda":
            mine = torch.from_numpy(counts)
            lst = [torch.empty_like(mine) for _ in range(self.dc.world)]
            dist.all_gather(lst, mine, group=self.group)
            return torch.stack(lst).numpy()
        st = self._side_stream()
        with torch.cuda.stream(st):
            mine = torch.from_numpy(counts).to(self.device, non_blocking=True)
            out = torch.empty((self.dc.world, len(counts)), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(out, mine, group=self.group)
            host = out.to("cpu", non_blocking=True)
            st.synchronize()
        return host.numpy()

    def _ptr(self, t):
        return t.data_ptr()

    def _buffer(self, which, n, device):
        """persistent, grow-only message buffers (one for all outgoing, one for all incoming messages of an exchange): no
        allocator traffic in the step loop.  Reuse across exchanges is safe: every exchange ends with import_done, which
        returns after the engine has consumed the receive buffer, and the sends have been waited for."""
        if not hasattr(self, "_bufs"):
            self._bufs = {}
        key = (which, str(device))
        t = self._bufs.get(key)
        if t is None or t.numel() < n:
            t = self._torch.empty(int(n * 1.25) + 1024, dtype=self._torch.float64, device=device)
            self._bufs[key] = t
        return t[:n]

    def exchange(self, kind: int, overlap_fn=None):
        """Counts -> all_gather; payload -> one message per peer.  `overlap_fn` (optional) is called after the sends
        and receives have been posted and before they are waited for (inner-cell force launch goes here)."""
        # Error handling is COLLECTIVE: the count table carries a status column.  An engine error on this rank (export /
        # halo capacity overflow, a molecule beyond the halo region — raised by export_counts / import_done) is not thrown
        # here but gathered, and every rank raises after the all_gather, before any point-to-point operation is posted.
        try:
            counts = self.engine.export_counts(kind).astype(np.int64)  # [27]
        except Exception as e:  # noqa: BLE001 (any engine failure must reach the other ranks)
            self._deferred = self._deferred or e
            counts = np.zeros(27, dtype=np.int64)
        status = np.int64(1 if self._deferred is not None else 0)
        allc = counts[None, :]
        if self.peers:
            table = self._gather_counts(np.concatenate([counts, [status]]))
            allc, stat = table[:, :27], table[:, 27]
            if np.any(stat != 0):
                bad = [int(r) for r in np.nonzero(stat)[0]]
                mine = f"; this rank: {self._deferred}" if self._deferred is not None else ""
                raise DecompositionError(f"rank(s) {bad} reported an engine error during the exchange of kind {kind}{mine}")
        elif self._deferred is not None:
            raise self._deferred
        if kind == HALO:
            self._halo_counts = (counts.copy(), np.array(allc, copy=True))  # what every position refresh until the next build repeats
        self._transfer(kind, counts, allc, overlap_fn)

    def exchange_refresh(self, overlap_fn=None):
        """List-reuse step: the halo copies of the last list build get their current positions.  Same messages as that
        build's halo exchange (same peers, directions, record order and counts) with 3 doubles per record — no count
        exchange, no host synchronisation on counters."""
        counts, allc = self._halo_counts
        self._transfer(REFRESH, counts, allc, overlap_fn)

    def _transfer(self, kind, counts, allc, overlap_fn=None):
        torch, dist = self._torch, self.dist
        w = RECORD_DOUBLES[kind]
        # ONE device buffer holds every outgoing message (directions grouped by peer) and ONE every incoming message:
        # one pack call / one import call / one stream synchronisation per exchange, whatever the number of peers
        n_out = {p: int(sum(counts[d] for d in self._outgoing[p])) for p in self.peers}
        n_in = {p: int(sum(allc[self.dc.real_rank(p)][d] for d in self._incoming[p])) for p in self.peers}
        tot_out, tot_in = sum(n_out.values()), sum(n_in.values())
        ops = []
        sbuf = rbuf = None
        self_copies = []  # (send view, peer) of loopback messages when the transport cannot send to the own rank (gloo)
        self_p2p = self._self_p2p_ok()
        if tot_out:
            sbuf = self._buffer("s", tot_out * w, self.device)
            order = [d for p in self.peers for d in self._outgoing[p] if counts[d]]
            self.engine.export_pack_dirs(kind, order, self._ptr(sbuf), tot_out)
            if self.stage:
                torch.cuda.synchronize()
                sbuf = sbuf.cpu()
            off = 0
            for p in self.peers:
                if n_out[p]:
                    view = sbuf[off * w:(off + n_out[p]) * w]
                    if self.dc.real_rank(p) == self.dc.rank and not self_p2p:
                        self_copies.append((view, p))
                    else:
                        ops.append(dist.P2POp(dist.isend, view, self.dc.real_rank(p), group=self.group))
                    off += n_out[p]
        if tot_in:
            rbuf = self._buffer("r", tot_in * w, "cpu" if self.stage else self.device)
            off = 0
            for p in self.peers:
                if n_in[p]:
                    view = rbuf[off * w:(off + n_in[p]) * w]
                    if self.dc.real_rank(p) == self.dc.rank and not self_p2p:
                        src = [v_ for v_, q_ in self_copies if q_ == p]
                        view.copy_(src[0])
                    else:
                        ops.append(dist.P2POp(dist.irecv, view, self.dc.real_rank(p), group=self.group))
                    off += n_in[p]
        # The transfers are issued from a dedicated side stream: RCCL orders its kernels after an event on the CURRENT
        # torch stream, and the default (null) stream shares its hardware queue with the engine's main stream — the
        # event, and with it the whole transfer, would wait for the inner-cell kernel (seen in the rocprof trace).
        if self.device.type == "cuda":
            st = self._side_stream()
            with torch.cuda.stream(st):
                reqs = dist.batch_isend_irecv(ops) if ops else []
                if overlap_fn is not None:
                    overlap_fn()
                for r in reqs:
                    r.wait()
                st.synchronize()
        else:
            reqs = dist.batch_isend_irecv(ops) if ops else []
            if overlap_fn is not None:
                overlap_fn()
            for r in reqs:
                r.wait()
        if tot_in:
            if self.stage:
                rbuf = rbuf.to(self.device)
            self.engine.import_records(kind, self._ptr(rbuf), tot_in)  # asynchronous: rbuf lives until import_done
        try:
            self.engine.import_done(kind)
        except Exception as e:  # noqa: BLE001: reported by all ranks together at the next count exchange
            if not self.peers:
                raise
            self._deferred = self._deferred or e


class DistributedSimulation:
    """One rank of the decomposed time loop (the overlapped variant of Simulation::simulate,
    Simulation.cpp:1015-1019,1301-1319 -> NonBlockingMPIMultiStepHandler::performOverlappingTasks)."""

    def __init__(self, decomp: CartesianDecomposition, engine, dist, device, group=None, stage_through_host=False):
        self.dc, self.engine, self.dist, self.device, self.group = decomp, engine, dist, device, group
        self.stage = bool(stage_through_host)
        self.ex = HaloExchanger(decomp, engine, dist, device, group, stage_through_host)
        self.grid_desc = decomp.describe()
        self.n_global = None
        self._torch = __import__("torch")

    def initial_forces(self):
        e = self.engine
        e.rebin()
        self.ex.exchange(LEAVING)
        e.forces(1, want_macro=False)
        e.halo()
        self.ex.exchange(HALO)
        return e.forces(2)

    def _exchange_and_forces(self, want, fuse_dt=None):
        """re-bin + migration, then the inner-cell traversal is launched FIRST (it needs the owned molecules only) and
        the whole halo phase — image generation, count exchange, packing, RCCL transfer, import, sort — runs while
        it computes: the engine moves the halo phase to its second (high-priority) stream as long as an inner pass
        is in flight, and none of the host synchronisations of the exchange waits for the inner kernel.  The boundary
        traversal waits (on the device) for the populated halo.  With fuse_dt the force passes also integrate
        (ls1hip_forces_kick_drift): no F round trip, no integrator pass."""
        e = self.engine
        e.rebin()
        self.ex.exchange(LEAVING)
        if fuse_dt is None:
            e.forces(1, want_macro=False)
        else:
            e.forces_kick_drift(1, fuse_dt)
        e.halo()
        self.ex.exchange(HALO)
        if fuse_dt is None:
            return e.forces(2, want_macro=want)
        return e.forces_kick_drift(2, fuse_dt, want_macro=want)

    def step(self, dt, want=False):
        e = self.engine
        e.kick_drift(dt)
        macro = self._exchange_and_forces(want)
        kin = e.kick(0.5 * dt, want_sums=want)
        return macro, kin

    def run(self, dt, nsteps, fuse=True, lists=None):
        """nsteps full time steps.  Between two steps the post-force kick and the next pre-force kick + drift are
        either done by the force pass itself (fused / reduced-memory mode, when the engine offers it) or by one pass
        over the molecules (ls1hip_kick_then_kick_drift); both are bitwise the same as the separate calls.  The last
        step is unfused: forces and kinetic sums are needed for the global values."""
        e = self.engine
        fuse = bool(fuse) and getattr(e, "can_fuse_integration", lambda: False)()
        if fuse and lists is not False and getattr(e, "can_verlet", lambda: False)():
            return self.run_lists(dt, nsteps)
        out = None
        advanced = False
        for s in range(nsteps):
            last = s == nsteps - 1
            if advanced:
                pass  # positions already advanced by the previous force pass
            elif s == 0:
                e.kick_drift(dt)
            else:
                e.kick_then_kick_drift(dt)
            advanced = fuse and not last
            macro = self._exchange_and_forces(last, dt if advanced else None)
            if last:
                kin = e.kick(0.5 * dt, want_sums=True)
                out = self.reduce_globals(macro, kin)
        return out

    def _collective_rebuild(self):
        """Has ANY rank's displacement bound exceeded skin / 2?  (A rebuild moves molecules between ranks: all or none.)"""
        torch = self._torch
        need = 1.0 if self.engine.verlet_poll() else 0.0
        if self.dc.world == 1:
            return need > 0
        t = torch.tensor([need], dtype=torch.float64, device="cpu" if (self.stage or self.device.type != "cuda") else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item()) > 0

    def run_lists(self, dt, nsteps):
        """The decomposed loop in list mode (ls1hip_set_verlet): between two rebuilds no migration, no re-binning, no halo
        regeneration and no count exchange — the halo copies receive their current positions through the messages of the
        build-time halo exchange (3 doubles per record), overlapped with the inner-brick pass.  Rebuild steps (all ranks
        together, when any rank's displacement bound exceeds skin / 2) run the full exchange and rebuild the lists."""
        e = self.engine
        out = None
        advanced = False
        for s in range(nsteps):
            last = s == nsteps - 1
            if advanced:
                rebuild = self._collective_rebuild()
            else:
                e.kick_drift(dt)  # only the first step of a run integrates separately
                rebuild = True
            advanced = not last
            fdt = dt if advanced else 0.0
            if rebuild:
                e.rebin()
                self.ex.exchange(LEAVING)
                e.halo()
                self.ex.exchange(HALO)
                e.verlet_build()
                macro = e.forces_list(0, fdt, want_macro=last)
            else:
                e.forces_list(1, fdt)        # inner bricks: owned positions only
                e.halo_refresh()             # second stream: local images + packing for the peers
                self.ex.exchange_refresh()   # ... while the inner pass computes
                macro = e.forces_list(2, fdt, want_macro=last)
            if last:
                kin = e.kick(0.5 * dt, want_sums=True)
                out = self.reduce_globals(macro, kin)
        return out

    def reduce_globals(self, macro, kin):
        """Domain::calculateGlobalValues: one all_reduce of {U_pot, virial, sum mv^2, sum Iw^2, N, rotDOF} (+ the error
        status of the last import of this rank, so that a deferred engine error ends the run on every rank)."""
        torch = self._torch
        err = 1.0 if self.ex._deferred is not None else 0.0
        t = torch.tensor([macro[0], macro[1], kin[0], kin[1], float(kin[2]), float(kin[3]), err], dtype=torch.float64,
                         device="cpu" if self.stage else self.device)
        self.dist.all_reduce(t, group=self.group)
        v = t.cpu().numpy()
        if v[6] != 0:
            mine = f"; this rank: {self.ex._deferred}" if self.ex._deferred is not None else ""
            raise DecompositionError(f"{int(v[6])} rank(s) reported an engine error in the last exchange{mine}")
        return dict(upot=float(v[0]), virial=float(v[1]), summv2=float(v[2]), sumIw2=float(v[3]), n=int(v[4]),
                    rot_dof=int(v[5]))


def build_strong_scaling_box(comps, rc, n_per_dim, world, rank, local_rank, rho, temp, cic=None, kernel=0,
                             stage_through_host=False, loopback=False, options=None, skin=None):
    """bench.py helper: the GLOBAL jittered bcc liquid of 2*n^3 molecules (synth.py) split over the rank grid; every
    rank generates exactly the molecules of its own sub-box, in device memory, chunk by chunk."""
    import torch
    import torch.distributed as dist

    from . import synth
    from .engine import DeviceEngine

    grid = dims_create(world)
    L = synth.box_length(n_per_dim, rho)
    global_len = np.array([L, L, L])
    dc = CartesianDecomposition(world, rank, global_len, grid, loopback=loopback)
    lo, hi = dc.bounding_box()
    eng = DeviceEngine(local_rank)
    eng.set_components(comps, rc)
    if cic:
        eng.set_option("cells_in_cutoff", cic)
    eng.set_option("force_kernel", kernel)
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    if skin:
        eng.set_verlet(skin)
    eng.set_domain(global_len, lo, hi, rank, dc.neighbor_table())
    dev = torch.device("cuda", local_rank)
    # upper bound of the sub-box population (uniform density + the jitter band on every face)
    vol = float(np.prod(hi - lo + 0.2))
    eng.upload_begin(int(vol * rho * 1.02) + 4096)
    for ids_t, r_t, v_t in synth.bcc_chunks_device(torch, dev, n_per_dim, lo, hi, rho=rho, temp=temp):
        torch.cuda.synchronize()
        eng.upload_chunk_device(ids_t.numel(), ids_t.data_ptr(), 0, r_t.data_ptr(), v_t.data_ptr())
    eng.upload_end()
    torch.cuda.empty_cache()
    sim = DistributedSimulation(dc, eng, dist, dev, stage_through_host=stage_through_host)
    n_mine = torch.tensor([eng.count()[0]], dtype=torch.int64, device="cpu" if stage_through_host else dev)
    if world > 1:
        dist.all_reduce(n_mine)
    sim.n_global = int(n_mine.item())
    assert sim.n_global == 2 * n_per_dim ** 3, (sim.n_global, 2 * n_per_dim ** 3)
    sim.initial_forces()
    return sim
