// DomainDecompRccl.hpp — the decomposed time loop in C++17 directly on RCCL: regular 3-D rank grid, one process per GPU,
// direct full-shell exchange of packed records over point-to-point ncclSend / ncclRecv (xGMI), inner / boundary overlap.
//
// The compiled-language counterpart of ls1-mardyn_amd/decomp.py (same protocol, same message layout, same loops), for hosts
// that are not Python.  What it stands in for in the reference:
//   parallel/DomainDecomposition.cpp:19-41,84-123     MPI_Dims_create / MPI_Cart_create, bounding boxes      -> CartDecomp
//   parallel/NeighbourCommunicationScheme.cpp:115-136 direct scheme: LEAVING_ONLY, then HALO_COPIES, per neighbour -> HaloExchangerRccl
//   parallel/CommunicationPartner.cpp:139-227,266-389 packing / unpacking of the molecule records            -> ls1hip_export_* / ls1hip_import
//   parallel/NonBlockingMPIMultiStepHandler.cpp:30-97 communication overlapped with the inner-cell traversal -> DecomposedLoop
//   Domain.cpp:151-181                                calculateGlobalValues (all-reduce of the sums)          -> reduce_globals
// An MPI host (class DomainDecompHip : DomainDecompBase inside the reference) would keep HaloExchangerRccl's structure and
// replace the four transport calls (all_gather_counts, send, recv, all_reduce) by MPI ones on the same device buffers.
//
// xGMI notes: on a 2x2x2 periodic grid every GPU has 7 distinct peers = its 7 links; the 26 directions are merged per PEER
// into one message (one ncclSend + one ncclRecv per peer and phase inside one ncclGroup), buffers are persistent (grow-only),
// the transfers run on their own high-priority stream while the engine's main stream computes the inner bricks.
// Error handling is collective: the count exchange carries a status word, every rank throws together.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <ctime>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "IdHandOver.hpp"
#include "ls1hip.h"

namespace ls1hip {

struct DecompositionError : std::runtime_error {
	using std::runtime_error::runtime_error;
};

#define LS1_HIPCHECK(call)                                                                                              \
	do {                                                                                                               \
		hipError_t e_ = (call);                                                                                       \
		if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_));           \
	} while (0)
#define LS1_NCCLCHECK(call)                                                                                             \
	do {                                                                                                               \
		ncclResult_t r_ = (call);                                                                                     \
		if (r_ != ncclSuccess) throw std::runtime_error(std::string(#call) + ": " + ncclGetErrorString(r_));         \
	} while (0)

// ---- geometry of the rank grid ---------------------------------------------------------------------------------------------
struct CartDecomp {
	int world = 1, rank = 0;
	int grid[3] = {1, 1, 1}, coords[3] = {0, 0, 0};
	double L[3] = {0., 0., 0.};
	bool loopback = false;  // rehearsal: periodic images of the own sub-box travel through the transport (alias id world + rank)

	// balanced factorisation, largest factor first (what MPI_Dims_create returns for 1, 2, 4, 8, ...)
	static void dims_create(int world, int g[3]) {
		g[0] = g[1] = g[2] = 1;
		std::vector<int> f;
		for (int n = world, p = 2; n > 1; ++p)
			while (n % p == 0) {
				f.push_back(p);
				n /= p;
			}
		std::sort(f.rbegin(), f.rend());
		for (int p : f) *std::min_element(g, g + 3) *= p;
		std::sort(g, g + 3, [](int a, int b) { return a > b; });
	}
	CartDecomp(int world_, int rank_, const double len[3], bool loopback_ = false) : world(world_), rank(rank_), loopback(loopback_) {
		dims_create(world, grid);
		for (int d = 0; d < 3; ++d) L[d] = len[d];
		coords_of(rank, coords);
	}
	void coords_of(int r, int c[3]) const {
		c[0] = r % grid[0];
		c[1] = (r / grid[0]) % grid[1];
		c[2] = r / (grid[0] * grid[1]);
	}
	int rank_of(const int c[3]) const { return (c[2] * grid[1] + c[1]) * grid[0] + c[0]; }
	// [c L / g, (c + 1) L / g), the last rank's upper bound exactly L (DomainDecomposition.cpp:114-123)
	void bounding_box(double lo[3], double hi[3], int r = -1) const {
		int c[3];
		coords_of(r < 0 ? rank : r, c);
		for (int d = 0; d < 3; ++d) {
			lo[d] = c[d] * L[d] / grid[d];
			hi[d] = c[d] + 1 < grid[d] ? (c[d] + 1) * L[d] / grid[d] : L[d];
		}
	}
	// neighbor_rank[27], index (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1); all sides periodic
	void neighbor_table(int tab[27], int r = -1) const {
		const int owner = r < 0 ? rank : r;
		int c[3];
		coords_of(owner, c);
		for (int sz = -1; sz <= 1; ++sz)
			for (int sy = -1; sy <= 1; ++sy)
				for (int sx = -1; sx <= 1; ++sx) {
					const int s[3] = {sx, sy, sz};
					int n[3];
					for (int d = 0; d < 3; ++d) n[d] = ((c[d] + s[d]) % grid[d] + grid[d]) % grid[d];
					tab[(sz + 1) * 9 + (sy + 1) * 3 + (sx + 1)] = rank_of(n);
				}
		if (loopback) {
			for (int k = 0; k < 27; ++k)
				if (tab[k] == owner) tab[k] = world + owner;
			tab[13] = owner;
		}
	}
	int real_rank(int peer) const { return peer >= world ? peer - world : peer; }
	std::vector<int> peers() const {
		int tab[27];
		neighbor_table(tab);
		std::set<int> s;
		for (int k = 0; k < 27; ++k)
			if (tab[k] >= 0 && tab[k] != rank) s.insert(tab[k]);
		return {s.begin(), s.end()};
	}
};

// ---- transport interface (duck-typed; HaloExchangerT / DecomposedLoopT are templates over it) ----------------------------------
//   int world(), rank();  std::vector<int64_t> all_gather(const int64_t* mine, int n);  void all_reduce(double* v, int n, ReduceOp);
//   void group_start(); void send(const double* dev, size_t count, int peer); void recv(double* dev, size_t count, int peer);
//   void group_end();   // returns when every transfer of the group has completed
// Implementations: RcclTransport (below: xGMI, one GPU per rank) and MailboxTransport (MailboxTransport.hpp: host-staged, any
// number of ranks per GPU — tests, and hosts without a fabric).
enum class ReduceOp { Sum, Max };

// ---- transport: one RCCL communicator, its own stream --------------------------------------------------------------------
class RcclTransport {
public:
	// id_source: where rank 0 hands the ncclUniqueId to the others (IdHandOver.hpp: "tcp:<host>:<port>" or a file); world 1 needs none
	RcclTransport(int world, int rank, int device, const std::string& id_source) : _world(world), _rank(rank) {
		LS1_HIPCHECK(hipSetDevice(device));
		int lo = 0, hi = 0;
		LS1_HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
		LS1_HIPCHECK(hipStreamCreateWithPriority(&_stream, hipStreamNonBlocking, hi));
		ncclUniqueId id;
		if (rank == 0) LS1_NCCLCHECK(ncclGetUniqueId(&id));
		std::string id_file;
		if (world > 1) {
			if (id_source.rfind("tcp:", 0) != 0) id_file = id_source;
			hand_over_bytes(world, rank, id_source, &id, sizeof(id));
		}
		LS1_NCCLCHECK(ncclCommInitRank(&_comm, world, id, rank));
		if (rank == 0 && !id_file.empty()) (void)remove(id_file.c_str());  // every rank has read it: the init above is collective
		LS1_HIPCHECK(hipMalloc(&_d_small, 64 * sizeof(double) * (size_t)std::max(world, 1)));
		LS1_HIPCHECK(hipHostMalloc(&_h_small, 64 * sizeof(double) * (size_t)std::max(world, 1)));
	}
	~RcclTransport() {
		if (_comm) ncclCommDestroy(_comm);
		if (_d_small) (void)hipFree(_d_small);
		if (_d_bytes) (void)hipFree(_d_bytes);
		if (_h_small) (void)hipHostFree(_h_small);
		if (_stream) (void)hipStreamDestroy(_stream);
	}
	RcclTransport(const RcclTransport&) = delete;
	RcclTransport& operator=(const RcclTransport&) = delete;

	int world() const { return _world; }
	int rank() const { return _rank; }
	hipStream_t stream() const { return _stream; }
	ncclComm_t comm() const { return _comm; }
	void sync() { LS1_HIPCHECK(hipStreamSynchronize(_stream)); }

	// [world][n] table of every rank's n int64 values (n <= 32)
	std::vector<int64_t> all_gather(const int64_t* mine, int n) {
		int64_t* h = reinterpret_cast<int64_t*>(_h_small);
		int64_t* d = reinterpret_cast<int64_t*>(_d_small);
		std::memcpy(h, mine, n * sizeof(int64_t));
		LS1_HIPCHECK(hipMemcpyAsync(d, h, n * sizeof(int64_t), hipMemcpyHostToDevice, _stream));
		LS1_NCCLCHECK(ncclAllGather(d, d + 32, (size_t)n, ncclInt64, _comm, _stream));
		LS1_HIPCHECK(hipMemcpyAsync(h, d + 32, (size_t)_world * n * sizeof(int64_t), hipMemcpyDeviceToHost, _stream));
		sync();
		return std::vector<int64_t>(h, h + (size_t)_world * n);
	}
	void all_reduce(double* v, int n, ReduceOp op) { all_reduce(v, n, op == ReduceOp::Sum ? ncclSum : ncclMax); }
	// [world][n] table of every rank's n bytes (the reference's typed collectives: DomainDecompHip::collComm*)
	void all_gather_bytes(const void* mine, size_t n, std::vector<char>& all) {
		all.assign((size_t)_world * n, 0);
		if (_world == 1) {
			std::memcpy(all.data(), mine, n);
			return;
		}
		const size_t need = n * ((size_t)_world + 1);
		if (need > _bytes_cap) {
			if (_d_bytes) (void)hipFree(_d_bytes);
			_bytes_cap = need + 4096;
			LS1_HIPCHECK(hipMalloc(&_d_bytes, _bytes_cap));
		}
		char* d = static_cast<char*>(_d_bytes);
		LS1_HIPCHECK(hipMemcpyAsync(d, mine, n, hipMemcpyHostToDevice, _stream));
		LS1_NCCLCHECK(ncclAllGather(d, d + n, n, ncclChar, _comm, _stream));
		LS1_HIPCHECK(hipMemcpyAsync(all.data(), d + n, (size_t)_world * n, hipMemcpyDeviceToHost, _stream));
		sync();
	}
	void barrier() {
		double v = 0.;
		all_reduce(&v, 1, ncclSum);
	}
	void group_start() { LS1_NCCLCHECK(ncclGroupStart()); }
	void send(const double* dev, size_t count, int peer) { LS1_NCCLCHECK(ncclSend(dev, count, ncclDouble, peer, _comm, _stream)); }
	void recv(double* dev, size_t count, int peer) { LS1_NCCLCHECK(ncclRecv(dev, count, ncclDouble, peer, _comm, _stream)); }
	void group_end() {
		LS1_NCCLCHECK(ncclGroupEnd());
		sync();
	}
	// in-place sum / max of n <= 16 doubles
	void all_reduce(double* v, int n, ncclRedOp_t op) {
		if (_world == 1) return;
		double* h = reinterpret_cast<double*>(_h_small);
		double* d = reinterpret_cast<double*>(_d_small);
		std::memcpy(h, v, n * sizeof(double));
		LS1_HIPCHECK(hipMemcpyAsync(d, h, n * sizeof(double), hipMemcpyHostToDevice, _stream));
		LS1_NCCLCHECK(ncclAllReduce(d, d, (size_t)n, ncclDouble, op, _comm, _stream));
		LS1_HIPCHECK(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, _stream));
		sync();
		std::memcpy(v, h, n * sizeof(double));
	}

private:
	int _world, _rank;
	hipStream_t _stream = nullptr;
	ncclComm_t _comm = nullptr;
	void* _d_small = nullptr;
	void* _h_small = nullptr;
	void* _d_bytes = nullptr;
	size_t _bytes_cap = 0;
};

// ---- per-peer merged exchange of packed records --------------------------------------------------------------------------
template <class TR>
class HaloExchangerT {
public:
	enum Kind { LEAVING = 0, HALO = 1, REFRESH = 2 };
	static int record_doubles(int kind) { return kind == LEAVING ? LS1HIP_LEAVING_DOUBLES : kind == HALO ? LS1HIP_HALO_DOUBLES : LS1HIP_REFRESH_DOUBLES; }

	HaloExchangerT(const CartDecomp& dc, ls1hip_ctx* ctx, TR& tr) : _dc(dc), _ctx(ctx), _tr(tr) {
		_dc.neighbor_table(_nbr);
		_peers = _dc.peers();
		for (int p : _peers) {
			int t[27];
			_dc.neighbor_table(t, _dc.real_rank(p));
			// a message from peer p carries the directions of p's table that point at me (at my alias for a loopback)
			const int me = _dc.real_rank(p) == _dc.rank ? p : _dc.rank;
			for (int d = 0; d < 27; ++d) {
				if (d == 13) continue;
				if (t[d] == me) _incoming[p].push_back(d);
				if (_nbr[d] == p) _outgoing[p].push_back(d);
			}
		}
	}
	~HaloExchangerT() {
		if (_sbuf) (void)hipFree(_sbuf);
		if (_rbuf) (void)hipFree(_rbuf);
	}
	bool has_deferred_error() const { return !_deferred.empty(); }
	const std::string& deferred_error() const { return _deferred; }

	// counts -> all_gather (with status word) -> payload, one message per peer
	void exchange(int kind) {
		uint64_t c64[27] = {0};
		if (ls1hip_export_counts(_ctx, kind, c64) != LS1HIP_OK) {
			if (_deferred.empty()) _deferred = ls1hip_last_error(_ctx);
			std::fill(c64, c64 + 27, 0);
		}
		int64_t mine[28];
		for (int d = 0; d < 27; ++d) mine[d] = (int64_t)c64[d];
		mine[27] = _deferred.empty() ? 0 : 1;
		std::vector<int64_t> table(mine, mine + 28);
		if (!_peers.empty()) {
			table = _tr.all_gather(mine, 28);
			std::string bad;
			for (int r = 0; r < _dc.world; ++r)
				if (table[(size_t)r * 28 + 27] != 0) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
			if (!bad.empty())
				throw DecompositionError("rank(s) " + bad + " reported an engine error during the exchange of kind " + std::to_string(kind) +
										 (_deferred.empty() ? "" : "; this rank: " + _deferred));
		} else if (!_deferred.empty()) {
			throw DecompositionError(_deferred);
		}
		if (kind == HALO) _halo_table = table;  // what every position refresh until the next build repeats
		transfer(kind, table);
	}
	// list-reuse step: same messages as the last build's halo exchange, 3 doubles per record, no count exchange
	void exchange_refresh() { transfer(REFRESH, _halo_table); }

private:
	void reserve(void** buf, size_t* cap, size_t bytes) {
		if (bytes <= *cap) return;
		if (*buf) LS1_HIPCHECK(hipFree(*buf));
		*cap = bytes + bytes / 4 + 4096;
		LS1_HIPCHECK(hipMalloc(buf, *cap));
	}
	void transfer(int kind, const std::vector<int64_t>& table) {
		const int w = record_doubles(kind);
		const size_t mine = (size_t)_dc.rank * 28;  // (the table has one row per rank)
		std::map<int, size_t> n_out, n_in;
		size_t tot_out = 0, tot_in = 0;
		std::vector<int> order;
		for (int p : _peers) {
			size_t o = 0, i = 0;
			for (int d : _outgoing[p]) {
				o += (size_t)table[mine + d];
				if (table[mine + d]) order.push_back(d);
			}
			for (int d : _incoming[p]) i += (size_t)table[(size_t)_dc.real_rank(p) * 28 + d];
			n_out[p] = o;
			n_in[p] = i;
			tot_out += o;
			tot_in += i;
		}
		if (tot_out) {
			reserve(&_sbuf, &_scap, tot_out * w * sizeof(double));
			// one pack call for all messages (directions grouped by peer); returns when the records are in the buffer
			if (ls1hip_export_pack_dirs(_ctx, kind, order.data(), (int)order.size(), _sbuf, tot_out) != LS1HIP_OK)
				throw std::runtime_error(std::string("ls1hip_export_pack_dirs: ") + ls1hip_last_error(_ctx));
		}
		if (tot_in) reserve(&_rbuf, &_rcap, tot_in * w * sizeof(double));
		if (tot_out || tot_in) {
			_tr.group_start();
			size_t so = 0, ro = 0;
			for (int p : _peers) {
				if (n_out[p]) _tr.send(static_cast<double*>(_sbuf) + so * w, n_out[p] * w, _dc.real_rank(p));
				if (n_in[p]) _tr.recv(static_cast<double*>(_rbuf) + ro * w, n_in[p] * w, _dc.real_rank(p));
				so += n_out[p];
				ro += n_in[p];
			}
			_tr.group_end();
		}
		if (tot_in && ls1hip_import(_ctx, kind, _rbuf, tot_in) != LS1HIP_OK && _deferred.empty()) _deferred = ls1hip_last_error(_ctx);
		if (ls1hip_import_done(_ctx, kind) != LS1HIP_OK) {
			if (_peers.empty()) throw DecompositionError(ls1hip_last_error(_ctx));
			if (_deferred.empty()) _deferred = ls1hip_last_error(_ctx);  // reported by all ranks together at the next count exchange
		}
	}

	const CartDecomp& _dc;
	ls1hip_ctx* _ctx;
	TR& _tr;
	int _nbr[27];
	std::vector<int> _peers;
	std::map<int, std::vector<int>> _incoming, _outgoing;
	std::vector<int64_t> _halo_table;
	std::string _deferred;
	void* _sbuf = nullptr;
	void* _rbuf = nullptr;
	size_t _scap = 0, _rcap = 0;
};

using HaloExchangerRccl = HaloExchangerT<RcclTransport>;

// ---- one rank of the decomposed time loop ----------------------------------------------------------------------------------
struct GlobalValues {
	double upot = 0., virial = 0., summv2 = 0., sumIw2 = 0.;
	uint64_t n = 0, rot_dof = 0;
};

template <class TR>
class DecomposedLoopT {
	using Ex = HaloExchangerT<TR>;

public:
	DecomposedLoopT(const CartDecomp& dc, ls1hip_ctx* ctx, TR& tr) : _ctx(ctx), _tr(tr), _ex(dc, ctx, tr) {}

	GlobalValues initial_forces() {
		chk(ls1hip_rebin(_ctx), "ls1hip_rebin");
		_ex.exchange(Ex::LEAVING);
		chk(ls1hip_forces(_ctx, 1, nullptr, nullptr), "ls1hip_forces");
		chk(ls1hip_halo(_ctx), "ls1hip_halo");
		_ex.exchange(Ex::HALO);
		double u = 0., w = 0.;
		chk(ls1hip_forces(_ctx, 2, &u, &w), "ls1hip_forces");
		size_t n = 0, h = 0;
		ls1hip_count(_ctx, &n, &h);
		return reduce_globals(u, w, 0., 0., n, 0);
	}

	// nsteps full time steps; the force passes integrate (reduced-memory mode) except in the last step, whose forces and
	// kinetic sums feed the global values.  List mode (ls1hip_set_verlet) when the engine offers it.
	GlobalValues run(double dt, unsigned long nsteps) {
		long fuse = 0, lists = 0;
		ls1hip_get_option(_ctx, "can_fuse_integration", &fuse);
		ls1hip_get_option(_ctx, "verlet_lists", &lists);
		if (fuse && lists) return run_lists(dt, nsteps);
		GlobalValues out;
		bool advanced = false;
		for (unsigned long s = 0; s < nsteps; ++s) {
			const bool last = s + 1 == nsteps;
			if (!advanced) chk(s == 0 ? ls1hip_kick_drift(_ctx, dt) : ls1hip_kick_then_kick_drift(_ctx, dt), "kick_drift");
			advanced = fuse && !last;
			// re-bin + migration, the inner traversal FIRST (owned molecules only), the whole halo phase while it computes
			chk(ls1hip_rebin(_ctx), "ls1hip_rebin");
			_ex.exchange(Ex::LEAVING);
			double u = 0., w = 0.;
			if (advanced) chk(ls1hip_forces_kick_drift(_ctx, 1, dt, nullptr, nullptr), "ls1hip_forces_kick_drift");
			else chk(ls1hip_forces(_ctx, 1, nullptr, nullptr), "ls1hip_forces");
			chk(ls1hip_halo(_ctx), "ls1hip_halo");
			_ex.exchange(Ex::HALO);
			if (advanced) chk(ls1hip_forces_kick_drift(_ctx, 2, dt, nullptr, nullptr), "ls1hip_forces_kick_drift");
			else chk(ls1hip_forces(_ctx, 2, last ? &u : nullptr, last ? &w : nullptr), "ls1hip_forces");
			if (last) out = finish(dt, u, w);
		}
		return out;
	}

	// between two rebuilds: no migration, no re-binning, no halo regeneration, no count exchange — the halo copies receive
	// their current positions through the messages of the build-time halo exchange, overlapped with the inner-brick pass
	GlobalValues run_lists(double dt, unsigned long nsteps) {
		GlobalValues out;
		bool advanced = false;
		for (unsigned long s = 0; s < nsteps; ++s) {
			const bool last = s + 1 == nsteps;
			bool rebuild = true;
			if (advanced) rebuild = collective_rebuild();
			else chk(ls1hip_kick_drift(_ctx, dt), "ls1hip_kick_drift");  // only the first step of a run integrates separately
			advanced = !last;
			const double fdt = advanced ? dt : 0.;
			double u = 0., w = 0.;
			if (rebuild) {
				chk(ls1hip_rebin(_ctx), "ls1hip_rebin");
				_ex.exchange(Ex::LEAVING);
				chk(ls1hip_halo(_ctx), "ls1hip_halo");
				_ex.exchange(Ex::HALO);
				chk(ls1hip_verlet_build(_ctx), "ls1hip_verlet_build");
				chk(ls1hip_forces_list(_ctx, 0, fdt, last ? &u : nullptr, last ? &w : nullptr), "ls1hip_forces_list");
			} else {
				chk(ls1hip_forces_list(_ctx, 1, fdt, nullptr, nullptr), "ls1hip_forces_list");  // inner bricks: owned positions only
				chk(ls1hip_halo_refresh(_ctx), "ls1hip_halo_refresh");                          // second stream: local images + packing
				_ex.exchange_refresh();                                                         // ... while the inner pass computes
				chk(ls1hip_forces_list(_ctx, 2, fdt, last ? &u : nullptr, last ? &w : nullptr), "ls1hip_forces_list");
			}
			if (last) out = finish(dt, u, w);
		}
		return out;
	}

	// Domain::calculateGlobalValues: one all-reduce of {U_pot, virial, sum m v^2, sum I w^2, N, rotDOF} + the error status
	GlobalValues reduce_globals(double upot, double virial, double summv2, double sumIw2, uint64_t n, uint64_t rot_dof) {
		double v[7] = {upot, virial, summv2, sumIw2, (double)n, (double)rot_dof, _ex.has_deferred_error() ? 1. : 0.};
		_tr.all_reduce(v, 7, ReduceOp::Sum);
		if (v[6] != 0.)
			throw DecompositionError(std::to_string((int)v[6]) + " rank(s) reported an engine error in the last exchange" +
									 (_ex.has_deferred_error() ? "; this rank: " + _ex.deferred_error() : ""));
		GlobalValues g;
		g.upot = v[0]; g.virial = v[1]; g.summv2 = v[2]; g.sumIw2 = v[3];
		g.n = (uint64_t)v[4]; g.rot_dof = (uint64_t)v[5];
		return g;
	}

private:
	void chk(int rc, const char* what) {
		if (rc != LS1HIP_OK) throw std::runtime_error(std::string(what) + ": " + ls1hip_last_error(_ctx));
	}
	GlobalValues finish(double dt, double u, double w) {
		double mv2 = 0., iw2 = 0.;
		uint64_t n = 0, rd = 0;
		chk(ls1hip_kick(_ctx, 0.5 * dt, &mv2, &iw2, &n, &rd), "ls1hip_kick");
		return reduce_globals(u, w, mv2, iw2, n, rd);
	}
	// has ANY rank's displacement bound exceeded skin / 2?  (a rebuild moves molecules between ranks: all or none)
	bool collective_rebuild() {
		int need = 0;
		chk(ls1hip_verlet_poll(_ctx, &need), "ls1hip_verlet_poll");
		double v = need ? 1. : 0.;
		_tr.all_reduce(&v, 1, ReduceOp::Max);
		return v > 0.;
	}

	ls1hip_ctx* _ctx;
	TR& _tr;
	Ex _ex;
};
using DecomposedLoop = DecomposedLoopT<RcclTransport>;

}  // namespace ls1hip
