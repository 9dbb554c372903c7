// DomainDecompHip.cpp — see DomainDecompHip.h.
#include "DomainDecompHip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "Domain.h"
#include "Simulation.h"
#include "utils/Logger.h"

#include "DomainDecompRccl.hpp"
#include "LinkedCellsHip.h"
#include "MailboxTransport.hpp"

using Log::global_log;
using namespace ls1hip;

static int env_int(const char* a, const char* b, int dflt) {
	if (const char* e = getenv(a)) return atoi(e);
	if (b)
		if (const char* e = getenv(b)) return atoi(e);
	return dflt;
}

struct DomainDecompHip::Impl {
	std::unique_ptr<MailboxTransport> mailbox;
	std::unique_ptr<RcclTransport> rccl;
	std::unique_ptr<CartDecomp> cart;
	std::unique_ptr<HaloExchangerT<MailboxTransport>> exMailbox;
	std::unique_ptr<HaloExchangerT<RcclTransport>> exRccl;
	ls1hip_ctx* ctx = nullptr;

	void gather(const void* mine, size_t n, std::vector<char>& all) {
		if (mailbox) mailbox->all_gather_bytes(mine, n, all);
		else rccl->all_gather_bytes(mine, n, all);
	}
	void barrier() {
		if (mailbox) mailbox->barrier();
		else if (rccl) rccl->barrier();
	}
	bool loopback = false;
	CartDecomp& decomp(int world, int rank, const double L[3]) {
		if (!cart) cart.reset(new CartDecomp(world, rank, L, loopback));
		return *cart;
	}
};

DomainDecompHip::DomainDecompHip() : DomainDecompBase(), _impl(new Impl) {
	_rank = env_int("LS1HIP_RANK", "RANK", 0);
	_world = env_int("LS1HIP_WORLD_SIZE", "WORLD_SIZE", 1);
	_device = env_int("LS1HIP_DEVICE", "LOCAL_RANK", 0);
	if (_world < 1 || _rank < 0 || _rank >= _world) {
		global_log->error() << "DomainDecompHip: bad rank / world size (" << _rank << " / " << _world << ")" << std::endl;
		Simulation::exit(690);
	}
	// LS1HIP_LOOPBACK=1 (one rank): the periodic images this rank would make locally travel through the transport instead — all 26
	// directions are exported, packed, sent to the own rank, received and imported exactly as between GPUs (rehearsal of the
	// multi-rank seam on one GPU; decomp.py / bench.py --loopback do the same for the handed-over loops)
	if (const char* e = getenv("LS1HIP_LOOPBACK")) _impl->loopback = atoi(e) != 0 && _world == 1;
	_decomposed = _world > 1 || _impl->loopback;
	if (!_decomposed) return;
	std::string transport = "rccl";
	if (const char* e = getenv("LS1HIP_TRANSPORT")) transport = e;
	try {
		if (transport == "mailbox") {
			const char* dir = getenv("LS1HIP_COMM_DIR");
			_impl->mailbox.reset(new MailboxTransport(_world, _rank, dir ? dir : ""));
		} else if (transport == "rccl") {
			_impl->rccl.reset(new RcclTransport(_world, _rank, _device, _world > 1 ? rccl_id_source_from_env() : std::string()));
		} else {
			global_log->error() << "DomainDecompHip: unknown LS1HIP_TRANSPORT '" << transport << "' (rccl | mailbox)" << std::endl;
			Simulation::exit(691);
		}
	} catch (const std::exception& e) {
		global_log->error() << "DomainDecompHip: transport '" << transport << "': " << e.what() << std::endl;
		Simulation::exit(692);
	}
	int g[3];
	CartDecomp::dims_create(_world, g);
	global_log->info() << "DomainDecompHip: rank " << _rank << " of " << _world << ", grid " << g[0] << " x " << g[1] << " x " << g[2]
					   << ", transport " << transport << ", device " << _device << (_impl->loopback ? ", LOOPBACK (own images through the transport)" : "")
					   << std::endl;
}

DomainDecompHip::~DomainDecompHip() = default;

void DomainDecompHip::barrier() const { _impl->barrier(); }

static void global_length(Domain* domain, double L[3]) {
	for (int d = 0; d < 3; ++d) L[d] = domain->getGlobalLength(d);
}

double DomainDecompHip::getBoundingBoxMin(int dimension, Domain* domain) {
	if (_world == 1) return DomainDecompBase::getBoundingBoxMin(dimension, domain);
	double L[3], lo[3], hi[3];
	global_length(domain, L);
	_impl->decomp(_world, _rank, L).bounding_box(lo, hi);
	return lo[dimension];
}
double DomainDecompHip::getBoundingBoxMax(int dimension, Domain* domain) {
	if (_world == 1) return DomainDecompBase::getBoundingBoxMax(dimension, domain);
	double L[3], lo[3], hi[3];
	global_length(domain, L);
	_impl->decomp(_world, _rank, L).bounding_box(lo, hi);
	return hi[dimension];
}

void DomainDecompHip::neighbourTable(const double globalLength[3], int nbr[27]) {
	_impl->decomp(_world, _rank, globalLength).neighbor_table(nbr);
}

void DomainDecompHip::exchange(ls1hip_ctx* ctx, const double globalLength[3], int kind) {
	CartDecomp& dc = _impl->decomp(_world, _rank, globalLength);
	try {
		if (_impl->ctx != ctx) {  // (a new device context: the exchangers hold it)
			_impl->exMailbox.reset();
			_impl->exRccl.reset();
			_impl->ctx = ctx;
		}
		if (_impl->mailbox) {
			if (!_impl->exMailbox) _impl->exMailbox.reset(new HaloExchangerT<MailboxTransport>(dc, ctx, *_impl->mailbox));
			if (kind == 2) _impl->exMailbox->exchange_refresh();
			else _impl->exMailbox->exchange(kind);
		} else {
			if (!_impl->exRccl) _impl->exRccl.reset(new HaloExchangerT<RcclTransport>(dc, ctx, *_impl->rccl));
			if (kind == 2) _impl->exRccl->exchange_refresh();
			else _impl->exRccl->exchange(kind);
		}
	} catch (const std::exception& e) {
		global_log->error() << "DomainDecompHip: exchange of kind " << kind << " failed: " << e.what() << std::endl;
		Simulation::exit(693);
	}
}

// variable-length all-gather of raw records (the mirror snapshot's migration of molecules that await theirs on the device):
// every rank's `n` records of `bytes` each -> all of them, rank by rank
void DomainDecompHip::gatherRecords(const void* mine, size_t n, size_t bytes, std::vector<char>& all, std::vector<size_t>& counts) {
	counts.assign((size_t)_world, n);
	all.clear();
	if (_world == 1) {
		all.assign(static_cast<const char*>(mine), static_cast<const char*>(mine) + n * bytes);
		return;
	}
	try {
		uint64_t c = n;
		std::vector<char> tab;
		_impl->gather(&c, sizeof(c), tab);
		size_t mx = 0;
		for (int r = 0; r < _world; ++r) {
			uint64_t v;
			std::memcpy(&v, tab.data() + (size_t)r * sizeof(v), sizeof(v));
			counts[(size_t)r] = (size_t)v;
			mx = std::max(mx, (size_t)v);
		}
		if (mx == 0) return;
		std::vector<char> pad(mx * bytes, 0), g;
		if (n) std::memcpy(pad.data(), mine, n * bytes);
		_impl->gather(pad.data(), pad.size(), g);
		for (int r = 0; r < _world; ++r) all.insert(all.end(), g.begin() + (size_t)r * mx * bytes, g.begin() + (size_t)r * mx * bytes + counts[(size_t)r] * bytes);
	} catch (const std::exception& e) {
		global_log->error() << "DomainDecompHip: gather of snapshot records: " << e.what() << std::endl;
		Simulation::exit(697);
	}
}

// a rebuild moves molecules between ranks: all ranks rebuild, or none
bool DomainDecompHip::anyRank(bool mine) {
	if (_world == 1) return mine;
	double v = mine ? 1. : 0.;
	try {
		if (_impl->mailbox) _impl->mailbox->all_reduce(&v, 1, ReduceOp::Max);
		else _impl->rccl->all_reduce(&v, 1, ReduceOp::Max);
	} catch (const std::exception& e) {
		global_log->error() << "DomainDecompHip: rebuild decision: " << e.what() << std::endl;
		Simulation::exit(695);
	}
	return v > 0.;
}

// LinkedCells::update has classified the molecules (ls1hip_rebin: leavers packed per direction); here they travel, then the
// halo copies (NeighbourCommunicationScheme.cpp:115-136: LEAVING_ONLY, then HALO_COPIES)
void DomainDecompHip::balanceAndExchange(double lastTraversalTime, bool forceRebalancing, ParticleContainer* moleculeContainer,
										 Domain* domain) {
	if (!_decomposed) {
		DomainDecompBase::balanceAndExchange(lastTraversalTime, forceRebalancing, moleculeContainer, domain);
		return;
	}
	LinkedCellsHip* cont = dynamic_cast<LinkedCellsHip*>(moleculeContainer);
	if (!cont) {
		global_log->error() << "DomainDecompHip: multi-rank runs need the device container (LinkedCellsHip)" << std::endl;
		Simulation::exit(694);
	}
	cont->exchangeAcrossRanks(*this, domain);
}

void DomainDecompHip::assertDisjunctivity(ParticleContainer* moleculeContainer) const {
	if (LinkedCellsHip* cont = dynamic_cast<LinkedCellsHip*>(moleculeContainer)) cont->snapshotForReaders();
}

// ---- the reference's typed collectives ------------------------------------------------------------------------------------------
void DomainDecompHip::collCommInit(int numValues, int /*key*/) {
	_values.clear();
	_values.reserve((size_t)numValues);
	_getter = 0;
}
void DomainDecompHip::collCommFinalize() {
	_values.clear();
	_getter = 0;
}
void DomainDecompHip::collCommAppendInt(int x) {
	Value v;
	std::memset(&v, 0, sizeof(v));
	v.type = 0;
	v.v.i = x;
	_values.push_back(v);
}
void DomainDecompHip::collCommAppendUnsLong(unsigned long x) {
	Value v;
	std::memset(&v, 0, sizeof(v));
	v.type = 1;
	v.v.ul = x;
	_values.push_back(v);
}
void DomainDecompHip::collCommAppendFloat(float x) {
	Value v;
	std::memset(&v, 0, sizeof(v));
	v.type = 2;
	v.v.f = x;
	_values.push_back(v);
}
void DomainDecompHip::collCommAppendDouble(double x) {
	Value v;
	std::memset(&v, 0, sizeof(v));
	v.type = 3;
	v.v.d = x;
	_values.push_back(v);
}
void DomainDecompHip::collCommAppendLongDouble(long double x) {
	Value v;
	std::memset(&v, 0, sizeof(v));
	v.type = 4;
	v.v.ld = x;
	_values.push_back(v);
}
int DomainDecompHip::collCommGetInt() { return _values.at(_getter++).v.i; }
unsigned long DomainDecompHip::collCommGetUnsLong() { return _values.at(_getter++).v.ul; }
float DomainDecompHip::collCommGetFloat() { return _values.at(_getter++).v.f; }
double DomainDecompHip::collCommGetDouble() { return _values.at(_getter++).v.d; }
long double DomainDecompHip::collCommGetLongDouble() { return _values.at(_getter++).v.ld; }

void DomainDecompHip::allGatherValues(std::vector<std::vector<Value>>& perRank) {
	const size_t n = _values.size();
	std::vector<char> all;
	_impl->gather(_values.data(), n * sizeof(Value), all);
	perRank.assign((size_t)_world, std::vector<Value>(n));
	for (int r = 0; r < _world; ++r) std::memcpy(perRank[r].data(), all.data() + (size_t)r * n * sizeof(Value), n * sizeof(Value));
}

// every rank combines the gathered values in rank order: identical results everywhere, native types (no detour through double)
void DomainDecompHip::combine(int mode, int root) {
	_getter = 0;
	if (_world == 1 || _values.empty()) return;
	std::vector<std::vector<Value>> t;
	allGatherValues(t);
	for (size_t k = 0; k < _values.size(); ++k) {
		Value acc = t[mode == 4 ? (size_t)root : 0][k];
		if (mode != 4) {
			const int last = mode == 3 ? _rank : _world - 1;  // inclusive scan: ranks 0 .. me
			for (int r = 1; r <= last; ++r) {
				const Value& x = t[(size_t)r][k];
				switch (acc.type) {
#define LS1_COMBINE(field)                                                                  \
	if (mode == 0 || mode == 3) acc.v.field += x.v.field;                                   \
	else if (mode == 1) acc.v.field = x.v.field < acc.v.field ? x.v.field : acc.v.field;    \
	else acc.v.field = x.v.field > acc.v.field ? x.v.field : acc.v.field;
					case 0: LS1_COMBINE(i) break;
					case 1: LS1_COMBINE(ul) break;
					case 2: LS1_COMBINE(f) break;
					case 3: LS1_COMBINE(d) break;
					default: LS1_COMBINE(ld) break;
#undef LS1_COMBINE
				}
			}
		}
		_values[k] = acc;
	}
}
void DomainDecompHip::collCommAllreduceSum() { combine(0, 0); }
void DomainDecompHip::collCommAllreduceSumAllowPrevious() { combine(0, 0); }
void DomainDecompHip::collCommAllreduceCustom(ReduceType type) { combine(type == SUM ? 0 : type == MIN ? 1 : 2, 0); }
void DomainDecompHip::collCommScanSum() { combine(3, 0); }
void DomainDecompHip::collCommBroadcast(int root) { combine(4, root); }
