// DomainDecompHip.h — seam B, multi-rank: the domain decomposition the UNMODIFIED reference driver constructs in a non-MPI
// build (Simulation.cpp:1356 `_domainDecomposition = new DomainDecompBase();`, mapped by seam_b_register.h), one process per
// GPU, launched by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun, mpirun-less shell loops, srun).
//
//   class DomainDecompHip : public DomainDecompBase        (parallel/DomainDecompBase.h:51-341)
//
// What it overrides and what it stands in for:
//   getRank / getNumProcs / barrier                      parallel/DomainDecompMPIBase.cpp (MPI_Comm_rank / size / Barrier)
//   getBoundingBoxMin / Max                              parallel/DomainDecomposition.cpp:114-123 (regular grid, MPI_Dims_create order)
//   balanceAndExchange                                   parallel/DomainDecompMPIBase.cpp:181-214 + NeighbourCommunicationScheme.cpp:115-136
//                                                        (direct scheme: leaving molecules, then halo copies, one merged message per
//                                                        neighbour) — through the export / import entry points of include/ls1hip.h on the
//                                                        device container, transport = RCCL over xGMI (or the host-staged mailbox)
//   collCommInit ... collCommBroadcast                   parallel/CollectiveCommunication.h (typed value list, all-reduce / scan / broadcast);
//                                                        every global reduction of a non-MPI build goes through these virtuals
//                                                        (Domain.cpp:151-181 calculateGlobalValues, generators, thermostats)
// With one rank it behaves exactly as DomainDecompBase (the sequential periodic boundary is then handled inside the device
// container).  Environment: LS1HIP_TRANSPORT = rccl (default when every rank has its own GPU) | mailbox (host-staged files in
// LS1HIP_COMM_DIR: several ranks per GPU, tests).  The ncclUniqueId travels from rank 0 over a TCP rendezvous on the launcher's
// MASTER_ADDR : MASTER_PORT + 17 (LS1HIP_RCCL_PORT overrides; LS1HIP_RCCL_ID_FILE = a file instead) — per job, nothing left on disk.
// LS1HIP_LOOPBACK=1 with one rank routes the own periodic images through the transport (rehearsal of the exchange on one GPU).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "parallel/DomainDecompBase.h"

#include "ls1hip.h"

class LinkedCellsHip;

class DomainDecompHip : public DomainDecompBase {
public:
	DomainDecompHip();
	~DomainDecompHip() override;

	int getRank() const override { return _rank; }
	int getNumProcs() const override { return _world; }
	void barrier() const override;
	double getBoundingBoxMin(int dimension, Domain* domain) override;
	double getBoundingBoxMax(int dimension, Domain* domain) override;
	void balanceAndExchange(double lastTraversalTime, bool forceRebalancing, ParticleContainer* moleculeContainer, Domain* domain) override;

	void collCommInit(int numValues, int key = 0) override;
	void collCommFinalize() override;
	void collCommAppendInt(int intValue) override;
	void collCommAppendUnsLong(unsigned long unsLongValue) override;
	void collCommAppendFloat(float floatValue) override;
	void collCommAppendDouble(double doubleValue) override;
	void collCommAppendLongDouble(long double longDoubleValue) override;
	int collCommGetInt() override;
	unsigned long collCommGetUnsLong() override;
	float collCommGetFloat() override;
	double collCommGetDouble() override;
	long double collCommGetLongDouble() override;
	void collCommAllreduceSum() override;
	void collCommAllreduceSumAllowPrevious() override;
	void collCommAllreduceCustom(ReduceType type) override;
	void collCommScanSum() override;
	void collCommBroadcast(int root = 0) override;
	std::string getName() override { return "DomainDecompHip"; }
	// Called by EVERY rank at the head of Domain::writeCheckpoint (Domain.cpp:600), before the ranks write their molecules one
	// after the other (DomainDecompBase::writeMoleculesToFile, DomainDecompBase.cpp:505-541: a rank iterates while the others wait in
	// a barrier): the collective point at which the device container refills its host mirror for the readers to come (in
	// multi-rank list mode that snapshot moves records between ranks, LinkedCellsHip::syncMirrorFromDevice).
	void assertDisjunctivity(ParticleContainer* moleculeContainer) const override;

	// ---- used by LinkedCellsHip ---------------------------------------------------------------------------------------------
	int localDevice() const { return _device; }
	void neighbourTable(const double globalLength[3], int nbr[27]);                      // neighbor_rank[27] of ls1hip_set_domain
	void exchange(ls1hip_ctx* ctx, const double globalLength[3], int kind);              // 0 leaving molecules, 1 halo copies
	// kind 2 = the position refresh of a list-reuse step (the records of the last halo exchange, 3 doubles each, no count exchange)
	bool anyRank(bool mine);                                                             // logical OR over the ranks (rebuild decision)
	bool decomposed() const { return _decomposed; }  // the container's molecules cross rank boundaries through exchange(): > 1 rank, or the 1-rank loopback rehearsal
	// every rank's n records of `bytes` bytes each -> all records rank by rank (+ per-rank counts); collective
	void gatherRecords(const void* mine, size_t n, size_t bytes, std::vector<char>& all, std::vector<size_t>& counts);

private:
	struct Impl;
	struct Value {
		int type;  // 0 int, 1 unsigned long, 2 float, 3 double, 4 long double
		union {
			int i;
			unsigned long ul;
			float f;
			double d;
			long double ld;
		} v;
	};
	void allGatherValues(std::vector<std::vector<Value>>& perRank);
	void combine(int mode, int root);  // 0 sum, 1 min, 2 max, 3 inclusive scan, 4 broadcast
	int _rank = 0, _world = 1, _device = 0;
	bool _decomposed = false;
	std::vector<Value> _values;
	size_t _getter = 0;
	std::unique_ptr<Impl> _impl;
};
