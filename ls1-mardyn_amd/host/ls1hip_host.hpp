// ls1hip_host.hpp — C++17 host side of seam B above the C ABI (include/ls1hip.h): the reference's plug-in interfaces for
// this path with the reference's class and method names, argument meaning and call order, each a thin forwarder to
// libls1hip.  A maintainer derives these from the reference's abstract bases (ParticleContainer, CellProcessor,
// Integrator, DomainDecompBase) and registers them in Simulation::readXML; here they are free-standing
// so that they compile without the reference's headers and can be driven by the parity tests
// (tests/hostcpp/host_sim.cpp -> tests/test_gpu_hostcpp.py).  The Python classes in ls1-mardyn_amd/mirror.py are the
// same layer for the Python tests.
//
//   reference (file:line under /root/reference/src)                      here
//   Domain (Domain.h: setLocalUpot/Virial/Summv2/...)                    ls1hip::Domain
//   CellProcessor (particleContainer/adapter/CellProcessor.h:29-94)      ls1hip::CellProcessor
//   VectorizedCellProcessor (adapter/VectorizedCellProcessor.cpp:21-157) ls1hip::VectorizedCellProcessor
//   ParticleContainer / LinkedCells (ParticleContainer.h:69-278,
//       LinkedCells.cpp:243-356,564-628)                                 ls1hip::LinkedCells
//   DomainDecompBase::balanceAndExchange (DomainDecompBase.cpp:51-86)    ls1hip::DomainDecompBase
//   Integrator / Leapfrog (integrators/Leapfrog.cpp:35-150)              ls1hip::Leapfrog
//   Simulation::prepare_start / simulate (Simulation.cpp:813-892,979-1167) ls1hip::simulate
//
// Error convention: the reference logs and calls Simulation::exit(code); these classes throw ls1hip::Error carrying the
// ABI's code and ls1hip_last_error text, which the adapter's caller maps to global_log->error() + Simulation::exit.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ls1hip.h"

namespace ls1hip {

struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// flat component description exactly as ls1hip_set_components takes it (see the header for the strides)
struct ComponentTables {
	int ncomp = 0;
	std::vector<int> nlj, nc, nd, nq;
	std::vector<double> lj, ch, dp, qp, mass, I, mix;
	double eps_rf = 1e10;
};

class Domain {
public:
	explicit Domain(const std::array<double, 3>& globalLength) : _globalLength(globalLength) {}
	double getGlobalLength(int d) const { return _globalLength[d]; }
	void setLocalUpot(double u) { _localUpot = u; }
	double getLocalUpot() const { return _localUpot; }
	void setLocalVirial(double v) { _localVirial = v; }
	double getLocalVirial() const { return _localVirial; }
	void setLocalSummv2(double s, int = 0) { _summv2 = s; }
	void setLocalSumIw2(double s, int = 0) { _sumIw2 = s; }
	void setLocalNrotDOF(int, unsigned long N, unsigned long rotDOF) {
		_N = N;
		_rotDOF = rotDOF;
	}
	double getLocalSummv2(int = 0) const { return _summv2; }
	double getLocalSumIw2(int = 0) const { return _sumIw2; }
	unsigned long getLocalN() const { return _N; }
	unsigned long getLocalRotDOF() const { return _rotDOF; }

private:
	std::array<double, 3> _globalLength;
	double _localUpot = 0., _localVirial = 0., _summv2 = 0., _sumIw2 = 0.;
	unsigned long _N = 0, _rotDOF = 0;
};

class CellProcessor {
public:
	CellProcessor(double cutoffRadius, double LJCutoffRadius) : _cutoffRadius(cutoffRadius), _LJCutoffRadius(LJCutoffRadius) {}
	virtual ~CellProcessor() = default;
	double getCutoffRadius() const { return _cutoffRadius; }
	double getLJCutoffRadius() const { return _LJCutoffRadius; }
	double getCutoffRadiusSquare() const { return _cutoffRadius * _cutoffRadius; }
	double getLJCutoffRadiusSquare() const { return _LJCutoffRadius * _LJCutoffRadius; }
	virtual void initTraversal() = 0;
	virtual void endTraversal() = 0;

protected:
	double _cutoffRadius, _LJCutoffRadius;
};

// The traversal itself happens on the device (LinkedCells::traverseCells); this class keeps the reference's role of
// accumulating the macroscopic values of a traversal and publishing them to Domain in endTraversal().
class VectorizedCellProcessor : public CellProcessor {
public:
	VectorizedCellProcessor(Domain& domain, double cutoffRadius, double LJcutoffRadius)
		: CellProcessor(cutoffRadius, LJcutoffRadius), _domain(domain) {}
	void initTraversal() override { _upot = _virial = 0.; }
	void accumulate(double upot, double virial) {  // called by the container for every device pass of the traversal
		_upot += upot;
		_virial += virial;
	}
	void endTraversal() override {  // VectorizedCellProcessor.cpp:124-157
		_domain.setLocalUpot(_upot);
		_domain.setLocalVirial(_virial);
	}

private:
	Domain& _domain;
	double _upot = 0., _virial = 0.;
};

class LinkedCells {
public:
	LinkedCells(const std::array<double, 3>& bBoxMin, const std::array<double, 3>& bBoxMax, double cutoffRadius,
				const ComponentTables& components, double LJCutoffRadius = -1., int cellsInCutoffRadius = 1, int device = 0)
		: _bBoxMin(bBoxMin), _bBoxMax(bBoxMax), _cutoffRadius(cutoffRadius) {
		int rc = ls1hip_create(device, &_ctx);
		if (rc) throw Error(rc, std::string("ls1hip_create failed: ") + ls1hip_last_error(nullptr));
		const ComponentTables& t = components;
		auto p = [](const std::vector<double>& v) { return v.empty() ? nullptr : v.data(); };
		static const double zero[1] = {0.};
		check(ls1hip_set_components(_ctx, t.ncomp, t.nlj.data(), t.nc.data(), t.nd.data(), t.nq.data(), p(t.lj) ? p(t.lj) : zero,
									p(t.ch) ? p(t.ch) : zero, p(t.dp) ? p(t.dp) : zero, p(t.qp) ? p(t.qp) : zero, t.mass.data(),
									t.I.data(), p(t.mix) ? p(t.mix) : zero, t.eps_rf, cutoffRadius,
									LJCutoffRadius > 0. ? LJCutoffRadius : cutoffRadius));
		check(ls1hip_set_option(_ctx, "cells_in_cutoff", cellsInCutoffRadius));
		// The engine's global box has its origin at 0 (ls1hip_set_domain); the sequential driver's box does too
		// (Domain / DomainDecompBase).  A shifted box would silently move the frame of every uploaded coordinate.
		for (int d = 0; d < 3; ++d)
			if (bBoxMin[d] != 0.) throw Error(LS1HIP_EINVAL, "LinkedCells: the bounding box must start at the origin");
		double len[3], lo[3], hi[3];
		for (int d = 0; d < 3; ++d) {
			len[d] = bBoxMax[d] - bBoxMin[d];
			lo[d] = 0.;
			hi[d] = len[d];
		}
		int nbr[27];
		for (int& n : nbr) n = 0;  // sequential, fully periodic: every neighbour is this rank (DomainDecompBase)
		check(ls1hip_set_domain(_ctx, len, lo, hi, 0, nbr));
	}
	~LinkedCells() {
		if (_ctx) ls1hip_destroy(_ctx);
	}
	LinkedCells(const LinkedCells&) = delete;
	LinkedCells& operator=(const LinkedCells&) = delete;

	// ParticleContainer::addParticles (ParticleContainer.h:108-130): r, v [n][3], q [n][4], D [n][3]
	void addParticles(size_t n, const uint64_t* id, const int32_t* cid, const double* r, const double* v, const double* q,
					  const double* D) {
		check(ls1hip_upload(_ctx, n, id, cid, r, v, q, D));
	}
	void update() { check(ls1hip_rebin(_ctx)); }  // LinkedCells::update, LinkedCells.cpp:243-356
	void updateMoleculeCaches() {}                 // the device SoA is the cache (LinkedCells.cpp:1054-1086)
	void traverseCells(CellProcessor& cellProcessor) {  // LinkedCells.cpp:564-575
		cellProcessor.initTraversal();
		double u = 0., w = 0.;
		check(ls1hip_forces(_ctx, 0, &u, &w));
		if (auto* vcp = dynamic_cast<VectorizedCellProcessor*>(&cellProcessor)) vcp->accumulate(u, w);
		cellProcessor.endTraversal();
	}
	// overlap split of NonBlockingMPIMultiStepHandler (LinkedCells.cpp:577-609): initTraversal / endTraversal are the
	// caller's, exactly as in the reference
	void traversePartialInnermostCells(CellProcessor&, unsigned stage, int stageCount) {
		if (stage == 0 && stageCount >= 1) check(ls1hip_forces(_ctx, 1, nullptr, nullptr));
	}
	void traverseNonInnermostCells(CellProcessor& cellProcessor) {
		double u = 0., w = 0.;
		check(ls1hip_forces(_ctx, 2, &u, &w));  // sums of both passes
		if (auto* vcp = dynamic_cast<VectorizedCellProcessor*>(&cellProcessor)) vcp->accumulate(u, w);
	}
	void deleteOuterParticles() {}  // the halo segment is rebuilt by the next exchange (LinkedCells.cpp:611-628)
	bool requiresForceExchange() const { return false; }
	unsigned long getNumberOfParticles() {
		size_t n = 0, h = 0;
		check(ls1hip_count(_ctx, &n, &h));
		return (unsigned long)n;
	}
	double getCutoff() const { return _cutoffRadius; }
	double getBoundingBoxMin(int d) const { return _bBoxMin[d]; }
	double getBoundingBoxMax(int d) const { return _bBoxMax[d]; }

	// iterator(ONLY_INNER_AND_BOUNDARY) read access: host copies in device order
	struct Molecules {
		std::vector<uint64_t> id;
		std::vector<int32_t> cid;
		std::vector<double> r, v, q, D, F, M;
	};
	Molecules molecules(bool with_forces = true) {
		Molecules m;
		const size_t n = getNumberOfParticles();
		m.id.resize(n); m.cid.resize(n);
		m.r.resize(3 * n); m.v.resize(3 * n); m.q.resize(4 * n); m.D.resize(3 * n);
		check(ls1hip_download_state(_ctx, n, m.id.data(), m.cid.data(), m.r.data(), m.v.data(), m.q.data(), m.D.data()));
		if (with_forces) {
			m.F.resize(3 * n); m.M.resize(3 * n);
			check(ls1hip_download_forces(_ctx, n, m.F.data(), m.M.data(), nullptr));
		}
		return m;
	}

	ls1hip_ctx* context() { return _ctx; }
	void check(int rc) {
		if (rc) throw Error(rc, std::string("ls1hip error ") + std::to_string(rc) + ": " + ls1hip_last_error(_ctx));
	}

private:
	std::array<double, 3> _bBoxMin, _bBoxMax;
	double _cutoffRadius;
	ls1hip_ctx* _ctx = nullptr;
};

// sequential periodic boundary: leaving molecules are wrapped by update(); halo copies are created here
class DomainDecompBase {
public:
	void balanceAndExchange(double /*lastTraversalTime*/, bool /*forceRebalancing*/, LinkedCells& moleculeContainer, Domain&) {
		exchangeMolecules(moleculeContainer);
	}
	void exchangeMolecules(LinkedCells& moleculeContainer) { moleculeContainer.check(ls1hip_halo(moleculeContainer.context())); }
};

class Integrator {
public:
	explicit Integrator(double timestepLength = 0.) : _timestepLength(timestepLength) {}
	virtual ~Integrator() = default;
	double getTimestepLength() const { return _timestepLength; }
	void setTimestepLength(double dt) { _timestepLength = dt; }
	virtual void eventNewTimestep(LinkedCells& molCont, Domain& domain) = 0;
	virtual void eventForcesCalculated(LinkedCells& molCont, Domain& domain) = 0;

protected:
	double _timestepLength;
};

class Leapfrog : public Integrator {
public:
	using Integrator::Integrator;
	void eventNewTimestep(LinkedCells& molCont, Domain&) override {  // Leapfrog.cpp:42-64 -> upd_preF
		molCont.check(ls1hip_kick_drift(molCont.context(), _timestepLength));
	}
	void eventForcesCalculated(LinkedCells& molCont, Domain& domain) override {  // Leapfrog.cpp:35-40,66-150 -> upd_postF
		double summv2 = 0., sumIw2 = 0.;
		uint64_t N = 0, rotDOF = 0;
		molCont.check(ls1hip_kick(molCont.context(), 0.5 * _timestepLength, &summv2, &sumIw2, &N, &rotDOF));
		domain.setLocalSummv2(summv2, 0);
		domain.setLocalSumIw2(sumIw2, 0);
		domain.setLocalNrotDOF(0, (unsigned long)N, (unsigned long)rotDOF);
	}
};

// The hot-path part of Simulation::prepare_start / simulate (Simulation.cpp:813-892, 979-1167), call for call.
inline void simulate(LinkedCells& container, DomainDecompBase& decomp, CellProcessor& cellProcessor, Integrator& integrator,
					 Domain& domain, unsigned long nsteps, bool initial_forces = true) {
	if (initial_forces) {
		container.update();
		decomp.balanceAndExchange(1.0, false, container, domain);
		container.updateMoleculeCaches();
		container.traverseCells(cellProcessor);
		container.deleteOuterParticles();
	}
	for (unsigned long s = 0; s < nsteps; ++s) {
		integrator.eventNewTimestep(container, domain);
		container.update();
		decomp.balanceAndExchange(0.0, false, container, domain);
		container.updateMoleculeCaches();
		container.traverseCells(cellProcessor);
		container.deleteOuterParticles();
		integrator.eventForcesCalculated(container, domain);
	}
}

}  // namespace ls1hip
