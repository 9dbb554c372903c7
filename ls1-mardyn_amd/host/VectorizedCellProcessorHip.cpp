// VectorizedCellProcessorHip.cpp — seam A: a drop-in translation unit for the reference's
//   src/particleContainer/adapter/VectorizedCellProcessor.cpp
// It is compiled AGAINST THE REFERENCE'S OWN HEADER (particleContainer/adapter/VectorizedCellProcessor.h:29-321) and
// linked INSTEAD of the reference TU, so Simulation.cpp:768-779 (`new VectorizedCellProcessor(*_domain, rc, rcLJ)`),
// LinkedCells::traverseCells (LinkedCells.cpp:564-575), Simulation::updateForces (Simulation.cpp:752-762) and every
// other caller stay byte-for-byte unchanged.  Host C++17; all compute goes through the C ABI of libls1hip
// (include/ls1hip.h, ls1hip_soa_forces) to the HIP kernels.
//
// Contract kept (adapter/CellProcessor.h:29-94, threading contract SURVEY.md 8b):
//   initTraversal()                       called by one thread: reset
//   processCell / processCellPair         called concurrently from the traversal's OpenMP threads: re-entrant no-ops —
//                                         the per-cell-pair work of one traversal is ONE kernel launch at endTraversal
//   endTraversal()                        gather r,q,cid of all molecules (owned + halo copies) from the container,
//                                         run the device traversal, add F / M / Vi to the molecules, publish
//                                         U_pot / virial via Domain::setLocalUpot / setLocalVirial
//                                         (VectorizedCellProcessor.cpp:155-156)
// The per-site SoA accumulators stay zero (LinkedCells::updateMoleculeCaches cleared them), so the unchanged
// FullMolecule::calcFM (FullMolecule.cpp:526-629) adds nothing on top of the molecule-level F/M/Vi written here.
// Errors follow the reference convention: log + Simulation::exit(code) (Simulation.cpp:155-158).
#include "particleContainer/adapter/VectorizedCellProcessor.h"

#include <cmath>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "Domain.h"
#include "Simulation.h"
#include "ensemble/EnsembleBase.h"
#include "molecules/Molecule.h"
#include "particleContainer/ParticleContainer.h"
#include "utils/Logger.h"

#include "ls1hip.h"
#include "ls1hip_components.hpp"

using Log::global_log;

namespace {
struct HipState {
	ls1hip_ctx* ctx = nullptr;
	double rc = 0.;
};
std::mutex g_mu;
std::map<const VectorizedCellProcessor*, HipState> g_state;  // the reference header fixes the data members

void die(ls1hip_ctx* ctx, const char* what, int rc) {
	global_log->error() << "VectorizedCellProcessor(hip): " << what << " failed (" << rc << "): "
						<< ls1hip_last_error(ctx) << std::endl;
	Simulation::exit(670 - rc);
}
}  // namespace

VectorizedCellProcessor::VectorizedCellProcessor(Domain& domain, double cutoffRadius, double LJcutoffRadius)
	: CellProcessor(cutoffRadius, LJcutoffRadius), _domain(domain),
	  _epsRFInvrc3(2. * (domain.getepsilonRF() - 1.) / ((cutoffRadius * cutoffRadius * cutoffRadius) * (2. * domain.getepsilonRF() + 1.))),
	  _eps_sig(), _shift6(), _upot6lj(0.0), _upotXpoles(0.0), _virial(0.0), _myRF(0.0) {
	global_log->info() << "VectorizedCellProcessor: MI355X/HIP back end (" << ls1hip_version() << ")" << std::endl;
	HipState st;
	st.rc = cutoffRadius;
	int device = 0;
	if (const char* e = getenv("LS1HIP_DEVICE")) device = atoi(e);
	int rc = ls1hip_create(device, &st.ctx);
	if (rc) die(nullptr, "ls1hip_create", rc);
	// component set -> flat tables of the C ABI (same columns as the .inp component block)
	rc = ls1hip_set_components_from(st.ctx, *(_simulation.getEnsemble()->getComponents()), domain, cutoffRadius, LJcutoffRadius);
	if (rc) die(st.ctx, "ls1hip_set_components", rc);
	_numThreads = 0;
	std::lock_guard<std::mutex> lk(g_mu);
	g_state[this] = st;
}

VectorizedCellProcessor::~VectorizedCellProcessor() {
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_state.find(this);
	if (it != g_state.end()) {
		ls1hip_destroy(it->second.ctx);
		g_state.erase(it);
	}
}

void VectorizedCellProcessor::initTraversal() {
	_virial = 0.0;
	_upot6lj = 0.0;
	_upotXpoles = 0.0;
	_myRF = 0.0;
}

void VectorizedCellProcessor::processCell(ParticleCell&) {}
void VectorizedCellProcessor::processCellPair(ParticleCell&, ParticleCell&, bool) {}

void VectorizedCellProcessor::endTraversal() {
	HipState st;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		st = g_state.at(this);
	}
	ParticleContainer* cont = global_simulation->getMoleculeContainer();
	// cell grid of the container (LinkedCells::rebuild, LinkedCells.cpp:150-170) incl. its one-cell halo layer
	double bmin[3], bmax[3], clen[3];
	int box[3], dims[3];
	const float rcf = (float)st.rc;
	for (int d = 0; d < 3; ++d) {
		bmin[d] = cont->getBoundingBoxMin(d);
		bmax[d] = cont->getBoundingBoxMax(d);
		box[d] = (int)std::floor((bmax[d] - bmin[d]) / rcf);
		dims[d] = box[d] + 2;
		clen[d] = (bmax[d] - bmin[d]) / box[d];
	}
	std::vector<Molecule*> mols;
	for (auto m = cont->iterator(ParticleIterator::ALL_CELLS); m.isValid(); ++m) mols.push_back(&(*m));
	const size_t n = mols.size();
	const size_t ncells = (size_t)dims[0] * dims[1] * dims[2];
	std::vector<uint32_t> cell(n), cell_start(ncells + 1, 0), fill;
	for (size_t i = 0; i < n; ++i) {
		int c[3];
		for (int d = 0; d < 3; ++d) {
			const double x = mols[i]->r(d);
			int k;
			if (x < bmin[d]) k = 0;
			else if (x >= bmax[d]) k = dims[d] - 1;
			else {
				k = (int)std::floor((x - bmin[d]) / clen[d]);
				if (k < 0) k = 0;
				if (k > box[d] - 1) k = box[d] - 1;
				k += 1;
			}
			c[d] = k;
		}
		cell[i] = (uint32_t)((c[2] * dims[1] + c[1]) * dims[0] + c[0]);
		cell_start[cell[i] + 1]++;
	}
	for (size_t c = 0; c < ncells; ++c) cell_start[c + 1] += cell_start[c];
	fill.assign(cell_start.begin(), cell_start.end() - 1);
	std::vector<uint32_t> order(n);
	for (size_t i = 0; i < n; ++i) order[fill[cell[i]]++] = (uint32_t)i;
	std::vector<double> r(3 * n), q(4 * n), F(3 * n), M(3 * n), Vi(3 * n);
	std::vector<int32_t> cid(n);
	for (size_t k = 0; k < n; ++k) {
		const Molecule* m = mols[order[k]];
		for (int d = 0; d < 3; ++d) r[3 * k + d] = m->r(d);
		q[4 * k] = m->q().qw(); q[4 * k + 1] = m->q().qx(); q[4 * k + 2] = m->q().qy(); q[4 * k + 3] = m->q().qz();
		cid[k] = (int32_t)m->componentid();
	}
	double upot = 0., virial = 0.;
	int rc = ls1hip_soa_forces(st.ctx, dims, cell_start.data(), n, r.data(), q.data(), cid.data(), F.data(), M.data(),
							   Vi.data(), &upot, &virial);
	if (rc) die(st.ctx, "ls1hip_soa_forces", rc);
	for (size_t k = 0; k < n; ++k) {
		Molecule* m = mols[order[k]];
		m->Fadd(&F[3 * k]);
		m->Madd(&M[3 * k]);
		m->Viadd(&Vi[3 * k]);
	}
	_virial = virial;
	_upot6lj = 0.0;
	_upotXpoles = upot;
	_domain.setLocalVirial(virial);
	_domain.setLocalUpot(upot);
}
