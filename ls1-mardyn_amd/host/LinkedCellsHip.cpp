// LinkedCellsHip.cpp — seam B (see LinkedCellsHip.h): the ParticleContainer and Integrator the unmodified
// Simulation.cpp constructs, forwarding the hot path to libls1hip (include/ls1hip.h) and every host-side service to a
// real reference LinkedCells object kept as the lazily synced mirror.
//
// One time step as the unmodified driver runs it (Simulation.cpp:979-1167) and where each call lands:
//   _integrator->eventNewTimestep          LeapfrogHip: ls1hip_scale_kick_drift (velocity scaling with the driver's betas + kick + drift, one pass)
//   _moleculeContainer->update             ls1hip_rebin (wrap + counting sort on the device)
//   _domainDecomposition->balanceAndExchange   the reference's host loops find an empty region iterator (no-op); the
//                                          periodic wrap / halo copies they would make are done by ls1hip_rebin / ls1hip_halo
//   _moleculeContainer->updateMoleculeCaches   ls1hip_halo
//   _moleculeContainer->traverseCells      ls1hip_forces -> Domain::setLocalUpot / setLocalVirial (as AutoPasContainer does,
//                                          particleContainer/AutoPasContainer.cpp:348-395)
//   updateForces (calcFM loop)             iterates the emptied mirror: no-op (site reduction is fused in the force kernel)
//   _moleculeContainer->deleteOuterParticles   no-op (the halo segment is rebuilt every step)
//   _integrator->eventForcesCalculated     LeapfrogHip: ls1hip_kick -> Domain::setLocalSummv2 / SumIw2 / NrotDOF
//   _domain->calculateGlobalValues         unchanged reference code on those sums (beta factors)
//   _velocityScalingThermostat.apply       iterates the emptied mirror: no-op; the betas are applied on the device at the
//                                          head of the next eventNewTimestep (same position in the sequence of kicks)
#include "LinkedCellsHip.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include "Domain.h"
#include "Simulation.h"
#include "ensemble/EnsembleBase.h"
#include "molecules/Molecule.h"
#include "parallel/DomainDecompBase.h"
#include "particleContainer/adapter/VectorizedCellProcessor.h"
#include "plugins/PluginBase.h"
#include "utils/Logger.h"
#include "utils/xmlfileUnits.h"

#include "DomainDecompHip.h"
#include "ls1hip_components.hpp"

using Log::global_log;

// LS1HIP_PROFILE=1: host wall time spent inside the device container / integrator entry points, printed at the last step — what of
// the driver's per-step time is ours (queueing + waiting for the device) and what is the driver's own host work
#include <chrono>
namespace {
struct HostProfile {
	bool on = getenv("LS1HIP_PROFILE") != nullptr;
	std::map<std::string, std::pair<double, unsigned long>> t;
	void report() {
		if (!on) return;
		for (auto& it : t)
			global_log->info() << "LS1HIP_PROFILE " << it.first << ": " << it.second.first << " s in " << it.second.second << " calls" << std::endl;
	}
} g_prof;
struct ProfScope {
	const char* name;
	std::chrono::steady_clock::time_point t0;
	explicit ProfScope(const char* n) : name(n) {
		if (g_prof.on) t0 = std::chrono::steady_clock::now();
	}
	~ProfScope() {
		if (!g_prof.on) return;
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		// iterator() is called by every thread of the driver's parallel regions (LinkedCells.h:245-250): one writer at a time
#if defined(_OPENMP)
#pragma omp critical(ls1hip_profile_map)
#endif
		{
			auto& e = g_prof.t[name];
			e.first += dt;
			e.second++;
		}
	}
};
}  // namespace

void LinkedCellsHip::die(const char* what, int rc) const {
	global_log->error() << "LinkedCellsHip: " << what << " failed (" << rc << "): " << ls1hip_last_error(_ctx) << std::endl;
	Simulation::exit(680 - rc);
}

LinkedCellsHip::LinkedCellsHip() : ParticleContainer(), _mirror() {
	global_log->info() << "LinkedCellsHip: device-resident container, MI355X/HIP back end (" << ls1hip_version() << ")" << std::endl;
}

LinkedCellsHip::LinkedCellsHip(double bBoxMin[3], double bBoxMax[3], double cutoffRadius)
	: ParticleContainer(bBoxMin, bBoxMax), _mirror(bBoxMin, bBoxMax, cutoffRadius) {}

LinkedCellsHip::~LinkedCellsHip() {
	if (_ctx) ls1hip_destroy(_ctx);
}

void LinkedCellsHip::readXML(XMLfileUnits& xmlconfig) { mirror().readXML(xmlconfig); }

bool LinkedCellsHip::rebuild(double bBoxMin[3], double bBoxMax[3]) {
	for (int d = 0; d < 3; ++d) {
		_boundingBoxMin[d] = bBoxMin[d];
		_boundingBoxMax[d] = bBoxMax[d];
	}
	_hostDirty = true;  // the device grid is derived from the box at the next upload
	return mirror().rebuild(bBoxMin, bBoxMax);
}

// ---- host-side population: everything goes to the mirror ---------------------------------------------------------------
bool LinkedCellsHip::addParticle(Molecule& particle, bool inBoxCheckedAlready, bool checkWhetherDuplicate,
								 const bool& rebuildCaches) {
	if (_inExchange) return false;  // (unreachable: the exchange window sees empty region iterators)
	if (!_mirrorFresh) syncMirrorFromDevice();
	_hostDirty = true;
	return mirror().addParticle(particle, inBoxCheckedAlready, checkWhetherDuplicate, rebuildCaches);
}

bool LinkedCellsHip::addHaloParticle(Molecule& particle, bool inBoxCheckedAlready, bool checkWhetherDuplicate,
									 const bool& rebuildCaches) {
	// halo copies are generated on the device (ls1hip_halo); host-side copies would only matter to host traversals
	(void)particle; (void)inBoxCheckedAlready; (void)checkWhetherDuplicate; (void)rebuildCaches;
	return false;
}

void LinkedCellsHip::addParticles(std::vector<Molecule>& particles, bool checkWhetherDuplicate) {
	if (!_mirrorFresh) syncMirrorFromDevice();
	_hostDirty = true;
	mirror().addParticles(particles, checkWhetherDuplicate);
}

unsigned long LinkedCellsHip::initCubicGrid(std::array<unsigned long, 3> numMoleculesPerDimension,
											std::array<double, 3> simBoxLength, size_t seed_offset) {
	_hostDirty = true;
	_mirrorFresh = true;
	return mirror().initCubicGrid(numMoleculesPerDimension, simBoxLength, seed_offset);
}

void LinkedCellsHip::clear() {
	mirror().clear();
	_mirrorFresh = true;
	_hostDirty = true;
	_uploaded = false;
}

void LinkedCellsHip::deleteMolecule(ParticleIterator& moleculeIter, const bool& rebuildCaches) {
	// the iterator can only point into a fresh mirror (a stale one is empty inside the host-loop windows and refilled outside)
	mirror().deleteMolecule(moleculeIter, rebuildCaches);
	_hostDirty = true;
}

double LinkedCellsHip::getEnergy(ParticlePairsHandler* particlePairsHandler, Molecule* m1, CellProcessor& cellProcessor) {
	if (!_mirrorFresh) syncMirrorFromDevice();
	return mirror().getEnergy(particlePairsHandler, m1, cellProcessor);  // grand-canonical insertions: host path of the reference
}

std::variant<ParticleIterator, SingleCellIterator<ParticleCell>> LinkedCellsHip::getMoleculeAtPosition(const double pos[3]) {
	if (!_mirrorFresh) syncMirrorFromDevice();  // a point query has no per-step caller in the driver: always the real molecules
	return mirror().getMoleculeAtPosition(pos);
}

// Who iterates a stale mirror?  Inside the two windows below it is the driver's own per-step host loops, whose work the device has
// done (they must find nothing); anywhere else it is a consumer of the molecules — an end-of-step plugin (CheckpointWriter with a
// write frequency, MmpldWriter, MaxCheck ...), the timed or final checkpoint (Simulation.cpp:1169-1175,1216-1225), a finishing
// plugin — and it gets the real molecules: the mirror is refilled from the device ON DEMAND, never silently empty.
//   window 1: eventNewTimestep ... eventForcesCalculated  (exchangeMolecules, updateForces / calcFM, Simulation.cpp:995-1099)
//   window 2: eventForcesCalculated ... advanceSimulationTime  (VelocityScalingThermostat::apply, Simulation.cpp:1108-1136): the
//             simulation time is still the one eventForcesCalculated saw
bool LinkedCellsHip::inHostLoopWindow() {
	if (_stepOpen) return true;
	if (_quietArmed) {
		if (global_simulation->getSimulationTime() == _quietTime) return true;
		_quietArmed = false;  // the driver advanced the time: the step's host loops are over
	}
	return false;
}

void LinkedCellsHip::stepClosed() {
	_stepOpen = false;
	_quietArmed = true;
	_quietTime = global_simulation->getSimulationTime();
	_betaPending = !global_simulation->getDomain()->NVE();
}

void LinkedCellsHip::ensureMirror() {
	ProfScope prof_("ensureMirror");
	if (_mirrorFresh || !_uploaded || inHostLoopWindow()) return;
	// iterator() may be called by every thread of a parallel region at once (LinkedCells.h:245-250): one of them refills
#if defined(_OPENMP)
#pragma omp critical(ls1hip_mirror_sync)
#endif
	{
		// past the thermostat's window: the scaling of the step just finished (VelocityScalingThermostat::apply, which found an
		// empty mirror) is still pending on the device — it is folded into the next kick + drift pass — so the snapshot applies it
		if (!_mirrorFresh) syncMirrorFromDevice(_betaPending);
	}
}

void LinkedCellsHip::joinClear() {
	if (_clearJob.valid()) _clearJob.get();
}

ParticleIterator LinkedCellsHip::iterator(ParticleIterator::Type t) {
	ProfScope prof_("iterator");
	ensureMirror();
	if (!_mirrorFresh) return ParticleIterator();  // the driver's per-step host loops over a stale mirror: nothing to visit
	return mirror().iterator(t);
}

RegionParticleIterator LinkedCellsHip::regionIterator(const double startCorner[3], const double endCorner[3],
													   ParticleIterator::Type t) {
	if (_inExchange) return RegionParticleIterator();  // DomainDecompBase::exchangeMolecules: done on the device
	ensureMirror();
	if (!_mirrorFresh) return RegionParticleIterator();
	return mirror().regionIterator(startCorner, endCorner, t);
}

unsigned long LinkedCellsHip::getNumberOfParticles() {
	ProfScope prof_("getNumberOfParticles");
	if (_uploaded && !_hostDirty) {
		size_t n = 0, h = 0;
		ls1hip_count(_ctx, &n, &h);
		return (unsigned long)n;
	}
	return mirror().getNumberOfParticles();
}

std::vector<unsigned long> LinkedCellsHip::getParticleCellStatistics() {
	ensureMirror();
	return mirror().getParticleCellStatistics();
}
std::string LinkedCellsHip::getConfigurationAsString() { return mirror().getConfigurationAsString() + " (device-resident, libls1hip)"; }
size_t LinkedCellsHip::getTotalSize() { return mirror().getTotalSize(); }
void LinkedCellsHip::printSubInfo(int offset) { mirror().printSubInfo(offset); }

// ---- device side ------------------------------------------------------------------------------------------------------------
void LinkedCellsHip::uploadFromMirror() {
	Simulation* sim = global_simulation;
	Domain* domain = sim->getDomain();
	// multi-rank: the decomposition the driver constructed must be the device-side one (seam_b_register.h maps it); it supplies
	// the rank grid, this rank's neighbours and the transport (RCCL over xGMI, or the host-staged mailbox)
	DomainDecompHip* dd = dynamic_cast<DomainDecompHip*>(&sim->domainDecomposition());
	_multiRank = sim->domainDecomposition().getNumProcs() > 1 || (dd && dd->decomposed());  // (decomposed: > 1 rank, or the loopback rehearsal)
	if (_multiRank && !dd) {
		global_log->error() << "LinkedCellsHip: a multi-rank run needs DomainDecompHip (seam_b_register.h / INTEGRATION.md)" << std::endl;
		Simulation::exit(681);
	}
	int rc;
	if (!_ctx) {
		int device = dd ? dd->localDevice() : 0;
		if (const char* e = getenv("LS1HIP_DEVICE")) device = atoi(e);
		if ((rc = ls1hip_create(device, &_ctx))) die("ls1hip_create", rc);
	}
	if ((rc = ls1hip_set_components_from(_ctx, *(sim->getEnsemble()->getComponents()), *domain, sim->getcutoffRadius(),
										 sim->getLJCutoff())))
		die("ls1hip_set_components", rc);
	if ((rc = ls1hip_set_option(_ctx, "cells_in_cutoff", mirror().getHaloWidthNumCells()))) die("ls1hip_set_option", rc);
	// neighbour lists with a skin (ls1hip_set_verlet): on by default, LS1HIP_SKIN=<length> overrides, LS1HIP_SKIN=0 keeps the
	// reference's search-every-step scheme.  Default: 8 % of the cutoff for the single-centre LJ lists (every listed pair costs a
	// full pair evaluation), 15 % for multi-site sets (their list pass filters the listed pairs by the cutoff first, a pair of the
	// skin costs a few instructions: profiles/r4_ms_skin_sweep.txt, r4_seam_b_speed.txt).  The engine falls back by itself where
	// lists do not apply (two cells per cutoff, regions beyond its staging capacity).
	bool single_centre = true;
	for (const Component& comp : *(sim->getEnsemble()->getComponents())) single_centre = single_centre && comp.numSites() == 1 && comp.numLJcenters() == 1;
	single_centre = single_centre && sim->getEnsemble()->getComponents()->size() == 1;
	_skin = (single_centre ? 0.08 : 0.15) * sim->getcutoffRadius();
	if (const char* e = getenv("LS1HIP_SKIN")) _skin = atof(e);
	if (mirror().getHaloWidthNumCells() != 1) _skin = 0.;
	// Multi-rank runs use the lists too (round 4: the default; LS1HIP_MULTIRANK_LISTS=0 searches every step).  Between two rebuilds a
	// molecule may sit up to skin / 2 outside its owner's box, where the host mirror — a real LinkedCells with the rank's bounding
	// box — cannot hold it: a snapshot for a reader of the mirror (syncMirrorFromDevice) therefore hands such molecules to the rank
	// whose box they are in, for the snapshot only; the device state and the lists are untouched.
	if (_multiRank && getenv("LS1HIP_MULTIRANK_LISTS") && atoi(getenv("LS1HIP_MULTIRANK_LISTS")) == 0) _skin = 0.;
	_overlap = !(getenv("LS1HIP_OVERLAP") && atoi(getenv("LS1HIP_OVERLAP")) == 0);
	if ((rc = ls1hip_set_verlet(_ctx, _skin > 0. ? 1 : 0, _skin))) die("ls1hip_set_verlet", rc);
	double glen[3], bmin[3], bmax[3];
	int nbr[27];
	for (int d = 0; d < 3; ++d) {
		glen[d] = domain->getGlobalLength(d);
		bmin[d] = mirror().getBoundingBoxMin(d);
		bmax[d] = mirror().getBoundingBoxMax(d);
	}
	for (int k = 0; k < 27; ++k) nbr[k] = 0;  // DomainDecompBase: every side is periodic onto this rank
	if (_multiRank) dd->neighbourTable(glen, nbr);
	if ((rc = ls1hip_set_domain(_ctx, glen, bmin, bmax, _multiRank ? dd->getRank() : 0, nbr))) die("ls1hip_set_domain", rc);
	// streamed in chunks straight from the mirror's iterator (a full copy in six vectors next to the mirror's Molecule objects
	// cost ~17 GB of host memory more at 10^8 molecules): the announced total is an upper bound that sizes the device arrays
	const unsigned long n0 = mirror().getNumberOfParticles();
	constexpr size_t CHUNK = 1u << 20;
	std::vector<uint64_t> id(CHUNK);
	std::vector<int32_t> cid(CHUNK);
	std::vector<double> r(3 * CHUNK), v(3 * CHUNK), q(4 * CHUNK), D(3 * CHUNK);
	if ((rc = ls1hip_upload_begin(_ctx, n0))) die("ls1hip_upload_begin", rc);
	size_t k = 0;
	auto flush = [&]() {
		if (k && (rc = ls1hip_upload_chunk(_ctx, k, id.data(), cid.data(), r.data(), v.data(), q.data(), D.data()))) die("ls1hip_upload_chunk", rc);
		k = 0;
	};
	for (auto m = mirror().iterator(ParticleIterator::ONLY_INNER_AND_BOUNDARY); m.isValid(); ++m) {
		id[k] = m->getID();
		cid[k] = (int32_t)m->componentid();
		for (int d = 0; d < 3; ++d) {
			r[3 * k + d] = m->r(d);
			v[3 * k + d] = m->v(d);
			D[3 * k + d] = m->D(d);
		}
		q[4 * k] = m->q().qw(); q[4 * k + 1] = m->q().qx(); q[4 * k + 2] = m->q().qy(); q[4 * k + 3] = m->q().qz();
		if (++k == CHUNK) flush();
	}
	flush();
	if ((rc = ls1hip_upload_end(_ctx))) die("ls1hip_upload_end", rc);
	_uploaded = true;
	_hostDirty = false;
}

void LinkedCellsHip::update() {
	ProfScope prof_("update");
	if (_hostDirty || !_uploaded) {
		if (!_mirrorFresh) {
			global_log->error() << "LinkedCellsHip: molecules were changed on the host while the mirror was stale" << std::endl;
			Simulation::exit(682);
		}
		mirror().update();
		uploadFromMirror();
	}
	if (_multiRank) {
		// List mode (the piecewise scheme of ls1hip.h "list mode", as the handed-over decomposed loop runs it): between two rebuilds
		// no molecule changes its owner and every halo copy keeps its slot — this step only refreshes positions (see
		// exchangeAcrossRanks).  The ranks rebuild together: each polls its displacement bound (advanced on the device by the
		// integrator's drift), one logical OR over the ranks decides.
		long ready = 0;
		ls1hip_get_option(_ctx, "verlet_ready", &ready);
		bool rebuild = true;
		if (_skin > 0. && ready) {
			long pending = 0;
			ls1hip_get_option(_ctx, "verlet_bound_pending", &pending);
			int need = 0;
			if (pending) {  // (no drift since the lists were built or last asked: they are still good)
				int rcp = ls1hip_verlet_poll(_ctx, &need);
				if (rcp) die("ls1hip_verlet_poll", rcp);
			}
			rebuild = need != 0;
		}
		DomainDecompHip* dd = dynamic_cast<DomainDecompHip*>(&global_simulation->domainDecomposition());
		_rebuildStep = dd ? dd->anyRank(rebuild) : rebuild;
		if (_rebuildStep) {
			// classification only (leavers packed per direction); DomainDecompHip::balanceAndExchange moves them, then the halo copies
			int rc = ls1hip_rebin(_ctx);
			if (rc) die("ls1hip_rebin", rc);
		}
		_inExchange = true;
		return;
	}
	// list-aware: re-bin + halo (+ list build), or — while the lists are alive — only a refresh of the halo positions
	int rc = ls1hip_update(_ctx, nullptr);
	if (rc) die("ls1hip_update", rc);
	_inExchange = true;
}

// DomainDecompHip::balanceAndExchange (multi-rank): leaving molecules, halo generation, halo copies — the direct scheme of the
// reference (NeighbourCommunicationScheme.cpp:115-136) over the export / import entry points
//
// Overlap (round 4; the reference's own split: C08CellPairTraversal.h:78-192 traverseCellPairsInner / Outer driven by
// NonBlockingMPIMultiStepHandler.cpp:30-97, which a non-MPI build of the driver never reaches): the INNER pass — bricks whose
// cutoff shell holds owned molecules only — is queued on the engine's main stream as soon as the owned molecules are final, the
// whole halo phase (generation or refresh, packing, the transfers on the transport's stream, import) runs while it computes, and
// the driver's traverseCells then launches the boundary pass only, which waits on the device for the imported halo.
// LS1HIP_OVERLAP=0 keeps one complete pass behind a blocking exchange.
void LinkedCellsHip::exchangeAcrossRanks(DomainDecompHip& dd, Domain* domain) {
	double glen[3];
	for (int d = 0; d < 3; ++d) glen[d] = domain->getGlobalLength(d);
	int rc;
	_innerLaunched = false;
	if (!_rebuildStep) {  // list-reuse step: current positions of the build-time halo records, nothing else travels
		if (_overlap) {
			deviceForces(1);  // inner bricks: owned positions only
			_innerLaunched = true;
		}
		if ((rc = ls1hip_halo_refresh(_ctx))) die("ls1hip_halo_refresh", rc);
		dd.exchange(_ctx, glen, 2);
		return;
	}
	dd.exchange(_ctx, glen, 0);
	long lists = 0;
	ls1hip_get_option(_ctx, "verlet_lists", &lists);
	if (_overlap && !lists) {  // search every step: the owned segment is sorted once the immigrants are in
		deviceForces(1);
		_innerLaunched = true;
	}
	if ((rc = ls1hip_halo(_ctx))) die("ls1hip_halo", rc);
	dd.exchange(_ctx, glen, 1);
	if (lists && (rc = ls1hip_verlet_build(_ctx))) die("ls1hip_verlet_build", rc);  // (a rebuild step: the build needs the halo)
}

void LinkedCellsHip::updateMoleculeCaches() {
	_inExchange = false;  // (the halo was populated / refreshed by ls1hip_update)
	if (_mirrorFresh) mirror().updateMoleculeCaches();  // keeps calcFM on mirror molecules (prepare_start) well defined
}
void LinkedCellsHip::updateInnerMoleculeCaches() {}
void LinkedCellsHip::updateBoundaryAndHaloMoleculeCaches() { updateMoleculeCaches(); }

void LinkedCellsHip::deleteOuterParticles() {
	if (_mirrorFresh) mirror().deleteOuterParticles();  // device: the halo segment is rebuilt by every ls1hip_halo
}

void LinkedCellsHip::deviceForces(int which) {
	ProfScope prof_("deviceForces");
	double upot = 0., virial = 0.;
	const bool want = which != 1;
	long lists = 0;
	ls1hip_get_option(_ctx, "verlet_ready", &lists);
	// The kernels are only queued here.  With the integrator's post-force kick armed (LeapfrogHip::eventNewTimestep) it is
	// queued right behind the traversal, and the host waits for the traversal's sums only: the kick runs on the device while the
	// driver does its host work between traverseCells and eventForcesCalculated (long-range correction, timers, plugins).
	// single-centre LJ list path, complete traversal: the pass does the armed post-force kick and its kinetic sum in its own
	// epilogue (ls1hip_forces_list_kick) — one pass over the molecules less per step
	long fold = 0;
	if (lists && which == 0 && _armedKick > 0.) ls1hip_get_option(_ctx, "list_kick_available", &fold);
	int rc;
	if (fold) {
		rc = ls1hip_forces_list_kick(_ctx, _armedKick, nullptr, nullptr);
		_armedKick = 0.;
		_kickQueued = true;
	} else {
		rc = lists ? ls1hip_forces_list(_ctx, which, 0., nullptr, nullptr) : ls1hip_forces(_ctx, which, nullptr, nullptr);
	}
	if (rc) die("ls1hip_forces", rc);
	if (want) {
		if ((rc = ls1hip_traversal_mark(_ctx))) die("ls1hip_traversal_mark", rc);
		if (_armedKick > 0.) {
			if ((rc = ls1hip_kick(_ctx, _armedKick, nullptr, nullptr, nullptr, nullptr))) die("ls1hip_kick", rc);
			_armedKick = 0.;
			_kickQueued = true;
		}
		if ((rc = ls1hip_traversal_sums(_ctx, &upot, &virial))) die("ls1hip_traversal_sums", rc);
		// what VectorizedCellProcessor::endTraversal publishes (VectorizedCellProcessor.cpp:155-156)
		Domain* domain = global_simulation->getDomain();
		domain->setLocalUpot(upot);
		domain->setLocalVirial(virial);
	}
}

static void require_vectorized(CellProcessor& cp) {
	if (dynamic_cast<VectorizedCellProcessor*>(&cp) == nullptr) {
		global_log->error() << "LinkedCellsHip: only the force traversal (VectorizedCellProcessor) runs on the device container; "
							   "other cell processors need the host container" << std::endl;
		Simulation::exit(683);
	}
}

void LinkedCellsHip::traverseCells(CellProcessor& cellProcessor) {
	require_vectorized(cellProcessor);
	// multi-rank with overlap: the inner pass was queued inside balanceAndExchange, ahead of the halo phase
	deviceForces(_innerLaunched ? 2 : 0);
	_innerLaunched = false;
}
void LinkedCellsHip::traverseNonInnermostCells(CellProcessor& cellProcessor) {
	require_vectorized(cellProcessor);
	deviceForces(2);
}
void LinkedCellsHip::traversePartialInnermostCells(CellProcessor& cellProcessor, unsigned int stage, int stageCount) {
	require_vectorized(cellProcessor);
	(void)stageCount;
	if (stage == 0) deviceForces(1);  // the whole inner pass is one launch: it runs entirely at stage 0
}

// Plugins whose per-step hooks run INSIDE the windows in which the host mirror is empty (beforeEventNewTimestep, beforeForces,
// siteWiseForces, afterForces: Simulation.cpp:1001-1082) would silently find no molecule on the device path.  Known ones stop the
// run with an explanation instead; plugins that act at endStep / finish (writers, samplers) get the refilled mirror.
// LS1HIP_ALLOW_PLUGINS=1 lets a user take responsibility for a plugin whose overrides are empty.
static void check_plugins_once() {
	static const std::set<std::string> in_window = {
		"CavityWriter", "CommunicationPartnerWriter", "FlopRateWriter", "HaloParticleWriter", "ODF", "RDF", "COMaligner", "DirectedPM",
		"Dropaccelerator", "Dropaligner", "ExamplePlugin", "FixRegion", "InMemoryCheckpointing", "MaxCheck", "Mirror", "MirrorSystem",
		"TestPlugin", "WallPotential", "DistControl", "DriftCtrl", "ExtractPhase", "MettDeamon", "MettDeamonFeedrateDirector", "PosNegComp",
		"RegionSampling"};
	if (getenv("LS1HIP_ALLOW_PLUGINS") && atoi(getenv("LS1HIP_ALLOW_PLUGINS")) != 0) return;
	std::list<PluginBase*>* plugins = global_simulation->getPluginList();
	if (!plugins) return;
	for (PluginBase* p : *plugins) {
		if (!p) continue;
		const std::string name = p->getPluginName();
		if (in_window.count(name)) {
			global_log->error() << "LinkedCellsHip: plugin '" << name << "' acts on the molecules between eventNewTimestep and the thermostat "
								   "(beforeForces / siteWiseForces / afterForces ...): on the device container those hooks see an empty host "
								   "mirror and would silently do nothing.  Run it with the host container, or set LS1HIP_ALLOW_PLUGINS=1 if its "
								   "hooks are known to be empty." << std::endl;
			Simulation::exit(698);
		}
	}
}

void LinkedCellsHip::deviceAdvanced() {
	if (!_pluginsChecked) {
		_pluginsChecked = true;
		check_plugins_once();
	}
	++_stepIndex;
	_stepOpen = true;
	_quietArmed = false;
	_betaPending = false;  // (the integrator has just applied the factors on the device)
	if (_mirrorFresh) {
		// stale from here on: iterator() yields nothing without touching the mirror.  LinkedCells::clear (LinkedCells.cpp:590-595)
		// walks the cells serially — 50 ms at 10^7 molecules, inside the driver's timed loop (emptying the cells through the
		// container's parallel iterator was measured: 84 ms) — so it runs on a helper thread; every other user of the mirror
		// waits for it (joinClear)
		_clearJob = std::async(std::launch::async, [this] { _mirror.clear(); });
		_mirrorFresh = false;
	}
}

void LinkedCellsHip::syncMirrorFromDevice(bool applyPendingBeta) {
	if (!_uploaded) return;
	std::vector<Component>& comps0 = *(global_simulation->getEnsemble()->getComponents());
	std::vector<double> betaT(comps0.size(), 1.), betaR(comps0.size(), 1.);
	if (applyPendingBeta) {  // the factors the next eventNewTimestep will apply on the device (global or per thermostat)
		Domain* domain = global_simulation->getDomain();
		for (size_t k = 0; k < comps0.size(); ++k) {
			const bool several = domain->severalThermostats();
			betaT[k] = several ? domain->getGlobalBetaTrans(domain->getThermostat((int)k)) : domain->getGlobalBetaTrans();
			betaR[k] = several ? domain->getGlobalBetaRot(domain->getThermostat((int)k)) : domain->getGlobalBetaRot();
		}
	}
	size_t n = 0, h = 0;
	ls1hip_count(_ctx, &n, &h);
	std::vector<uint64_t> id(n);
	std::vector<int32_t> cid(n);
	std::vector<double> r(3 * n), v(3 * n), q(4 * n), D(3 * n), F(3 * n), M(3 * n);
	int rc = ls1hip_download_state(_ctx, n, id.data(), cid.data(), r.data(), v.data(), q.data(), D.data());
	if (rc) die("ls1hip_download_state", rc);
	const bool haveF = ls1hip_download_forces(_ctx, n, F.data(), M.data(), nullptr) == LS1HIP_OK;
	std::vector<Component>& comps = *(global_simulation->getEnsemble()->getComponents());
	mirror().clear();
	std::vector<Molecule> mols;
	mols.reserve(n);
	// Multi-rank list mode: a molecule that awaits its migration (up to skin / 2 outside this rank's box until the next list rebuild)
	// is handed, FOR THIS SNAPSHOT ONLY, to the rank whose box holds its periodically wrapped position — the union of the ranks'
	// mirrors is every molecule exactly once, each inside its holder's box, whatever step a reader asks at.  Collective: every rank
	// reaches this point together (the driver is SPMD; the mirror's readers run on all ranks).
	struct Stray {
		uint64_t id;
		int64_t cid;
		double r[3], v[3], q[4], D[3], F[3], M[3];
	};
	std::vector<Stray> strays;
	std::vector<char> mine_is_stray(n, 0);
	const bool migrate = _multiRank && _skin > 0.;
	if (migrate) {
		Domain* domain = global_simulation->getDomain();
		for (size_t i = 0; i < n; ++i) {
			bool out = false;
			for (int d = 0; d < 3; ++d) out = out || r[3 * i + d] < _mirror.getBoundingBoxMin(d) || r[3 * i + d] >= _mirror.getBoundingBoxMax(d);
			if (!out) continue;
			mine_is_stray[i] = 1;
			Stray s;
			s.id = id[i];
			s.cid = cid[i];
			for (int d = 0; d < 3; ++d) {
				const double L = domain->getGlobalLength(d);
				double x = r[3 * i + d];
				x -= L * std::floor(x / L);
				if (x >= L) x = std::nextafter(L, 0.);
				s.r[d] = x;
				s.v[d] = betaT[cid[i]] * v[3 * i + d];
				s.D[d] = betaR[cid[i]] * D[3 * i + d];
				s.F[d] = haveF ? F[3 * i + d] : 0.;
				s.M[d] = haveF ? M[3 * i + d] : 0.;
			}
			for (int d = 0; d < 4; ++d) s.q[d] = q[4 * i + d];
			strays.push_back(s);
		}
	}
	for (size_t i = 0; i < n; ++i) {
		if (mine_is_stray[i]) continue;
		const double betaTrans = betaT[cid[i]], betaRot = betaR[cid[i]];
		Molecule m(id[i], &comps[cid[i]], r[3 * i], r[3 * i + 1], r[3 * i + 2], betaTrans * v[3 * i], betaTrans * v[3 * i + 1],
				   betaTrans * v[3 * i + 2], q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3], betaRot * D[3 * i],
				   betaRot * D[3 * i + 1], betaRot * D[3 * i + 2]);
		if (haveF) {
			m.setF(&F[3 * i]);
			m.setM(&M[3 * i]);
		}
		mols.push_back(m);
	}
	if (migrate) {
		DomainDecompHip* dd = dynamic_cast<DomainDecompHip*>(&global_simulation->domainDecomposition());
		std::vector<char> all;
		std::vector<size_t> counts;
		dd->gatherRecords(strays.data(), strays.size(), sizeof(Stray), all, counts);
		const size_t total = all.size() / sizeof(Stray);
		size_t taken = 0;
		for (size_t k = 0; k < total; ++k) {
			Stray s;
			std::memcpy(&s, all.data() + k * sizeof(Stray), sizeof(Stray));
			bool in = true;
			for (int d = 0; d < 3; ++d) in = in && s.r[d] >= _mirror.getBoundingBoxMin(d) && s.r[d] < _mirror.getBoundingBoxMax(d);
			if (!in) continue;
			Molecule m(s.id, &comps[(size_t)s.cid], s.r[0], s.r[1], s.r[2], s.v[0], s.v[1], s.v[2], s.q[0], s.q[1], s.q[2], s.q[3], s.D[0], s.D[1], s.D[2]);
			if (haveF) {
				m.setF(s.F);
				m.setM(s.M);
			}
			mols.push_back(m);
			++taken;
		}
		if (getenv("LS1HIP_LOG_SNAPSHOTS")) {
			double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
			for (size_t i = 0; i < n; ++i)
				for (int d = 0; d < 3; ++d) {
					lo[d] = std::min(lo[d], r[3 * i + d]);
					hi[d] = std::max(hi[d], r[3 * i + d]);
				}
			long nb = 0, ns = 0;
			ls1hip_get_option(_ctx, "verlet_builds", &nb);
			ls1hip_get_option(_ctx, "verlet_steps", &ns);
			global_log->info() << "LinkedCellsHip: snapshot extent x [" << lo[0] << ", " << hi[0] << "] y [" << lo[1] << ", " << hi[1] << "] z [" << lo[2]
							   << ", " << hi[2] << "] box x [" << _mirror.getBoundingBoxMin(0) << ", " << _mirror.getBoundingBoxMax(0) << ") builds " << nb
							   << " list steps " << ns << std::endl;
		}
		if (!strays.empty() || taken || getenv("LS1HIP_LOG_SNAPSHOTS"))
			global_log->info() << "LinkedCellsHip: snapshot at step " << global_simulation->getSimulationStep() << ": " << strays.size()
							   << " molecule(s) awaiting migration handed over, " << taken << " taken into this rank's box" << std::endl;
	}
	mirror().addParticles(mols);
	mirror().updateMoleculeCaches();
	_mirrorFresh = true;
	_hostDirty = false;
}

// ---- integrator -------------------------------------------------------------------------------------------------------------
static LinkedCellsHip* device_container(ParticleContainer* c) {
	LinkedCellsHip* p = dynamic_cast<LinkedCellsHip*>(c);
	if (!p || !p->deviceReady()) {
		global_log->error() << "LeapfrogHip needs the device container (LinkedCellsHip) with its molecules uploaded" << std::endl;
		Simulation::exit(684);
	}
	return p;
}

void LeapfrogHip::readXML(XMLfileUnits& xmlconfig) {
	_timestepLength = 0;
	xmlconfig.getNodeValueReduced("timestep", _timestepLength);
	global_log->info() << "Timestep: " << _timestepLength << " (LeapfrogHip: integration on the device)" << std::endl;
}

void LeapfrogHip::init() { _state = STATE_POST_FORCE_CALCULATION; }

void LeapfrogHip::eventNewTimestep(ParticleContainer* moleculeContainer, Domain* domain) {
	ProfScope prof_("eventNewTimestep");
	if (_state != STATE_POST_FORCE_CALCULATION) return;
	LinkedCellsHip* cont = device_container(moleculeContainer);
	ls1hip_ctx* ctx = cont->context();
	int rc;
	if (_haveBeta && !domain->NVE()) {
		// VelocityScalingThermostat::apply of the step just finished (Simulation.cpp:1108-1131), global thermostat — folded into
		// the same pass over the molecules as transition3to1 + transition1to2 (FullMolecule::upd_preF, Leapfrog.cpp:48-64)
		if (domain->severalThermostats()) {
			// componentwise branch of VelocityScalingThermostat::apply (VelocityScalingThermostat.cpp:45-69) with the factors the
			// driver hands it (Simulation.cpp:1111-1127): one pair per component = that of the component's thermostat
			std::vector<Component>& comps = *(global_simulation->getEnsemble()->getComponents());
			std::vector<double> bt(comps.size(), 1.), br(comps.size(), 1.);
			for (size_t cid = 0; cid < comps.size(); ++cid) {
				const int th = domain->getThermostat((int)cid);
				for (int d = 0; d < 3; ++d)
					if (domain->getThermostatDirectedVelocity(th, d) != 0.) {
						global_log->error() << "LeapfrogHip: thermostats with a directed velocity are not available on the device path" << std::endl;
						Simulation::exit(685);
					}
				bt[cid] = domain->getGlobalBetaTrans(th);
				br[cid] = domain->getGlobalBetaRot(th);
			}
			if ((rc = ls1hip_scale_kick_drift_components(ctx, (int)comps.size(), bt.data(), br.data(), _timestepLength))) {
				global_log->error() << "ls1hip_scale_kick_drift_components: " << ls1hip_last_error(ctx) << std::endl;
				Simulation::exit(686);
			}
		} else if ((rc = [&] { ProfScope p2("eNT.scale_kick_drift"); return ls1hip_scale_kick_drift(ctx, domain->getGlobalBetaTrans(), domain->getGlobalBetaRot(), _timestepLength); }())) {
			global_log->error() << "ls1hip_scale_kick_drift: " << ls1hip_last_error(ctx) << std::endl;
			Simulation::exit(686);
		}
	} else if ((rc = ls1hip_kick_drift(ctx, _timestepLength))) {
		global_log->error() << "ls1hip_kick_drift: " << ls1hip_last_error(ctx) << std::endl;
		Simulation::exit(687);
	}
	{
		ProfScope p3("eNT.deviceAdvanced");
		cont->deviceAdvanced();
	}
	cont->armPostForceKick(0.5 * _timestepLength);
	_state = STATE_PRE_FORCE_CALCULATION;
}

void LeapfrogHip::eventForcesCalculated(ParticleContainer* moleculeContainer, Domain* domain) {
	ProfScope prof_("eventForcesCalculated");
	if (_state != STATE_PRE_FORCE_CALCULATION) return;
	LinkedCellsHip* cont = device_container(moleculeContainer);
	ls1hip_ctx* ctx = cont->context();
	double summv2 = 0., sumIw2 = 0.;
	uint64_t n = 0, rotdof = 0;
	// transition2to3: upd_postF + the kinetic sums of thermostat 0 (Leapfrog.cpp:66-150); normally already queued behind the
	// traversal by the container (see LinkedCellsHip::deviceForces) and finished by now
	int rc = cont->takeQueuedKick() ? ls1hip_kinetic_sums(ctx, &summv2, &sumIw2, &n, &rotdof)
									: ls1hip_kick(ctx, 0.5 * _timestepLength, &summv2, &sumIw2, &n, &rotdof);
	if (rc) {
		global_log->error() << "ls1hip_kick: " << ls1hip_last_error(ctx) << std::endl;
		Simulation::exit(688);
	}
	if (domain->severalThermostats()) {
		// Leapfrog::transition2to3, several thermostats (Leapfrog.cpp:84-104, 142-146): sums keyed by the thermostat of the
		// molecule's component; thermostat 0 (the whole system) is formed by Domain::calculateGlobalValues from the others
		std::vector<Component>& comps = *(global_simulation->getEnsemble()->getComponents());
		const int nc = (int)comps.size();
		std::vector<double> mv2(nc), iw2(nc);
		std::vector<uint64_t> nn(nc), rd(nc);
		if ((rc = ls1hip_kinetic_sums_by_component(ctx, nc, mv2.data(), iw2.data(), nn.data(), rd.data()))) {
			global_log->error() << "ls1hip_kinetic_sums_by_component: " << ls1hip_last_error(ctx) << std::endl;
			Simulation::exit(688);
		}
		std::map<int, double> tmv2, tiw2;
		std::map<int, unsigned long> tn, trd;
		for (int cid = 0; cid < nc; ++cid) {
			if (nn[cid] == 0) continue;  // (the reference's maps only hold thermostats that own molecules on this rank)
			const int th = domain->getThermostat(cid);
			tmv2[th] += mv2[cid];
			tiw2[th] += iw2[cid];
			tn[th] += nn[cid];
			trd[th] += rd[cid];
		}
		for (auto& it : tmv2) {
			domain->setLocalSummv2(it.second, it.first);
			domain->setLocalSumIw2(tiw2[it.first], it.first);
			domain->setLocalNrotDOF(it.first, tn[it.first], trd[it.first]);
		}
	} else {
		domain->setLocalSummv2(summv2, 0);
		domain->setLocalSumIw2(sumIw2, 0);
		domain->setLocalNrotDOF(0, n, rotdof);
	}
	_haveBeta = true;
	_state = STATE_POST_FORCE_CALCULATION;
	++_stepsDone;
	cont->stepClosed();
	// lazily synced mirror: whoever iterates the container outside the driver's per-step host loops gets it refilled on demand
	// (LinkedCellsHip::ensureMirror); LS1HIP_MIRROR_SYNC_INTERVAL additionally refills every N steps
	long interval = 0;
	if (const char* e = getenv("LS1HIP_MIRROR_SYNC_INTERVAL")) interval = atol(e);
	Simulation* sim = global_simulation;
	const bool last = sim->getSimulationStep() >= sim->getNumTimesteps();
	if (last) {
		long on = 0, builds = 0, evals = 0;
		ls1hip_get_option(ctx, "verlet_lists", &on);
		ls1hip_get_option(ctx, "verlet_builds", &builds);
		ls1hip_get_option(ctx, "verlet_steps", &evals);
		g_prof.report();
		global_log->info() << "LinkedCellsHip: neighbour lists " << (on ? "on" : "off") << ": " << builds << " builds for " << evals
						   << " list force evaluations" << std::endl;
	}
	if (interval > 0 && _stepsDone % (unsigned long)interval == 0) cont->syncMirrorFromDevice();
}
