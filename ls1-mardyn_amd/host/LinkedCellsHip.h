// LinkedCellsHip.h — seam B: device-resident drop-ins for the reference's LinkedCells container and Leapfrog integrator,
// compiled AGAINST THE REFERENCE'S HEADERS and constructed by the UNMODIFIED Simulation.cpp.
//
//   class LinkedCellsHip : public ParticleContainer   (particleContainer/ParticleContainer.h:69-278, all 30 methods)
//   class LeapfrogHip    : public Integrator          (integrators/Integrator.h:32-83)
//
// How the unmodified driver picks them up (zero edits to the reference's sources): Simulation.cpp is compiled with
//     -include host/seam_b_register.h
// which includes the reference's own LinkedCells.h / Leapfrog.h first (so their later #include is a no-op) and then maps
// the class names used at the two construction sites (Simulation.cpp:177 `new Leapfrog()`, :422 `new LinkedCells()`) to
// the classes below.  The type strings of the XML config ("LinkedCells", "Leapfrog") are untouched; a maintainer who
// prefers an explicit registration adds the 3-line `else if(datastructuretype == "LinkedCellsHip")` branch instead
// (INTEGRATION.md).
//
// State model.  The molecules live on the device (libls1hip, include/ls1hip.h).  The HOST MIRROR is a real reference
// LinkedCells object owned by the container: while the driver fills the container (phase-space readers, generators ->
// addParticle / addParticles / initCubicGrid) and during prepare_start it is authoritative and every ParticleContainer
// method behaves exactly as the reference's; the first update() uploads it.  Once the integrator has advanced the device
// state the mirror is EMPTIED (its iterators are valid and yield nothing), which is what turns the driver's per-step host
// loops — Simulation::updateForces (calcFM), DomainDecompBase::exchangeMolecules, VelocityScalingThermostat::apply — into
// no-ops without touching them: their work happens on the device (site reduction inside the force kernels, periodic
// wrap + halo copies in ls1hip_rebin / ls1hip_halo, velocity scaling by the beta factors the driver's own
// Domain::calculateGlobalValues computed).  OUTSIDE those per-step windows (eventNewTimestep .. advanceSimulationTime) a reader
// of the stale mirror — end-of-step plugins such as CheckpointWriter, the timed / final checkpoint, finishing plugins — makes
// the container refill it from the device first (lazily synced ON DEMAND, read-only snapshot): such a reader never sees an
// empty container.  INSIDE the windows every caller of iterator() / regionIterator() finds nothing — the driver's own loops by
// design, but also plugin hooks that run there (beforeForces, siteWiseForces, afterForces) and TemperatureControl's loops: plugins
// known to use those hooks stop the run with an explanation (LinkedCellsHip.cpp: check_plugins_once) rather than silently doing
// nothing; the XML thermostat type TemperatureControl is not served by the device path (use VelocityScaling).
#pragma once
#include <array>
#include <future>
#include <string>
#include <variant>
#include <vector>

#include "integrators/Integrator.h"
#include "particleContainer/LinkedCells.h"
#include "particleContainer/ParticleContainer.h"

#include "ls1hip.h"

class Domain;
class DomainDecompHip;

class LinkedCellsHip : public ParticleContainer {
public:
	LinkedCellsHip();
	LinkedCellsHip(double bBoxMin[3], double bBoxMax[3], double cutoffRadius);
	~LinkedCellsHip() override;

	void readXML(XMLfileUnits& xmlconfig) override;
	bool rebuild(double bBoxMin[3], double bBoxMax[3]) override;
	void update() override;
	bool addParticle(Molecule& particle, bool inBoxCheckedAlready = false, bool checkWhetherDuplicate = false,
					 const bool& rebuildCaches = false) override;
	bool addHaloParticle(Molecule& particle, bool inBoxCheckedAlready = false, bool checkWhetherDuplicate = false,
						 const bool& rebuildCaches = false) override;
	void addParticles(std::vector<Molecule>& particles, bool checkWhetherDuplicate = false) override;
	void traverseCells(CellProcessor& cellProcessor) override;
	void traverseNonInnermostCells(CellProcessor& cellProcessor) override;
	void traversePartialInnermostCells(CellProcessor& cellProcessor, unsigned int stage, int stageCount) override;
	ParticleIterator iterator(ParticleIterator::Type t) override;
	RegionParticleIterator regionIterator(const double startCorner[3], const double endCorner[3],
										  ParticleIterator::Type t) override;
	unsigned long getNumberOfParticles() override;
	double getBoundingBoxMin(int dimension) const override { return _mirror.getBoundingBoxMin(dimension); }
	double getBoundingBoxMax(int dimension) const override { return _mirror.getBoundingBoxMax(dimension); }
	bool isInBoundingBox(double r[3]) const override { return _mirror.isInBoundingBox(r); }
	int getHaloWidthNumCells() override { return _mirror.getHaloWidthNumCells(); }
	void clear() override;
	void deleteOuterParticles() override;
	double get_halo_L(int index) const override { return _mirror.get_halo_L(index); }
	double getCutoff() const override { return _mirror.getCutoff(); }
	void setCutoff(double rc) override { _mirror.setCutoff(rc); }
	void deleteMolecule(ParticleIterator& moleculeIter, const bool& rebuildCaches) override;
	double getEnergy(ParticlePairsHandler* particlePairsHandler, Molecule* m1, CellProcessor& cellProcessor) override;
	void updateInnerMoleculeCaches() override;
	void updateBoundaryAndHaloMoleculeCaches() override;
	void updateMoleculeCaches() override;
	std::variant<ParticleIterator, SingleCellIterator<ParticleCell>> getMoleculeAtPosition(const double pos[3]) override;
	bool requiresForceExchange() const override { return false; }  // full shell: every rank computes complete forces
	unsigned long initCubicGrid(std::array<unsigned long, 3> numMoleculesPerDimension, std::array<double, 3> simBoxLength,
								size_t seed_offset) override;
	double* getCellLength() override { return _mirror.getCellLength(); }
	std::vector<unsigned long> getParticleCellStatistics() override;
	std::string getConfigurationAsString() override;
	// MemoryProfilable
	size_t getTotalSize() override;
	void printSubInfo(int offset) override;
	std::string getName() override { return "LinkedCellsHip"; }

	// ---- used by LeapfrogHip ------------------------------------------------------------------------------------------
	ls1hip_ctx* context() { return _ctx; }
	bool deviceReady() const { return _uploaded; }
	void deviceAdvanced();            // the integrator moved the molecules on the device: the mirror is stale -> emptied
	// refill the mirror with the device state (read-only snapshot for plugins / writers); applyPendingBeta: with the velocity
	// scaling of the finished step applied, which the device folds into its next kick + drift pass
	void syncMirrorFromDevice(bool applyPendingBeta = false);
	bool mirrorFresh() const { return _mirrorFresh; }
	void snapshotForReaders() { ensureMirror(); }  // collective entry (DomainDecompHip::assertDisjunctivity): refill a stale mirror now
	void stepClosed();                // eventForcesCalculated is through: only the thermostat's host loop follows in this step
	void exchangeAcrossRanks(DomainDecompHip& dd, Domain* domain);  // multi-rank: leaving molecules + halo copies through dd's transport
	void armPostForceKick(double dt_half) { _armedKick = dt_half; }  // the next complete traversal queues the kick behind itself
	bool takeQueuedKick() {
		const bool q = _kickQueued;
		_kickQueued = false;
		return q;
	}

private:
	void die(const char* what, int rc) const;
	bool inHostLoopWindow();          // the driver's per-step host loops (device did their work): a stale mirror stays empty
	void ensureMirror();              // any other reader of a stale mirror: refill from the device first
	void uploadFromMirror();
	void deviceForces(int which);

	LinkedCells _mirror;  // host mirror: the reference's own container (all host-side semantics)
	// LinkedCells::clear of a stale mirror runs on a helper thread (LinkedCellsHip::deviceAdvanced); everything that touches the
	// mirror's molecules goes through mirror(), which waits for it.  (Declared after _mirror: destroyed — i.e. joined — before it.)
	std::future<void> _clearJob;
	void joinClear();
	LinkedCells& mirror() {
		joinClear();
		return _mirror;
	}
	ls1hip_ctx* _ctx = nullptr;
	double _skin = 0.;          // neighbour-list skin handed to ls1hip_set_verlet (0: search every step)
	double _armedKick = 0.;     // dt / 2 of the integrator's post-force kick, queued behind the next traversal
	bool _kickQueued = false;   // ... and it has been queued
	bool _uploaded = false;     // the device holds the molecule set
	bool _mirrorFresh = true;   // the mirror holds the current molecule set
	unsigned long _stepIndex = 0;  // steps the integrator has started (eventNewTimestep)
	bool _rebuildStep = true;   // multi-rank list mode: this step re-bins, migrates and rebuilds the lists (decided by all ranks)
	bool _hostDirty = true;     // molecules were added / removed through the host interface since the last upload
	bool _multiRank = false;    // more than one rank (or the loopback rehearsal): update() only classifies, DomainDecompHip exchanges
	bool _overlap = true;       // multi-rank: the inner pass is queued ahead of the halo phase (LS1HIP_OVERLAP=0: one pass behind the exchange)
	bool _innerLaunched = false;  // ... and it has been, for the traversal about to come
	bool _pluginsChecked = false;
	bool _inExchange = false;   // between update() and updateMoleculeCaches(): the driver's exchangeMolecules window
	bool _stepOpen = false;     // between eventNewTimestep and eventForcesCalculated
	bool _quietArmed = false;   // after eventForcesCalculated until the driver advances the simulation time (thermostat loop)
	double _quietTime = 0.;
	bool _betaPending = false;  // thermostat factors of the finished step not yet applied on the device
};

class LeapfrogHip : public Integrator {
public:
	LeapfrogHip() = default;
	explicit LeapfrogHip(double timestepLength) : Integrator(timestepLength) {}
	~LeapfrogHip() override = default;
	void readXML(XMLfileUnits& xmlconfig) override;
	void init() override;
	void eventForcesCalculated(ParticleContainer* moleculeContainer, Domain* domain) override;
	void eventNewTimestep(ParticleContainer* moleculeContainer, Domain* domain) override;

private:
	enum { STATE_NEW_TIMESTEP = 1, STATE_PRE_FORCE_CALCULATION = 2, STATE_POST_FORCE_CALCULATION = 3 };
	int _state = STATE_POST_FORCE_CALCULATION;
	bool _haveBeta = false;  // Domain::calculateGlobalValues has produced scaling factors for the step just finished
	unsigned long _stepsDone = 0;
};
