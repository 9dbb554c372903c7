"""Host-side mirror of the reference's plug-in interfaces for the linked-cell pair-force path.

Same class / method names, argument meaning and call order as the reference so that tests read like the
reference's own (``/root/reference/src/particleContainer/tests/LinkedCellsTest.cpp:511-600``):

    container.update(); decomp.balanceAndExchange(0., False, container, domain)
    container.updateMoleculeCaches(); container.traverseCells(cellProcessor)

Everything here is plumbing over the C ABI (``engine.DeviceEngine``); the compute is HIP.

Mirrored interfaces (reference file:line):
  ParticleContainer / LinkedCells   particleContainer/ParticleContainer.h:69-278, LinkedCells.cpp:243-356,564-628
  CellProcessor / VectorizedCellProcessor   adapter/CellProcessor.h:29-94, VectorizedCellProcessor.cpp:111-157
  Integrator / Leapfrog             integrators/Integrator.h:32-83, Leapfrog.cpp:35-150
  DomainDecompBase                  parallel/DomainDecompBase.cpp:51-86 (sequential periodic boundary)
  Domain (local macroscopic values) Domain.h:129-150, Domain.cpp:126-134
"""
from __future__ import annotations

import numpy as np

from .engine import DeviceEngine
from .inp import ComponentSet


class Domain:
    """The slice of the reference's Domain this path talks to (setLocal*/getLocal*)."""

    def __init__(self, global_length):
        self._globalLength = np.asarray(global_length, dtype=np.float64).copy()
        self._localUpot = 0.0
        self._localVirial = 0.0
        self._local2KETrans = {0: 0.0}
        self._local2KERot = {0: 0.0}
        self._localN = {0: 0}
        self._localRotDOF = {0: 0}

    def getGlobalLength(self, d):
        return float(self._globalLength[d])

    def setLocalUpot(self, u):
        self._localUpot = float(u)

    def getLocalUpot(self):
        return self._localUpot

    def setLocalVirial(self, v):
        self._localVirial = float(v)

    def getLocalVirial(self):
        return self._localVirial

    def setLocalSummv2(self, summv2, thermostat=0):
        self._local2KETrans[thermostat] = float(summv2)

    def setLocalSumIw2(self, sumIw2, thermostat=0):
        self._local2KERot[thermostat] = float(sumIw2)

    def setLocalNrotDOF(self, thermostat, N, rotDOF):
        self._localN[thermostat] = int(N)
        self._localRotDOF[thermostat] = int(rotDOF)

    def getLocalSummv2(self, thermostat=0):
        return self._local2KETrans[thermostat]

    # --- Domain::calculateGlobalValues, thermostat 0, single rank (Domain.cpp:151-240)
    def setGlobalTemperature(self, T):
        self._universalTargetTemperature = float(T)

    def calculateGlobalValues(self, domainDecomp=None, particleContainer=None, collectThermostatVelocities=True,
                              Tfactor=1.0):
        self._globalUpot = self._localUpot
        self._globalVirial = self._localVirial
        N, rotDOF = self._localN[0], self._localRotDOF[0]
        summv2 = self._local2KETrans[0]
        sumIw2 = self._local2KERot[0] if rotDOF > 0 else 0.0
        self._globalTemperature = (summv2 + sumIw2) / (3 * N + rotDOF) if N > 0 else 0.0
        Ti = Tfactor * getattr(self, "_universalTargetTemperature", 0.0)
        if Ti > 0.0 and N > 0:
            self._universalBTrans = (3.0 * N * Ti / summv2) ** 0.4
            self._universalBRot = 1.0 if sumIw2 == 0.0 else (rotDOF * Ti / sumIw2) ** 0.4
        else:
            self._universalBTrans = self._universalBRot = 1.0

    def getGlobalBetaTrans(self):
        return self._universalBTrans

    def getGlobalBetaRot(self):
        return self._universalBRot

    def getGlobalCurrentTemperature(self):
        return self._globalTemperature

    def getLocalSumIw2(self, thermostat=0):
        return self._local2KERot[thermostat]


class CellProcessor:
    """adapter/CellProcessor.h:29-94 — only the traversal-level hooks exist on the device path: the per-cell
    callbacks (processCell / processCellPair) are fused into one kernel launch per traversal."""

    def __init__(self, cutoffRadius, LJCutoffRadius):
        self._cutoffRadiusSquare = cutoffRadius * cutoffRadius
        self._LJCutoffRadiusSquare = LJCutoffRadius * LJCutoffRadius

    def getCutoffRadiusSquare(self):
        return self._cutoffRadiusSquare

    def getLJCutoffRadiusSquare(self):
        return self._LJCutoffRadiusSquare

    def initTraversal(self):
        raise NotImplementedError

    def endTraversal(self):
        raise NotImplementedError


class VectorizedCellProcessor(CellProcessor):
    """VectorizedCellProcessor(Domain&, cutoffRadius, LJcutoffRadius) — adapter/VectorizedCellProcessor.h:40-71."""

    def __init__(self, domain: Domain, cutoffRadius: float, LJcutoffRadius: float):
        super().__init__(cutoffRadius, LJcutoffRadius)
        self._domain = domain
        self._upot = 0.0
        self._virial = 0.0

    def initTraversal(self):
        self._upot = 0.0
        self._virial = 0.0

    def _accumulate(self, upot, virial):
        self._upot, self._virial = upot, virial

    def endTraversal(self):
        # VectorizedCellProcessor.cpp:155-156
        self._domain.setLocalVirial(self._virial)
        self._domain.setLocalUpot(self._upot)


class LinkedCells:
    """Device-resident replacement of the reference's LinkedCells (seam B of SURVEY.md 8b).

    LinkedCells(bBoxMin, bBoxMax, cutoffRadius) as in particleContainer/LinkedCells.h; the extra keyword
    arguments carry what the reference takes from global_simulation (component set, LJ cutoff, device)."""

    def __init__(self, bBoxMin, bBoxMax, cutoffRadius, *, components: ComponentSet, globalLength=None,
                 LJCutoffRadius=None, device: int = 0, cellsInCutoffRadius: int = 1, my_rank: int = 0,
                 neighbor_rank=None, periodic: bool = True, engine: DeviceEngine | None = None):
        self._bmin = np.asarray(bBoxMin, dtype=np.float64).copy()
        self._bmax = np.asarray(bBoxMax, dtype=np.float64).copy()
        self._cutoffRadius = float(cutoffRadius)
        self.engine = engine or DeviceEngine(device)
        self.engine.set_components(components, cutoffRadius, LJCutoffRadius)
        self.engine.set_option("cells_in_cutoff", cellsInCutoffRadius)
        gl = self._bmax if globalLength is None else np.asarray(globalLength, dtype=np.float64)
        self.engine.set_domain(gl, self._bmin, self._bmax, my_rank, neighbor_rank, periodic)
        self._components = components

    # --- ParticleContainer.h:108-130
    def addParticles(self, ids, cid, r, v, q=None, D=None):
        self.engine.upload(ids, cid, r, v, q, D)

    # --- LinkedCells::update, LinkedCells.cpp:243-302
    def update(self):
        self.engine.rebin()

    # --- LinkedCells::updateMoleculeCaches, LinkedCells.cpp:1054-1086: the device SoA *is* the cache
    def updateMoleculeCaches(self):
        pass

    # --- LinkedCells::traverseCells, LinkedCells.cpp:564-575
    def traverseCells(self, cellProcessor: VectorizedCellProcessor):
        cellProcessor.initTraversal()
        cellProcessor._accumulate(*self.engine.forces(0))
        cellProcessor.endTraversal()

    # --- comm/compute overlap split, LinkedCells.cpp:577-609 (traversePartialInnermostCells / NonInnermost)
    def traversePartialInnermostCells(self, cellProcessor, stage: int, stageCount: int):
        if stage == 0:
            cellProcessor.initTraversal()
            self.engine.forces(1, want_macro=False)

    def traverseNonInnermostCells(self, cellProcessor):
        cellProcessor._accumulate(*self.engine.forces(2))
        cellProcessor.endTraversal()

    # --- LinkedCells::deleteOuterParticles, LinkedCells.cpp:611-628: halo copies live in their own segment and
    # are rebuilt by the next exchange; nothing to delete.
    def deleteOuterParticles(self):
        pass

    def requiresForceExchange(self):
        return False  # full-shell: C08CellPairTraversal.h:35

    def getNumberOfParticles(self):
        return self.engine.count()[0]

    def getCutoff(self):
        return self._cutoffRadius

    def getBoundingBoxMin(self, d):
        return float(self._bmin[d])

    def getBoundingBoxMax(self, d):
        return float(self._bmax[d])

    def getCellLength(self):
        return self.engine.grid()[1]

    def get_halo_L(self, d):
        dims, clen, hw = self.engine.grid()
        return float(clen[d] * hw)

    # host views (the reference's iterator(ONLY_INNER_AND_BOUNDARY) read access)
    def molecules(self):
        return self.engine.download_state()

    def forces(self, with_vi=False):
        return self.engine.download_forces(with_vi)


class DomainDecompBase:
    """Sequential periodic boundary: parallel/DomainDecompBase.cpp:51-86."""

    def balanceAndExchange(self, lastTraversalTime, forceRebalancing, moleculeContainer: LinkedCells, domain):
        self.exchangeMolecules(moleculeContainer, domain)

    def exchangeMolecules(self, moleculeContainer: LinkedCells, domain):
        # leaving molecules were wrapped by update(); populate the halo layer with copies
        moleculeContainer.engine.halo()

    def getBoundingBoxMin(self, d, domain):
        return 0.0

    def getBoundingBoxMax(self, d, domain):
        return domain.getGlobalLength(d)


class Integrator:
    def __init__(self, timestepLength=0.0):
        self._timestepLength = float(timestepLength)

    def getTimestepLength(self):
        return self._timestepLength

    def setTimestepLength(self, dt):
        self._timestepLength = float(dt)


class Leapfrog(Integrator):
    """integrators/Leapfrog.cpp:17-150 — same three-state machine."""

    STATE_UNKNOWN, STATE_NEW_TIMESTEP, STATE_PRE_FORCE_CALCULATION, STATE_POST_FORCE_CALCULATION = 0, 1, 2, 3

    def __init__(self, timestepLength=0.0):
        super().__init__(timestepLength)
        self.init()

    def init(self):
        self._state = self.STATE_POST_FORCE_CALCULATION

    def eventNewTimestep(self, molCont: LinkedCells, domain: Domain):
        if self._state == self.STATE_POST_FORCE_CALCULATION:
            self._state = self.STATE_NEW_TIMESTEP  # transition3to1
            molCont.engine.kick_drift(self._timestepLength)  # transition1to2 -> upd_preF
            self._state = self.STATE_PRE_FORCE_CALCULATION

    def eventForcesCalculated(self, molCont: LinkedCells, domain: Domain):
        if self._state == self.STATE_PRE_FORCE_CALCULATION:
            summv2, sumIw2, N, rotDOF = molCont.engine.kick(0.5 * self._timestepLength)  # transition2to3 -> upd_postF
            domain.setLocalSummv2(summv2, 0)
            domain.setLocalSumIw2(sumIw2, 0)
            domain.setLocalNrotDOF(0, N, rotDOF)
            self._state = self.STATE_POST_FORCE_CALCULATION


class VelocityScalingThermostat:
    """thermostats/VelocityScalingThermostat.cpp:9-96, global (non component-wise) branch."""

    def __init__(self):
        self._globalBetaTrans = 1.0
        self._globalBetaRot = 1.0

    def setGlobalBetaTrans(self, beta):
        self._globalBetaTrans = float(beta)

    def setGlobalBetaRot(self, beta):
        self._globalBetaRot = float(beta)

    def apply(self, moleculeContainer: LinkedCells):
        moleculeContainer.engine.scale_velocities(self._globalBetaTrans, self._globalBetaRot)


def simulate(container: LinkedCells, decomp: DomainDecompBase, cellProcessor: VectorizedCellProcessor,
             integrator: Leapfrog, domain: Domain, nsteps: int, initial_forces: bool = True,
             thermostat: VelocityScalingThermostat | None = None):
    """The hot-path part of Simulation::prepare_start / simulate (Simulation.cpp:813-892, 979-1167), call for call."""
    if initial_forces:
        container.update()
        decomp.balanceAndExchange(1.0, False, container, domain)
        container.updateMoleculeCaches()
        container.traverseCells(cellProcessor)
        container.deleteOuterParticles()
    for _ in range(nsteps):
        integrator.eventNewTimestep(container, domain)
        container.update()
        decomp.balanceAndExchange(0.0, False, container, domain)
        container.updateMoleculeCaches()
        container.traverseCells(cellProcessor)
        container.deleteOuterParticles()
        integrator.eventForcesCalculated(container, domain)
        if thermostat is not None:  # Simulation.cpp:1099-1131
            domain.calculateGlobalValues(decomp, container, True, 1.0)
            thermostat.setGlobalBetaTrans(domain.getGlobalBetaTrans())
            thermostat.setGlobalBetaRot(domain.getGlobalBetaRot())
            thermostat.apply(container)
