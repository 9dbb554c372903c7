/*
 * ls1hip.h — C ABI of libls1hip, the MI355X-native linked-cell pair-force engine.
 *
 * This is the drop-in boundary for ONE hot path of ls1-MarDyn (reference: reinago/ls1-mardyn):
 *   LinkedCells traversal -> VectorizedCellProcessor pair kernel -> Leapfrog integrator (+ ghost-cell halo).
 * Every entry point names the reference interface it replaces (paths relative to /root/reference/src).
 * Host adapters (C++17 classes with the reference's ParticleContainer / CellProcessor / Integrator signatures,
 * or the Python mirror in ls1-mardyn_amd/) call ONLY these functions.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, FP64 only (the reference's MARDYN_DPDP build).
 *   - every function returns 0 on success, <0 on error (LS1HIP_E*); ls1hip_last_error() gives the text.
 *     (reference convention: log + Simulation::exit(code), Simulation.cpp:155-158 — adapters translate.)
 *   - the caller owns every host buffer; the library owns device memory and HIP streams; no pointer passed in
 *     is retained after the call returns.  `dev_*` arguments are DEVICE pointers (e.g. torch tensor data_ptr()).
 *   - entry points are main-thread only and internally asynchronous on the context's HIP streams
 *     (reference threading contract: SURVEY.md 8b).
 *   - molecule arrays are AoS [n][k] row-major exactly like the reference's Molecule fields:
 *     r[3], v[3], q[4]=(w,x,y,z), D[3] (angular momentum), F[3], M[3], Vi[3]; cid is 0-based.
 */
#ifndef LS1HIP_H_
#define LS1HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ls1hip_ctx ls1hip_ctx;

enum {
	LS1HIP_OK = 0,
	LS1HIP_EINVAL = -1,   /* bad argument / call order                                   */
	LS1HIP_EHIP = -2,     /* HIP runtime error                                           */
	LS1HIP_ENOMEM = -3,   /* device allocation failed / capacity exceeded                */
	LS1HIP_ELOST = -4,    /* a molecule left the halo region (reference: "particle lost")*/
	LS1HIP_ENODEV = -5    /* no usable gfx950 device                                     */
};

/* site table strides (doubles per site), same column order as the reference's .inp component block
 * (io/ASCIIReader.cpp:178-206) with the LJ shift already reduced to shift6 (molecules/Component.cpp:105-118) */
#define LS1HIP_LJ_STRIDE 7 /* x y z m eps sigma shift6 */
#define LS1HIP_CH_STRIDE 5 /* x y z m q                */
#define LS1HIP_DP_STRIDE 7 /* x y z ex ey ez absMy     */
#define LS1HIP_QP_STRIDE 7 /* x y z ex ey ez absQ      */

/* force-kernel variants (ls1hip_set_option "force_kernel") */
#define LS1HIP_FK_AUTO 0      /* fastest parity-green kernel for the component set           */
#define LS1HIP_FK_GENERIC 1   /* thread-per-molecule, global-memory neighbours (any density) */
#define LS1HIP_FK_LDS_LIST 2  /* brick-tiled, LDS-staged, per-lane neighbour lists (1CLJ)    */
#define LS1HIP_FK_MS_BRICK 3  /* multi-site, LDS-staged molecule pairs; bitwise == GENERIC (AUTO's choice for multi-site sets) */
#define LS1HIP_FK_MS_SITES 4  /* multi-site, LDS tables + cached own sites + FMA bodies; 1e-12 of GENERIC; on request only   */
#define LS1HIP_FK_NEIGHBOUR_LIST 5 /* reported by "last_force_kernel" only: forces from the stored neighbour lists (ls1hip_set_verlet) */

/* ---- lifetime ------------------------------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (one process per GPU; multi-GPU = one context per rank).
 * Replaces: construction of LinkedCells + VectorizedCellProcessor + Leapfrog in Simulation::readXML /
 * prepare_start (Simulation.cpp:411-455, 768-779, 168-186). */
int ls1hip_create(int device, ls1hip_ctx** out);
int ls1hip_destroy(ls1hip_ctx* ctx);
/* Text of the last error on this context (never NULL).  ctx may be NULL for creation errors. */
const char* ls1hip_last_error(const ls1hip_ctx* ctx);
/* Library version / build target string, e.g. "ls1hip 0.1 gfx950" ("... +variant": see option "build_variant"). */
const char* ls1hip_version(void);

/* Integer options: "force_kernel" (LS1HIP_FK_*), "cells_in_cutoff" (1|2, LinkedCells <cellsInCutoffRadius>,
 * particleContainer/LinkedCells.h:81-106), "compute_vi" (0|1 per-molecule virial Vi output),
 * "deterministic" (0|1 canonical in-cell order by molecule id),
 * "count_pairs" (0|1 tally molecule pairs / site interactions inside the cutoff for ls1hip_pair_stats — the
 * counters of adapter/FlopCounter.cpp:20-76; forces then use the generic kernel),
 * "build_variant" (read only: 0 = the regular library; 1 = the library contains an object built by tools/ab_variant.sh as a
 *   timing / layout variant of a kernel — such builds may compute WRONG forces by construction and are for same-box A/B timing only;
 *   ls1hip_version() then ends in "+variant"),
 * "last_force_kernel" (read only: kernel family of the last force launch — LS1HIP_FK_*: 1 generic, 2 single-centre LJ brick
 *   kernels, 3 multi-site brick kernel, 4 multi-site site kernel, 5 neighbour-list force pass; lets callers / tests see a fallback to the generic kernel),
 * "precision" (list force pass of the single-centre LJ path: 0 = FP64 (default), 1 = SPDP, 2 = SPSP — the reference's
 *   MARDYN_SPDP / MARDYN_SPSP build modes, vectorization/RealVec.h, RealAccumVecSPDP.h: pair arithmetic in FP32, sums in
 *   FP64 / FP32; molecule state, integration and reductions stay FP64; used while every brick of the last list build is
 *   regular, read-only "precision_in_use" tells),
 * "local_rebuild" (neighbour-list loop of a single periodic domain, fused passes: 1 (default) = the lists are rebuilt when
 * the pair-displacement bound of some brick neighbourhood exceeds skin / 2 — sum over the steps of dt * (s1 + s2) / 2, s1 >= s2
 * the two largest speeds among the molecules of the 27 bricks around a brick; 0 = global bound, sum of dt * v_max; both are
 * rigorous, the local one is never earlier),
 * "overlap_halo" (ls1hip_run: 0 = halo, then one traversal of all cells (default, fastest on a single GPU); 1 = inner
 *   cells first, the halo built on a second stream meanwhile, then the boundary cells — the order a transport-driven
 *   multi-rank loop uses; 2 = halo, inner cells, boundary cells on one stream),
 * "fuse_integration" (0|1, default 1: ls1hip_run lets the force pass do the integration between steps, see
 * ls1hip_forces_kick_drift), "can_fuse_integration" (read only), "can_fuse_rigid_lists" (read only: a single-component
 *   rigid multi-site set under neighbour lists — ls1hip_run lets the pair-stream list pass integrate its own molecules,
 *   ls1hip_forces_list with dt > 0 does it piecewise; bitwise the separate kick + kick + drift pass),
 * "lj_split" (variant of the single-centre LJ fast path; results are the same to rounding, only speed differs:
 *   0 = choose from the mean cell occupancy (default); 1 | 2 = list kernel with 1 | 2 lanes per molecule;
 *   4 = FP32 MFMA distance-tile pre-filter + exact FP64 evaluation, 1x4x4-cell bricks, 512 threads;
 *   5 = the same with 256 threads (larger staging area); 6 = 1x4x2-cell bricks (dense cells).
 *   4-6 need cells_in_cutoff = 1 and fall back to the list kernel otherwise). */
int ls1hip_set_option(ls1hip_ctx* ctx, const char* name, long value);
int ls1hip_get_option(const ls1hip_ctx* ctx, const char* name, long* value);

/* ---- model ---------------------------------------------------------------------------------------------------- */

/* Component set + cutoffs -> device parameter tables.
 * Replaces: Comp2Param::initialize (molecules/Comp2Param.cpp:10-187), the table build in
 * VectorizedCellProcessor::VectorizedCellProcessor (adapter/VectorizedCellProcessor.cpp:21-83) and
 * Ensemble::setComponentLookUpIDs (ensemble/EnsembleBase.cpp:88-101).
 * nlj/nc/nd/nq: [ncomp] site counts; lj/ch/dp/qp: flat site tables (component-major, strides above);
 * mass: [ncomp]; I: [ncomp][3] principal moments; mix: [ncomp*(ncomp-1)/2][2] = (xi, eta) for i<j in reader
 * order (io/ASCIIReader.cpp:221-230); eps_rf: reaction-field epsilon; rc / rc_lj: cutoffs. */
int ls1hip_set_components(ls1hip_ctx* ctx, int ncomp, const int* nlj, const int* nc, const int* nd, const int* nq,
						  const double* lj, const double* ch, const double* dp, const double* qp, const double* mass,
						  const double* I, const double* mix, double eps_rf, double rc, double rc_lj);

/* Rotational degrees of freedom per component as the reference counts them: Component::getRotationalDegreesOfFreedom
 * (molecules/Component.cpp:140-167 — from the moments of the SITE masses; the I line of an .inp, io/ASCIIReader.cpp:208-212, or
 * <momentsofinertia>, Component.cpp:88-97, may override the VALUES of the moments afterwards without changing this count).
 * Optional: ls1hip_set_components derives the count of non-zero moments of `I`, which is the same number unless such an override
 * turned a zero moment into a non-zero one.  Only the thermostat's degree-of-freedom sums use it (integrators/Leapfrog.cpp:100,126
 * -> Domain.cpp:204-240), never the motion.  Call after ls1hip_set_components. */
int ls1hip_set_rot_dof(ls1hip_ctx* ctx, int ncomp, const int* rot_dof);

/* Read back the derived LJ table ([ncenters][ncenters] each; ncenters = sum nlj) — for parity tests of the
 * parameter derivation.  Any pointer may be NULL. */
int ls1hip_get_lj_table(const ls1hip_ctx* ctx, int* ncenters, double* eps24, double* sig2, double* shift6);

/* Domain of THIS rank: bounding box [box_min, box_max) inside the global periodic box of edge global_len
 * (origin 0).  neighbor_rank[27]: rank owning the region in direction (sx,sy,sz) in {-1,0,1}^3, index
 * (sz+1)*9+(sy+1)*3+(sx+1); an entry equal to `my_rank` means "periodic image handled locally", -1 means
 * open boundary (no halo from that side).  For a single GPU pass all entries = my_rank (periodic) or -1 (open).
 * Replaces: LinkedCells::LinkedCells / rebuild (particleContainer/LinkedCells.cpp:42-109,136-204),
 * DomainDecompBase::getBoundingBoxMin/Max and DomainDecomposition grid setup
 * (parallel/DomainDecomposition.cpp:19-41,112-123). */
int ls1hip_set_domain(ls1hip_ctx* ctx, const double global_len[3], const double box_min[3], const double box_max[3],
					  int my_rank, const int neighbor_rank[27]);
/* Cell grid chosen for the domain: dims[3] incl. halo layers, cell_len[3]. */
int ls1hip_get_grid(const ls1hip_ctx* ctx, int dims[3], double cell_len[3], int* halo_width);

/* ---- molecules ------------------------------------------------------------------------------------------------ */

/* Replace the molecule set of this rank (all must lie inside [box_min, box_max)).
 * Replaces: ParticleContainer::addParticles (particleContainer/ParticleContainer.h:108-130).
 * q and D may be NULL (unit quaternion / zero angular momentum). */
int ls1hip_upload(ls1hip_ctx* ctx, size_t n, const uint64_t* id, const int32_t* cid, const double* r, const double* v,
				  const double* q, const double* D);
/* Streaming form of ls1hip_upload for phase spaces that should never exist twice on the host (the 10^8 restart case,
 * SURVEY.md 8f-3): announce an upper bound of the total (it sizes the device arrays), hand over any number of chunks,
 * finish; the molecules actually handed over become the molecule set.  Chunks are copied raw to the device
 * and transposed into the device SoA there (no host-side Molecule objects, no host-side SoA copies).
 * ls1hip_upload_records takes the records of the reference's binary checkpoint as they lie in the file:
 *   LS1HIP_REC_ICRVQD 116 B {id u64, cid u32 (1-based), r 3 f64, v 3 f64, q 4 f64, D 3 f64}
 *                     (FullMolecule::writeBinary, molecules/FullMolecule.cpp:451-473; Domain.cpp:572-595),
 *   LS1HIP_REC_ICRV    60 B {id, cid, r, v},  LS1HIP_REC_IRV 56 B {id, r, v}   (io/BinaryReader.cpp:103-108,179-213;
 *                     q = (1,0,0,0), D = 0, and cid = 1 for IRV).
 * Replaces: BinaryReader::readPhaseSpace -> ParticleContainer::addParticle (io/BinaryReader.cpp:140-260).
 * Errors (molecule outside the bounding box, wrong component id) are reported by ls1hip_upload_end. */
#define LS1HIP_REC_ICRVQD 0
#define LS1HIP_REC_ICRV 1
#define LS1HIP_REC_IRV 2
int ls1hip_upload_begin(ls1hip_ctx* ctx, size_t n_total);
int ls1hip_upload_chunk(ls1hip_ctx* ctx, size_t n, const uint64_t* id, const int32_t* cid, const double* r,
						const double* v, const double* q, const double* D);
int ls1hip_upload_records(ls1hip_ctx* ctx, size_t n, const void* records, int format);
/* ls1hip_upload_chunk for producers that already live on the GPU (generators, RCCL receive buffers): the arrays are
 * DEVICE pointers on the context's device, complete when the call is made (synchronise the producing stream first);
 * they may be released when the call returns. */
int ls1hip_upload_chunk_device(ls1hip_ctx* ctx, size_t n, const uint64_t* dev_id, const int32_t* dev_cid,
							   const double* dev_r, const double* dev_v, const double* dev_q, const double* dev_D);
int ls1hip_upload_end(ls1hip_ctx* ctx);
/* Owned molecules [first, first + n) in device order as ICRVQD records (the payload of the reference's binary
 * checkpoint, written by ParticleContainer iteration + FullMolecule::writeBinary in io/CheckpointWriter /
 * Domain::writeCheckpoint, Domain.cpp:572-595); positions are wrapped into the box. */
int ls1hip_download_records(ls1hip_ctx* ctx, size_t first, size_t n, void* records);

/* Number of molecules owned by this rank / halo copies currently held. */
int ls1hip_count(const ls1hip_ctx* ctx, size_t* n_owned, size_t* n_halo);

/* Download owned molecules in device (cell-sorted) order.  Any pointer may be NULL.
 * Replaces: ParticleContainer::iterator(ONLY_INNER_AND_BOUNDARY) read access (ParticleContainer.h:150-170). */
int ls1hip_download_state(ls1hip_ctx* ctx, size_t cap, uint64_t* id, int32_t* cid, double* r, double* v, double* q,
						  double* D);
/* Forces of the last ls1hip_forces call, same order as ls1hip_download_state.
 * Replaces: Molecule::F / M / Vi after Simulation::updateForces -> calcFM (Simulation.cpp:752-762). */
int ls1hip_download_forces(ls1hip_ctx* ctx, size_t cap, double* F, double* M, double* Vi);

/* ---- the time step, piecewise (mirrors Simulation::simulate, Simulation.cpp:995-1099) ----------------------- */

/* Leapfrog::eventNewTimestep -> FullMolecule::upd_preF (integrators/Leapfrog.cpp:48-64,
 * molecules/FullMolecule.cpp:334-364): v += dt/2m F; r += dt v; quaternion / L half steps. */
int ls1hip_kick_drift(ls1hip_ctx* ctx, double dt);
/* ls1hip_scale_velocities(beta_trans, beta_rot) followed by ls1hip_kick_drift(dt) in ONE pass over the molecules (what
 * Simulation::simulate does between two steps of an NVT run: VelocityScalingThermostat::apply, Simulation.cpp:1108-1131, then
 * Integrator::eventNewTimestep); the arithmetic is that of the two separate calls, operation for operation. */
int ls1hip_scale_kick_drift(ls1hip_ctx* ctx, double beta_trans, double beta_rot, double dt);

/* LinkedCells::update (LinkedCells.cpp:243-356): re-sort owned molecules into cells; molecules that left the box
 * through a side whose neighbor_rank is this rank are wrapped (DomainDecompBase::handleDomainLeavingParticles,
 * parallel/DomainDecompBase.cpp:174-225); molecules leaving towards other ranks are packed for export
 * (see ls1hip_export_*). */
int ls1hip_rebin(ls1hip_ctx* ctx);

/* Halo population for all directions handled locally (DomainDecompBase::populateHaloLayerWithCopies,
 * parallel/DomainDecompBase.cpp:293-348) and packing of halo copies for remote directions; then
 * LinkedCells::updateMoleculeCaches (LinkedCells.cpp:1054-1086: site expansion, F/M/Vi cleared). */
int ls1hip_halo(ls1hip_ctx* ctx);

/* LinkedCells::traverseCells(VectorizedCellProcessor) + Simulation::updateForces (LinkedCells.cpp:564-575,
 * VectorizedCellProcessor.cpp:111-157,796-2821; FullMolecule::calcFM molecules/FullMolecule.cpp:526-629).
 * upot / virial = the values the reference passes to Domain::setLocalUpot / setLocalVirial
 * (VectorizedCellProcessor.cpp:155-156) for THIS rank; each pair that crosses a rank/periodic boundary is
 * counted half on each side, so the sum over ranks equals the reference's global value.
 * which: 0 = all cells, 1 = inner cells only (no halo cell in the neighbourhood;
 * LinkedCells::traversePartialInnermostCells), 2 = the remaining (boundary) cells
 * (LinkedCells::traverseNonInnermostCells) — the reference's comm/compute overlap split
 * (parallel/NonBlockingMPIMultiStepHandler.cpp:30-97).  Results of 1 then 2 accumulate. */
int ls1hip_forces(ls1hip_ctx* ctx, int which, double* upot, double* virial);

/* Leapfrog::eventForcesCalculated -> FullMolecule::upd_postF (Leapfrog.cpp:66-150, FullMolecule.cpp:366-389):
 * v += dt_half/m F; L += dt_half M; returns sum m v^2, sum I w^2, N, rotational DOF (thermostat 0). */
int ls1hip_kick(ls1hip_ctx* ctx, double dt_half, double* summv2, double* sumIw2, uint64_t* n, uint64_t* rot_dof);
/* The kinetic sums of the last ls1hip_kick, for callers that queued it with NULL outputs (asynchronously) and fetch later. */
int ls1hip_kinetic_sums(ls1hip_ctx* ctx, double* summv2, double* sumIw2, uint64_t* n, uint64_t* rot_dof);
/* Component-wise thermostats (Domain::severalThermostats()): the sums of Leapfrog::transition2to3 (integrators/Leapfrog.cpp:84-104)
 * per COMPONENT — arrays of ncomp entries: sum m v^2, sum I w^2, N, rotational DOF of the component's owned molecules, from the
 * current velocities (call after ls1hip_kick).  The caller folds components into thermostats (Domain::getThermostat). */
int ls1hip_kinetic_sums_by_component(ls1hip_ctx* ctx, int ncomp, double* summv2, double* sumIw2, uint64_t* n, uint64_t* rot_dof);
/* Overlap of host and device work around a traversal: ls1hip_forces / ls1hip_forces_list with upot = virial = NULL only queue
 * the kernels.  ls1hip_traversal_mark queues a copy of the traversal's sums behind them; ls1hip_traversal_sums waits for THAT
 * copy only — whatever was queued after the mark (typically the post-force ls1hip_kick with NULL outputs) keeps running on
 * the device while the host goes on with U_pot and the virial (what VectorizedCellProcessor::endTraversal publishes). */
int ls1hip_traversal_mark(ls1hip_ctx* ctx);
int ls1hip_traversal_sums(ls1hip_ctx* ctx, double* upot, double* virial);

/* Force pass that consumes the forces at once (reduced-memory mode; the reference's counterpart is the RMM pair
 * VCP1CLJRMM::processCell* + LeapfrogRMM, particleContainer/adapter/VCP1CLJRMM.cpp:241-369, integrators/LeapfrogRMM.cpp):
 * ls1hip_forces(which) fused with the post-force kick of this step and the pre-force kick + drift of the NEXT step,
 * arithmetic identical to ls1hip_forces + ls1hip_kick_then_kick_drift(dt) (bitwise).  F is never stored (-48 B of
 * HBM traffic per molecule, no separate integrator pass); the advanced positions are picked up by the following
 * ls1hip_rebin.  Afterwards velocities are at the half step, forces are NOT available (ls1hip_download_forces fails
 * until the next ls1hip_forces).  which = 0, or 1 followed by 2 (overlap split), as for ls1hip_forces; U_pot / virial
 * as there.  Available when ls1hip_get_option("can_fuse_integration") is 1: single-centre LJ fast path, no per-molecule
 * virial, no device thermostat.  ls1hip_run uses it between steps unless option "fuse_integration" is 0. */
int ls1hip_forces_kick_drift(ls1hip_ctx* ctx, int which, double dt, double* upot, double* virial);

/* eventForcesCalculated of step n immediately followed by eventNewTimestep of step n+1 (Leapfrog.cpp:66-150 then
 * :48-64) in ONE pass over the molecules: v += dt/m F; r += dt v (and the rotational counterparts) — bitwise the same
 * as ls1hip_kick(dt/2) + ls1hip_kick_drift(dt), for steps whose kinetic sums are not needed (NVE, no output). */
int ls1hip_kick_then_kick_drift(ls1hip_ctx* ctx, double dt);
/* ls1hip_scale_kick_drift with one factor pair per COMPONENT: VelocityScalingThermostat::apply, componentwise branch
 * (thermostats/VelocityScalingThermostat.cpp:45-69; the driver sets the factors of a component's thermostat,
 * Simulation.cpp:1111-1127), folded into the pre-force kick + drift pass. */
int ls1hip_scale_kick_drift_components(ls1hip_ctx* ctx, int ncomp, const double* beta_trans, const double* beta_rot, double dt);

/* VelocityScalingThermostat::apply, global branch (thermostats/VelocityScalingThermostat.cpp:80-96): v *= beta_trans,
 * D *= beta_rot for every owned molecule. */
int ls1hip_scale_velocities(ls1hip_ctx* ctx, double beta_trans, double beta_rot);

/* Global velocity-scaling thermostat inside ls1hip_run: after every post-force kick the device computes
 * beta_trans = (3 N T / sum m v^2)^0.4, beta_rot = (rotDOF T / sum I w^2)^0.4 exactly as Domain::calculateGlobalValues
 * does for thermostat 0 (Domain.cpp:204-240) and applies them, without a host round trip.  enabled = 0 -> NVE.
 * (Single-rank domains; multi-rank hosts reduce the sums and call ls1hip_scale_velocities.) */
int ls1hip_set_thermostat(ls1hip_ctx* ctx, int enabled, double target_temperature);

/* Homogeneous long-range correction: the constants Homogeneous::init / calculateLongRange compute and
 * hand to Domain::setUpotCorr / setVirialCorr (longRange/Homogeneous.cpp:21-135; LJ tail integrals of Lustig 1988,
 * :137-180; dipole reaction-field self term).  Host arithmetic on the component tables of this context.
 * n_per_component[ncomp]: global molecule count per component; global_rho = N / V. */
int ls1hip_long_range_homogeneous(ls1hip_ctx* ctx, const uint64_t* n_per_component, double global_rho,
								  double* upot_corr, double* virial_corr);

/* Neighbour-list reuse for ls1hip_run (single-centre LJ fast path, single rank, one cell per cutoff): the cell grid, the
 * halo shell and per-molecule neighbour lists are built with rc + skin and kept for as many steps as no molecule can have
 * moved by more than skin / 2 — the fused force pass reports max |v| of every step, the device accumulates the
 * displacement bound sum(dt * vmax) and ls1hip_run re-bins / rebuilds exactly when it exceeds skin / 2 (no guessed
 * interval; forces, U_pot and virial are the per-step kernels' to rounding, every listed pair is re-tested against rc in
 * FP64).  Between rebuilds there is no search, no re-binning and no halo regeneration.  Must be called after
 * ls1hip_set_components and before ls1hip_set_domain.  Positions reported by the download calls are wrapped into the box.
 * Precedent in the reference: the Verlet-list containers behind particleContainer/AutoPasContainer.cpp:281-346
 * (verletSkinRadius / verletRebuildFrequency).  enabled = 1: used when the mean population of a brick's region fits the
 * LDS staging area of the list kernels (otherwise ls1hip_run keeps the per-step kernels); 2: always (bricks that do not fit
 * are evaluated from global memory: correct, slow — for tests).  Read-only options "verlet_lists", "verlet_builds", "verlet_steps",
 * "verlet_ready" (lists built and alive), "verlet_bound_pending" (a drift since the lists were built / last polled: ls1hip_verlet_poll
 * may be asked), "verlet_irregular_bricks", "list_kick_available" (ls1hip_forces_list_kick applies). */
int ls1hip_set_verlet(ls1hip_ctx* ctx, int enabled, double skin);

/* The list mode piecewise — what a transport-driven multi-rank loop calls between two rebuilds (ls1hip_run is the
 * single-rank loop built from the same pieces).  Between rebuilds no molecule changes its owner (it may sit up to
 * skin / 2 outside its rank's box) and every halo copy keeps its slot; per step only POSITIONS travel:
 *   rebuild step:  ls1hip_rebin -> leaving exchange -> ls1hip_halo -> halo exchange (kind 1) -> ls1hip_verlet_build
 *                  -> ls1hip_forces_list
 *   reuse step:    ls1hip_forces_list(which=1) [inner bricks, needs owned positions only] || ls1hip_halo_refresh ->
 *                  refresh exchange (kind 2: LS1HIP_REFRESH_DOUBLES = 3 doubles per record, the records of the build-time
 *                  halo exchange in the same order and number: no count exchange) -> ls1hip_forces_list(which=2)
 * ls1hip_forces_list: dt > 0 fuses the pass with the integration (as ls1hip_forces_kick_drift), dt = 0 leaves F.
 * ls1hip_verlet_poll: after a fused pass — has this rank's displacement bound exceeded skin / 2?  The ranks must agree on
 * a rebuild (reduce the flags with a maximum).
 * Replaces, for these steps: DomainDecompBase::exchangeMolecules + LinkedCells::update (no migration, no re-sort) and the
 * HALO_COPIES message of NeighbourCommunicationScheme.cpp:115-136 by a position-only message. */
#define LS1HIP_REFRESH_DOUBLES 3
int ls1hip_verlet_build(ls1hip_ctx* ctx);
int ls1hip_halo_refresh(ls1hip_ctx* ctx);
/* Single-rank convenience: LinkedCells::update + DomainDecompBase::balanceAndExchange + updateMoleculeCaches in one call, list-aware.
 * While lists are alive and the displacement bound allows it (every drifting pass — the fused force pass as well as
 * ls1hip_kick_drift / ls1hip_kick_then_kick_drift — advances the bound on the device) only the halo positions are refreshed;
 * otherwise ls1hip_rebin + ls1hip_halo (+ ls1hip_verlet_build in list mode).  *rebuilt (may be NULL): 1 if it re-binned.
 * Follow with ls1hip_forces_list (list mode) or ls1hip_forces. */
int ls1hip_update(ls1hip_ctx* ctx, int* rebuilt);
int ls1hip_forces_list(ls1hip_ctx* ctx, int which, double dt, double* upot, double* virial);
/* List traversal of ALL cells that also does the step's post-force kick (Leapfrog::transition2to3 -> FullMolecule::upd_postF,
 * integrators/Leapfrog.cpp:66-150) in its epilogue: v += dt_half / m F with the kinetic sum of thermostat 0 (fetch it with
 * ls1hip_kinetic_sums), F is kept — the pass ls1hip_run takes on unfused steps (NVT, the last step), for drivers that call the
 * pieces themselves (the reference driver through LinkedCellsHip / LeapfrogHip: one pass over the molecules less per step).
 * Single-centre LJ list path, and the pair-stream list pass of ONE rigid multi-site component (v, D kicked, sum m v^2 and sum I w^2
 * of the step; same state as ls1hip_forces_list + ls1hip_kick bit for bit); read-only option "list_kick_available".  Otherwise
 * (several components, per-molecule virial) LS1HIP_EINVAL — use ls1hip_forces_list + ls1hip_kick. */
int ls1hip_forces_list_kick(ls1hip_ctx* ctx, double dt_half, double* upot, double* virial);
int ls1hip_verlet_poll(ls1hip_ctx* ctx, int* need_rebuild);

/* nsteps full time steps entirely on the device (single rank, all directions local), no host round trip
 * inside: the loop body of Simulation::simulate (Simulation.cpp:979-1167) for an NVE run without plugins.
 * out6 (may be NULL) = {upot, virial, summv2, sumIw2, N, rotDOF} of the LAST step. */
int ls1hip_run(ls1hip_ctx* ctx, double dt, unsigned long nsteps, double* out6);
/* The global values of EVERY step of the last ls1hip_run, one row {upot, virial, summv2, sumIw2, N, rotDOF} per step,
 * oldest first (the last 4096 steps are kept): what Leapfrog::transition2to3 hands to Domain::setLocalSummv2 / SumIw2
 * (integrators/Leapfrog.cpp:115-150) and VectorizedCellProcessor::endTraversal to setLocalUpot / Virial in every step,
 * i.e. the inputs of Domain::calculateGlobalValues (Domain.cpp:151-181, called per step at Simulation.cpp:1099-1103).
 * The rows are written by the reduction kernels on the device, no host round trip inside the run; in the fused mode
 * the kinetic sum comes out of the force pass's own epilogue.  Columns 2-5 are NaN for steps whose kinetic sums were not
 * computed (unfused NVE steps before the last).  rows may be NULL to query *nrows. */
int ls1hip_run_log(ls1hip_ctx* ctx, size_t cap_rows, double* rows, size_t* nrows);

/* ---- multi-GPU plumbing: packed buffers a host transport (RCCL via torch.distributed, MPI, ...) moves ---------
 * Wire formats (little endian, FP64): leaving molecule = 15 doubles {id(as bits), cid(as bits), r3, v3, q4, D3}
 * (the reference's 116-byte record, parallel/CommunicationBuffer.cpp:131-145, padded to 120); halo molecule =
 * 9 doubles {id, cid, r3, q4} (reference 68-byte record, CommunicationBuffer.cpp:167-176, padded to 72).
 * Positions are already shifted into the receiver's frame.
 * Replaces: CommunicationPartner::initSend / unpack (parallel/CommunicationPartner.cpp:139-227,266-389). */
#define LS1HIP_LEAVING_DOUBLES 15
#define LS1HIP_HALO_DOUBLES 9
/* kind: 0 = leaving molecules (valid after ls1hip_rebin), 1 = halo copies (valid after ls1hip_halo), 2 = position refresh of
 * the halo copies of the last list build (valid after ls1hip_halo_refresh; the counts are those of that build).
 * counts[27]: molecules packed per direction (0 for local / open directions). */
int ls1hip_export_counts(ls1hip_ctx* ctx, int kind, uint64_t counts[27]);
/* Copy the packed records of direction `dir` into the DEVICE buffer dev_buf (capacity in records). */
int ls1hip_export_pack(ls1hip_ctx* ctx, int kind, int dir, void* dev_buf, size_t cap);
/* The same for a list of directions, records back to back in the order given: one message per PEER (all directions
 * that point at the same neighbour rank), one stream synchronisation per message instead of one per direction.
 * Replaces the per-neighbour send loop of DirectNeighbourCommunicationScheme::initExchangeMoleculesMPI
 * (parallel/NeighbourCommunicationScheme.cpp:115-136). */
int ls1hip_export_pack_dirs(ls1hip_ctx* ctx, int kind, const int* dirs, int ndirs, void* dev_buf, size_t cap);
/* Append `n` received records from DEVICE buffer dev_buf: kind 0 -> owned molecules (then call ls1hip_rebin
 * again is NOT needed: they are binned on arrival), kind 1 -> halo copies.  The call is asynchronous: dev_buf is read
 * on the engine's stream and must stay valid until ls1hip_import_done(kind) returns.  Call ls1hip_import_done(kind)
 * after the last import of a kind. */
int ls1hip_import(ls1hip_ctx* ctx, int kind, const void* dev_buf, size_t n);
int ls1hip_import_done(ls1hip_ctx* ctx, int kind);

/* ---- seam A: CellProcessor-level drop-in (no container replacement) -------------------------------------------
 * One call = one complete traversal over the cells the reference's LinkedCells holds, given as a flat cell-major
 * molecule list.  Replaces the body of VectorizedCellProcessor::processCell / processCellPair / endTraversal
 * (adapter/VectorizedCellProcessor.cpp:124-157,2734-2821) when the adapter batches at endTraversal
 * (seam A).  cell_dims[3]: the reference's _cellsPerDimension incl. halo; cell_start[ncells+1]:
 * first molecule of each cell in the flat arrays; r/q/cid: molecule data incl. halo molecules;
 * outputs F/M/Vi [n][3] (halo entries are written as 0) and upot/virial with the reference's macroscopic rule. */
int ls1hip_soa_forces(ls1hip_ctx* ctx, const int cell_dims[3], const uint32_t* cell_start, size_t n, const double* r,
					  const double* q, const int32_t* cid, double* F, double* M, double* Vi, double* upot,
					  double* virial);

/* ---- measurement ---------------------------------------------------------------------------------------------- */

/* Device time (ms, HIP events on the context's compute stream) and launch count accumulated per kernel class
 * since the last reset: names[] in {"force","integrate","rebin","halo","build" (neighbour-list construction)}. */
int ls1hip_timing(ls1hip_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);
int ls1hip_timing_reset(ls1hip_ctx* ctx);
/* Per-launch HIP-event timing: 0 = off, 1 = every phase, 2 = force passes only (each timed scope puts two event markers
 * into the stream, ~10 us of device idle per scope: a measured run that needs the force-kernel duration only uses 2). */
int ls1hip_timing_enable(ls1hip_ctx* ctx, int on);
/* Pair statistics of the last force call: distance checks and in-range molecule pairs (as FlopCounter counts
 * them, adapter/FlopCounter.cpp:20-76); requires option "count_pairs"=1. */
int ls1hip_pair_stats(ls1hip_ctx* ctx, uint64_t* dist_checks, uint64_t* pairs_in_range);

#ifdef __cplusplus
}
#endif
#endif /* LS1HIP_H_ */
