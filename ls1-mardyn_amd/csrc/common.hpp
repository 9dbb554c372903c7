// common.hpp — context, device-resident data layout and kernel launch interface of libls1hip.
//
// Data layout in HBM (design stance of SURVEY.md §7: device-resident, cell-sorted global SoA):
//   owned molecules  [0, n_real)                sorted by inner/boundary cell (x-fastest linear cell index),
//   halo copies      [n_real, n_real + n_halo)  sorted by halo cell,
// one FP64 array per coordinate (x,y,z | vx,vy,vz | q0..q3 | Dx,Dy,Dz | Fx.. | Mx.. | Vix..), u64 id, i32 cid.
// cell_begin[c] / cell_end[c] give the slice of every cell of the full grid (incl. halo cells) in that index
// space.  Keeping the halo copies in their own segment is what lets the inner-cell force launch run while the
// halo segment is still being exchanged/sorted (the reference's NonBlockingMPIMultiStepHandler split).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ls1hip.h"
#include "grid.hpp"
#include "molpair.hpp"

// A library that contains an object compiled as a timing / layout VARIANT (tools/ab_variant.sh: -DLS1_BUILD_VARIANT) carries this
// symbol; the regular build does not define it anywhere, so the weak reference below resolves to null
// (ls1hip_get_option "build_variant", ls1hip_version).
#ifdef LS1_BUILD_VARIANT
extern "C" __attribute__((weak, visibility("default"))) const int ls1hip_variant_marker = 1;
#else
extern "C" {
extern __attribute__((weak)) const int ls1hip_variant_marker;
}
#endif

namespace ls1 {

// The 64-byte per-step record of the multi-site pair-stream force pass (kernels_force_mslist.hip): {x, y, z, component id, q0, q1,
// q2, q3 (normalised once per molecule and step: FullMolecule::setupSoACache normalises q before rotating, FullMolecule.cpp:720)}.  Written by k_msl_pack and by the rigid-body kick + drift passes (IntegArgs::pk) — ONE body, contraction off, so that the
// record is the same bits whichever pass wrote it.
__device__ __forceinline__ void msl_write_record(double* __restrict__ pk, uint32_t p, double x, double y, double z, double q0, double q1,
												  double q2, double q3, bool with_rot, int cid) {
#pragma clang fp contract(off)
	double w = 1., qx = 0., qy = 0., qz = 0.;
	if (with_rot) {
		const double inv = 1. / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
		w = q0 * inv; qx = q1 * inv; qy = q2 * inv; qz = q3 * inv;
	}
	double2* const rec = reinterpret_cast<double2*>(pk + (size_t)8 * p);
	rec[0] = make_double2(x, y);
	rec[1] = make_double2(z, __hiloint2double(0, cid));  // (the first 32 bytes are all the cutoff filter of the force pass reads)
	rec[2] = make_double2(w, qx);
	rec[3] = make_double2(qy, qz);
}

struct MolSoA {  // one set of state arrays (two sets exist: the rebin gathers from one into the other)
	double *x, *y, *z, *vx, *vy, *vz;
	double *q0, *q1, *q2, *q3, *Dx, *Dy, *Dz;  // only allocated when the component set rotates
	uint64_t* id;
	int32_t* cid;
};

struct ForceSoA {
	double *Fx, *Fy, *Fz, *Mx, *My, *Mz, *Vix, *Viy, *Viz;
};

struct HaloStage {  // unsorted halo copies (local images + imported records)
	double *x, *y, *z, *q0, *q1, *q2, *q3;
	uint64_t* id;
	int32_t* cid;
	uint32_t *key, *rank;
	uint32_t* src;  // owned index of the molecule a LOCAL image copies (0xffffffff for imported records)
	uint8_t* dir;   // direction (0..26) of the image: its shift is shift[dir]
};

// device-side counters / flags, one small struct in device memory (no host sync needed inside the step)
struct DevCounters {
	uint32_t n_real;       // owned molecules after the last rebin
	uint32_t n_halo;       // halo copies sorted into the halo segment
	uint32_t n_halo_staged;// halo copies staged (local images + imported), before sorting
	uint32_t n_stay;       // scratch: owned molecules that stay on this rank
	uint32_t exp_leave[27];
	uint32_t exp_halo[27];
	uint32_t err_lost;     // molecules that left the halo region
	uint32_t err_overflow; // capacity overflows
	uint32_t err_ingest;   // uploaded molecules outside the bounding box of this rank / with a wrong component id
	uint32_t err_ingest_first;  // index (in upload order) of one offending molecule
	uint32_t vl_irregular;      // list build: bricks without a complete set of stored lists (unstaged, overflow)
	uint32_t vl_local_excess;   // local rebuild criterion: some brick neighbourhood's pair-displacement bound exceeds skin / 2
	unsigned long long dist_checks, pairs_in_range;
	unsigned long long msl_total;  // multi-site pair lists: pairs (incl. block padding) of the last build
	double vmax2;          // list-reuse mode: max |v|^2 of the drift velocities of the current step
	double vl_bound;       // upper bound of the displacement of any molecule since the neighbour lists were built
	double vl_base;        // local criterion: the part of that bound added by UNFUSED drifts (they only know the global v_max)
	double macro[4];       // u6, uX, rf, virial of the current traversal
	double kin[2];         // sum m v^2, sum I w^2
	double beta[2];        // thermostat factors derived from kin (beta_trans, beta_rot)
	unsigned long long kin_n, kin_rotdof;
};

struct ForceParams {
	Grid g;
	const double *x, *y, *z, *q0, *q1, *q2, *q3;
	const int32_t* cid;
	const uint32_t *cell_begin, *cell_end, *ckey;
	double *Fx, *Fy, *Fz, *Mx, *My, *Mz, *Vix, *Viy, *Viz;
	const CompTable* ct;
	DevCounters* cnt;
	double* partials;  // [gridDim][4]
	uint32_t n_real_cap;  // launch bound (n_real is read from cnt for device-driven loops)
	int which;         // 0 all, 1 inner cells, 2 boundary cells, 3 all non-halo cells (seam A)
	uint32_t n_fixed;  // if != 0: number of molecules (overrides cnt->n_real; seam A)
	// multi-site pair-stream lists: molecule of every group slot (groups sorted by component inside windows of 8 groups), nullptr =
	// slot s is molecule s (one component)
	const uint32_t* msl_gm = nullptr;
	int msl_g = 128;  // molecules per group (msl_group_size)
	// fused pair-stream pass of rigid bodies (fuse = 1): the epilogue integrates the group's molecules (leapfrog_body.hpp) — x y z,
	// q, v (vx..), D in place, next step's records to msl_pk_out (the pass reads the other record buffer), max |v|^2 per group
	double *Dx = nullptr, *Dy = nullptr, *Dz = nullptr;
	double *ox = nullptr, *oy = nullptr, *oz = nullptr, *oq0 = nullptr, *oq1 = nullptr, *oq2 = nullptr, *oq3 = nullptr;  // x y z q, writable
	double* msl_pk_out = nullptr;
	double* msl_vmax = nullptr;
	int count_pairs;
	const uint32_t* brick_list;  // brick kernels: bricks of this pass (boundary), nullptr = all bricks / inner box
	uint32_t n_list;
	// neighbour-list kernels: where the per-brick data (record, lists, own indices) of the brick of launch slot k lives —
	// 0: at the brick's linear index; 1: at k (the launch runs over ALL bricks in the blocked order, and the data is stored in
	// that order: nothing at the head of a workgroup waits for the brick_list lookup); 2: at brick_did[k] (inner / boundary passes)
	int did_mode;
	const uint32_t* brick_did;
	// inner pass: the inner bricks form a box of bricks [lo, lo + n) per dimension — indexed arithmetically (a list lookup
	// puts a dependent global load in front of every workgroup: ~1 us x 105 rounds on the 2.4 ms inner pass)
	int inner_box, inner_lo[3], inner_n[3];
	// fused force -> kick -> kick -> drift (LJ fast path only): v updated in place, new positions written to Fx/Fy/Fz
	int fuse;
	double dt, dt_inv2m, mass;
	double *vx, *vy, *vz;
	// 1CLJ fast-path scalars
	double eps24, sig2, shift6, rc2;
	// neighbour-list reuse (kernels_force_verlet.hip): 0 off, 1 build the lists (no forces), 2 forces from the lists
	int vl_mode;
	double vl_rc2;       // (rc + skin)^2: list cutoff
	uint64_t* vl_words;  // [brick][tile][word][64 lanes]: 4 u16 LDS byte offsets per word
	uint8_t* vl_nw;      // [brick][tile]: words per lane in use (wave maximum), 0xff = overflow -> direct evaluation
	int precision;       // list force pass: 0 FP64, 1 FP32 pair arithmetic with FP64 sums (SPDP), 2 FP32 sums too (SPSP)
	uint32_t* vl_rec;    // [brick][verlet_record_words()]: region cell table + flags, written by the build
	uint16_t* vl_ii;     // [brick][tiles * 64]: LDS slot of every owned molecule
	uint32_t* vl_gi;     // [brick][tiles * 64]: global index of every owned molecule
	double* vl_top2;     // [brick][2] or nullptr: the two largest |v_drift|^2 of the brick's owned molecules (local rebuild criterion)
};

// inner / boundary brick lists of the LJ brick kernels for the current grid and brick shape (kernels_force_lj.hip)
struct BrickLists {
	int shape[3] = {0, 0, 0}, dims[3] = {0, 0, 0}, hw = 0;
	// [0] inner bricks, [1] boundary bricks, [2] all bricks (each in blocked order); [3] / [4]: for every entry of [0] / [1] its
	// position in [2] (the index of the brick's list data, see ForceParams::did_mode)
	uint32_t* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	uint32_t n[5] = {0, 0, 0, 0, 0};
};

struct Timer {
	std::vector<hipEvent_t> ev;  // start/stop pairs
	size_t used = 0;
	double total_ms = 0;
	uint64_t launches = 0;
};

}  // namespace ls1

struct ls1hip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	// second (high-priority) stream: while an inner-cell force pass (which = 1) is in flight on `stream`, the halo phase
	// (generation, packing, import, sort) runs here; ev_owned marks the state the halo phase may read (recorded on
	// `stream` just before the inner pass), ev_halo the populated halo the boundary pass (which = 2) waits for
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_owned = nullptr, ev_halo = nullptr;
	bool inner_in_flight = false;
	std::string err;
	// options
	long opt_force_kernel = LS1HIP_FK_AUTO, opt_cic = 1, opt_vi = 0, opt_det = 1, opt_count_pairs = 0, opt_lj_split = 0;
	ls1::BrickLists brick_lists;
	long last_force_kernel = 0;  // family of the last force launch: 1 generic, 2 LJ brick kernels, 3 multi-site brick kernel
	// ls1hip_run halo mode: 0 halo, then one pass over all cells (fastest on one GPU: measured 3.50 / 3.52 / 3.60 ms for
	// modes 0 / 1 / 2); 1 inner-cell pass first with the halo phase on the second stream meanwhile; 2 halo, inner, boundary
	long opt_overlap_halo = 0;
	long opt_precision = 0;       // list force pass: 0 FP64 | 1 SPDP | 2 SPSP (used while every brick is regular)
	bool vl_all_regular = false;  // result of the last list build
	long opt_fuse = 1;       // ls1hip_run: fuse force + integration between steps when possible
	// where the CURRENT positions live when not in mol[cur].x/y/z: the force arrays after a fused per-step pass, the second
	// position buffer (alt_*) in the list-reuse mode; nullptr = mol[cur]
	double *pos_x = nullptr, *pos_y = nullptr, *pos_z = nullptr;
	int fused_split = 0;     // a fused which=1 pass is waiting for its which=2 pass
	// model
	bool have_comp = false, have_domain = false;
	ls1::CompTable h_ct;
	ls1::CompTable* d_ct = nullptr;
	double rc = 0, rc_lj = 0;
	bool vl_top2_pending = false;  // the last force pass left per-brick drift-speed bounds for the coming unfused drift
	bool one_clj = false;
	// domain
	ls1::Grid g;
	double global_len[3] = {0, 0, 0};
	double dom_min[3] = {0, 0, 0}, dom_max[3] = {0, 0, 0};  // bounding box as passed to ls1hip_set_domain
	int my_rank = 0;
	int nbr[27];
	double shift[27][3];  // position shift applied to a copy/leaver sent in direction d
	bool has_remote = false;
	// molecules
	size_t cap_real = 0, cap_halo = 0;
	size_t n_real = 0;  // host view (valid when !dirty_counts)
	size_t n_halo = 0;
	uint32_t pending_in = 0;  // molecules in the source set during a split rebin (owned + imported)
	ls1::MolSoA mol[2];
	int cur = 0;
	ls1::ForceSoA frc;
	ls1::HaloStage hs;
	uint32_t *d_key = nullptr, *d_rank = nullptr, *d_perm = nullptr, *d_ckey = nullptr;
	uint64_t* d_idk = nullptr;
	uint32_t *d_count = nullptr, *d_cell_begin = nullptr, *d_cell_end = nullptr, *d_blocksum = nullptr;
	size_t cells_alloc = 0;
	uint32_t* d_shell = nullptr;
	uint32_t n_shell = 0;
	ls1::DevCounters* d_cnt = nullptr;
	ls1::DevCounters* h_cnt = nullptr;  // pinned mirror
	ls1::DevCounters* h_mark = nullptr;  // pinned mirror of ls1hip_traversal_mark
	hipEvent_t ev_mark = nullptr;
	double* d_partials = nullptr;
	double* d_stage = nullptr;  // [128][4] second-stage reduction buffer
	size_t partials_cap = 0;
	double* d_exp_leave = nullptr;  // per-direction slices, LS1HIP_LEAVING_DOUBLES per record
	double* d_exp_halo = nullptr;   // per-direction slices, LS1HIP_HALO_DOUBLES per record
	uint32_t exp_off_leave[28], exp_off_halo[28];
	bool forces_valid = false, halo_valid = false, binned = false;
	bool thermostat_on = false;
	double thermostat_T = 0.;
	// timing
	int timing_on = 0;  // 0 off, 1 all phases, 2 force passes only
	ls1::Timer t_force, t_integrate, t_rebin, t_halo, t_build;  // "build" = neighbour-list construction
	std::vector<void*> allocs;
	// neighbour-list reuse (ls1hip_set_verlet; kernels_force_verlet.hip)
	bool vl_on = false, vl_force = false;
	double vl_skin = 0.;
	double rc_list = 0.;  // rc + skin: cutoff of the cell grid, the halo shell and the lists
	uint64_t* d_vl_words = nullptr;
	uint8_t* d_vl_nw = nullptr;
	char* seam_a_buf = nullptr;  // persistent arena of ls1hip_soa_forces (seam A)
	size_t seam_a_cap = 0;
	uint32_t* d_vl_rec = nullptr;
	uint16_t* d_vl_ii = nullptr;
	uint32_t* d_vl_gi = nullptr;
	double* d_vl_top2 = nullptr;  // [brick][2] two largest |v_drift|^2 per brick (local rebuild criterion)
	double* d_vl_acc = nullptr;   // [brick] accumulated pair-displacement bound of the brick's neighbourhood
	long opt_local_rebuild = 1;   // single periodic domain, fused FP64 list passes: rebuild by the local criterion
	size_t vl_words_cap = 0, vl_tiles_cap = 0;
	double *alt_x = nullptr, *alt_y = nullptr, *alt_z = nullptr;  // second position buffer (owned + halo segment)
	bool vl_valid = false;       // the lists match the current binning and the displacement bound is tracked on the device
	uint32_t *d_halo_src = nullptr;
	uint8_t* d_halo_dir = nullptr;
	uint32_t *d_exp_halo_src = nullptr, *d_imp_slot = nullptr, *d_s2s = nullptr;
	double* d_exp_refresh = nullptr;  // 3 doubles per export slot
	uint32_t vl_exp_counts[27] = {0};  // halo export counts of the exchange the lists were built from
	uint32_t vl_imp_total = 0;         // halo records imported in that exchange
	uint32_t halo_import_at = 0;      // halo records imported since the last ls1hip_halo (build) / ls1hip_halo_refresh
	bool vl_ready = false;            // lists built for the current binning (piecewise / multi-rank list mode)
	bool vl_fresh = false;            // no fused list pass since the build: the displacement bound restarts with the next one
	bool vl_bound_pending = false;    // a fused list pass has published a rebuild flag that was not read yet
	volatile uint32_t* h_flag = nullptr;  // host-visible {seq << 1 | rebuild needed}, written by the step's last reduction
	uint32_t* d_flag = nullptr;
	uint32_t vl_seq = 0;
	unsigned long vl_builds = 0, vl_steps = 0;
	// multi-site neighbour lists (kernels_force_mslist.hip): per group of 128 owned molecules one block of molecule pairs
	uint32_t *d_msl_cnt = nullptr, *d_msl_off = nullptr, *d_msl_j = nullptr;
	uint8_t* d_msl_il = nullptr;
	uint32_t* d_msl_scratch = nullptr;  // [msl_capture_cap()][msl_stride]: hits captured by the count kernel
	uint16_t* d_msl_mcnt = nullptr;     // [msl_stride]: hits per molecule
	uint32_t* d_msl_gm = nullptr;       // [groups * 128]: molecule of every group slot (k_msl_groups; several components only)
	double* d_msl_pk2 = nullptr;        // the record buffer a fused pass writes (the two trade places after it)
	double* d_msl_pk = nullptr;         // [msl_stride][8]: packed per-step state of the owned molecules (k_msl_pack)
	bool msl_pk_fresh = false;          // ... and it holds the CURRENT positions / orientations (written by the last kick + drift pass)
	size_t msl_stride = 0;
	size_t msl_groups_cap = 0, msl_pairs_cap = 0;
	unsigned long long msl_pairs = 0;  // pairs of the current lists (incl. padding)
	double* d_shift27 = nullptr;        // device copy of shift[27][3]
	// per-step globals of ls1hip_run (ls1hip_run_log): rows of 6 doubles, written by the reduction kernels
	double* d_steplog = nullptr;
	double *log_row = nullptr, *log_row_kin = nullptr;  // rows the next force / kinetic reduction refreshes (null outside ls1hip_run)
	size_t steplog_steps = 0;
	// streaming upload (ls1hip_upload_begin / _chunk / _records / _end)
	bool ingest_open = false;
	size_t ingest_total = 0, ingest_at = 0;
	void* d_ingest = nullptr;  // device staging of one chunk
	size_t ingest_bytes = 0;
};

namespace ls1 {

// ---- kernel launchers (definitions in kernels_*.hip) --------------------------------------------------------------
struct RebinArgs {
	Grid g;
	MolSoA src, dst;
	bool has_rot;
	uint32_t *key, *rank, *perm, *ckey, *count, *cell_begin, *cell_end, *blocksum;
	uint64_t* idk;  // molecule ids in slot order (canonical in-cell ordering)
	DevCounters* cnt;
	uint32_t n_in;      // molecules in src (launch bound)
	int nbr[27];
	int my_rank;
	double shift[27][3];
	double* exp_leave;
	uint32_t exp_off[28];  // record offsets of each direction's slice in exp_leave (exp_off[27] = total)
	uint32_t cap_real;
	int deterministic;
};
void launch_rebin_classify(const RebinArgs& a, hipStream_t s);
void launch_rebin_sort_gather(const RebinArgs& a, hipStream_t s);

struct HaloArgs {
	Grid g;
	MolSoA mol;  // current set; halo segment is written at [n_real, ...)
	HaloStage hs;
	uint32_t* hsrc;  // per sorted halo copy: owned index of its source molecule / direction of its shift (list-reuse
	uint8_t* hdir;   // mode refreshes the halo positions from these instead of regenerating the images)
	// multi-rank list-reuse: what a refresh needs to route positions without re-sorting
	uint32_t* exp_src;   // per export slot (layout of exp_halo): owned index of the exported molecule
	uint32_t* imp_slot;  // per received record (arrival order over the imports of this exchange): its staging slot
	uint32_t imp_at;     // records received before this import call
	uint32_t* s2s;       // per staging slot: index of the copy inside the sorted halo segment
	bool has_rot;
	uint32_t *perm, *count, *cell_begin, *cell_end, *blocksum;
	uint64_t* idk;  // ids of the staged halo copies in slot order (canonical in-cell order by counting, as in k_gather)
	const uint32_t* shell;  // owned cells within 2*hw of a face (the only ones that can feed the halo)
	uint32_t nshell;
	DevCounters* cnt;
	uint32_t n_real_cap, cap_halo;
	int nbr[27];
	int my_rank;
	double shift[27][3];
	double rc;
	double* exp_halo;
	uint32_t exp_off[28];
	int deterministic;
};
void launch_halo_generate(const HaloArgs& a, hipStream_t s);
void launch_halo_import(const HaloArgs& a, const double* dev_records, uint32_t n, hipStream_t s);
void launch_halo_finalize(const HaloArgs& a, hipStream_t s);
// positions of the halo copies recomputed from their source molecules: dst[n_real + k] = src[hsrc[k]] + shift[hdir[k]]
void launch_halo_refresh(const HaloArgs& a, const double* sx, const double* sy, const double* sz, double* dx, double* dy,
						 double* dz, hipStream_t s);
// positions (receiver frame) of the molecules exported as halo copies at build time -> out[slot * 3 ..], export slot layout
void launch_refresh_pack(const HaloArgs& a, const double* sx, const double* sy, const double* sz, double* out, hipStream_t s);
// received refresh records (3 doubles each, build-time arrival order) -> halo segment of the position buffer dx / dy / dz
void launch_refresh_import(const HaloArgs& a, const double* rec, uint32_t n, double* dx, double* dy, double* dz, hipStream_t s);
void launch_leave_import(const RebinArgs& a, const double* dev_records, uint32_t n, uint32_t at, hipStream_t s);
void launch_pack_copy(double* dst, const double* src, uint32_t ndoubles, hipStream_t s);
// up to 27 (source offset -> destination offset) runs of doubles copied by one launch; dst_off is ascending
struct PackSegments {
	int n;
	uint64_t total;
	uint64_t src_off[27], dst_off[27];
};
void launch_pack_segments(const PackSegments& seg, const double* src, double* dst, hipStream_t s);

// device-side ingest / egress (kernels_ingest.hip)
struct IngestArgs {
	MolSoA dst;
	bool has_rot;
	int ncomp;
	double bmin[3], bmax[3];
	DevCounters* cnt;
	uint32_t at;     // first destination index
	uint32_t first;  // index of the chunk's first molecule in upload order (error reporting)
	uint32_t n;
};
void launch_ingest_aos(const IngestArgs& a, const uint64_t* id, const int32_t* cid, const double* r, const double* v,
					   const double* q, const double* D, hipStream_t s);
void launch_ingest_records(const IngestArgs& a, const void* rec, int fmt, hipStream_t s);
struct EgressArgs {
	MolSoA src;
	const double *x, *y, *z;  // current positions (may differ from src.x/y/z in the fused / list-reuse modes)
	bool has_rot;
	int periodic[3];
	double bmin[3], bmax[3], len[3];
	uint32_t first, n;
};
void launch_egress_records(const EgressArgs& a, void* rec, hipStream_t s);

void launch_force_generic(const ForceParams& p, bool one_clj, bool with_vi, bool has_rot, hipStream_t s, uint32_t* nblocks,
						  double expected_neighbours = 0.);
// LDS-tiled 1CLJ kernel (kernels_force_lj.hip); returns false if it cannot handle the configuration
bool launch_force_lj(const ForceParams& p, hipStream_t s, uint32_t* nblocks, double* partials, size_t partials_cap,
					 int split, double mean_per_cell, BrickLists* bl);
// neighbour-list (Verlet) variant of the 1CLJ fast path (kernels_force_verlet.hip): p.vl_mode 1 builds the lists of ALL
// bricks, 2 evaluates the forces of the bricks of pass p.which from them
bool launch_force_verlet(const ForceParams& p, hipStream_t s, uint32_t* nblocks, size_t partials_cap, BrickLists* bl);
void verlet_geometry(const Grid& g, long* nbricks, size_t* words_per_brick, size_t* tiles_per_brick);
int verlet_region_capacity();  // molecules of a brick's region the list kernels can stage in LDS
int verlet_region_cells();
void verlet_brick_shape(int shape[3]);
int verlet_record_words();
// speed_factor: the speeds in top2 are multiplied by it (1: they are drift speeds; max(beta, 1) for the bounds a post-kick pass left;
// < 0: max(cnt->beta[0], 1) read on the device)
void launch_bound_local(const Grid& g, const double* top2, double* acc, DevCounters* cnt, double dt, double limit, hipStream_t s,
						double speed_factor = 1.);
long verlet_brick_count(const Grid& g);  // cells per brick edge of the list kernels
// brick-tiled multi-site kernel (kernels_force_ms.hip); returns false if it cannot handle the configuration
bool launch_force_ms(const ForceParams& p, bool with_vi, bool has_rot, bool one_component, hipStream_t s, uint32_t* nblocks,
					 size_t partials_cap, double mean_per_cell, BrickLists* bl);
// site kernel (kernels_force_sites.hip): LDS-resident tables, cached own sites, LPM lanes per molecule, launch-time brick shape
bool launch_force_sites(const ForceParams& p, const CompTable& hct, bool with_vi, hipStream_t s, uint32_t* nblocks,
						size_t partials_cap, double mean_per_cell, double mean_neighbours, BrickLists* bl);
// multi-site neighbour lists (kernels_force_mslist.hip): count + offsets, fill, force pass over the pair stream
int msl_group_size(bool lj_only, int ncomp);
uint32_t msl_groups(uint32_t n_real, int g);
int msl_capture_cap();  // hits per molecule the count kernel keeps for the fill kernel (scratch[k][stride])
void launch_msl_groups(const ForceParams& p, uint32_t* gm, int ncomp, hipStream_t s);
void launch_msl_count(const ForceParams& p, uint32_t* grp_cnt, uint32_t* off, uint32_t* scratch, uint16_t* mcnt, uint32_t stride,
					  hipStream_t s);
void launch_msl_fill(const ForceParams& p, const uint32_t* off, const uint32_t* hsrc, const uint8_t* hdir, uint32_t* out_j,
					 uint8_t* out_il, int ncomp, const uint32_t* scratch, const uint16_t* mcnt, uint32_t stride, hipStream_t s);
bool launch_force_ms_list(const ForceParams& p, bool has_rot, bool lj_only, bool linear, int ncomp, const uint32_t* off, const uint32_t* pj, const uint8_t* pil,
						  const double* shift27, double* pk, hipStream_t s, uint32_t* nblocks, size_t partials_cap, bool pk_fresh = false);
// kin_in_slot1: the partials' slot 1 carries sum m v^2 of a fused force + integration pass (goes to cnt->kin[0], not to
// the macroscopic sums); log (may be null): the step-log row {U_pot, virial, sum m v^2, sum I w^2, N, rotDOF} to refresh
struct ReduceMode {
	bool overwrite = false;     // first pass of a traversal: the sums start here
	bool kin_in_slot1 = false;  // slot 1 = sum m v^2 of a fused pass
	double target_T = 0.;       // ... and, if > 0, the thermostat factors are derived from it (Domain.cpp:225-240)
	double* log = nullptr;      // step-log row to refresh
	// list-reuse mode: slot 2 = max |v_drift|^2 (combined by max).  On the LAST pass of a step the displacement bound is
	// advanced by dt * sqrt(vmax2) (after being reset when the lists were rebuilt in this step) and the rebuild flag
	// {seq, bound > limit} is published to the host-visible word `flag`.
	bool vmax_in_slot2 = false, last_pass = false, lists_rebuilt = false;
	bool local_criterion = false;  // the rebuild flag comes from cnt->vl_local_excess (k_bound_local ran before the reduction)
	double dt = 0., limit = 0.;
	uint32_t seq = 0;
	volatile uint32_t* flag = nullptr;
};
void launch_force_reduce(DevCounters* cnt, const double* partials, uint32_t nblocks, double* stage, hipStream_t s,
						 const ReduceMode& m);
void launch_clear_macro(DevCounters* cnt, hipStream_t s);

struct IntegArgs {
	MolSoA mol;
	ForceSoA frc;
	const CompTable* ct;
	DevCounters* cnt;
	double* partials;
	uint32_t n_cap;
	bool has_rot;
	double dt;
	double* vmax_part;  // list mode: per-workgroup max |v_drift|^2 of a kick + drift pass (null otherwise)
	// kick + drift passes: velocity scaling of the thermostat applied to v / D as they are read (same arithmetic as a separate
	// k_scale pass before: v * beta, then the kick), 0 = off, 1 = factors below, 2 = factors from cnt->beta
	// 3 = per-component factors (component-wise thermostats: VelocityScalingThermostat::apply, componentwise branch)
	int pre_scale = 0;
	double pre_bt = 1., pre_br = 1.;
	double pre_bt_c[MAXC], pre_br_c[MAXC];
	// multi-site neighbour lists: a kick + drift pass of rotating molecules also writes the packed per-step record of the pair-stream
	// force pass (msl_write_record) — the separate k_msl_pack pass over the molecules (0.27 ms at 10^7) is then skipped
	double* pk = nullptr;
	int pk_ncomp = 1;
};
// kinetic sums per COMPONENT of the current velocities / angular momenta: out[c][4] = {sum m v^2, sum I w^2, N, rotational DOF}
// (the host folds components into thermostats: Leapfrog.cpp:84-104, Domain::getThermostat)
void launch_kin_by_component(const IntegArgs& a, int ncomp, double* scratch, double* out, hipStream_t s);
void launch_bound_update(DevCounters* cnt, const double* vmax_part, uint32_t nblocks, double dt, double limit, bool fresh, uint32_t seq,
						 volatile uint32_t* flag, hipStream_t s, bool local_criterion = false);
void launch_kick_drift(const IntegArgs& a, hipStream_t s);
void launch_kick(const IntegArgs& a, hipStream_t s, uint32_t* nblocks);
void launch_kick_then_kick_drift(const IntegArgs& a, hipStream_t s);
void launch_kin_reduce(DevCounters* cnt, const double* partials, uint32_t nblocks, hipStream_t s, double target_T,
					   double* log = nullptr);
void launch_scale(const IntegArgs& a, double beta_trans, double beta_rot, bool from_device, hipStream_t s);

}  // namespace ls1
