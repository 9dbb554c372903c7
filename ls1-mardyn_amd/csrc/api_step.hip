// api_step.hip — the C ABI of libls1hip (include/ls1hip.h), part 2: the pieces of one time step (re-bin, halo, force traversals,
// Leapfrog passes, thermostat scaling), the neighbour lists and their upkeep, and ls1hip_run (the whole loop without host round trips).
#include "api_internal.hpp"

// ---- step pieces ---------------------------------------------------------------------------------------------------
RebinArgs rebin_args(ls1hip_ctx* c, uint32_t n_in) {
	RebinArgs a;
	a.g = c->g;
	a.src = c->mol[c->cur];
	if (c->pos_x) {  // a fused force pass left the advanced positions of the owned molecules elsewhere (force arrays / alt buffer)
		a.src.x = c->pos_x;
		a.src.y = c->pos_y;
		a.src.z = c->pos_z;
	}
	a.dst = c->mol[c->cur ^ 1];
	a.has_rot = c->h_ct.has_rot;
	a.key = c->d_key; a.rank = c->d_rank; a.perm = c->d_perm; a.ckey = c->d_ckey; a.idk = c->d_idk;
	a.count = c->d_count; a.cell_begin = c->d_cell_begin; a.cell_end = c->d_cell_end; a.blocksum = c->d_blocksum;
	a.cnt = c->d_cnt;
	a.n_in = n_in;
	memcpy(a.nbr, c->nbr, sizeof(a.nbr));
	a.my_rank = c->my_rank;
	memcpy(a.shift, c->shift, sizeof(a.shift));
	a.exp_leave = c->d_exp_leave;
	memcpy(a.exp_off, c->exp_off_leave, sizeof(a.exp_off));
	a.cap_real = (uint32_t)c->cap_real;
	a.deterministic = (int)c->opt_det;
	return a;
}

HaloArgs halo_args(ls1hip_ctx* c) {
	HaloArgs a;
	a.g = c->g;
	a.mol = c->mol[c->cur];
	a.hs = c->hs;
	a.has_rot = c->h_ct.has_rot;
	a.perm = c->d_perm; a.count = c->d_count; a.cell_begin = c->d_cell_begin; a.cell_end = c->d_cell_end;
	a.blocksum = c->d_blocksum;
	a.idk = c->d_idk;
	a.hsrc = c->d_halo_src;
	a.hdir = c->d_halo_dir;
	a.exp_src = c->d_exp_halo_src;
	a.imp_slot = c->d_imp_slot;
	a.imp_at = c->halo_import_at;
	a.s2s = c->d_s2s;
	a.shell = c->d_shell;
	a.nshell = c->n_shell;
	a.cnt = c->d_cnt;
	a.n_real_cap = (uint32_t)c->n_real;
	a.cap_halo = (uint32_t)c->cap_halo;
	memcpy(a.nbr, c->nbr, sizeof(a.nbr));
	a.my_rank = c->my_rank;
	memcpy(a.shift, c->shift, sizeof(a.shift));
	a.rc = c->rc_list;
	a.exp_halo = c->d_exp_halo;
	memcpy(a.exp_off, c->exp_off_halo, sizeof(a.exp_off));
	a.deterministic = (int)c->opt_det;
	return a;
}

int do_rebin_finish(ls1hip_ctx* c, uint32_t n_in) {
	RebinArgs a = rebin_args(c, n_in);
	launch_rebin_sort_gather(a, c->stream);
	HIPCHK(c, hipGetLastError());
	c->cur ^= 1;
	c->pos_x = c->pos_y = c->pos_z = nullptr;
	c->vl_ready = false;  // a new binning: the neighbour lists (offsets into the old staging order) are void
	c->vl_bound_pending = false;
	c->binned = true;
	c->halo_valid = false;
	c->forces_valid = false;
	c->msl_pk_fresh = false;
	return LS1HIP_OK;
}

extern "C" int ls1hip_rebin(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_domain && c->cap_real, "no molecules uploaded");
	HIPCHK(c, hipSetDevice(c->device));
	REQUIRE(c, !c->fused_split, "a fused inner pass (which=1) is waiting for its which=2 pass");
	if (c->inner_in_flight) {  // an inner pass that was never completed by a boundary pass: leave the two-stream mode cleanly
		HIPCHK(c, hipStreamSynchronize(c->stream2));
		c->inner_in_flight = false;
	}
	TimedScope ts(c, c->t_rebin);
	RebinArgs a = rebin_args(c, (uint32_t)c->n_real);
	launch_rebin_classify(a, c->stream);
	HIPCHK(c, hipGetLastError());
	c->pending_in = (uint32_t)c->n_real;
	if (!c->has_remote) return do_rebin_finish(c, (uint32_t)c->n_real);
	c->binned = false;
	return LS1HIP_OK;
}

extern "C" int ls1hip_halo(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->binned, "ls1hip_rebin (and import_done(0) on multi-rank domains) must precede ls1hip_halo");
	HIPCHK(c, hipSetDevice(c->device));
	hipStream_t hs = halo_stream(c);
	if (c->inner_in_flight) HIPCHK(c, hipStreamWaitEvent(hs, c->ev_owned, 0));  // the re-binned owned molecules
	TimedScope ts(c, c->t_halo, hs);
	c->halo_import_at = 0;
	c->vl_ready = false;
	if (c->vl_on && c->cap_halo) HIPCHK(c, hipMemsetAsync(c->d_s2s, 0xff, c->cap_halo * sizeof(uint32_t), hs));
	HaloArgs a = halo_args(c);
	launch_halo_generate(a, hs);
	if (!c->has_remote) {
		launch_halo_finalize(a, hs);
		c->halo_valid = true;
		if (c->inner_in_flight) HIPCHK(c, hipEventRecord(c->ev_halo, hs));
	}
	HIPCHK(c, hipGetLastError());
	c->forces_valid = false;
	c->msl_pk_fresh = false;
	return LS1HIP_OK;
}

// Stream discipline of the force passes.  which = 1 (inner cells) marks the state the halo phase may read and leaves the
// pass in flight: until the matching which = 2 call the halo phase (ls1hip_halo, export / import of kind 1) runs on the
// second, high-priority stream, concurrently with the inner-cell kernel, and the host is never blocked by that kernel.
// which = 2 (boundary cells) first waits for the populated halo.
static int before_force_pass(ls1hip_ctx* c, int which) {
	if (which == 1) {
		HIPCHK(c, hipEventRecord(c->ev_owned, c->stream));
	} else if (c->inner_in_flight) {
		if (which == 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_halo, 0));
		c->inner_in_flight = false;
	}
	return LS1HIP_OK;
}

// vl: 0 = per-step kernels (search every step), 2 = forces from the stored neighbour lists (kernels_force_verlet.hip).
// In the list mode the current positions are read from the buffer the previous fused pass wrote (c->pos_*, else
// mol[cur]) and a fused pass writes the advanced positions to the OTHER of the two position buffers.
struct ForcePass {
	int which = 0;
	bool fuse = false;
	bool post_kick = false;  // list mode: the pass does the post-force kick (+ sum m v^2) itself and keeps F (fuse must be false)
	double dt = 0.;
	int vl = 0;
	bool lists_rebuilt = false;  // list mode: the lists were rebuilt in this step (the displacement bound restarts)
};

static bool msl_lj_only(const ls1hip_ctx* c) {
	bool lj_only = true;
	for (int k = 0; k < c->h_ct.ncomp; ++k) lj_only = lj_only && c->h_ct.nc[k] == 0 && c->h_ct.nd[k] == 0 && c->h_ct.nq[k] == 0;
	return lj_only;
}
static void fill_force_params(ls1hip_ctx* c, ForceParams& P, int which) {
	memset(&P, 0, sizeof(P));
	const MolSoA& m = c->mol[c->cur];
	P.g = c->g;
	P.x = m.x; P.y = m.y; P.z = m.z; P.q0 = m.q0; P.q1 = m.q1; P.q2 = m.q2; P.q3 = m.q3;
	P.cid = m.cid;
	P.cell_begin = c->d_cell_begin; P.cell_end = c->d_cell_end; P.ckey = c->d_ckey;
	P.Fx = c->frc.Fx; P.Fy = c->frc.Fy; P.Fz = c->frc.Fz; P.Mx = c->frc.Mx; P.My = c->frc.My; P.Mz = c->frc.Mz;
	P.Vix = c->frc.Vix; P.Viy = c->frc.Viy; P.Viz = c->frc.Viz;
	P.ct = c->d_ct;
	P.cnt = c->d_cnt;
	P.partials = c->d_partials;
	P.n_real_cap = (uint32_t)c->n_real;
	P.which = which;
	P.count_pairs = (int)c->opt_count_pairs;
	P.eps24 = c->h_ct.eps24[0];
	P.sig2 = c->h_ct.sig2[0];
	P.shift6 = c->h_ct.shift6[0];
	P.rc2 = c->h_ct.rc2;
	P.vl_rc2 = c->rc_list * c->rc_list;
	P.vl_words = c->d_vl_words;
	P.vl_nw = c->d_vl_nw;
	P.precision = (c->opt_precision && c->vl_all_regular) ? (int)c->opt_precision : 0;
	P.vl_rec = c->d_vl_rec;
	P.vl_ii = c->d_vl_ii;
	P.vl_gi = c->d_vl_gi;
	P.msl_gm = c->h_ct.ncomp > 1 ? c->d_msl_gm : nullptr;
	P.msl_g = msl_group_size(msl_lj_only(c), c->h_ct.ncomp);
}

static int launch_forces(ls1hip_ctx* c, const ForcePass& fp) {
	const int which = fp.which;
	const bool fuse = fp.fuse;
	ForceParams P;
	fill_force_params(c, P, which);
	const MolSoA& m = c->mol[c->cur];
	if (fuse || fp.post_kick) {
		P.fuse = fuse ? 1 : 2;
		P.dt = fp.dt;
		P.dt_inv2m = (.5 * fp.dt) / c->h_ct.mass[0];  // as k_kick_then_kick_drift: dt_halve / mass
		P.mass = c->h_ct.mass[0];
		P.vx = m.vx; P.vy = m.vy; P.vz = m.vz;
	}
	if (fp.vl) {
		P.vl_mode = 2;
		if (c->pos_x) {  // current positions (owned + refreshed halo) live in the second buffer
			P.x = c->pos_x; P.y = c->pos_y; P.z = c->pos_z;
		}
		if (fuse && c->one_clj) {  // the advanced positions go to the other buffer
			const bool in_alt = c->pos_x == c->alt_x;
			P.Fx = in_alt ? m.x : c->alt_x;
			P.Fy = in_alt ? m.y : c->alt_y;
			P.Fz = in_alt ? m.z : c->alt_z;
		}
	}
	uint32_t nblocks = 0;
	// the first pass of a traversal starts the macroscopic sums: the reduction overwrites them (pair counting, a
	// diagnostic mode, also needs its counters cleared before the kernel)
	const bool first_pass = which == 0 || which == 1;
	if (first_pass && c->opt_count_pairs) launch_clear_macro(c->d_cnt, c->stream);
	bool done = false;
	int family = LS1HIP_FK_LDS_LIST;
	const double ncell = (double)c->g.box[0] * c->g.box[1] * c->g.box[2];
	const double mean_per_cell = ncell > 0 ? (double)c->n_real / ncell : 0.;
	// local rebuild criterion (kernels_force_verlet.hip, k_bound_local): complete fused traversals of a single periodic domain
	const bool local_ok = fp.vl && c->one_clj && which == 0 && !c->has_remote && c->opt_local_rebuild && c->d_vl_top2;
	const bool local_crit = local_ok && fuse;
	// A pass that does the post-force kick but leaves the drift to a separate kick + drift pass (NVT; piecewise drivers) reports,
	// per brick, the two largest BOUNDS of the coming drift speed: |beta v + dt/2m F| <= max(beta, 1) (|v| + |dt/2m F|), the
	// thermostat factor beta being known only after this pass.  track_unfused_drift turns them into the local criterion.
	const bool local_post = local_ok && !fuse && fp.post_kick;
	if (local_crit || local_post) P.vl_top2 = c->d_vl_top2;
	c->vl_top2_pending = local_post;
	if (fp.vl && !c->one_clj) {
		if (which != 0) FAIL(c, LS1HIP_EINVAL, "multi-site neighbour lists serve complete traversals (which = 0)");
		const bool lj_only = msl_lj_only(c);
		// linear molecules (every LJ centre on the body z axis: ethane, the 2CLJ family): the axis form of the orientation
		bool linear = lj_only;
		for (int k = 0; k < c->h_ct.ncenters && linear; ++k) linear = c->h_ct.ljpos[k][0] == 0. && c->h_ct.ljpos[k][1] == 0.;
		if (fuse || fp.post_kick) {
			// fuse: the pass integrates its molecules itself — state in place, next step's records to the other record buffer, one
			// drift speed per group; post_kick: it does the post-force kick and leaves {sum m v^2, sum I w^2, rot. DOF} per group
			const MolSoA& ms = c->mol[c->cur];
			P.Dx = ms.Dx; P.Dy = ms.Dy; P.Dz = ms.Dz;
			P.ox = c->pos_x ? c->pos_x : ms.x; P.oy = c->pos_x ? c->pos_y : ms.y; P.oz = c->pos_x ? c->pos_z : ms.z;  // (= P.x y z, writable)
			P.oq0 = ms.q0; P.oq1 = ms.q1; P.oq2 = ms.q2; P.oq3 = ms.q3;
			P.msl_pk_out = c->d_msl_pk2;
			P.msl_vmax = c->d_partials + (size_t)4 * msl_groups((uint32_t)c->n_real, P.msl_g);  // behind the macroscopic partials
		}
		done = launch_force_ms_list(P, c->h_ct.has_rot != 0, lj_only, linear, c->h_ct.ncomp, c->d_msl_off, c->d_msl_j, c->d_msl_il, c->d_shift27, c->d_msl_pk,
									c->stream, &nblocks, c->partials_cap, c->msl_pk_fresh);
		if (!done) FAIL(c, LS1HIP_EINVAL, "multi-site neighbour-list force pass could not be launched");
		if (fuse) {
			// the displacement bound of the lists, from the drift speeds the epilogue left per group (as track_unfused_drift)
			launch_bound_update(c->d_cnt, P.msl_vmax, nblocks, fp.dt, 0.5 * c->vl_skin, c->vl_fresh, ++c->vl_seq, c->d_flag, c->stream, false);
			std::swap(c->d_msl_pk, c->d_msl_pk2);
		} else if (fp.post_kick) {
			launch_kin_reduce(c->d_cnt, P.msl_vmax, nblocks, c->stream, c->thermostat_on ? c->thermostat_T : 0., c->log_row);
		}
		family = LS1HIP_FK_NEIGHBOUR_LIST;
	} else if (fp.vl) {
		done = launch_force_verlet(P, c->stream, &nblocks, c->partials_cap, &c->brick_lists);
		if (!done) FAIL(c, LS1HIP_EINVAL, "neighbour-list force pass could not be launched");
		family = LS1HIP_FK_NEIGHBOUR_LIST;
		if (local_crit) launch_bound_local(c->g, c->d_vl_top2, c->d_vl_acc, c->d_cnt, fp.dt, 0.5 * c->vl_skin, c->stream);
	} else if (c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi && !c->opt_count_pairs) {
		done = launch_force_lj(P, c->stream, &nblocks, c->d_partials, c->partials_cap, (int)c->opt_lj_split, mean_per_cell,
							   &c->brick_lists);
	} else if (!c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_count_pairs && which != 3) {
		double vol = 1.;
		for (int d = 0; d < 3; ++d) vol *= c->g.bmax[d] - c->g.bmin[d];
		const double nbrs = vol > 0. ? (double)c->n_real / vol * 4.18879 * c->rc * c->rc * c->rc : 0.;
		// AUTO = the molecule-pair brick kernel: at the 2-5 molecules per cell of the multi-site fixtures both kernels are bound
		// by the per-brick staging / search chain, not by the pair bodies, and the site kernel (cheaper bodies, more phases)
		// measured 10-20 % slower; it is there on request
		if (c->opt_force_kernel == LS1HIP_FK_MS_SITES) {
			done = launch_force_sites(P, c->h_ct, c->opt_vi != 0, c->stream, &nblocks, c->partials_cap, mean_per_cell, nbrs,
									  &c->brick_lists);
			if (done) family = LS1HIP_FK_MS_SITES;
		}
		if (!done) {
			done = launch_force_ms(P, c->opt_vi != 0, c->h_ct.has_rot != 0, c->h_ct.ncomp == 1, c->stream, &nblocks, c->partials_cap,
								   mean_per_cell, &c->brick_lists);
			if (done) family = LS1HIP_FK_MS_BRICK;
		}
	}
	if (!done && fuse) FAIL(c, LS1HIP_EINVAL, "fused force + integration needs the single-centre LJ fast path");
	c->last_force_kernel = done ? family : LS1HIP_FK_GENERIC;
	if (!done) {
		double vol = 1.;
		for (int d = 0; d < 3; ++d) vol *= c->g.bmax[d] - c->g.bmin[d];
		const double nbrs = vol > 0. ? (double)c->n_real / vol * 4.18879 * c->rc * c->rc * c->rc : 0.;
		launch_force_generic(P, c->one_clj, c->opt_vi != 0, c->h_ct.has_rot != 0, c->stream, &nblocks, nbrs);
	}
	ReduceMode rm;
	rm.overwrite = first_pass && !c->opt_count_pairs;
	rm.kin_in_slot1 = (fuse || fp.post_kick) && c->one_clj;  // (multi-site list passes: their own partial buffers, see above)
	rm.target_T = (fp.post_kick && c->thermostat_on && c->one_clj) ? c->thermostat_T : 0.;
	rm.log = c->log_row;
	if (fp.vl && fuse && c->one_clj) {
		rm.vmax_in_slot2 = true;
		rm.last_pass = which != 1;
		rm.lists_rebuilt = fp.lists_rebuilt;
		rm.dt = fp.dt;
		rm.limit = 0.5 * c->vl_skin;
		rm.local_criterion = local_crit;
		rm.seq = ++c->vl_seq;
		rm.flag = c->d_flag;
	}
	launch_force_reduce(c->d_cnt, c->d_partials, nblocks, c->d_stage, c->stream, rm);
	HIPCHK(c, hipGetLastError());
	return LS1HIP_OK;
}
static int launch_forces(ls1hip_ctx* c, int which, bool fuse = false, double dt = 0.) {
	ForcePass fp;
	fp.which = which;
	fp.fuse = fuse;
	fp.dt = dt;
	return launch_forces(c, fp);
}

void macro_to_upot_virial(const DevCounters* h, double* upot, double* virial) {
	// VectorizedCellProcessor::endTraversal, VectorizedCellProcessor.cpp:155-156
	if (upot) *upot = h->macro[0] / 6.0 + h->macro[1] + h->macro[2];
	if (virial) *virial = h->macro[3] + 3.0 * h->macro[2];
}

extern "C" int ls1hip_forces(ls1hip_ctx* c, int which, double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, which >= 0 && which <= 2, "which must be 0, 1 or 2");
	REQUIRE(c, c->binned, "molecules are not binned (call ls1hip_rebin)");
	REQUIRE(c, which == 1 || c->halo_valid, "halo not populated (call ls1hip_halo / import_done(1))");
	REQUIRE(c, !c->fused_split, "a fused inner pass (ls1hip_forces_kick_drift which=1) must be completed by its which=2 pass");
	HIPCHK(c, hipSetDevice(c->device));
	{
		int rc = before_force_pass(c, which);
		if (rc) return rc;
		TimedScope ts(c, c->t_force);
		if ((rc = launch_forces(c, which))) return rc;
	}
	if (which == 1) c->inner_in_flight = !c->halo_valid;  // halo already populated (old call order): nothing to overlap
	if (which != 1) c->forces_valid = true;
	if (upot || virial) {
		int rc = sync_counters(c);
		if (rc) return rc;
		macro_to_upot_virial(c->h_cnt, upot, virial);
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_forces_kick_drift(ls1hip_ctx* c, int which, double dt, double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, which >= 0 && which <= 2, "which must be 0, 1 or 2");
	REQUIRE(c, c->binned, "molecules are not binned (call ls1hip_rebin)");
	REQUIRE(c, which == 1 || c->halo_valid, "halo not populated (call ls1hip_halo / import_done(1))");
	REQUIRE(c, can_fuse(c), "fused force + integration: single-centre LJ fast path, no per-molecule virial, no device thermostat");
	REQUIRE(c, which == 2 ? c->fused_split == 1 : c->fused_split == 0, "fused passes must be which=0, or which=1 followed by which=2");
	HIPCHK(c, hipSetDevice(c->device));
	{
		int rc = before_force_pass(c, which);
		if (rc) return rc;
		TimedScope ts(c, c->t_force);
		if ((rc = launch_forces(c, which, true, dt))) return rc;
	}
	if (which == 1) {
		c->inner_in_flight = !c->halo_valid;
		c->fused_split = 1;
	} else {
		// velocities are at t + dt/2 of the NEXT step and the advanced positions wait in the force arrays for ls1hip_rebin
		c->fused_split = 0;
		c->pos_x = c->frc.Fx; c->pos_y = c->frc.Fy; c->pos_z = c->frc.Fz;
		c->vl_ready = false;
		c->binned = false;
		c->halo_valid = false;
		c->forces_valid = false;
		c->msl_pk_fresh = false;
	}
	if (upot || virial) {
		int rc = sync_counters(c);
		if (rc) return rc;
		macro_to_upot_virial(c->h_cnt, upot, virial);
	}
	return LS1HIP_OK;
}

// positions parked in the force arrays by a fused pass -> back into the molecule arrays (readers other than ls1hip_rebin)
int materialise_positions(ls1hip_ctx* c) {
	if (!c->pos_x) return LS1HIP_OK;
	const MolSoA& m = c->mol[c->cur];
	const uint32_t n = (uint32_t)c->n_real;
	launch_pack_copy(m.x, c->pos_x, n, c->stream);
	launch_pack_copy(m.y, c->pos_y, n, c->stream);
	launch_pack_copy(m.z, c->pos_z, n, c->stream);
	HIPCHK(c, hipGetLastError());
	c->pos_x = c->pos_y = c->pos_z = nullptr;
	return LS1HIP_OK;
}

static IntegArgs integ_args(ls1hip_ctx* c, double dt) {
	IntegArgs a;
	a.mol = c->mol[c->cur];
	a.frc = c->frc;
	a.ct = c->d_ct;
	a.cnt = c->d_cnt;
	a.partials = c->d_partials;
	a.n_cap = (uint32_t)c->n_real;
	a.has_rot = c->h_ct.has_rot;
	a.dt = dt;
	a.vmax_part = nullptr;
	return a;
}

// List mode: a separate kick + drift pass moves the molecules in place in the CURRENT position buffer, reports the
// maximum drift speed and advances the displacement bound (the lists stay valid; the next ls1hip_update decides).
static IntegArgs integ_args_lists(ls1hip_ctx* c, double dt) {
	IntegArgs a = integ_args(c, dt);
	if (c->pos_x) {
		a.mol.x = c->pos_x;
		a.mol.y = c->pos_y;
		a.mol.z = c->pos_z;
	}
	a.vmax_part = c->d_partials;
	// rigid bodies under pair-stream lists: the pass also writes the records the next force pass reads (instead of k_msl_pack)
	if (!c->one_clj && c->h_ct.has_rot && c->vl_ready && c->d_msl_pk && (size_t)c->n_real <= c->msl_stride) {
		a.pk = c->d_msl_pk;
		a.pk_ncomp = c->h_ct.ncomp;
	}
	return a;
}
// beta: the thermostat factor the drift pass applied to the velocities (host value), or < 0: it took cnt->beta[0] on the device
static int track_unfused_drift(ls1hip_ctx* c, double dt, double beta = 1.) {
	const uint32_t nb = ((uint32_t)c->n_real + 255u) / 256u;
	// local criterion (single periodic domain): the force pass before this drift left every brick's two largest drift-speed
	// bounds (launch_forces, local_post); the global bound then only decides together with the brick neighbourhoods' pair bounds
	const bool local = c->vl_top2_pending && c->vl_ready && c->one_clj && !c->has_remote && c->opt_local_rebuild && c->d_vl_top2 && c->d_vl_acc;
	c->vl_top2_pending = false;
	if (local) launch_bound_local(c->g, c->d_vl_top2, c->d_vl_acc, c->d_cnt, dt, 0.5 * c->vl_skin, c->stream, beta < 0. ? -1. : std::max(beta, 1.));
	launch_bound_update(c->d_cnt, c->d_partials, nb, dt, 0.5 * c->vl_skin, c->vl_fresh, ++c->vl_seq, c->d_flag, c->stream, local);
	HIPCHK(c, hipGetLastError());
	c->vl_fresh = false;
	c->vl_bound_pending = true;
	return LS1HIP_OK;
}

static int kick_drift_impl(ls1hip_ctx* c, double dt, int pre_scale, double bt, double br);
extern "C" int ls1hip_kick_drift(ls1hip_ctx* c, double dt) { return kick_drift_impl(c, dt, 0, 1., 1.); }
extern "C" int ls1hip_scale_kick_drift(ls1hip_ctx* c, double beta_trans, double beta_rot, double dt) {
	return kick_drift_impl(c, dt, 1, beta_trans, beta_rot);
}
static int kick_drift_impl(ls1hip_ctx* c, double dt, int pre_scale, double bt, double br) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->cap_real, "no molecules uploaded");
	REQUIRE(c, !c->fused_split, "a fused inner pass is waiting for its boundary pass");
	REQUIRE(c, !c->pos_x || c->forces_valid, "positions were already advanced by ls1hip_forces_kick_drift (call ls1hip_rebin)");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->pos_x && !c->vl_ready) {  // positions parked in another buffer and no lists to keep alive
		int rcm = materialise_positions(c);
		if (rcm) return rcm;
	}
	TimedScope ts(c, c->t_integrate);
	bool wrote_pk = false;
	if (c->vl_ready) {
		IntegArgs a = integ_args_lists(c, dt);
		a.pre_scale = pre_scale; a.pre_bt = bt; a.pre_br = br;
		launch_kick_drift(a, c->stream);
		wrote_pk = a.pk != nullptr;
		int rcb = track_unfused_drift(c, dt, pre_scale == 0 ? 1. : pre_scale == 2 ? -1. : bt);
		if (rcb) return rcb;
	} else {
		IntegArgs a = integ_args(c, dt);
		a.pre_scale = pre_scale; a.pre_bt = bt; a.pre_br = br;
		launch_kick_drift(a, c->stream);
	}
	HIPCHK(c, hipGetLastError());
	c->binned = false;
	c->halo_valid = false;
	c->forces_valid = false;
	c->msl_pk_fresh = wrote_pk;
	return LS1HIP_OK;
}

extern "C" int ls1hip_kick_then_kick_drift(ls1hip_ctx* c, double dt) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->forces_valid, "forces are not valid (call ls1hip_forces)");
	REQUIRE(c, !c->thermostat_on, "with the device thermostat the two half kicks are separate passes (kick, scale, kick_drift)");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->pos_x && !c->vl_ready) {
		int rcm = materialise_positions(c);
		if (rcm) return rcm;
	}
	TimedScope ts(c, c->t_integrate);
	bool wrote_pk = false;
	if (c->vl_ready) {
		const IntegArgs a = integ_args_lists(c, dt);
		launch_kick_then_kick_drift(a, c->stream);
		wrote_pk = a.pk != nullptr;
		int rcb = track_unfused_drift(c, dt);
		if (rcb) return rcb;
	} else {
		launch_kick_then_kick_drift(integ_args(c, dt), c->stream);
	}
	HIPCHK(c, hipGetLastError());
	c->binned = false;
	c->halo_valid = false;
	c->forces_valid = false;
	c->msl_pk_fresh = wrote_pk;
	return LS1HIP_OK;
}

extern "C" int ls1hip_kick(ls1hip_ctx* c, double dt_half, double* summv2, double* sumIw2, uint64_t* n,
						   uint64_t* rot_dof) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->forces_valid, "forces are not valid (call ls1hip_forces)");
	HIPCHK(c, hipSetDevice(c->device));
	c->vl_top2_pending = false;  // (a kick after the pass that formed the per-brick bounds: they no longer bound the drift speed)
	{
		TimedScope ts(c, c->t_integrate);
		uint32_t nb = 0;
		launch_kick(integ_args(c, dt_half), c->stream, &nb);
		launch_kin_reduce(c->d_cnt, c->d_partials, nb, c->stream, c->thermostat_on ? c->thermostat_T : 0., c->log_row_kin);
	}
	if (summv2 || sumIw2 || n || rot_dof) {
		int rc = sync_counters(c);
		if (rc) return rc;
		if (summv2) *summv2 = c->h_cnt->kin[0];
		if (sumIw2) *sumIw2 = c->h_cnt->kin[1];
		if (n) *n = c->h_cnt->kin_n;
		if (rot_dof) *rot_dof = c->h_cnt->kin_rotdof;
	}
	return LS1HIP_OK;
}

// Component-wise thermostats (Domain::severalThermostats): the kinetic sums of Leapfrog::transition2to3 (Leapfrog.cpp:84-104) per
// COMPONENT, from the current velocities / angular momenta (i.e. after ls1hip_kick); the caller folds components into thermostats
// (Domain::getThermostat).  A separate pass over v, D, q, cid (only taken by runs with several thermostats).
extern "C" int ls1hip_kinetic_sums_by_component(ls1hip_ctx* c, int ncomp, double* summv2, double* sumIw2, uint64_t* n, uint64_t* rot_dof) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp && ncomp == c->h_ct.ncomp, "ncomp must be the number of components (%d)", c->h_ct.ncomp);
	REQUIRE(c, c->cap_real, "no molecules uploaded");
	HIPCHK(c, hipSetDevice(c->device));
	{
		TimedScope ts(c, c->t_integrate);
		launch_kin_by_component(integ_args(c, 0.), ncomp, c->d_partials, c->d_stage, c->stream);
		HIPCHK(c, hipGetLastError());
	}
	double h[MAXC * 4];
	HIPCHK(c, hipMemcpyAsync(h, c->d_stage, (size_t)ncomp * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for (int k = 0; k < ncomp; ++k) {
		if (summv2) summv2[k] = h[4 * k];
		if (sumIw2) sumIw2[k] = h[4 * k + 1];
		if (n) n[k] = (uint64_t)(h[4 * k + 2] + 0.5);
		if (rot_dof) rot_dof[k] = (uint64_t)(h[4 * k + 3] + 0.5);
	}
	return LS1HIP_OK;
}

// VelocityScalingThermostat::apply, componentwise branch (thermostats/VelocityScalingThermostat.cpp:45-69: v *= beta_trans, D *=
// beta_rot of the molecule's thermostat) folded into the pre-force kick + drift, with one factor pair per component
extern "C" int ls1hip_scale_kick_drift_components(ls1hip_ctx* c, int ncomp, const double* beta_trans, const double* beta_rot, double dt) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp && ncomp == c->h_ct.ncomp && beta_trans && beta_rot, "one (beta_trans, beta_rot) pair per component (%d)", c->h_ct.ncomp);
	REQUIRE(c, c->cap_real, "no molecules uploaded");
	REQUIRE(c, !c->fused_split, "a fused inner pass is waiting for its boundary pass");
	REQUIRE(c, !c->pos_x || c->forces_valid, "positions were already advanced by ls1hip_forces_kick_drift (call ls1hip_rebin)");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->pos_x && !c->vl_ready) {
		int rcm = materialise_positions(c);
		if (rcm) return rcm;
	}
	TimedScope ts(c, c->t_integrate);
	IntegArgs a = c->vl_ready ? integ_args_lists(c, dt) : integ_args(c, dt);
	a.pre_scale = 3;
	for (int k = 0; k < MAXC; ++k) {
		a.pre_bt_c[k] = k < ncomp ? beta_trans[k] : 1.;
		a.pre_br_c[k] = k < ncomp ? beta_rot[k] : 1.;
	}
	launch_kick_drift(a, c->stream);
	if (c->vl_ready) {
		double bmax = 1.;
		for (int k = 0; k < ncomp; ++k) bmax = std::max(bmax, beta_trans[k]);
		int rcb = track_unfused_drift(c, dt, bmax);
		if (rcb) return rcb;
	}
	HIPCHK(c, hipGetLastError());
	c->binned = false;
	c->halo_valid = false;
	c->forces_valid = false;
	c->msl_pk_fresh = c->vl_ready && a.pk != nullptr;
	return LS1HIP_OK;
}

extern "C" int ls1hip_kinetic_sums(ls1hip_ctx* c, double* summv2, double* sumIw2, uint64_t* n, uint64_t* rot_dof) {
	if (!c) return LS1HIP_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	int rc = sync_counters(c);
	if (rc) return rc;
	if (summv2) *summv2 = c->h_cnt->kin[0];
	if (sumIw2) *sumIw2 = c->h_cnt->kin[1];
	if (n) *n = c->h_cnt->kin_n;
	if (rot_dof) *rot_dof = c->h_cnt->kin_rotdof;
	return LS1HIP_OK;
}

extern "C" int ls1hip_traversal_mark(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->h_mark) {
		void* h = nullptr;
		HIPCHK(c, hipHostMalloc(&h, sizeof(DevCounters), hipHostMallocDefault));
		c->h_mark = (DevCounters*)h;
		HIPCHK(c, hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming));
	}
	HIPCHK(c, hipMemcpyAsync(c->h_mark, c->d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipEventRecord(c->ev_mark, c->stream));
	return LS1HIP_OK;
}

extern "C" int ls1hip_traversal_sums(ls1hip_ctx* c, double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->h_mark, "no traversal was marked (ls1hip_traversal_mark)");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipEventSynchronize(c->ev_mark));
	if (c->h_mark->err_overflow)
		FAIL(c, LS1HIP_ENOMEM, "device buffer overflow (%u records dropped): halo/export capacity exceeded", c->h_mark->err_overflow);
	if (c->h_mark->err_lost) FAIL(c, LS1HIP_ELOST, "%u molecule(s) left the halo region of this rank", c->h_mark->err_lost);
	macro_to_upot_virial(c->h_mark, upot, virial);
	return LS1HIP_OK;
}

extern "C" int ls1hip_scale_velocities(ls1hip_ctx* c, double beta_trans, double beta_rot) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->cap_real, "no molecules uploaded");
	HIPCHK(c, hipSetDevice(c->device));
	// the per-brick drift-speed bounds a post-kick list pass left behind (launch_forces, local_post) were formed from the velocities
	// of that pass: a scaling in between makes them stale — the coming drift counts with the global bound it measures itself
	c->vl_top2_pending = false;
	TimedScope ts(c, c->t_integrate);
	launch_scale(integ_args(c, 0.), beta_trans, beta_rot, false, c->stream);
	HIPCHK(c, hipGetLastError());
	return LS1HIP_OK;
}

// ---- neighbour-list reuse ---------------------------------------------------------------------------------------------
extern "C" int ls1hip_set_verlet(ls1hip_ctx* c, int enabled, double skin) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, !enabled || skin > 0., "the skin must be positive");
	REQUIRE(c, !c->have_domain, "ls1hip_set_verlet must be called before ls1hip_set_domain (the cell grid depends on rc + skin)");
	c->vl_on = enabled != 0;
	c->vl_force = enabled == 2;
	c->vl_skin = enabled ? skin : 0.;
	c->rc_list = c->rc + c->vl_skin;
	c->vl_ready = false;
	return LS1HIP_OK;
}

// the list-reuse loop serves what the fused per-step loop serves, on a single rank with one cell per cutoff
// (a domain whose mean brick region would not fit the LDS staging area has had its lists switched off at upload time)
static bool can_verlet(const ls1hip_ctx* c) { return c->vl_on && c->g.hw == 1 && !c->has_remote; }

// host-visible word the step's last reduction / the drift pass publishes {sequence, rebuild needed} to
static int ensure_rebuild_flag(ls1hip_ctx* c) {
	if (!c->h_flag) {
		void* h = nullptr;
		HIPCHK(c, hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
		c->h_flag = (volatile uint32_t*)h;
		*c->h_flag = 0;
		void* d = nullptr;
		HIPCHK(c, hipHostGetDevicePointer(&d, h, 0));
		c->d_flag = (uint32_t*)d;
	}
	return LS1HIP_OK;
}

static int ensure_verlet_buffers(ls1hip_ctx* c) {
	long nbricks;
	size_t wpb, tpb;
	verlet_geometry(c->g, &nbricks, &wpb, &tpb);
	const size_t words = (size_t)nbricks * wpb, tiles = (size_t)nbricks * tpb;
	if (words > c->vl_words_cap || tiles > c->vl_tiles_cap) {
		dfree(c->d_vl_words);
		dfree(c->d_vl_nw);
		dfree(c->d_vl_rec);
		dfree(c->d_vl_ii);
		dfree(c->d_vl_gi);
		dfree(c->d_vl_top2);
		dfree(c->d_vl_acc);
		c->vl_words_cap = c->vl_tiles_cap = 0;
		int rc;
		if ((rc = dalloc(c, &c->d_vl_words, words)) || (rc = dalloc(c, &c->d_vl_nw, tiles)) ||
			(rc = dalloc(c, &c->d_vl_rec, (size_t)nbricks * verlet_record_words())) || (rc = dalloc(c, &c->d_vl_ii, tiles * 64)) ||
			(rc = dalloc(c, &c->d_vl_gi, tiles * 64)) || (rc = dalloc(c, &c->d_vl_top2, (size_t)nbricks * 2)) ||
			(rc = dalloc(c, &c->d_vl_acc, (size_t)nbricks)))
			return rc;
		c->vl_words_cap = words;
		c->vl_tiles_cap = tiles;
	}
	return ensure_rebuild_flag(c);
}

// lists of all bricks from the freshly binned molecules + halo copies in mol[cur]
static int verlet_build(ls1hip_ctx* c) {
	int rc = ensure_verlet_buffers(c);
	if (rc) return rc;
	TimedScope ts(c, c->t_build);  // list construction: its own timer ("build"), next to the re-binning of a rebuild step
	ForceParams P;
	fill_force_params(c, P, 0);
	P.vl_mode = 1;
	uint32_t nb = 0;
	HIPCHK(c, hipMemsetAsync(&c->d_cnt->vl_irregular, 0, sizeof(uint32_t), c->stream));
	// local rebuild criterion: the per-brick bounds, the share of unfused drifts and the verdict start at zero with the lists
	HIPCHK(c, hipMemsetAsync(c->d_vl_acc, 0, (size_t)verlet_brick_count(c->g) * sizeof(double), c->stream));
	HIPCHK(c, hipMemsetAsync(&c->d_cnt->vl_base, 0, sizeof(double), c->stream));
	HIPCHK(c, hipMemsetAsync(&c->d_cnt->vl_local_excess, 0, sizeof(uint32_t), c->stream));
	if (!launch_force_verlet(P, c->stream, &nb, c->partials_cap, &c->brick_lists))
		FAIL(c, LS1HIP_EINVAL, "neighbour lists could not be built for this grid");
	HIPCHK(c, hipGetLastError());
	c->vl_builds++;
	c->vl_all_regular = false;
	if (c->opt_precision) {
		// the single-precision force pass serves regular bricks only: it is used while the build reports none of the other kind
		// (one host round trip per list build, only in this mode)
		int rs = sync_counters(c);
		if (rs) return rs;
		c->vl_all_regular = c->h_cnt->vl_irregular == 0;
	}
	return LS1HIP_OK;
}

// halo positions of the current position buffer from their source molecules (no re-binning, no image generation)
static int verlet_refresh_halo(ls1hip_ctx* c) {
	TimedScope ts(c, c->t_halo);
	HaloArgs a = halo_args(c);
	const MolSoA& m = c->mol[c->cur];
	double *x = c->pos_x ? c->pos_x : m.x, *y = c->pos_x ? c->pos_y : m.y, *z = c->pos_x ? c->pos_z : m.z;
	launch_halo_refresh(a, x, y, z, x, y, z, c->stream);
	HIPCHK(c, hipGetLastError());
	c->halo_valid = true;
	return LS1HIP_OK;
}

// result of the step's last reduction: does the displacement bound exceed skin / 2?  (host-visible word, polled: the
// kernels of the next step cannot be chosen before it is known; a stream synchronisation costs ~10 us more)
static int verlet_poll_rebuild(ls1hip_ctx* c, bool* need) {
	const uint32_t want = c->vl_seq;
	for (long spin = 0;; ++spin) {
		const uint32_t f = *c->h_flag;
		if ((f >> 1) == want) {
			*need = (f & 1u) != 0;
			return LS1HIP_OK;
		}
		if ((spin & 0xffff) == 0xffff) {  // a failed launch would never publish: ask the runtime now and then
			hipError_t e = hipStreamQuery(c->stream);
			if (e == hipSuccess) {
				const uint32_t g = *c->h_flag;
				if ((g >> 1) == want) {
					*need = (g & 1u) != 0;
					return LS1HIP_OK;
				}
				FAIL(c, LS1HIP_EHIP, "the rebuild flag of step sequence %u was never published", want);
			}
			if (e != hipErrorNotReady) FAIL(c, LS1HIP_EHIP, "stream error while waiting for the rebuild flag: %s", hipGetErrorString(e));
		}
	}
}

// ---- list mode, piecewise (multi-rank loops drive these; ls1hip_run is the single-rank loop) -------------------------------
// lists serve the single-centre LJ fast path on a one-cell-per-cutoff grid; FUSED list passes additionally need can_fuse()
static bool can_list_lj(const ls1hip_ctx* c) {
	return c->vl_on && c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi && !c->opt_count_pairs && c->g.hw == 1;
}
// multi-site component sets: per-wave pair streams (kernels_force_mslist.hip); single-rank domains, complete traversals
static bool can_list_ms(const ls1hip_ctx* c) {
	return c->vl_on && c->have_comp && !c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi && !c->opt_count_pairs &&
		   c->g.hw == 1 && !c->has_remote && c->n_real + c->cap_halo < (size_t)0x07ffffff;
}
static bool can_list(const ls1hip_ctx* c) { return can_list_lj(c) || can_list_ms(c); }

// pair streams of all groups from the freshly binned molecules + halo copies in mol[cur]
static int msl_build(ls1hip_ctx* c) {
	int rc = ensure_rebuild_flag(c);
	if (rc) return rc;
	TimedScope ts(c, c->t_build);
	const uint32_t ng = msl_groups((uint32_t)c->n_real, msl_group_size(msl_lj_only(c), c->h_ct.ncomp));
	if ((size_t)ng + 1 > c->msl_groups_cap) {
		dfree(c->d_msl_cnt);
		dfree(c->d_msl_off);
		dfree(c->d_msl_gm);
		c->msl_groups_cap = 0;
		if ((rc = dalloc(c, &c->d_msl_cnt, (size_t)ng + 1)) || (rc = dalloc(c, &c->d_msl_off, (size_t)ng + 2)) ||
			(rc = dalloc(c, &c->d_msl_gm, ((size_t)ng + 1) * 128)))
			return rc;
		c->msl_groups_cap = (size_t)ng + 1;
	}
	if (c->n_real > c->msl_stride) {
		dfree(c->d_msl_scratch);
		dfree(c->d_msl_mcnt);
		dfree(c->d_msl_pk);
		dfree(c->d_msl_pk2);
		c->msl_stride = 0;
		const size_t stride = (c->cap_real + 63) & ~(size_t)63;
		if ((rc = dalloc(c, &c->d_msl_scratch, stride * (size_t)msl_capture_cap())) || (rc = dalloc(c, &c->d_msl_mcnt, stride)) ||
			(rc = dalloc(c, &c->d_msl_pk, stride * 8)) || (rc = dalloc(c, &c->d_msl_pk2, stride * 8)))
			return rc;
		c->msl_stride = stride;
	}
	ForceParams P;
	fill_force_params(c, P, 0);
	if (P.msl_gm) launch_msl_groups(P, c->d_msl_gm, c->h_ct.ncomp, c->stream);
	launch_msl_count(P, c->d_msl_cnt, c->d_msl_off, c->d_msl_scratch, c->d_msl_mcnt, (uint32_t)c->msl_stride, c->stream);
	HIPCHK(c, hipGetLastError());
	// one host round trip per list build: the pair count sizes the stream
	unsigned long long total = 0;
	HIPCHK(c, hipMemcpyAsync(&total, &c->d_cnt->msl_total, sizeof(total), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	REQUIRE(c, total < 0xffffffc0ull, "multi-site neighbour lists: %llu pairs exceed the 32-bit pair index", total);
	if (total > c->msl_pairs_cap) {
		dfree(c->d_msl_j);
		dfree(c->d_msl_il);
		c->msl_pairs_cap = 0;
		const size_t want = (size_t)(total + total / 8 + 4096);
		if ((rc = dalloc(c, &c->d_msl_j, want)) || (rc = dalloc(c, &c->d_msl_il, want))) return rc;
		c->msl_pairs_cap = want;
	}
	c->msl_pairs = total;
	launch_msl_fill(P, c->d_msl_off, c->d_halo_src, c->d_halo_dir, c->d_msl_j, c->d_msl_il, c->h_ct.ncomp, c->d_msl_scratch, c->d_msl_mcnt,
					(uint32_t)c->msl_stride, c->stream);
	HIPCHK(c, hipGetLastError());
	c->vl_builds++;
	c->vl_all_regular = false;
	c->msl_pk_fresh = false;  // new order of the owned molecules
	return LS1HIP_OK;
}

extern "C" int ls1hip_verlet_build(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, can_list(c), "neighbour lists need ls1hip_set_verlet and one cell per cutoff (single-centre LJ fast path, or a multi-site set on a single-rank domain)");
	REQUIRE(c, c->binned && c->halo_valid, "neighbour lists are built from binned molecules and a populated halo");
	REQUIRE(c, !c->inner_in_flight && !c->fused_split, "a split force pass is in flight");
	HIPCHK(c, hipSetDevice(c->device));
	int rc = can_list_lj(c) ? verlet_build(c) : msl_build(c);
	if (rc) return rc;
	// the export counts / import total of this halo exchange are what every refresh until the next build repeats
	if ((rc = sync_counters(c))) return rc;
	for (int d = 0; d < 27; ++d) c->vl_exp_counts[d] = c->h_cnt->exp_halo[d];
	c->vl_imp_total = c->halo_import_at;
	c->vl_ready = true;
	c->vl_fresh = true;
	c->vl_bound_pending = false;
	return LS1HIP_OK;
}

extern "C" int ls1hip_halo_refresh(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
	HIPCHK(c, hipSetDevice(c->device));
	hipStream_t hs = halo_stream(c);
	if (c->inner_in_flight) HIPCHK(c, hipStreamWaitEvent(hs, c->ev_owned, 0));  // positions of the owned molecules
	TimedScope ts(c, c->t_halo, hs);
	c->halo_import_at = 0;
	HaloArgs a = halo_args(c);
	const MolSoA& m = c->mol[c->cur];
	double *x = c->pos_x ? c->pos_x : m.x, *y = c->pos_x ? c->pos_y : m.y, *z = c->pos_x ? c->pos_z : m.z;
	launch_halo_refresh(a, x, y, z, x, y, z, hs);
	if (c->has_remote) {
		launch_refresh_pack(a, x, y, z, c->d_exp_refresh, hs);
	} else {
		c->halo_valid = true;
		if (c->inner_in_flight) HIPCHK(c, hipEventRecord(c->ev_halo, hs));
	}
	HIPCHK(c, hipGetLastError());
	return LS1HIP_OK;
}

static int forces_list_impl(ls1hip_ctx* c, int which, double dt, bool post_kick, double* upot, double* virial);
extern "C" int ls1hip_forces_list(ls1hip_ctx* c, int which, double dt, double* upot, double* virial) {
	return forces_list_impl(c, which, dt, false, upot, virial);
}
extern "C" int ls1hip_forces_list_kick(ls1hip_ctx* c, double dt_half, double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, dt_half > 0., "dt_half must be > 0");
	REQUIRE(c, c->one_clj || can_list_kick_ms(c),
			"the post-force kick is folded into the single-centre LJ list pass and into the pair-stream pass of ONE rigid component (otherwise: ls1hip_forces_list + ls1hip_kick)");
	return forces_list_impl(c, 0, 2. * dt_half, true, upot, virial);
}
// post_kick (ls1hip_run, ls1hip_forces_list_kick): dt is the time step, the pass is NOT fused with the drift but does the post-force kick and
// the kinetic sum of the step itself (F is stored); which must be 0
static int forces_list_impl(ls1hip_ctx* c, int which, double dt, bool post_kick, double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, which >= 0 && which <= 2, "which must be 0, 1 or 2");
	REQUIRE(c, c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
	REQUIRE(c, which == 1 || c->halo_valid, "halo positions are not current (ls1hip_halo_refresh / import_done(2))");
	REQUIRE(c, dt >= 0., "dt must be >= 0 (0: forces only, > 0: fused with kick + kick + drift)");
	const bool fuse = dt > 0. && !post_kick;
	REQUIRE(c, !post_kick || which == 0, "the post-force kick is folded into complete traversals only");
	REQUIRE(c, !fuse || (c->one_clj ? can_fuse(c) : (can_fuse_ms(c) && which == 0)),
			"fused list passes: single-centre LJ or rigid multi-site bodies (complete traversals), no per-molecule virial, no device thermostat");
	REQUIRE(c, (fuse && which == 2) ? c->fused_split == 1 : c->fused_split == 0,
			"fused list passes must be which=0, or which=1 followed by which=2");
	HIPCHK(c, hipSetDevice(c->device));
	{
		int rc = before_force_pass(c, which);
		if (rc) return rc;
		TimedScope ts(c, c->t_force);
		ForcePass fp;
		fp.which = which;
		fp.fuse = fuse;
		fp.dt = dt;
		fp.vl = 2;
		fp.post_kick = post_kick;
		fp.lists_rebuilt = c->vl_fresh;
		if ((rc = launch_forces(c, fp))) return rc;
	}
	if (which == 1) {
		c->inner_in_flight = !c->halo_valid;
		if (fuse) c->fused_split = 1;
	} else if (fuse) {
		// velocities are at t + dt/2 of the next step; the advanced positions wait in the other position buffer (single-centre
		// LJ), or are in place with the next step's records in the record buffer (rigid bodies: launch_forces swapped the two)
		if (c->one_clj) {
			const bool in_alt = c->pos_x == c->alt_x;
			c->pos_x = in_alt ? nullptr : c->alt_x;
			c->pos_y = in_alt ? nullptr : c->alt_y;
			c->pos_z = in_alt ? nullptr : c->alt_z;
		}
		c->fused_split = 0;
		c->vl_fresh = false;
		c->vl_bound_pending = true;
		c->halo_valid = false;
		c->forces_valid = false;
		c->msl_pk_fresh = !c->one_clj;
		c->vl_steps++;
	} else {
		c->forces_valid = true;
		c->vl_steps++;
	}
	if (upot || virial) {
		int rc = sync_counters(c);
		if (rc) return rc;
		macro_to_upot_virial(c->h_cnt, upot, virial);
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_verlet_poll(ls1hip_ctx* c, int* need_rebuild) {
	if (!c || !need_rebuild) return LS1HIP_EINVAL;
	REQUIRE(c, c->vl_bound_pending, "no fused list pass has published a displacement bound");
	bool need = true;
	int rc = verlet_poll_rebuild(c, &need);
	if (rc) return rc;
	*need_rebuild = need ? 1 : 0;
	return LS1HIP_OK;
}

// LinkedCells::update + DomainDecompBase::balanceAndExchange + updateMoleculeCaches of a SINGLE-RANK domain in one call, list-aware:
// while neighbour lists are alive and the displacement bound allows it, the molecules keep their cells and the halo copies
// their slots — only the halo positions are refreshed; otherwise re-bin, regenerate the halo and (list mode) rebuild.
extern "C" int ls1hip_update(ls1hip_ctx* c, int* rebuilt) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, !c->has_remote, "ls1hip_update serves single-rank domains (multi-rank: rebin / exchange / halo / verlet_build)");
	REQUIRE(c, c->have_domain && c->cap_real, "no molecules uploaded");
	HIPCHK(c, hipSetDevice(c->device));
	int rc;
	bool rebuild = true;
	if (c->vl_ready && can_list(c)) {
		rebuild = false;
		if (c->vl_bound_pending && (rc = verlet_poll_rebuild(c, &rebuild))) return rc;
	}
	if (rebuilt) *rebuilt = rebuild ? 1 : 0;
	if (!rebuild) return verlet_refresh_halo(c);
	if ((rc = ls1hip_rebin(c)) || (rc = ls1hip_halo(c))) return rc;
	if (can_list(c)) return ls1hip_verlet_build(c);
	return LS1HIP_OK;
}

extern "C" int ls1hip_run(ls1hip_ctx* c, double dt, unsigned long nsteps, double* out6) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, !c->has_remote, "ls1hip_run drives single-rank domains only (use the piecewise calls with a transport)");
	REQUIRE(c, c->forces_valid, "initial forces required (rebin, halo, forces) before ls1hip_run");
	HIPCHK(c, hipSetDevice(c->device));
	// Between two steps of an NVE run on the LJ fast path the force pass does the integration itself (fused mode, the
	// reference's reduced-memory scheme); the last step is unfused so that F and the kinetic sums are available.
	const bool fuse = c->opt_fuse && (can_fuse(c) || (can_verlet(c) && can_list_ms(c) && can_fuse_ms(c)));
	// neighbour-list loop (fused or not: NVT and unfused NVE steps advance the displacement bound in their kick + drift pass)
	const bool verlet = can_verlet(c) && can_list(c);
	// the single-centre list pass does the post-force kick (+ sum m v^2) itself; the multi-site one leaves it to the integrator passes
	const bool list_kick = verlet && (c->one_clj || (can_list_ms(c) && can_list_kick_ms(c)));
	bool advanced = false;  // the previous force pass already did kick + kick + drift
	// step log: one row {U_pot, virial, sum m v^2, sum I w^2, N, rotDOF} per step, written by the reductions on the device
	if (!c->d_steplog) {
		int rc0 = dalloc(c, &c->d_steplog, STEPLOG_ROWS * 6);
		if (rc0) return rc0;
	}
	HIPCHK(c, hipMemsetAsync(c->d_steplog, 0xff, std::min<size_t>(nsteps, STEPLOG_ROWS) * 6 * sizeof(double), c->stream));  // NaN = not computed
	struct LogGuard {
		ls1hip_ctx* c;
		~LogGuard() { c->log_row = c->log_row_kin = nullptr; }
	} log_guard{c};
	c->steplog_steps = 0;
	for (unsigned long s = 0; s < nsteps; ++s) {
		int rc;
		c->log_row = c->d_steplog + (s % STEPLOG_ROWS) * 6;                         // forces of step s
		c->log_row_kin = s ? c->d_steplog + ((s - 1) % STEPLOG_ROWS) * 6 : nullptr;  // a kick at the head of step s ends step s-1
		if (advanced) {
			// nothing to integrate: positions wait in the force arrays for the re-binning pass
		} else if (s == 0) {
			if ((rc = ls1hip_kick_drift(c, dt))) return rc;
		} else if (c->thermostat_on) {
			// NVT: the scaling factors depend on the kinetic sums after the kick, so the two half kicks stay separate
			// passes: kick (+ sums, betas on the device) -> scale -> kick+drift   (Simulation.cpp:1099-1131)
			// (the scaling itself is folded into the kick + drift pass, with the betas the kick's reduction left on the device)
			// In list mode the force pass of step s-1 has done the post-force kick and the kinetic sum (betas on the device).
			if (!list_kick && (rc = ls1hip_kick(c, 0.5 * dt, nullptr, nullptr, nullptr, nullptr))) return rc;
			if ((rc = kick_drift_impl(c, dt, 2, 1., 1.))) return rc;
		} else if (list_kick) {
			if ((rc = ls1hip_kick_drift(c, dt))) return rc;  // (post-force kick already done by the list force pass)
		} else {
			// post-force kick of step s-1 fused with the pre-force kick+drift of step s (same F, one pass)
			if ((rc = ls1hip_kick_then_kick_drift(c, dt))) return rc;
		}
		advanced = fuse && s + 1 < nsteps;
		if (verlet) {
			// the lists, the binning and the halo slots live until the displacement bound of the molecules (accumulated on
			// the device by whichever pass drifts them) exceeds skin / 2; then re-bin, regenerate the halo, rebuild the lists
			if ((rc = ls1hip_update(c, nullptr))) return rc;
			// unfused steps (the last one; every step of an NVT run): the pass still does the post-force kick + sum m v^2
			rc = advanced ? ls1hip_forces_list(c, 0, dt, nullptr, nullptr)
						  : forces_list_impl(c, 0, list_kick ? dt : 0., list_kick, nullptr, nullptr);
		} else if (c->opt_overlap_halo == 2) {
			if ((rc = ls1hip_rebin(c))) return rc;
			// halo first, then the inner and the boundary cells as two passes of the same stream
			if ((rc = ls1hip_halo(c))) return rc;
			rc = advanced ? ls1hip_forces_kick_drift(c, 1, dt, nullptr, nullptr) : ls1hip_forces(c, 1, nullptr, nullptr);
			if (rc) return rc;
			rc = advanced ? ls1hip_forces_kick_drift(c, 2, dt, nullptr, nullptr) : ls1hip_forces(c, 2, nullptr, nullptr);
		} else if (c->opt_overlap_halo) {
			if ((rc = ls1hip_rebin(c))) return rc;
			// inner-cell pass first (it needs the owned molecules only); the periodic images are generated and sorted
			// on the second stream while it runs; the boundary pass waits for them on the device
			rc = advanced ? ls1hip_forces_kick_drift(c, 1, dt, nullptr, nullptr) : ls1hip_forces(c, 1, nullptr, nullptr);
			if (rc || (rc = ls1hip_halo(c))) return rc;
			rc = advanced ? ls1hip_forces_kick_drift(c, 2, dt, nullptr, nullptr) : ls1hip_forces(c, 2, nullptr, nullptr);
		} else {
			if ((rc = ls1hip_rebin(c))) return rc;
			if ((rc = ls1hip_halo(c))) return rc;
			rc = advanced ? ls1hip_forces_kick_drift(c, 0, dt, nullptr, nullptr) : ls1hip_forces(c, 0, nullptr, nullptr);
		}
		if (rc) return rc;
		if (s + 1 == nsteps) {
			c->log_row_kin = c->log_row;
			if (!list_kick && (rc = ls1hip_kick(c, 0.5 * dt, nullptr, nullptr, nullptr, nullptr))) return rc;
			if (c->thermostat_on) {
				TimedScope ts(c, c->t_integrate);
				launch_scale(integ_args(c, 0.), 1., 1., true, c->stream);
			}
		}
	}
	if (verlet && c->pos_x) {
		// leave the state where every other entry point expects it: positions (owned + halo) in mol[cur] — the second position
		// buffer has the size of the first, so the two simply trade places (a copy cost 0.9 ms per call at 10^8 molecules)
		MolSoA& m = c->mol[c->cur];
		if (c->pos_x == c->alt_x) {
			std::swap(m.x, c->alt_x);
			std::swap(m.y, c->alt_y);
			std::swap(m.z, c->alt_z);
		} else {
			const uint32_t n = (uint32_t)(c->n_real + c->cap_halo);
			launch_pack_copy(m.x, c->pos_x, n, c->stream);
			launch_pack_copy(m.y, c->pos_y, n, c->stream);
			launch_pack_copy(m.z, c->pos_z, n, c->stream);
			HIPCHK(c, hipGetLastError());
		}
		c->pos_x = c->pos_y = c->pos_z = nullptr;
	}
	c->steplog_steps = nsteps;
	int rc = sync_counters(c);
	if (rc) return rc;
	if (out6) {
		macro_to_upot_virial(c->h_cnt, &out6[0], &out6[1]);
		out6[2] = c->h_cnt->kin[0];
		out6[3] = c->h_cnt->kin[1];
		out6[4] = (double)c->h_cnt->kin_n;
		out6[5] = (double)c->h_cnt->kin_rotdof;
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_run_log(ls1hip_ctx* c, size_t cap_rows, double* rows, size_t* nrows) {
	if (!c) return LS1HIP_EINVAL;
	const size_t have = std::min<size_t>(c->steplog_steps, STEPLOG_ROWS);
	if (nrows) *nrows = have;
	if (!rows || have == 0) return LS1HIP_OK;
	REQUIRE(c, cap_rows >= have, "buffer too small: %zu < %zu rows", cap_rows, have);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	// oldest row first: the log is a ring over the step number (at most two contiguous runs)
	const size_t first = (c->steplog_steps - have) % STEPLOG_ROWS;
	const size_t n1 = std::min(have, STEPLOG_ROWS - first);
	HIPCHK(c, hipMemcpy(rows, c->d_steplog + 6 * first, n1 * 6 * sizeof(double), hipMemcpyDeviceToHost));
	if (have > n1) HIPCHK(c, hipMemcpy(rows + 6 * n1, c->d_steplog, (have - n1) * 6 * sizeof(double), hipMemcpyDeviceToHost));
	return LS1HIP_OK;
}

