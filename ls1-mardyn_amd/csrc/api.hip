// api.hip — the C ABI of libls1hip (include/ls1hip.h), part 1: context and options, parameter-table derivation, domain and
// cell grid, device memory and upload / download of the molecule set, timers.  Host C++17; every entry point cites the reference
// interface it replaces in the header.  The pieces of a time step live in api_step.hip, the multi-rank exchange, the long-range
// correction and seam A in api_exchange.hip; shared helpers in api_internal.hpp.
#include "api_internal.hpp"

static thread_local std::string g_create_err = "";

// ---- lifetime ------------------------------------------------------------------------------------------------------
extern "C" const char* ls1hip_version(void) { return (&ls1hip_variant_marker != nullptr) ? "ls1hip 0.1 gfx950 +variant" : "ls1hip 0.1 gfx950"; }

extern "C" const char* ls1hip_last_error(const ls1hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int ls1hip_create(int device, ls1hip_ctx** out) {
	if (!out) return LS1HIP_EINVAL;
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) {
		g_create_err = std::string("no HIP device available: ") + hipGetErrorString(e);
		return LS1HIP_ENODEV;
	}
	if (device < 0 || device >= ndev) {
		g_create_err = "device index out of range";
		return LS1HIP_EINVAL;
	}
	ls1hip_ctx* c = new ls1hip_ctx();
	c->device = device;
	memset(c->mol, 0, sizeof(c->mol));
	memset(&c->frc, 0, sizeof(c->frc));
	memset(&c->hs, 0, sizeof(c->hs));
	memset(&c->h_ct, 0, sizeof(c->h_ct));
	for (int i = 0; i < 27; ++i) c->nbr[i] = -1;
	int prio_lo = 0, prio_hi = 0;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
		(e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi)) != hipSuccess ||
		(e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi)) != hipSuccess ||
		(e = hipEventCreateWithFlags(&c->ev_owned, hipEventDisableTiming)) != hipSuccess ||
		(e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming)) != hipSuccess ||
		(e = hipMalloc((void**)&c->d_ct, sizeof(CompTable))) != hipSuccess ||
		(e = hipMalloc((void**)&c->d_cnt, sizeof(DevCounters))) != hipSuccess ||
		(e = hipMalloc((void**)&c->d_stage, 128 * 4 * sizeof(double))) != hipSuccess ||
		(e = hipHostMalloc((void**)&c->h_cnt, sizeof(DevCounters))) != hipSuccess ||
		(e = hipMemset(c->d_cnt, 0, sizeof(DevCounters))) != hipSuccess) {
		g_create_err = std::string("context setup failed: ") + hipGetErrorString(e);
		delete c;
		return LS1HIP_EHIP;
	}
	*out = c;
	return LS1HIP_OK;
}

extern "C" int ls1hip_destroy(ls1hip_ctx* c) {
	if (!c) return LS1HIP_OK;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream);
	free_mol(c);
	free_cells(c);
	for (int k = 0; k < 5; ++k) dfree(c->brick_lists.d[k]);
	dfree(c->d_ct);
	dfree(c->d_cnt);
	dfree(c->d_stage);
	dfree(c->d_steplog);
	if (c->d_ingest) hipFree(c->d_ingest);
	if (c->h_cnt) hipHostFree(c->h_cnt);
	if (c->h_flag) hipHostFree((void*)c->h_flag);
	timer_free(c->t_force); timer_free(c->t_integrate); timer_free(c->t_rebin); timer_free(c->t_halo); timer_free(c->t_build);
	hipStreamDestroy(c->stream);
	if (c->stream2) hipStreamDestroy(c->stream2);
	if (c->ev_owned) hipEventDestroy(c->ev_owned);
	if (c->ev_halo) hipEventDestroy(c->ev_halo);
	if (c->ev_mark) hipEventDestroy(c->ev_mark);
	if (c->h_mark) hipHostFree(c->h_mark);
	delete c;
	return LS1HIP_OK;
}

// the fused force -> kick -> drift pass exists for the single-centre LJ brick kernels only
bool can_fuse(const ls1hip_ctx* c) {
	return c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi && !c->opt_count_pairs && !c->thermostat_on &&
		   (c->g.hw == 1 || c->g.hw == 2);
}

// rigid bodies of ONE component under pair-stream lists: the list pass integrates its own molecules (kernels_force_mslist.hip,
// leapfrog_body.hpp).  Several components: measured, no gain — the groups go through the slot map there, the epilogue's 8-byte
// accesses are scattered over a window of 1024 molecules and cost what the separate (coalesced) integrator pass costs.
bool can_fuse_ms(const ls1hip_ctx* c) {
	return c->have_comp && !c->one_clj && c->h_ct.has_rot && c->h_ct.ncomp == 1 && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi &&
		   !c->opt_count_pairs && !c->thermostat_on && !c->has_remote;
}

// ... and the same sets may have the post-force kick + kinetic sums of a step folded into the list pass (thermostat or not)
bool can_list_kick_ms(const ls1hip_ctx* c) {
	return c->have_comp && !c->one_clj && c->h_ct.has_rot && c->h_ct.ncomp == 1 && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi &&
		   !c->opt_count_pairs && !c->has_remote;
}

extern "C" int ls1hip_set_option(ls1hip_ctx* c, const char* name, long v) {
	if (!c || !name) return LS1HIP_EINVAL;
	std::string n(name);
	if (n == "force_kernel") {
		REQUIRE(c, v >= 0 && v <= 4, "force_kernel must be 0..4 (LS1HIP_FK_*)");
		c->opt_force_kernel = v;
	} else if (n == "cells_in_cutoff") {
		REQUIRE(c, v == 1 || v == 2, "cells_in_cutoff must be 1 or 2");
		REQUIRE(c, !c->have_domain, "cells_in_cutoff must be set before ls1hip_set_domain");
		c->opt_cic = v;
	} else if (n == "compute_vi") {
		c->opt_vi = v ? 1 : 0;
	} else if (n == "deterministic") {
		c->opt_det = v ? 1 : 0;
	} else if (n == "count_pairs") {
		c->opt_count_pairs = v ? 1 : 0;
	} else if (n == "fuse_integration") {
		c->opt_fuse = v ? 1 : 0;
	} else if (n == "overlap_halo") {
		REQUIRE(c, v >= 0 && v <= 2, "overlap_halo must be 0, 1 or 2");
		c->opt_overlap_halo = v;
	} else if (n == "precision") {
		REQUIRE(c, v >= 0 && v <= 2, "precision must be 0 (FP64), 1 (SPDP) or 2 (SPSP)");
		c->opt_precision = v;
	} else if (n == "local_rebuild") {
		c->opt_local_rebuild = v ? 1 : 0;
	} else if (n == "lj_split") {
		REQUIRE(c, v == 0 || v == 1 || v == 2 || v == 4 || v == 5 || v == 6, "lj_split must be 0 (auto), 1, 2 (list kernel lanes per molecule), 4, 5 or 6 (MFMA pre-filter variants)");
		c->opt_lj_split = v;
	} else {
		FAIL(c, LS1HIP_EINVAL, "unknown option '%s'", name);
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_get_option(const ls1hip_ctx* c, const char* name, long* v) {
	if (!c || !name || !v) return LS1HIP_EINVAL;
	std::string n(name);
	if (n == "force_kernel") *v = c->opt_force_kernel;
	else if (n == "cells_in_cutoff") *v = c->opt_cic;
	else if (n == "compute_vi") *v = c->opt_vi;
	else if (n == "deterministic") *v = c->opt_det;
	else if (n == "count_pairs") *v = c->opt_count_pairs;
	else if (n == "lj_split") *v = c->opt_lj_split;
	else if (n == "fuse_integration") *v = c->opt_fuse;
	else if (n == "overlap_halo") *v = c->opt_overlap_halo;
	else if (n == "precision") *v = c->opt_precision;
	else if (n == "local_rebuild") *v = c->opt_local_rebuild;
	else if (n == "verlet_irregular_bricks") {
		// bricks of the last list build that do NOT run the production path of the list force pass (region beyond the LDS staging
		// area, list overflow, more owned molecules than the tiles cover): read back from the device on request, any precision
		*v = 0;
		if (c->vl_ready && c->d_cnt) {
			uint32_t k = 0;
			if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
				hipMemcpy(&k, &c->d_cnt->vl_irregular, sizeof(k), hipMemcpyDeviceToHost) != hipSuccess)
				return LS1HIP_EHIP;
			*v = (long)k;
		}
	}
	else if (n == "verlet_mean_words_x1000") {
		// diagnostics: mean number of list words (4 entries each) a tile of the last single-centre list build walks = its longest
		// list, over the tiles that hold molecules; x 1000
		*v = 0;
		if (c->vl_ready && c->one_clj && c->d_vl_nw && c->vl_tiles_cap) {
			std::vector<uint8_t> h(c->vl_tiles_cap);
			if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
				hipMemcpy(h.data(), c->d_vl_nw, h.size(), hipMemcpyDeviceToHost) != hipSuccess)
				return LS1HIP_EHIP;
			// tiles 0..7 of every brick always hold molecules in a liquid; tile 8 a few, tile 9 none: count tiles 0..7 only
			double sum = 0.;
			size_t cnt = 0;
			for (size_t t = 0; t < h.size(); ++t)
				if (t % 10 < 8 && h[t] != 0xff && h[t] != 0) {
					sum += h[t];
					++cnt;
				}
			*v = cnt ? (long)(1000. * sum / (double)cnt) : 0;
		}
	} else if (n == "precision_in_use") *v = (c->opt_precision && c->vl_ready && c->vl_all_regular) ? c->opt_precision : 0;
	else if (n == "can_fuse_integration") *v = can_fuse(c) ? 1 : 0;
	else if (n == "can_fuse_rigid_lists") *v = (c->vl_on && can_fuse_ms(c)) ? 1 : 0;
	else if (n == "list_kick_available") *v = (c->vl_ready && (c->one_clj || can_list_kick_ms(c))) ? 1 : 0;
	else if (n == "verlet_bound_pending") *v = c->vl_bound_pending ? 1 : 0;  // a drift since the last poll / build: ls1hip_verlet_poll may be asked
	else if (n == "last_force_kernel") *v = c->last_force_kernel;
	else if (n == "build_variant") *v = (&ls1hip_variant_marker != nullptr) ? 1 : 0;  // 0 = the regular build (no timing-variant object inside)
	else if (n == "verlet_lists") *v = c->vl_on ? 1 : 0;
	else if (n == "verlet_ready") *v = c->vl_ready ? 1 : 0;
	else if (n == "verlet_builds") *v = (long)c->vl_builds;
	else if (n == "verlet_steps") *v = (long)c->vl_steps;
	else return LS1HIP_EINVAL;
	return LS1HIP_OK;
}

// ---- model ---------------------------------------------------------------------------------------------------------
extern "C" int ls1hip_set_components(ls1hip_ctx* c, int ncomp, const int* nlj, const int* nc, const int* nd,
									 const int* nq, const double* lj, const double* ch, const double* dp,
									 const double* qp, const double* mass, const double* I, const double* mix,
									 double eps_rf, double rc, double rc_lj) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, ncomp >= 1 && ncomp <= MAXC, "ncomp must be in 1..%d", MAXC);
	REQUIRE(c, nlj && nc && nd && nq && mass && I, "null component arrays");
	REQUIRE(c, rc > 0. && rc_lj > 0., "cutoffs must be positive");
	CompTable& t = c->h_ct;
	memset(&t, 0, sizeof(t));
	t.ncomp = ncomp;
	int tl = 0, tc = 0, td = 0, tq = 0;
	bool rot = false;
	for (int k = 0; k < ncomp; ++k) {
		REQUIRE(c, nlj[k] >= 0 && nc[k] >= 0 && nd[k] >= 0 && nq[k] >= 0, "negative site count");
		t.nlj[k] = nlj[k]; t.nc[k] = nc[k]; t.nd[k] = nd[k]; t.nq[k] = nq[k];
		t.olj[k] = tl; t.oc[k] = tc; t.od[k] = td; t.oq[k] = tq;
		tl += nlj[k]; tc += nc[k]; td += nd[k]; tq += nq[k];
		t.maxsites = std::max(t.maxsites, nlj[k] + nc[k] + nd[k] + nq[k]);
		t.mass[k] = mass[k];
		int rdof = 0;
		for (int d = 0; d < 3; ++d) {
			t.I[k][d] = I[3 * k + d];
			t.invI[k][d] = (I[3 * k + d] != 0.) ? 1. / I[3 * k + d] : 0.;  // FullMolecule.h:88-101
			if (I[3 * k + d] != 0.) ++rdof;
		}
		t.rotdof[k] = rdof;
	}
	REQUIRE(c, tl <= MAXS && tc <= MAXS && td <= MAXS && tq <= MAXS, "more than %d sites of one type", MAXS);
	REQUIRE(c, (tl == 0 || lj) && (tc == 0 || ch) && (td == 0 || dp) && (tq == 0 || qp), "null site table");
	REQUIRE(c, ncomp == 1 || mix, "mixing coefficients required for more than one component");
	for (int k = 0; k < tl; ++k) {
		const double* s = lj + (size_t)k * LS1HIP_LJ_STRIDE;
		for (int d = 0; d < 3; ++d) {
			t.ljpos[k][d] = s[d];
			rot |= (s[d] != 0.);
		}
	}
	for (int k = 0; k < tc; ++k) {
		const double* s = ch + (size_t)k * LS1HIP_CH_STRIDE;
		for (int d = 0; d < 3; ++d) {
			t.chpos[k][d] = s[d];
			rot |= (s[d] != 0.);
		}
		t.chq[k] = s[4];
	}
	for (int k = 0; k < td; ++k) {
		const double* s = dp + (size_t)k * LS1HIP_DP_STRIDE;
		for (int d = 0; d < 3; ++d) {
			t.dppos[k][d] = s[d];
			t.dpe[k][d] = s[3 + d];
		}
		t.dpmy[k] = s[6];
		rot = true;
	}
	for (int k = 0; k < tq; ++k) {
		const double* s = qp + (size_t)k * LS1HIP_QP_STRIDE;
		for (int d = 0; d < 3; ++d) {
			t.qppos[k][d] = s[d];
			t.qpe[k][d] = s[3 + d];
		}
		t.qpQ[k] = s[6];
		rot = true;
	}
	t.has_rot = rot ? 1 : 0;
	t.ncenters = tl;
	// LJ pair table: Comp2Param::initialize (Comp2Param.cpp:17-97) laid out like VectorizedCellProcessor's
	// _eps_sig / _shift6 (VectorizedCellProcessor.cpp:41-83) on the global centre numbering.
	int mixpos = 0;
	for (int ci = 0; ci < ncomp; ++ci) {
		for (int a = 0; a < nlj[ci]; ++a) {
			const double* sa = lj + (size_t)(t.olj[ci] + a) * LS1HIP_LJ_STRIDE;
			for (int b = 0; b < nlj[ci]; ++b) {
				const double* sb = lj + (size_t)(t.olj[ci] + b) * LS1HIP_LJ_STRIDE;
				const int k = (t.olj[ci] + a) * tl + (t.olj[ci] + b);
				double sg = .5 * (sa[5] + sb[5]);
				t.eps24[k] = 24. * sqrt(sa[4] * sb[4]);
				t.sig2[k] = sg * sg;
				t.shift6[k] = sa[6];
			}
		}
		for (int cj = ci + 1; cj < ncomp; ++cj) {
			const double xi = mix[2 * mixpos], eta = mix[2 * mixpos + 1];
			++mixpos;
			for (int a = 0; a < nlj[ci]; ++a) {
				const double* sa = lj + (size_t)(t.olj[ci] + a) * LS1HIP_LJ_STRIDE;
				for (int b = 0; b < nlj[cj]; ++b) {
					const double* sb = lj + (size_t)(t.olj[cj] + b) * LS1HIP_LJ_STRIDE;
					const double e24 = 24. * xi * sqrt(sa[4] * sb[4]);
					double sg = eta * .5 * (sa[5] + sb[5]);
					const double sg2 = sg * sg;
					const double p2 = sg2 / (rc_lj * rc_lj);
					const double p6 = p2 * p2 * p2;
					const double sh = e24 * (p6 - p6 * p6);
					const int kij = (t.olj[ci] + a) * tl + (t.olj[cj] + b);
					const int kji = (t.olj[cj] + b) * tl + (t.olj[ci] + a);
					t.eps24[kij] = t.eps24[kji] = e24;
					t.sig2[kij] = t.sig2[kji] = sg2;
					t.shift6[kij] = t.shift6[kji] = sh;
				}
			}
		}
	}
	t.rc2 = rc * rc;
	t.rclj2 = rc_lj * rc_lj;
	t.epsRFInvrc3 = 2. * (eps_rf - 1.) / ((rc * rc * rc) * (2. * eps_rf + 1.));  // VectorizedCellProcessor.cpp:24
	c->rc = rc;
	c->rc_lj = rc_lj;
	c->rc_list = rc + (c->vl_on ? c->vl_skin : 0.);
	c->one_clj = (ncomp == 1 && tl == 1 && tc == 0 && td == 0 && tq == 0 && !rot && rc == rc_lj);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(c->d_ct, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->have_comp = true;
	c->have_domain = false;  // the cell grid depends on the cutoff
	return LS1HIP_OK;
}

// Component::getRotationalDegreesOfFreedom per component, where it differs from the count of non-zero principal moments that
// ls1hip_set_components derives: the reference counts the moments computed from the SITE masses (Component.cpp:140-167), the
// I line of an .inp / <momentsofinertia> of the XML may then override their values (ASCIIReader.cpp:208-212, Component.cpp:88-97)
// without changing that count.  Enters the thermostat's degree-of-freedom sums only (Leapfrog.cpp:100,126), never the motion.
extern "C" int ls1hip_set_rot_dof(ls1hip_ctx* c, int ncomp, const int* rot_dof) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp, "ls1hip_set_components must be called first");
	REQUIRE(c, rot_dof && ncomp == c->h_ct.ncomp, "one value per component of the current set (%d)", c->h_ct.ncomp);
	for (int k = 0; k < ncomp; ++k) REQUIRE(c, rot_dof[k] >= 0 && rot_dof[k] <= 3, "rotational degrees of freedom must be in 0..3");
	for (int k = 0; k < ncomp; ++k) c->h_ct.rotdof[k] = rot_dof[k];
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpyAsync(c->d_ct, &c->h_ct, sizeof(c->h_ct), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return LS1HIP_OK;
}

extern "C" int ls1hip_get_lj_table(const ls1hip_ctx* c, int* ncenters, double* eps24, double* sig2, double* shift6) {
	if (!c || !c->have_comp) return LS1HIP_EINVAL;
	const int n = c->h_ct.ncenters;
	if (ncenters) *ncenters = n;
	if (eps24) memcpy(eps24, c->h_ct.eps24, sizeof(double) * n * n);
	if (sig2) memcpy(sig2, c->h_ct.sig2, sizeof(double) * n * n);
	if (shift6) memcpy(shift6, c->h_ct.shift6, sizeof(double) * n * n);
	return LS1HIP_OK;
}

extern "C" int ls1hip_set_domain(ls1hip_ctx* c, const double global_len[3], const double box_min[3],
								 const double box_max[3], int my_rank, const int neighbor_rank[27]) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp, "ls1hip_set_components must be called first");
	REQUIRE(c, global_len && box_min && box_max && neighbor_rank, "null argument");
	for (int d = 0; d < 3; ++d)
		REQUIRE(c, box_max[d] > box_min[d] && box_min[d] >= 0. && box_max[d] <= global_len[d], "bad bounding box");
	Grid g;
	int align[3];
	verlet_brick_shape(align);
	const bool lists = c->vl_on && c->opt_cic == 1;
	if (!grid_init(g, box_min, box_max, lists ? c->rc_list : c->rc, (int)c->opt_cic, lists ? align : nullptr))
		FAIL(c, LS1HIP_EINVAL, "LinkedCells: region too small for the cutoff (or too many cells)");
	c->g = g;
	c->my_rank = my_rank;
	c->has_remote = false;
	for (int d = 0; d < 3; ++d) {
		c->dom_min[d] = box_min[d];
		c->dom_max[d] = box_max[d];
	}
	for (int d = 0; d < 3; ++d) c->global_len[d] = global_len[d];
	for (int sz = -1; sz <= 1; ++sz)
		for (int sy = -1; sy <= 1; ++sy)
			for (int sx = -1; sx <= 1; ++sx) {
				const int dir = (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1);
				c->nbr[dir] = neighbor_rank[dir];
				if (dir != 13 && neighbor_rank[dir] >= 0 && neighbor_rank[dir] != my_rank) c->has_remote = true;
				const int s[3] = {sx, sy, sz};
				for (int d = 0; d < 3; ++d) {
					// a copy sent through the low face of the GLOBAL box reappears shifted by +L, through the high
					// face by -L (DomainDecompBase.cpp:180-182,299-301); inside the global box no shift
					double sh = 0.;
					if (s[d] < 0 && box_min[d] == 0.) sh = global_len[d];
					if (s[d] > 0 && box_max[d] == global_len[d]) sh = -global_len[d];
					c->shift[dir][d] = sh;
				}
			}
	c->nbr[13] = my_rank;
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->d_shift27) {
		int rs = dalloc(c, &c->d_shift27, 81);
		if (rs) return rs;
	}
	HIPCHK(c, hipMemcpy(c->d_shift27, &c->shift[0][0], 81 * sizeof(double), hipMemcpyHostToDevice));
	const size_t nc = (size_t)g.ncells;
	if (nc > c->cells_alloc) {
		free_cells(c);
		int rc;
		if ((rc = dalloc(c, &c->d_count, nc)) || (rc = dalloc(c, &c->d_cell_begin, nc)) ||
			(rc = dalloc(c, &c->d_cell_end, nc)) || (rc = dalloc(c, &c->d_blocksum, nc / 1024 + 2)))
			return rc;
		c->cells_alloc = nc;
	}
	{
		// owned cells within 2*hw of a face: the only cells whose molecules can lie within rc of the boundary
		std::vector<uint32_t> shell;
		const int hw = g.hw;
		for (int cz = hw; cz < g.dims[2] - hw; ++cz)
			for (int cy = hw; cy < g.dims[1] - hw; ++cy)
				for (int cx = hw; cx < g.dims[0] - hw; ++cx)
					if (cx < 2 * hw || cy < 2 * hw || cz < 2 * hw || cx >= g.dims[0] - 2 * hw || cy >= g.dims[1] - 2 * hw ||
						cz >= g.dims[2] - 2 * hw)
						shell.push_back((uint32_t)cell_index(g, cx, cy, cz));
		dfree(c->d_shell);
		int rc = dalloc(c, &c->d_shell, shell.size());
		if (rc) return rc;
		c->n_shell = (uint32_t)shell.size();
		if (!shell.empty())
			HIPCHK(c, hipMemcpy(c->d_shell, shell.data(), shell.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	HIPCHK(c, hipMemsetAsync(c->d_count, 0, nc * sizeof(uint32_t), c->stream));
	HIPCHK(c, hipMemsetAsync(c->d_cell_begin, 0, nc * sizeof(uint32_t), c->stream));
	HIPCHK(c, hipMemsetAsync(c->d_cell_end, 0, nc * sizeof(uint32_t), c->stream));
	c->have_domain = true;
	c->binned = false;
	c->halo_valid = false;
	c->forces_valid = false;
	c->msl_pk_fresh = false;
	return LS1HIP_OK;
}

extern "C" int ls1hip_get_grid(const ls1hip_ctx* c, int dims[3], double cell_len[3], int* halo_width) {
	if (!c || !c->have_domain) return LS1HIP_EINVAL;
	for (int d = 0; d < 3; ++d) {
		if (dims) dims[d] = c->g.dims[d];
		if (cell_len) cell_len[d] = c->g.clen[d];
	}
	if (halo_width) *halo_width = c->g.hw;
	return LS1HIP_OK;
}

// ---- molecules -----------------------------------------------------------------------------------------------------
static int alloc_mol(ls1hip_ctx* c, size_t n) {
	const bool rot = c->h_ct.has_rot;
	// capacities: owned molecules may grow by migration; halo copies from the geometry of the rc-shell
	double vol = 1., vol_out = 1.;
	for (int d = 0; d < 3; ++d) {
		const double L = c->g.bmax[d] - c->g.bmin[d];
		vol *= L;
		vol_out *= L + 2. * c->rc_list;
	}
	const double dens = (double)n / vol;
	size_t cap_real = (size_t)(n * (c->has_remote ? 1.25 : 1.0)) + 1024;
	size_t cap_halo = (size_t)(dens * (vol_out - vol) * 1.6) + 4096;
	cap_halo = std::min(cap_halo, 26 * n + 4096);
	// per-workgroup partial sums: the generic kernels launch cap_real / 128 workgroups, the brick kernels one per brick;
	// the smallest brick is 8 cells (1x4x2 or 2x2x2), rounded up per dimension, + 16 for the XCD-aligned grid.  (Sized by
	// molecules only, small / sparse boxes silently fell back to the generic kernel: found by the random-box sweep.)
	const size_t max_bricks = (size_t)c->g.box[0] * ((c->g.box[1] + 1) / 2) * ((c->g.box[2] + 1) / 2) + 16;
	const size_t partials_cap = std::max(cap_real / 32 + 16, max_bricks);  // generic kernel: up to 4 lanes per molecule, 128 threads
	if (cap_real <= c->cap_real && cap_halo <= c->cap_halo && partials_cap <= c->partials_cap) return 0;
	free_mol(c);
	const size_t tot = cap_real + cap_halo;
	int rc = 0;
	for (int k = 0; k < 2 && !rc; ++k) {
		MolSoA& m = c->mol[k];
		// set 0/1 both carry the halo segment (the rebin flips between them)
		(rc = dalloc(c, &m.x, tot)) || (rc = dalloc(c, &m.y, tot)) || (rc = dalloc(c, &m.z, tot)) ||
			(rc = dalloc(c, &m.vx, cap_real)) || (rc = dalloc(c, &m.vy, cap_real)) || (rc = dalloc(c, &m.vz, cap_real)) ||
			(rc = dalloc(c, &m.id, tot)) || (rc = dalloc(c, &m.cid, tot));
		if (!rc && rot)
			(rc = dalloc(c, &m.q0, tot)) || (rc = dalloc(c, &m.q1, tot)) || (rc = dalloc(c, &m.q2, tot)) ||
				(rc = dalloc(c, &m.q3, tot)) || (rc = dalloc(c, &m.Dx, cap_real)) || (rc = dalloc(c, &m.Dy, cap_real)) ||
				(rc = dalloc(c, &m.Dz, cap_real));
	}
	if (rc) return rc;
	ForceSoA& f = c->frc;
	(rc = dalloc(c, &f.Fx, cap_real)) || (rc = dalloc(c, &f.Fy, cap_real)) || (rc = dalloc(c, &f.Fz, cap_real)) ||
		(rc = dalloc(c, &f.Vix, cap_real)) || (rc = dalloc(c, &f.Viy, cap_real)) || (rc = dalloc(c, &f.Viz, cap_real));
	if (!rc && rot) (rc = dalloc(c, &f.Mx, cap_real)) || (rc = dalloc(c, &f.My, cap_real)) || (rc = dalloc(c, &f.Mz, cap_real));
	if (rc) return rc;
	HaloStage& h = c->hs;
	(rc = dalloc(c, &h.x, cap_halo)) || (rc = dalloc(c, &h.y, cap_halo)) || (rc = dalloc(c, &h.z, cap_halo)) ||
		(rc = dalloc(c, &h.id, cap_halo)) || (rc = dalloc(c, &h.cid, cap_halo)) || (rc = dalloc(c, &h.key, cap_halo)) ||
		(rc = dalloc(c, &h.rank, cap_halo)) || (rc = dalloc(c, &h.src, cap_halo)) || (rc = dalloc(c, &h.dir, cap_halo)) ||
		(rc = dalloc(c, &c->d_halo_src, cap_halo)) || (rc = dalloc(c, &c->d_halo_dir, cap_halo));
	if (!rc && c->vl_on) (rc = dalloc(c, &c->alt_x, tot)) || (rc = dalloc(c, &c->alt_y, tot)) || (rc = dalloc(c, &c->alt_z, tot));
	if (!rc && rot)
		(rc = dalloc(c, &h.q0, cap_halo)) || (rc = dalloc(c, &h.q1, cap_halo)) || (rc = dalloc(c, &h.q2, cap_halo)) ||
			(rc = dalloc(c, &h.q3, cap_halo));
	if (rc) return rc;
	(rc = dalloc(c, &c->d_key, cap_real)) || (rc = dalloc(c, &c->d_rank, cap_real)) ||
		(rc = dalloc(c, &c->d_perm, std::max(cap_real, cap_halo))) || (rc = dalloc(c, &c->d_ckey, cap_real)) ||
		(rc = dalloc(c, &c->d_idk, std::max(cap_real, cap_halo)));
	if (rc) return rc;
	c->partials_cap = partials_cap;
	if ((rc = dalloc(c, &c->d_partials, c->partials_cap * 4))) return rc;
	// export slices per remote direction, sized from the geometry of the region that feeds the direction
	uint32_t offL = 0, offH = 0;
	for (int sz = -1; sz <= 1; ++sz)
		for (int sy = -1; sy <= 1; ++sy)
			for (int sx = -1; sx <= 1; ++sx) {
				const int dir = (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1);
				c->exp_off_leave[dir] = offL;
				c->exp_off_halo[dir] = offH;
				if (dir == 13 || c->nbr[dir] < 0 || c->nbr[dir] == c->my_rank) continue;
				const int s[3] = {sx, sy, sz};
				double v = 1.;
				for (int d = 0; d < 3; ++d) v *= (s[d] != 0) ? c->rc_list : (c->g.bmax[d] - c->g.bmin[d]);
				const uint32_t cap = (uint32_t)(dens * v * 1.6) + 2048;
				offH += cap;
				offL += cap / 2 + 1024;
			}
	c->exp_off_leave[27] = offL;
	c->exp_off_halo[27] = offH;
	if ((rc = dalloc(c, &c->d_exp_leave, (size_t)offL * LS1HIP_LEAVING_DOUBLES)) ||
		(rc = dalloc(c, &c->d_exp_halo, (size_t)offH * LS1HIP_HALO_DOUBLES)) || (rc = dalloc(c, &c->d_exp_halo_src, (size_t)offH)) ||
		(rc = dalloc(c, &c->d_exp_refresh, (size_t)offH * 3)) || (rc = dalloc(c, &c->d_imp_slot, cap_halo)) ||
		(rc = dalloc(c, &c->d_s2s, cap_halo)))
		return rc;
	c->cap_real = cap_real;
	c->cap_halo = cap_halo;
	return 0;
}

// ---- streaming upload: chunks are copied raw into a device staging buffer and transposed into the SoA on the device ----

static IngestArgs ingest_args(ls1hip_ctx* c, size_t n) {
	IngestArgs a;
	a.dst = c->mol[0];
	a.has_rot = c->h_ct.has_rot;
	a.ncomp = c->h_ct.ncomp;
	for (int d = 0; d < 3; ++d) {
		a.bmin[d] = c->g.bmin[d];
		a.bmax[d] = c->g.bmax[d];
	}
	a.cnt = c->d_cnt;
	a.at = (uint32_t)c->ingest_at;
	a.first = (uint32_t)c->ingest_at;
	a.n = (uint32_t)n;
	return a;
}

extern "C" int ls1hip_upload_begin(ls1hip_ctx* c, size_t n) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_domain, "ls1hip_set_domain must be called first");
	REQUIRE(c, n < 0x7fff0000ull, "too many molecules for 32-bit indices");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	int rc = alloc_mol(c, n);
	if (rc) return rc;
	const size_t bytes = std::min(std::max<size_t>(n, 1), INGEST_CHUNK) * 116 + 64;
	if (bytes > c->ingest_bytes) {
		if (c->d_ingest) hipFree(c->d_ingest);
		c->d_ingest = nullptr;
		c->ingest_bytes = 0;
		hipError_t e = hipMalloc(&c->d_ingest, bytes);
		if (e != hipSuccess) FAIL(c, LS1HIP_ENOMEM, "hipMalloc(%zu bytes) for the upload staging failed: %s", bytes, hipGetErrorString(e));
		c->ingest_bytes = bytes;
	}
	c->cur = 0;
	HIPCHK(c, hipMemsetAsync(c->d_cnt, 0, sizeof(DevCounters), c->stream));
	c->ingest_open = true;
	c->ingest_total = n;
	c->ingest_at = 0;
	c->n_real = 0;
	c->n_halo = 0;
	c->binned = c->halo_valid = c->forces_valid = false;
	c->msl_pk_fresh = false;
	c->pos_x = c->pos_y = c->pos_z = nullptr;
	c->vl_ready = false;
	c->vl_bound_pending = false;
	c->fused_split = 0;
	return LS1HIP_OK;
}

extern "C" int ls1hip_upload_chunk(ls1hip_ctx* c, size_t n, const uint64_t* id, const int32_t* cid, const double* r,
								   const double* v, const double* q, const double* D) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->ingest_open, "ls1hip_upload_begin must be called first");
	REQUIRE(c, n == 0 || (id && r && v), "null molecule arrays");
	REQUIRE(c, c->ingest_at + n <= c->ingest_total, "more molecules than announced to ls1hip_upload_begin");
	HIPCHK(c, hipSetDevice(c->device));
	const bool rot = c->h_ct.has_rot;
	for (size_t o = 0; o < n; o += INGEST_CHUNK) {
		const size_t m = std::min(INGEST_CHUNK, n - o);
		char* b = (char*)c->d_ingest;
		uint64_t* did = (uint64_t*)b;
		double* dr = (double*)(b + 8 * m);
		double* dv = dr + 3 * m;
		double* dq = dv + 3 * m;
		double* dD = dq + 4 * m;
		int32_t* dc = (int32_t*)(dD + 3 * m);
		HIPCHK(c, hipMemcpyAsync(did, id + o, m * 8, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(dr, r + 3 * o, m * 24, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipMemcpyAsync(dv, v + 3 * o, m * 24, hipMemcpyHostToDevice, c->stream));
		if (rot && q) HIPCHK(c, hipMemcpyAsync(dq, q + 4 * o, m * 32, hipMemcpyHostToDevice, c->stream));
		if (rot && D) HIPCHK(c, hipMemcpyAsync(dD, D + 3 * o, m * 24, hipMemcpyHostToDevice, c->stream));
		if (cid) HIPCHK(c, hipMemcpyAsync(dc, cid + o, m * 4, hipMemcpyHostToDevice, c->stream));
		launch_ingest_aos(ingest_args(c, m), did, cid ? dc : nullptr, dr, dv, (rot && q) ? dq : nullptr, (rot && D) ? dD : nullptr,
						  c->stream);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next pass
		c->ingest_at += m;
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_upload_chunk_device(ls1hip_ctx* c, size_t n, const uint64_t* dev_id, const int32_t* dev_cid,
										  const double* dev_r, const double* dev_v, const double* dev_q, const double* dev_D) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->ingest_open, "ls1hip_upload_begin must be called first");
	REQUIRE(c, n == 0 || (dev_id && dev_r && dev_v), "null molecule arrays");
	REQUIRE(c, c->ingest_at + n <= c->ingest_total, "more molecules than announced to ls1hip_upload_begin");
	HIPCHK(c, hipSetDevice(c->device));
	const bool rot = c->h_ct.has_rot;
	launch_ingest_aos(ingest_args(c, n), dev_id, dev_cid, dev_r, dev_v, rot ? dev_q : nullptr, rot ? dev_D : nullptr, c->stream);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller may release its buffers when the call returns
	c->ingest_at += n;
	return LS1HIP_OK;
}

static size_t record_bytes(int format) { return format == LS1HIP_REC_ICRVQD ? 116 : (format == LS1HIP_REC_ICRV ? 60 : 56); }

extern "C" int ls1hip_upload_records(ls1hip_ctx* c, size_t n, const void* records, int format) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->ingest_open, "ls1hip_upload_begin must be called first");
	REQUIRE(c, format == LS1HIP_REC_ICRVQD || format == LS1HIP_REC_ICRV || format == LS1HIP_REC_IRV, "unknown record format %d", format);
	REQUIRE(c, n == 0 || records, "null record buffer");
	REQUIRE(c, c->ingest_at + n <= c->ingest_total, "more molecules than announced to ls1hip_upload_begin");
	HIPCHK(c, hipSetDevice(c->device));
	const size_t rb = record_bytes(format);
	for (size_t o = 0; o < n; o += INGEST_CHUNK) {
		const size_t m = std::min(INGEST_CHUNK, n - o);
		HIPCHK(c, hipMemcpyAsync(c->d_ingest, (const char*)records + o * rb, m * rb, hipMemcpyHostToDevice, c->stream));
		launch_ingest_records(ingest_args(c, m), c->d_ingest, format, c->stream);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipStreamSynchronize(c->stream));
		c->ingest_at += m;
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_upload_end(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->ingest_open, "ls1hip_upload_begin must be called first");
	HIPCHK(c, hipSetDevice(c->device));
	c->ingest_open = false;
	if (c->d_ingest) hipFree(c->d_ingest);  // the staging buffer is only needed while a phase space streams in
	c->d_ingest = nullptr;
	c->ingest_bytes = 0;
	const size_t n = c->ingest_at;  // the announced total is an upper bound (it sized the device arrays)
	if (c->vl_on && !c->vl_force && c->opt_cic == 1 && c->one_clj) {
		// (multi-site sets keep their lists in any case: their pair streams need no staging, kernels_force_mslist.hip)
		// Will the list kernels be able to stage a brick's region?  If the MEAN region already comes close to the capacity
		// (large skin, dense system) the per-step kernels are the better loop — and they want the reference's r_c grid,
		// not the r_c + skin one: the domain is set up again without the skin before anything is binned.
		const double ncell = (double)c->g.box[0] * c->g.box[1] * c->g.box[2];
		const double mean_region = ncell > 0. ? (double)n / ncell * verlet_region_cells() : 0.;
		if (mean_region * 1.06 > (double)verlet_region_capacity()) {
			c->vl_on = false;
			c->rc_list = c->rc;
			const double gl[3] = {c->global_len[0], c->global_len[1], c->global_len[2]};
			const double lo[3] = {c->dom_min[0], c->dom_min[1], c->dom_min[2]}, hi[3] = {c->dom_max[0], c->dom_max[1], c->dom_max[2]};
			int nbr[27];
			memcpy(nbr, c->nbr, sizeof(nbr));
			const int rc2 = ls1hip_set_domain(c, gl, lo, hi, c->my_rank, nbr);
			if (rc2) return rc2;
		}
	}
	const bool rot = c->h_ct.has_rot;
	// forces start at zero (a freshly read phase space has F = M = 0: FullMolecule.cpp:44-45)
	HIPCHK(c, hipMemsetAsync(c->frc.Fx, 0, c->cap_real * sizeof(double), c->stream));
	HIPCHK(c, hipMemsetAsync(c->frc.Fy, 0, c->cap_real * sizeof(double), c->stream));
	HIPCHK(c, hipMemsetAsync(c->frc.Fz, 0, c->cap_real * sizeof(double), c->stream));
	if (rot) {
		HIPCHK(c, hipMemsetAsync(c->frc.Mx, 0, c->cap_real * sizeof(double), c->stream));
		HIPCHK(c, hipMemsetAsync(c->frc.My, 0, c->cap_real * sizeof(double), c->stream));
		HIPCHK(c, hipMemsetAsync(c->frc.Mz, 0, c->cap_real * sizeof(double), c->stream));
	}
	HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (c->h_cnt->err_ingest)
		FAIL(c, LS1HIP_EINVAL, "%u uploaded molecule(s) lie outside the bounding box of this rank or carry a wrong component id (e.g. molecule %u)",
			 c->h_cnt->err_ingest, c->h_cnt->err_ingest_first);
	c->h_cnt->n_real = (uint32_t)n;
	HIPCHK(c, hipMemcpyAsync(&c->d_cnt->n_real, &c->h_cnt->n_real, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->n_real = n;
	return LS1HIP_OK;
}

extern "C" int ls1hip_upload(ls1hip_ctx* c, size_t n, const uint64_t* id, const int32_t* cid, const double* r,
							 const double* v, const double* q, const double* D) {
	int rc = ls1hip_upload_begin(c, n);
	if (rc) return rc;
	if ((rc = ls1hip_upload_chunk(c, n, id, cid, r, v, q, D))) {
		c->ingest_open = false;
		return rc;
	}
	return ls1hip_upload_end(c);
}

// stream of the halo phase: the second stream while an inner-cell force pass is in flight on the main one
hipStream_t halo_stream(ls1hip_ctx* c) { return c->inner_in_flight ? c->stream2 : c->stream; }

int sync_counters(ls1hip_ctx* c, hipStream_t s) {
	if (!s) s = c->stream;
	HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	c->n_halo = c->h_cnt->n_halo;
	if (c->h_cnt->err_overflow)
		FAIL(c, LS1HIP_ENOMEM, "device buffer overflow (%u records dropped): halo/export capacity exceeded", c->h_cnt->err_overflow);
	if (c->h_cnt->err_lost)
		FAIL(c, LS1HIP_ELOST, "%u molecule(s) left the halo region of this rank", c->h_cnt->err_lost);
	return LS1HIP_OK;
}

extern "C" int ls1hip_count(const ls1hip_ctx* c, size_t* n_owned, size_t* n_halo) {
	if (!c) return LS1HIP_EINVAL;
	if (n_owned) *n_owned = c->n_real;
	if (n_halo) *n_halo = c->n_halo;
	return LS1HIP_OK;
}

extern "C" int ls1hip_set_thermostat(ls1hip_ctx* c, int enabled, double target_temperature) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, !enabled || target_temperature > 0., "target temperature must be positive");
	c->thermostat_on = enabled != 0;
	c->thermostat_T = target_temperature;
	return LS1HIP_OK;
}

// ---- downloads -----------------------------------------------------------------------------------------------------
int d2h3(ls1hip_ctx* c, size_t n, const double* a, const double* b, const double* d, double* out, int stride, int o) {
	std::vector<double> t(n);
	const double* src[3] = {a, b, d};
	for (int k = 0; k < 3; ++k) {
		if (src[k]) {
			HIPCHK(c, hipMemcpy(t.data(), src[k], n * sizeof(double), hipMemcpyDeviceToHost));
			for (size_t i = 0; i < n; ++i) out[stride * i + o + k] = t[i];
		} else {
			for (size_t i = 0; i < n; ++i) out[stride * i + o + k] = 0.;
		}
	}
	return 0;
}

extern "C" int ls1hip_download_state(ls1hip_ctx* c, size_t cap, uint64_t* id, int32_t* cid, double* r, double* v,
									 double* q, double* D) {
	if (!c) return LS1HIP_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	REQUIRE(c, !c->fused_split, "state is half advanced (complete the fused which=2 pass first)");
	int rc;
	if ((rc = materialise_positions(c))) return rc;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const size_t n = c->n_real;
	REQUIRE(c, cap >= n, "buffer too small: %zu < %zu", cap, n);
	if (n == 0) return LS1HIP_OK;
	const MolSoA& m = c->mol[c->cur];
	if (id) HIPCHK(c, hipMemcpy(id, m.id, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
	if (cid) HIPCHK(c, hipMemcpy(cid, m.cid, n * sizeof(int32_t), hipMemcpyDeviceToHost));
	if (r && (rc = d2h3(c, n, m.x, m.y, m.z, r, 3, 0))) return rc;
	if (r && c->vl_on) {
		// between two rebuilds of the neighbour lists a molecule may sit up to skin / 2 outside the box: report it wrapped
		// (same rule and rounding clamps as the re-binning pass, DomainDecompBase.cpp:206-219)
		for (int k = 0; k < 3; ++k) {
			const int st = k == 0 ? 1 : (k == 1 ? 3 : 9);
			if (c->nbr[13 + st] != c->my_rank || c->nbr[13 - st] != c->my_rank) continue;
			const double lo = c->g.bmin[k], hi = c->g.bmax[k], len = c->global_len[k];
			for (size_t i = 0; i < n; ++i) {
				double& x = r[3 * i + k];
				if (x < lo) {
					x += len;
					if (x >= hi) x = std::nextafter(hi, lo);
				} else if (x >= hi) {
					x -= len;
					if (x <= lo) x = lo;
				}
			}
		}
	}
	if (v && (rc = d2h3(c, n, m.vx, m.vy, m.vz, v, 3, 0))) return rc;
	if (q) {
		if (c->h_ct.has_rot) {
			std::vector<double> t(n);
			HIPCHK(c, hipMemcpy(t.data(), m.q0, n * sizeof(double), hipMemcpyDeviceToHost));
			for (size_t i = 0; i < n; ++i) q[4 * i] = t[i];
			if ((rc = d2h3(c, n, m.q1, m.q2, m.q3, q, 4, 1))) return rc;
		} else {
			for (size_t i = 0; i < n; ++i) {
				q[4 * i] = 1.;
				q[4 * i + 1] = q[4 * i + 2] = q[4 * i + 3] = 0.;
			}
		}
	}
	if (D && (rc = d2h3(c, n, c->h_ct.has_rot ? m.Dx : nullptr, c->h_ct.has_rot ? m.Dy : nullptr,
						c->h_ct.has_rot ? m.Dz : nullptr, D, 3, 0)))
		return rc;
	return LS1HIP_OK;
}

extern "C" int ls1hip_download_records(ls1hip_ctx* c, size_t first, size_t n, void* records) {
	if (!c) return LS1HIP_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	REQUIRE(c, !c->fused_split, "state is half advanced (complete the fused which=2 pass first)");
	REQUIRE(c, first + n <= c->n_real, "record range [%zu, %zu) exceeds the %zu owned molecules", first, first + n, c->n_real);
	REQUIRE(c, n == 0 || records, "null record buffer");
	int rc;
	if ((rc = materialise_positions(c))) return rc;
	void* d = nullptr;
	const size_t chunk = std::min(std::max<size_t>(n, 1), INGEST_CHUNK);
	hipError_t e = hipMalloc(&d, chunk * 116);
	if (e != hipSuccess) FAIL(c, LS1HIP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", chunk * 116, hipGetErrorString(e));
	EgressArgs a;
	a.src = c->mol[c->cur];
	a.x = a.src.x; a.y = a.src.y; a.z = a.src.z;
	a.has_rot = c->h_ct.has_rot;
	for (int k = 0; k < 3; ++k) {
		a.bmin[k] = c->g.bmin[k];
		a.bmax[k] = c->g.bmax[k];
		a.len[k] = c->global_len[k];
		// a side is wrapped here when its image lives on this rank (single-rank periodic box)
		a.periodic[k] = c->nbr[13 + (k == 0 ? 1 : (k == 1 ? 3 : 9))] == c->my_rank && c->nbr[13 - (k == 0 ? 1 : (k == 1 ? 3 : 9))] == c->my_rank;
	}
	for (size_t o = 0; o < n && !rc; o += chunk) {
		const size_t m = std::min(chunk, n - o);
		a.first = (uint32_t)(first + o);
		a.n = (uint32_t)m;
		launch_egress_records(a, d, c->stream);
		if (hipMemcpyAsync((char*)records + o * 116, d, m * 116, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
			hipStreamSynchronize(c->stream) != hipSuccess)
			rc = LS1HIP_EHIP;
	}
	hipFree(d);
	if (rc) FAIL(c, rc, "record download failed");
	return LS1HIP_OK;
}

extern "C" int ls1hip_download_forces(ls1hip_ctx* c, size_t cap, double* F, double* M, double* Vi) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->forces_valid, "forces are not valid");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const size_t n = c->n_real;
	REQUIRE(c, cap >= n, "buffer too small: %zu < %zu", cap, n);
	if (n == 0) return LS1HIP_OK;
	int rc;
	const bool rot = c->h_ct.has_rot;
	if (F && (rc = d2h3(c, n, c->frc.Fx, c->frc.Fy, c->frc.Fz, F, 3, 0))) return rc;
	if (M && (rc = d2h3(c, n, rot ? c->frc.Mx : nullptr, rot ? c->frc.My : nullptr, rot ? c->frc.Mz : nullptr, M, 3, 0)))
		return rc;
	if (Vi) {
		REQUIRE(c, c->opt_vi, "per-molecule virial was not computed (set option compute_vi=1 before ls1hip_forces)");
		if ((rc = d2h3(c, n, c->frc.Vix, c->frc.Viy, c->frc.Viz, Vi, 3, 0))) return rc;
	}
	return LS1HIP_OK;
}

// ---- measurement ---------------------------------------------------------------------------------------------------
static Timer* timer_by_name(ls1hip_ctx* c, const char* name) {
	std::string n(name ? name : "");
	if (n == "force") return &c->t_force;
	if (n == "integrate") return &c->t_integrate;
	if (n == "rebin") return &c->t_rebin;
	if (n == "build") return &c->t_build;
	if (n == "halo") return &c->t_halo;
	return nullptr;
}

extern "C" int ls1hip_timing(ls1hip_ctx* c, const char* name, double* total_ms, uint64_t* launches) {
	if (!c) return LS1HIP_EINVAL;
	Timer* t = timer_by_name(c, name);
	REQUIRE(c, t, "unknown timer '%s'", name ? name : "(null)");
	HIPCHK(c, hipSetDevice(c->device));
	timer_collect(*t);
	if (total_ms) *total_ms = t->total_ms;
	if (launches) *launches = t->launches;
	return LS1HIP_OK;
}

extern "C" int ls1hip_timing_reset(ls1hip_ctx* c) {
	if (!c) return LS1HIP_EINVAL;
	for (Timer* t : {&c->t_force, &c->t_integrate, &c->t_rebin, &c->t_halo, &c->t_build}) {
		timer_collect(*t);
		t->total_ms = 0;
		t->launches = 0;
	}
	return LS1HIP_OK;
}

extern "C" int ls1hip_timing_enable(ls1hip_ctx* c, int on) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, on >= 0 && on <= 2, "timing mode must be 0 (off), 1 (all phases) or 2 (force passes only)");
	c->timing_on = on;
	return LS1HIP_OK;
}

extern "C" int ls1hip_pair_stats(ls1hip_ctx* c, uint64_t* dist_checks, uint64_t* pairs_in_range) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->opt_count_pairs, "option count_pairs=1 required");
	int rc = sync_counters(c);
	if (rc) return rc;
	if (dist_checks) *dist_checks = c->h_cnt->dist_checks;
	if (pairs_in_range) *pairs_in_range = c->h_cnt->pairs_in_range;
	return LS1HIP_OK;
}

