// api.hip — the C ABI of libls1hip (include/ls1hip.h): context, parameter-table derivation, device memory,
// and the orchestration of the kernels of one time step.  Host C++17; every entry point cites the reference
// interface it replaces in the header.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "common.hpp"

using namespace ls1;

static thread_local std::string g_create_err = "";

#define FAIL(ctx, code, ...)                                   \
	do {                                                       \
		char _b[512];                                          \
		snprintf(_b, sizeof(_b), __VA_ARGS__);                 \
		(ctx)->err = _b;                                       \
		return (code);                                         \
	} while (0)

#define HIPCHK(ctx, call)                                                                              \
	do {                                                                                               \
		hipError_t _e = (call);                                                                        \
		if (_e != hipSuccess) FAIL(ctx, LS1HIP_EHIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
	} while (0)

#define REQUIRE(ctx, cond, ...) \
	do {                        \
		if (!(cond)) FAIL(ctx, LS1HIP_EINVAL, __VA_ARGS__); \
	} while (0)

// ---- timing --------------------------------------------------------------------------------------------------------
constexpr size_t TIMER_MAX_PENDING = 4096;  // event pairs kept before they are folded into the total (bounds the event pool)
static void timer_fold(Timer& t) {
	for (size_t i = 0; i + 1 < t.used; i += 2) {
		float ms = 0.f;
		hipEventSynchronize(t.ev[i + 1]);
		if (hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]) == hipSuccess) t.total_ms += ms;
	}
	t.used = 0;
}
struct TimedScope {
	ls1hip_ctx* c;
	Timer* t;
	hipEvent_t stop = nullptr;
	hipStream_t s;
	TimedScope(ls1hip_ctx* ctx, Timer& tm, hipStream_t stream = nullptr) : c(ctx), t(&tm), s(stream ? stream : ctx->stream) {
		if (!c->timing_on || (c->timing_on == 2 && t != &c->t_force)) return;
		if (t->used >= TIMER_MAX_PENDING) timer_fold(*t);  // long runs with timing on: fold the finished pairs, reuse the events
		if (t->used + 2 > t->ev.size()) {
			hipEvent_t a, b;
			if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
			t->ev.push_back(a);
			t->ev.push_back(b);
		}
		hipEventRecord(t->ev[t->used], s);
		stop = t->ev[t->used + 1];
		t->used += 2;
		t->launches++;
	}
	~TimedScope() {
		if (stop) hipEventRecord(stop, s);
	}
};

static void timer_collect(Timer& t) { timer_fold(t); }
static void timer_free(Timer& t) {
	for (auto e : t.ev) hipEventDestroy(e);
	t.ev.clear();
	t.used = 0;
}

// ---- memory helpers ------------------------------------------------------------------------------------------------
template <class T>
static int dalloc(ls1hip_ctx* c, T** p, size_t n) {
	*p = nullptr;
	void* q = nullptr;
	hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
	if (e != hipSuccess) FAIL(c, LS1HIP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
	*p = (T*)q;
	return 0;
}
template <class T>
static void dfree(T*& p) {
	if (p) hipFree((void*)p);
	p = nullptr;
}

static void free_mol(ls1hip_ctx* c) {
	for (int k = 0; k < 2; ++k) {
		MolSoA& m = c->mol[k];
		dfree(m.x); dfree(m.y); dfree(m.z); dfree(m.vx); dfree(m.vy); dfree(m.vz);
		dfree(m.q0); dfree(m.q1); dfree(m.q2); dfree(m.q3); dfree(m.Dx); dfree(m.Dy); dfree(m.Dz);
		dfree(m.id); dfree(m.cid);
	}
	ForceSoA& f = c->frc;
	dfree(f.Fx); dfree(f.Fy); dfree(f.Fz); dfree(f.Mx); dfree(f.My); dfree(f.Mz); dfree(f.Vix); dfree(f.Viy); dfree(f.Viz);
	HaloStage& h = c->hs;
	dfree(h.x); dfree(h.y); dfree(h.z); dfree(h.q0); dfree(h.q1); dfree(h.q2); dfree(h.q3); dfree(h.id); dfree(h.cid);
	dfree(h.key); dfree(h.rank); dfree(h.src); dfree(h.dir);
	dfree(c->d_halo_src); dfree(c->d_halo_dir);
	dfree(c->d_exp_halo_src); dfree(c->d_imp_slot); dfree(c->d_s2s); dfree(c->d_exp_refresh);
	c->vl_ready = false;
	dfree(c->alt_x); dfree(c->alt_y); dfree(c->alt_z);
	dfree(c->d_vl_words); dfree(c->d_vl_nw); dfree(c->d_vl_rec); dfree(c->d_vl_ii); dfree(c->d_vl_gi);
	dfree(c->d_vl_top2); dfree(c->d_vl_acc);
	dfree(c->d_msl_cnt); dfree(c->d_msl_off); dfree(c->d_msl_j); dfree(c->d_msl_il); dfree(c->d_msl_scratch); dfree(c->d_msl_mcnt); dfree(c->d_msl_pk);
	c->msl_groups_cap = c->msl_pairs_cap = c->msl_stride = 0;
	dfree(c->seam_a_buf);
	c->seam_a_cap = 0;
	c->vl_words_cap = c->vl_tiles_cap = 0;
	c->pos_x = c->pos_y = c->pos_z = nullptr;
	dfree(c->d_key); dfree(c->d_rank); dfree(c->d_perm); dfree(c->d_ckey); dfree(c->d_idk);
	dfree(c->d_partials);
	dfree(c->d_exp_leave); dfree(c->d_exp_halo);
	c->cap_real = c->cap_halo = 0;
	c->partials_cap = 0;
}
static void free_cells(ls1hip_ctx* c) {
	dfree(c->d_count); dfree(c->d_cell_begin); dfree(c->d_cell_end); dfree(c->d_blocksum); dfree(c->d_shell);
	c->cells_alloc = 0;
	c->n_shell = 0;
}

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
static bool can_fuse(const ls1hip_ctx* c) {
	return c->one_clj && c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_vi && !c->opt_count_pairs && !c->thermostat_on &&
		   (c->g.hw == 1 || c->g.hw == 2);
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
	else if (n == "list_kick_available") *v = (c->vl_ready && c->one_clj) ? 1 : 0;
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
constexpr size_t STEPLOG_ROWS = 4096;  // per-step globals kept on the device for ls1hip_run_log (ring)
constexpr size_t INGEST_CHUNK = (size_t)1 << 22;  // molecules per staging pass (116 B each)

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
static hipStream_t halo_stream(ls1hip_ctx* c) { return c->inner_in_flight ? c->stream2 : c->stream; }

static int sync_counters(ls1hip_ctx* c, hipStream_t s = nullptr) {
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

// ---- step pieces ---------------------------------------------------------------------------------------------------
static RebinArgs rebin_args(ls1hip_ctx* c, uint32_t n_in) {
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

static HaloArgs halo_args(ls1hip_ctx* c) {
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

static int do_rebin_finish(ls1hip_ctx* c, uint32_t n_in) {
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
		if (fuse) {  // the advanced positions go to the other buffer
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
		bool lj_only = true;
		for (int k = 0; k < c->h_ct.ncomp; ++k) lj_only = lj_only && c->h_ct.nc[k] == 0 && c->h_ct.nd[k] == 0 && c->h_ct.nq[k] == 0;
		done = launch_force_ms_list(P, c->h_ct.has_rot != 0, lj_only, c->h_ct.ncomp, c->d_msl_off, c->d_msl_j, c->d_msl_il, c->d_shift27, c->d_msl_pk,
									c->stream, &nblocks, c->partials_cap);
		if (!done) FAIL(c, LS1HIP_EINVAL, "multi-site neighbour-list force pass could not be launched");
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
	rm.kin_in_slot1 = fuse || fp.post_kick;
	rm.target_T = (fp.post_kick && c->thermostat_on) ? c->thermostat_T : 0.;
	rm.log = c->log_row;
	if (fp.vl && fuse) {
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

static void macro_to_upot_virial(const DevCounters* h, double* upot, double* virial) {
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
	}
	if (upot || virial) {
		int rc = sync_counters(c);
		if (rc) return rc;
		macro_to_upot_virial(c->h_cnt, upot, virial);
	}
	return LS1HIP_OK;
}

// positions parked in the force arrays by a fused pass -> back into the molecule arrays (readers other than ls1hip_rebin)
static int materialise_positions(ls1hip_ctx* c) {
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
	if (c->vl_ready) {
		IntegArgs a = integ_args_lists(c, dt);
		a.pre_scale = pre_scale; a.pre_bt = bt; a.pre_br = br;
		launch_kick_drift(a, c->stream);
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
	if (c->vl_ready) {
		launch_kick_then_kick_drift(integ_args_lists(c, dt), c->stream);
		int rcb = track_unfused_drift(c, dt);
		if (rcb) return rcb;
	} else {
		launch_kick_then_kick_drift(integ_args(c, dt), c->stream);
	}
	HIPCHK(c, hipGetLastError());
	c->binned = false;
	c->halo_valid = false;
	c->forces_valid = false;
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

extern "C" int ls1hip_set_thermostat(ls1hip_ctx* c, int enabled, double target_temperature) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, !enabled || target_temperature > 0., "target temperature must be positive");
	c->thermostat_on = enabled != 0;
	c->thermostat_T = target_temperature;
	return LS1HIP_OK;
}

// ---- Homogeneous long-range correction: longRange/Homogeneous.cpp ---------------------------------------------------
// Tail (r > rc, homogeneous fluid) of one term (sigma^2 / r^2)^(-n) of a site-site potential, angle-averaged over the
// orientations of the two molecules; a, b = distances of the two sites from their molecules' centres.  The closed forms
// (Lustig 1988; longRange/Homogeneous.cpp:137-180) are written here through the differences of powers they are made of:
//   both sites central      u = -rc^m / (sigma^2n m),                              m = 2 n + 3
//   one eccentric site      first differences  d1(k) = (rc + a)^k - (rc - a)^k
//   two eccentric sites     second differences d2(k) over rc +- (a + b), rc +- (a - b)
// u: energy integral, v: virial integral.
namespace lrc {
struct Tail {
	double u, v;
};
static Tail tail_term(int n, double rc, double s2, double a, double b) {
	if (a < b) std::swap(a, b);  // (symmetric in the two sites)
	const int m = 2 * n + 3;
	const double sn = pow(s2, n);
	Tail t;
	if (a == 0.) {
		t.u = -pow(rc, m) / (sn * m);
		t.v = 2 * n * t.u;
	} else if (b == 0.) {
		auto d1 = [&](int k) { return pow(rc + a, k) - pow(rc - a, k); };
		const double w = 1. / (4 * sn * a * (n + 1));
		t.u = w * (d1(m + 1) / (m + 1) - rc * d1(m)) / m;
		t.v = -w * rc * rc * d1(m - 1) - 3 * t.u;
	} else {
		auto d2 = [&](int k) { return pow(rc + a + b, k) - pow(rc + a - b, k) - pow(rc - a + b, k) + pow(rc - a - b, k); };
		const double w = 1. / (8 * sn * a * b * (n + 1) * m);
		t.u = w * (d2(m + 2) / (m + 2) - rc * d2(m + 1)) / (m + 1);
		t.v = -w * rc * rc * d2(m) - 3 * t.u;
	}
	return t;
}
}  // namespace lrc

extern "C" int ls1hip_long_range_homogeneous(ls1hip_ctx* c, const uint64_t* nmol, double rho, double* upot_corr,
											 double* virial_corr) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp, "ls1hip_set_components must be called first");
	REQUIRE(c, nmol && rho > 0., "bad argument");
	const CompTable& t = c->h_ct;
	double U = 0., V = 0., self = 0., N = 0.;
	const double rc = c->rc_lj;
	for (int i = 0; i < t.ncomp; ++i) N += (double)nmol[i];
	REQUIRE(c, N > 0., "no molecules");
	for (int i = 0; i < t.ncomp; ++i) {
		// effective dipole of the component: point charges + point dipoles (Homogeneous.cpp:38-64)
		double cb[3] = {0., 0., 0.};
		for (int a = 0; a < t.nc[i]; ++a)
			for (int d = 0; d < 3; ++d) cb[d] += t.chq[t.oc[i] + a] * t.chpos[t.oc[i] + a][d];
		for (int a = 0; a < t.nd[i]; ++a) {
			const double* e = t.dpe[t.od[i] + a];
			const double norm = 1.0 / sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
			for (int d = 0; d < 3; ++d) cb[d] += t.dpmy[t.od[i] + a] * e[d] * norm;
		}
		self += (cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2]) * (double)nmol[i];
		for (int j = 0; j < t.ncomp; ++j)
			for (int a = 0; a < t.nlj[i]; ++a) {
				const double* pa = t.ljpos[t.olj[i] + a];
				const double tau1 = sqrt(pa[0] * pa[0] + pa[1] * pa[1] + pa[2] * pa[2]);
				for (int b = 0; b < t.nlj[j]; ++b) {
					const double* pb = t.ljpos[t.olj[j] + b];
					const double tau2 = sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
					REQUIRE(c, tau1 + tau2 < rc, "error calculating cutoff corrections, rc too small");  // :83-86
					const int k = (t.olj[i] + a) * t.ncenters + (t.olj[j] + b);
					if (t.shift6[k] != 0.0) continue;  // truncated-shifted pairs carry no tail correction (:93)
					const double fac = (double)nmol[i] * (double)nmol[j] * t.eps24[k], s2 = t.sig2[k];
					// LJ: 24 eps [(sigma/r)^12 - (sigma/r)^6] -> the n = -6 term minus the n = -3 term
					const lrc::Tail t12 = lrc::tail_term(-6, rc, s2, tau1, tau2), t6 = lrc::tail_term(-3, rc, s2, tau1, tau2);
					U += fac * (t12.u - t6.u);
					V += fac * (t12.v - t6.v);
				}
			}
	}
	// Homogeneous::calculateLongRange (:113-135)
	const double fac = M_PI * rho / (3. * N);
	const double selfterm = -0.5 * t.epsRFInvrc3 * self;
	if (upot_corr) *upot_corr = fac * U + selfterm;
	if (virial_corr) *virial_corr = -fac * V + 3. * selfterm;
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
	const uint32_t ng = msl_groups((uint32_t)c->n_real);
	if ((size_t)ng + 1 > c->msl_groups_cap) {
		dfree(c->d_msl_cnt);
		dfree(c->d_msl_off);
		c->msl_groups_cap = 0;
		if ((rc = dalloc(c, &c->d_msl_cnt, (size_t)ng + 1)) || (rc = dalloc(c, &c->d_msl_off, (size_t)ng + 2))) return rc;
		c->msl_groups_cap = (size_t)ng + 1;
	}
	if (c->n_real > c->msl_stride) {
		dfree(c->d_msl_scratch);
		dfree(c->d_msl_mcnt);
		dfree(c->d_msl_pk);
		c->msl_stride = 0;
		const size_t stride = (c->cap_real + 63) & ~(size_t)63;
		if ((rc = dalloc(c, &c->d_msl_scratch, stride * (size_t)msl_capture_cap())) || (rc = dalloc(c, &c->d_msl_mcnt, stride)) ||
			(rc = dalloc(c, &c->d_msl_pk, stride * 8)))
			return rc;
		c->msl_stride = stride;
	}
	ForceParams P;
	fill_force_params(c, P, 0);
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
	REQUIRE(c, c->one_clj, "the post-force kick is folded into the single-centre LJ list pass only (ls1hip_forces_list + ls1hip_kick)");
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
	REQUIRE(c, !fuse || can_fuse(c), "fused list passes: no per-molecule virial, no device thermostat");
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
		// velocities are at t + dt/2 of the next step; the advanced positions wait in the other position buffer
		const bool in_alt = c->pos_x == c->alt_x;
		c->pos_x = in_alt ? nullptr : c->alt_x;
		c->pos_y = in_alt ? nullptr : c->alt_y;
		c->pos_z = in_alt ? nullptr : c->alt_z;
		c->fused_split = 0;
		c->vl_fresh = false;
		c->vl_bound_pending = true;
		c->halo_valid = false;
		c->forces_valid = false;
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
	const bool fuse = c->opt_fuse && can_fuse(c);
	// neighbour-list loop (fused or not: NVT and unfused NVE steps advance the displacement bound in their kick + drift pass)
	const bool verlet = can_verlet(c) && can_list(c);
	// the single-centre list pass does the post-force kick (+ sum m v^2) itself; the multi-site one leaves it to the integrator passes
	const bool list_kick = verlet && c->one_clj;
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

// ---- downloads -----------------------------------------------------------------------------------------------------
static int d2h3(ls1hip_ctx* c, size_t n, const double* a, const double* b, const double* d, double* out, int stride, int o) {
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

// ---- multi-GPU plumbing --------------------------------------------------------------------------------------------
extern "C" int ls1hip_export_counts(ls1hip_ctx* c, int kind, uint64_t counts[27]) {
	if (!c || !counts) return LS1HIP_EINVAL;
	REQUIRE(c, kind >= 0 && kind <= 2, "kind must be 0, 1 or 2");
	HIPCHK(c, hipSetDevice(c->device));
	if (kind == 2) {  // position refresh of the halo copies exported when the lists were built: the counts are frozen
		REQUIRE(c, c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
		for (int d = 0; d < 27; ++d) counts[d] = c->vl_exp_counts[d];
		return LS1HIP_OK;
	}
	int rc = sync_counters(c, kind == 1 ? halo_stream(c) : c->stream);
	if (rc) return rc;
	for (int d = 0; d < 27; ++d) counts[d] = kind == 0 ? c->h_cnt->exp_leave[d] : c->h_cnt->exp_halo[d];
	return LS1HIP_OK;
}

static int export_pack_async(ls1hip_ctx* c, int kind, int dir, double* dst, size_t cap, uint32_t* n_out, hipStream_t st) {
	const uint32_t n = kind == 0 ? c->h_cnt->exp_leave[dir] : (kind == 1 ? c->h_cnt->exp_halo[dir] : c->vl_exp_counts[dir]);
	REQUIRE(c, cap >= n, "export buffer too small: %zu < %u records", cap, n);
	const int w = kind == 0 ? LS1HIP_LEAVING_DOUBLES : (kind == 1 ? LS1HIP_HALO_DOUBLES : LS1HIP_REFRESH_DOUBLES);
	const double* src = kind == 0 ? c->d_exp_leave + (size_t)c->exp_off_leave[dir] * w
								  : (kind == 1 ? c->d_exp_halo : c->d_exp_refresh) + (size_t)c->exp_off_halo[dir] * w;
	launch_pack_copy(dst, src, n * w, st);
	*n_out = n;
	return LS1HIP_OK;
}

extern "C" int ls1hip_export_pack(ls1hip_ctx* c, int kind, int dir, void* dev_buf, size_t cap) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, kind >= 0 && kind <= 2 && dir >= 0 && dir < 27 && dev_buf, "bad argument");
	REQUIRE(c, kind != 2 || c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
	HIPCHK(c, hipSetDevice(c->device));
	uint32_t n = 0;
	hipStream_t st = kind != 0 ? halo_stream(c) : c->stream;
	int rc = export_pack_async(c, kind, dir, (double*)dev_buf, cap, &n, st);
	if (rc) return rc;
	HIPCHK(c, hipStreamSynchronize(st));  // the transport runs on its own stream
	return LS1HIP_OK;
}

extern "C" int ls1hip_export_pack_dirs(ls1hip_ctx* c, int kind, const int* dirs, int ndirs, void* dev_buf, size_t cap) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, kind >= 0 && kind <= 2 && ndirs >= 0 && ndirs <= 27 && (ndirs == 0 || dirs) && (cap == 0 || dev_buf), "bad argument");
	REQUIRE(c, kind != 2 || c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
	HIPCHK(c, hipSetDevice(c->device));
	const int w = kind == 0 ? LS1HIP_LEAVING_DOUBLES : (kind == 1 ? LS1HIP_HALO_DOUBLES : LS1HIP_REFRESH_DOUBLES);
	PackSegments seg;
	seg.n = 0;
	size_t used = 0;
	for (int k = 0; k < ndirs; ++k) {
		REQUIRE(c, dirs[k] >= 0 && dirs[k] < 27, "direction %d out of range", dirs[k]);
		const uint32_t n = kind == 0 ? c->h_cnt->exp_leave[dirs[k]] : (kind == 1 ? c->h_cnt->exp_halo[dirs[k]] : c->vl_exp_counts[dirs[k]]);
		REQUIRE(c, cap - used >= n, "export buffer too small: %zu < %zu records", cap, used + n);
		if (n == 0) continue;
		seg.src_off[seg.n] = (uint64_t)(kind == 0 ? c->exp_off_leave[dirs[k]] : c->exp_off_halo[dirs[k]]) * w;
		seg.dst_off[seg.n] = (uint64_t)used * w;
		++seg.n;
		used += n;
	}
	seg.total = (uint64_t)used * w;
	hipStream_t st = kind != 0 ? halo_stream(c) : c->stream;
	launch_pack_segments(seg, kind == 0 ? c->d_exp_leave : (kind == 1 ? c->d_exp_halo : c->d_exp_refresh), (double*)dev_buf, st);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipStreamSynchronize(st));  // one synchronisation per message set: the transport runs on its own stream
	return LS1HIP_OK;
}

extern "C" int ls1hip_import(ls1hip_ctx* c, int kind, const void* dev_buf, size_t n) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, kind >= 0 && kind <= 2, "kind must be 0, 1 or 2");
	REQUIRE(c, n == 0 || dev_buf, "null buffer");
	HIPCHK(c, hipSetDevice(c->device));
	if (kind == 2) {
		REQUIRE(c, c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
		REQUIRE(c, c->halo_import_at + n <= c->vl_imp_total, "more refresh records than halo records were imported at build time");
		HaloArgs a = halo_args(c);
		const MolSoA& m = c->mol[c->cur];
		launch_refresh_import(a, (const double*)dev_buf, (uint32_t)n, c->pos_x ? c->pos_x : m.x, c->pos_x ? c->pos_y : m.y,
							  c->pos_x ? c->pos_z : m.z, halo_stream(c));
		c->halo_import_at += (uint32_t)n;
		return LS1HIP_OK;
	}
	if (kind == 0) {
		REQUIRE(c, c->pending_in + n <= c->cap_real, "owned-molecule capacity exceeded by immigration");
		RebinArgs a = rebin_args(c, c->pending_in);
		launch_leave_import(a, (const double*)dev_buf, (uint32_t)n, c->pending_in, c->stream);
		c->pending_in += (uint32_t)n;
	} else {
		HaloArgs a = halo_args(c);
		launch_halo_import(a, (const double*)dev_buf, (uint32_t)n, halo_stream(c));
		c->halo_import_at += (uint32_t)n;
	}
	// asynchronous: dev_buf is read on the engine's stream and must stay valid until ls1hip_import_done(kind) returns
	return LS1HIP_OK;
}

extern "C" int ls1hip_import_done(ls1hip_ctx* c, int kind) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, kind >= 0 && kind <= 2, "kind must be 0, 1 or 2");
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->has_remote) return LS1HIP_OK;  // purely local domain: ls1hip_rebin / ls1hip_halo already finished the phase
	if (kind == 2) {
		REQUIRE(c, c->vl_ready, "no neighbour lists (ls1hip_verlet_build)");
		REQUIRE(c, c->halo_import_at == c->vl_imp_total, "%u refresh records imported, %u halo records were imported at build time",
				c->halo_import_at, c->vl_imp_total);
		hipStream_t hs = halo_stream(c);
		c->halo_valid = true;
		if (c->inner_in_flight) HIPCHK(c, hipEventRecord(c->ev_halo, hs));
		HIPCHK(c, hipStreamSynchronize(hs));  // imported buffers may be released by the caller from here on
		return LS1HIP_OK;
	}
	if (kind == 0) {
		REQUIRE(c, !c->binned, "import_done(0) without a pending ls1hip_rebin");
		int rc = do_rebin_finish(c, c->pending_in);
		if (rc) return rc;
		if ((rc = sync_counters(c))) return rc;
		c->n_real = c->h_cnt->n_real;
	} else {
		REQUIRE(c, c->binned, "halo import before rebin");
		HaloArgs a = halo_args(c);
		hipStream_t hs = halo_stream(c);
		{
			TimedScope ts(c, c->t_halo, hs);
			launch_halo_finalize(a, hs);
		}
		c->halo_valid = true;
		if (c->inner_in_flight) HIPCHK(c, hipEventRecord(c->ev_halo, hs));
		HIPCHK(c, hipStreamSynchronize(hs));  // imported buffers may be released by the caller from here on
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

// ---- seam A --------------------------------------------------------------------------------------------------------
extern "C" int ls1hip_soa_forces(ls1hip_ctx* c, const int cell_dims[3], const uint32_t* cell_start, size_t n,
								 const double* r, const double* q, const int32_t* cid, double* F, double* M, double* Vi,
								 double* upot, double* virial) {
	if (!c) return LS1HIP_EINVAL;
	REQUIRE(c, c->have_comp, "ls1hip_set_components must be called first");
	REQUIRE(c, cell_dims && cell_start && (n == 0 || r), "null argument");
	REQUIRE(c, cell_dims[0] >= 3 && cell_dims[1] >= 3 && cell_dims[2] >= 3, "cell grid must include the halo layer");
	const size_t ncells = (size_t)cell_dims[0] * cell_dims[1] * cell_dims[2];
	REQUIRE(c, cell_start[ncells] == n, "cell_start[ncells] must equal n");
	HIPCHK(c, hipSetDevice(c->device));
	const bool rot = c->h_ct.has_rot;
	std::vector<double> hx(n), hy(n), hz(n), h0, h1, h2, h3;
	std::vector<int32_t> hc(n, 0);
	std::vector<uint32_t> hkey(n), hb(ncells), he(ncells);
	for (size_t i = 0; i < n; ++i) {
		hx[i] = r[3 * i];
		hy[i] = r[3 * i + 1];
		hz[i] = r[3 * i + 2];
		if (cid) hc[i] = cid[i];
	}
	if (rot) {
		h0.resize(n); h1.resize(n); h2.resize(n); h3.resize(n);
		for (size_t i = 0; i < n; ++i) {
			h0[i] = q ? q[4 * i] : 1.;
			h1[i] = q ? q[4 * i + 1] : 0.;
			h2[i] = q ? q[4 * i + 2] : 0.;
			h3[i] = q ? q[4 * i + 3] : 0.;
		}
	}
	for (size_t cc = 0; cc < ncells; ++cc) {
		REQUIRE(c, cell_start[cc] <= cell_start[cc + 1], "cell_start must be non-decreasing");
		hb[cc] = cell_start[cc];
		he[cc] = cell_start[cc + 1];
		for (uint32_t p = cell_start[cc]; p < cell_start[cc + 1]; ++p) hkey[p] = (uint32_t)cc;
	}
	double *dx = nullptr, *dy = nullptr, *dz = nullptr, *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
	double *fx = nullptr, *fy = nullptr, *fz = nullptr, *mx = nullptr, *my = nullptr, *mz = nullptr, *vx = nullptr,
		   *vy = nullptr, *vz = nullptr, *part = nullptr;
	int32_t* dc = nullptr;
	uint32_t *dk = nullptr, *db = nullptr, *de = nullptr;
	int rc = 0;
	const size_t npart = n / 64 + 16;
	// one persistent, grow-only device arena for the 21 arrays of a traversal (they were allocated and freed per
	// call — a few ms of hipMalloc / hipFree per time step of the driver)
	{
		auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
		const size_t nd = al(n * 8), ni = al(n * 4), nc4 = al(ncells * 4);
		const size_t need = nd * (3 + 3 + 3 + (rot ? 4 + 3 : 0)) + ni * 2 + nc4 * 2 + al(npart * 4 * 8);
		if (need > c->seam_a_cap) {
			dfree(c->seam_a_buf);
			c->seam_a_cap = 0;
			const size_t cap = need + need / 4;
			if ((rc = dalloc(c, &c->seam_a_buf, cap))) return rc;
			c->seam_a_cap = cap;
		}
		char* p = c->seam_a_buf;
		auto take = [&](size_t bytes) {
			char* q = p;
			p += bytes;
			return q;
		};
		dx = (double*)take(nd); dy = (double*)take(nd); dz = (double*)take(nd);
		fx = (double*)take(nd); fy = (double*)take(nd); fz = (double*)take(nd);
		vx = (double*)take(nd); vy = (double*)take(nd); vz = (double*)take(nd);
		if (rot) {
			d0 = (double*)take(nd); d1 = (double*)take(nd); d2 = (double*)take(nd); d3 = (double*)take(nd);
			mx = (double*)take(nd); my = (double*)take(nd); mz = (double*)take(nd);
		}
		dc = (int32_t*)take(ni); dk = (uint32_t*)take(ni);
		db = (uint32_t*)take(nc4); de = (uint32_t*)take(nc4);
		part = (double*)take(al(npart * 4 * 8));
	}
	auto cleanup = [&]() {};
	auto up = [&](void* d, const void* h, size_t bytes) { return bytes ? hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) : hipSuccess; };
	hipError_t e = hipSuccess;
	if ((e = up(dx, hx.data(), n * 8)) || (e = up(dy, hy.data(), n * 8)) || (e = up(dz, hz.data(), n * 8)) ||
		(e = up(dc, hc.data(), n * 4)) || (e = up(dk, hkey.data(), n * 4)) || (e = up(db, hb.data(), ncells * 4)) ||
		(e = up(de, he.data(), ncells * 4)) ||
		(rot && ((e = up(d0, h0.data(), n * 8)) || (e = up(d1, h1.data(), n * 8)) || (e = up(d2, h2.data(), n * 8)) ||
				 (e = up(d3, h3.data(), n * 8))))) {
		cleanup();
		FAIL(c, LS1HIP_EHIP, "upload failed: %s", hipGetErrorString(e));
	}
	for (double* z : {fx, fy, fz, vx, vy, vz}) hipMemsetAsync(z, 0, n * 8, c->stream);
	if (rot) for (double* z : {mx, my, mz}) hipMemsetAsync(z, 0, n * 8, c->stream);
	ForceParams P;
	memset(&P, 0, sizeof(P));
	P.g.dims[0] = cell_dims[0]; P.g.dims[1] = cell_dims[1]; P.g.dims[2] = cell_dims[2];
	P.g.hw = 1;
	P.g.ncells = (int)ncells;
	for (int d = 0; d < 3; ++d) P.g.box[d] = cell_dims[d] - 2;
	P.x = dx; P.y = dy; P.z = dz; P.q0 = d0; P.q1 = d1; P.q2 = d2; P.q3 = d3;
	P.cid = dc;
	P.cell_begin = db; P.cell_end = de; P.ckey = dk;
	P.Fx = fx; P.Fy = fy; P.Fz = fz; P.Mx = mx; P.My = my; P.Mz = mz; P.Vix = vx; P.Viy = vy; P.Viz = vz;
	P.ct = c->d_ct;
	P.cnt = c->d_cnt;
	P.partials = part;
	P.n_real_cap = (uint32_t)n;
	P.n_fixed = (uint32_t)n;
	P.which = 3;
	P.eps24 = c->h_ct.eps24[0]; P.sig2 = c->h_ct.sig2[0]; P.shift6 = c->h_ct.shift6[0]; P.rc2 = c->h_ct.rc2;
	uint32_t nblocks = 0;
	launch_clear_macro(c->d_cnt, c->stream);
	if (n) {
		// the brick kernels traverse exactly the non-halo cells (their which = 0), as the generic kernel does with which = 3;
		// the per-molecule virial (Vi) exists in the multi-site brick kernel and the generic kernel only
		bool done = false;
		const double inner_cells = (double)P.g.box[0] * P.g.box[1] * P.g.box[2];
		const double mean_per_cell = inner_cells > 0 ? (double)n / (double)ncells : 0.;
		if (c->opt_force_kernel != LS1HIP_FK_GENERIC && !c->opt_count_pairs) {
			ForceParams Q = P;
			Q.which = 0;
			if (c->one_clj && !Vi)
				done = launch_force_lj(Q, c->stream, &nblocks, part, npart, (int)c->opt_lj_split, mean_per_cell, &c->brick_lists);
			else if (!c->one_clj)
				done = launch_force_ms(Q, Vi != nullptr, rot, c->h_ct.ncomp == 1, c->stream, &nblocks, npart, mean_per_cell,
									   &c->brick_lists);
		}
		c->last_force_kernel = done ? (c->one_clj ? LS1HIP_FK_LDS_LIST : LS1HIP_FK_MS_BRICK) : LS1HIP_FK_GENERIC;
		if (!done) launch_force_generic(P, c->one_clj, true, rot, c->stream, &nblocks);
		launch_force_reduce(c->d_cnt, part, nblocks, c->d_stage, c->stream, ReduceMode());
	}
	rc = sync_counters(c);
	if (!rc) {
		macro_to_upot_virial(c->h_cnt, upot, virial);
		if (F) rc = d2h3(c, n, fx, fy, fz, F, 3, 0);
		if (!rc && M) rc = d2h3(c, n, rot ? mx : nullptr, rot ? my : nullptr, rot ? mz : nullptr, M, 3, 0);
		if (!rc && Vi) rc = d2h3(c, n, vx, vy, vz, Vi, 3, 0);
	}
	cleanup();
	return rc;
}
