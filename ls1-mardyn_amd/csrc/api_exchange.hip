// api_exchange.hip — the C ABI of libls1hip (include/ls1hip.h), part 3: homogeneous long-range correction, export / import of the
// multi-rank exchange (leaving molecules, halo copies, position refresh) and seam A (ls1hip_soa_forces).
#include "api_internal.hpp"

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

