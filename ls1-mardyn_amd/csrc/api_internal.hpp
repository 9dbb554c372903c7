// api_internal.hpp — shared by the translation units of the C ABI (api.hip: context, model, domain, upload / download, timing;
// api_step.hip: the pieces of a time step, neighbour lists, ls1hip_run; api_exchange.hip: long-range correction, export / import of
// the multi-rank exchange, seam A).  Round 4: split out of one 2 400-line api.hip; nothing here is exported from the
// library (LS1_INTERNAL = hidden visibility).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "common.hpp"

using namespace ls1;


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

#define LS1_INTERNAL __attribute__((visibility("hidden")))
constexpr size_t STEPLOG_ROWS = 4096;  // per-step globals kept on the device for ls1hip_run_log (ring)
constexpr size_t INGEST_CHUNK = (size_t)1 << 22;  // molecules per staging pass (116 B each)

// ---- timing --------------------------------------------------------------------------------------------------------
constexpr size_t TIMER_MAX_PENDING = 4096;  // event pairs kept before they are folded into the total (bounds the event pool)
static inline void timer_fold(Timer& t) {
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

static inline void timer_collect(Timer& t) { timer_fold(t); }
static inline void timer_free(Timer& t) {
	for (auto e : t.ev) hipEventDestroy(e);
	t.ev.clear();
	t.used = 0;
}

// ---- memory helpers ------------------------------------------------------------------------------------------------
template <class T>
static inline int dalloc(ls1hip_ctx* c, T** p, size_t n) {
	*p = nullptr;
	void* q = nullptr;
	hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
	if (e != hipSuccess) FAIL(c, LS1HIP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
	*p = (T*)q;
	return 0;
}
template <class T>
static inline void dfree(T*& p) {
	if (p) hipFree((void*)p);
	p = nullptr;
}

static inline void free_mol(ls1hip_ctx* c) {
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
	dfree(c->d_msl_cnt); dfree(c->d_msl_off); dfree(c->d_msl_j); dfree(c->d_msl_il); dfree(c->d_msl_scratch); dfree(c->d_msl_mcnt); dfree(c->d_msl_pk); dfree(c->d_msl_pk2); dfree(c->d_msl_gm);
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
static inline void free_cells(ls1hip_ctx* c) {
	dfree(c->d_count); dfree(c->d_cell_begin); dfree(c->d_cell_end); dfree(c->d_blocksum); dfree(c->d_shell);
	c->cells_alloc = 0;
	c->n_shell = 0;
}


// ---- helpers defined in one translation unit and used by another ---------------------------------------------------------------
LS1_INTERNAL bool can_fuse(const ls1hip_ctx* c);
LS1_INTERNAL bool can_fuse_ms(const ls1hip_ctx* c);
LS1_INTERNAL bool can_list_kick_ms(const ls1hip_ctx* c);
LS1_INTERNAL hipStream_t halo_stream(ls1hip_ctx* c);
LS1_INTERNAL int sync_counters(ls1hip_ctx* c, hipStream_t s = nullptr);
LS1_INTERNAL int d2h3(ls1hip_ctx* c, size_t n, const double* a, const double* b, const double* d, double* out, int stride, int o);
LS1_INTERNAL RebinArgs rebin_args(ls1hip_ctx* c, uint32_t n_in);
LS1_INTERNAL HaloArgs halo_args(ls1hip_ctx* c);
LS1_INTERNAL int do_rebin_finish(ls1hip_ctx* c, uint32_t n_in);
LS1_INTERNAL void macro_to_upot_virial(const DevCounters* h, double* upot, double* virial);
LS1_INTERNAL int materialise_positions(ls1hip_ctx* c);
