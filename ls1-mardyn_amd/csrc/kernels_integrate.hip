// kernels_integrate.hip — Leapfrog integrator as two streaming kernels over the device-resident SoA (HBM-bound:
// pre-force pass reads r,v,F and writes r,v = 120 B per molecule; post-force pass reads v,F, writes v = 72 B).
//
// Restates FullMolecule::upd_preF (/root/reference/src/molecules/FullMolecule.cpp:334-364) and upd_postF (:366-389)
// as driven by Leapfrog::transition1to2 / transition2to3 (/root/reference/src/integrators/Leapfrog.cpp:48-64,66-150).
#include "common.hpp"
#include "leapfrog_body.hpp"

namespace ls1 {

constexpr int ITPB = 256;

__device__ __forceinline__ void q_diff(const double q[4], V3 w, double dq[4]) {
	// Quaternion::differentiate, molecules/Quaternion.cpp:93-98
	dq[0] = .5 * (-q[1] * w.x - q[2] * w.y - q[3] * w.z);
	dq[1] = .5 * (q[0] * w.x - q[3] * w.y + q[2] * w.z);
	dq[2] = .5 * (q[3] * w.x + q[0] * w.y - q[1] * w.z);
	dq[3] = .5 * (-q[2] * w.x + q[1] * w.y + q[0] * w.z);
}

// list mode: the drift speed of this pass bounds how far any molecule moves in it; block maximum -> vmax_part[block]
__device__ __forceinline__ void block_vmax(double v2, double* vmax_part) {
	__shared__ double red[ITPB / 64];
	for (int o = 32; o > 0; o >>= 1) v2 = fmax(v2, __shfl_down(v2, o));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v2;
	__syncthreads();
	if (threadIdx.x == 0) {
		double m = 0.;
		for (int i = 0; i < ITPB / 64; ++i) m = fmax(m, red[i]);
		vmax_part[blockIdx.x] = m;
	}
}

template <bool HAS_ROT>
__global__ void __launch_bounds__(ITPB) k_kick_drift(IntegArgs a) {
	const uint32_t p = blockIdx.x * ITPB + threadIdx.x;
	double v2 = 0.;
	if (p < a.cnt->n_real) {
	const double dt = a.dt, dt_halve = .5 * dt;
	const int c = HAS_ROT ? a.mol.cid[p] : (a.ct->ncomp > 1 ? a.mol.cid[p] : 0);
	const double dtInv2m = dt_halve / a.ct->mass[c];
	double bt = 1., br = 1.;
	if (a.pre_scale) {  // VelocityScalingThermostat::apply folded into this pass (uniform branch)
		bt = a.pre_scale == 2 ? a.cnt->beta[0] : a.pre_scale == 3 ? a.pre_bt_c[c] : a.pre_bt;
		br = a.pre_scale == 2 ? a.cnt->beta[1] : a.pre_scale == 3 ? a.pre_br_c[c] : a.pre_br;
	}
	double vx = a.mol.vx[p], vy = a.mol.vy[p], vz = a.mol.vz[p];
	if (a.pre_scale) {
		vx *= bt;
		vy *= bt;
		vz *= bt;
	}
	vx += dtInv2m * a.frc.Fx[p];
	vy += dtInv2m * a.frc.Fy[p];
	vz += dtInv2m * a.frc.Fz[p];
	a.mol.vx[p] = vx;
	a.mol.vy[p] = vy;
	a.mol.vz[p] = vz;
	const double xn = a.mol.x[p] + dt * vx, yn = a.mol.y[p] + dt * vy, zn = a.mol.z[p] + dt * vz;
	a.mol.x[p] = xn;
	a.mol.y[p] = yn;
	a.mol.z[p] = zn;
	v2 = vx * vx + vy * vy + vz * vz;
	if (HAS_ROT) {
		double q[4] = {a.mol.q0[p], a.mol.q1[p], a.mol.q2[p], a.mol.q3[p]};
		V3 D = {a.mol.Dx[p], a.mol.Dy[p], a.mol.Dz[p]};
		if (a.pre_scale) D = {D.x * br, D.y * br, D.z * br};
		const V3 invI = {a.ct->invI[c][0], a.ct->invI[c][1], a.ct->invI[c][2]};
		V3 w = rotate_inv(rot_of(q[0], q[1], q[2], q[3]), D);
		w = {w.x * invI.x, w.y * invI.y, w.z * invI.z};
		double dq[4], qh[4];
		q_diff(q, w, dq);
		for (int k = 0; k < 4; ++k) qh[k] = dq[k] * dt_halve + q[k];
		double qcorr = 1. / sqrt(qh[0] * qh[0] + qh[1] * qh[1] + qh[2] * qh[2] + qh[3] * qh[3]);
		for (int k = 0; k < 4; ++k) qh[k] *= qcorr;
		D.x += dt_halve * a.frc.Mx[p];
		D.y += dt_halve * a.frc.My[p];
		D.z += dt_halve * a.frc.Mz[p];
		w = rotate_inv(rot_of(qh[0], qh[1], qh[2], qh[3]), D);
		w = {w.x * invI.x, w.y * invI.y, w.z * invI.z};
		q_diff(qh, w, dq);
		for (int k = 0; k < 4; ++k) q[k] += dq[k] * dt;
		qcorr = 1. / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
		const double qs0 = q[0] * qcorr, qs1 = q[1] * qcorr, qs2 = q[2] * qcorr, qs3 = q[3] * qcorr;
		a.mol.q0[p] = qs0;
		a.mol.q1[p] = qs1;
		a.mol.q2[p] = qs2;
		a.mol.q3[p] = qs3;
		a.mol.Dx[p] = D.x;
		a.mol.Dy[p] = D.y;
		a.mol.Dz[p] = D.z;
		// the record of the pair-stream force pass, from the values just stored (what k_msl_pack would read back)
		if (a.pk) msl_write_record(a.pk, p, xn, yn, zn, qs0, qs1, qs2, qs3, true, a.pk_ncomp > 1 ? c : 0);
	}
	}
	if (a.vmax_part) block_vmax(v2, a.vmax_part);  // every thread of the workgroup arrives here (one barrier inside)
}

// Inside ls1hip_run two consecutive Leapfrog events touch the same arrays back to back: upd_postF of step n
// (v += dt/2m F, L += dt/2 M) and upd_preF of step n+1 (v += dt/2m F; r += dt v; quaternion/L half steps with the SAME
// F, M).  Fused: v is read and written once, F read once (saves 72 B per molecule per step).  The arithmetic is the
// unfused sequence, operation for operation, so trajectories are bitwise identical to the piecewise calls.
template <bool HAS_ROT>
__global__ void __launch_bounds__(ITPB) k_kick_then_kick_drift(IntegArgs a) {
	const uint32_t p = blockIdx.x * ITPB + threadIdx.x;
	double v2 = 0.;
	if (p < a.cnt->n_real) {
	const int c = HAS_ROT ? a.mol.cid[p] : (a.ct->ncomp > 1 ? a.mol.cid[p] : 0);
	// (the arithmetic: leapfrog_body.hpp, shared with the fused epilogue of the pair-stream force pass)
	LeapState s;
	s.x = a.mol.x[p]; s.y = a.mol.y[p]; s.z = a.mol.z[p];
	s.vx = a.mol.vx[p]; s.vy = a.mol.vy[p]; s.vz = a.mol.vz[p];
	const V3 F = {a.frc.Fx[p], a.frc.Fy[p], a.frc.Fz[p]};
	V3 M = {0., 0., 0.}, invI = {0., 0., 0.};
	if (HAS_ROT) {
		s.q[0] = a.mol.q0[p]; s.q[1] = a.mol.q1[p]; s.q[2] = a.mol.q2[p]; s.q[3] = a.mol.q3[p];
		s.D = {a.mol.Dx[p], a.mol.Dy[p], a.mol.Dz[p]};
		M = {a.frc.Mx[p], a.frc.My[p], a.frc.Mz[p]};
		invI = {a.ct->invI[c][0], a.ct->invI[c][1], a.ct->invI[c][2]};
	}
	v2 = leap_post_pre<HAS_ROT>(a.dt, a.ct->mass[c], invI, F, M, s);
	a.mol.vx[p] = s.vx; a.mol.vy[p] = s.vy; a.mol.vz[p] = s.vz;
	a.mol.x[p] = s.x; a.mol.y[p] = s.y; a.mol.z[p] = s.z;
	if (HAS_ROT) {
		a.mol.q0[p] = s.q[0]; a.mol.q1[p] = s.q[1]; a.mol.q2[p] = s.q[2]; a.mol.q3[p] = s.q[3];
		a.mol.Dx[p] = s.D.x; a.mol.Dy[p] = s.D.y; a.mol.Dz[p] = s.D.z;
		// the record of the pair-stream force pass, from the values just stored (what k_msl_pack would read back)
		if (a.pk) msl_write_record(a.pk, p, s.x, s.y, s.z, s.q[0], s.q[1], s.q[2], s.q[3], true, a.pk_ncomp > 1 ? c : 0);
	}
	}
	if (a.vmax_part) block_vmax(v2, a.vmax_part);
}

void launch_kick_then_kick_drift(const IntegArgs& a, hipStream_t s) {
	if (a.n_cap == 0) return;
	const dim3 grid((a.n_cap + ITPB - 1) / ITPB);
	if (a.has_rot) hipLaunchKernelGGL(k_kick_then_kick_drift<true>, grid, dim3(ITPB), 0, s, a);
	else hipLaunchKernelGGL(k_kick_then_kick_drift<false>, grid, dim3(ITPB), 0, s, a);
}

__device__ __forceinline__ double wave_sum_i(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}

template <bool HAS_ROT>
__global__ void __launch_bounds__(ITPB) k_kick(IntegArgs a) {
	const uint32_t p = blockIdx.x * ITPB + threadIdx.x;
	double mv2 = 0., Iw2 = 0., rdof = 0.;
	if (p < a.cnt->n_real) {
		const int c = HAS_ROT ? a.mol.cid[p] : (a.ct->ncomp > 1 ? a.mol.cid[p] : 0);
		// (the arithmetic: leapfrog_body.hpp, shared with the post-kick epilogue of the pair-stream list pass)
		LeapState s;
		s.vx = a.mol.vx[p]; s.vy = a.mol.vy[p]; s.vz = a.mol.vz[p];
		const V3 F = {a.frc.Fx[p], a.frc.Fy[p], a.frc.Fz[p]};
		V3 M = {0., 0., 0.}, invI = {0., 0., 0.}, I = {0., 0., 0.};
		if (HAS_ROT) {
			s.q[0] = a.mol.q0[p]; s.q[1] = a.mol.q1[p]; s.q[2] = a.mol.q2[p]; s.q[3] = a.mol.q3[p];
			s.D = {a.mol.Dx[p], a.mol.Dy[p], a.mol.Dz[p]};
			M = {a.frc.Mx[p], a.frc.My[p], a.frc.Mz[p]};
			invI = {a.ct->invI[c][0], a.ct->invI[c][1], a.ct->invI[c][2]};
			I = {a.ct->I[c][0], a.ct->I[c][1], a.ct->I[c][2]};
		}
		leap_post<HAS_ROT>(a.dt, a.ct->mass[c], invI, I, F, M, s, mv2, Iw2);
		a.mol.vx[p] = s.vx; a.mol.vy[p] = s.vy; a.mol.vz[p] = s.vz;
		if (HAS_ROT) {
			a.mol.Dx[p] = s.D.x; a.mol.Dy[p] = s.D.y; a.mol.Dz[p] = s.D.z;
		}
		rdof = (double)a.ct->rotdof[c];
	}
	__shared__ double red[ITPB / 64][3];
	mv2 = wave_sum_i(mv2);
	Iw2 = wave_sum_i(Iw2);
	rdof = wave_sum_i(rdof);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) {
		red[w][0] = mv2;
		red[w][1] = Iw2;
		red[w][2] = rdof;
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		double s = 0.;
		for (int i = 0; i < ITPB / 64; ++i) s += red[i][threadIdx.x];
		a.partials[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
	}
}

// ---- kinetic sums per component (component-wise thermostats) -----------------------------------------------------------------------
template <bool HAS_ROT>
__global__ void __launch_bounds__(ITPB) k_kin_by_component(IntegArgs a, int ncomp, double* __restrict__ part) {
	const uint32_t p = blockIdx.x * ITPB + threadIdx.x;
	double mv2 = 0., Iw2 = 0., rdof = 0., one = 0.;
	int c = -1;
	if (p < a.cnt->n_real) {
		c = ncomp > 1 ? a.mol.cid[p] : 0;
		const double m = a.ct->mass[c];
		const double vx = a.mol.vx[p], vy = a.mol.vy[p], vz = a.mol.vz[p];
		mv2 = m * (vx * vx + vy * vy + vz * vz);
		rdof = (double)a.ct->rotdof[c];
		one = 1.;
		if (HAS_ROT) {
			const V3 D = {a.mol.Dx[p], a.mol.Dy[p], a.mol.Dz[p]};
			V3 w = rotate_inv(rot_of(a.mol.q0[p], a.mol.q1[p], a.mol.q2[p], a.mol.q3[p]), D);
			w = {w.x * a.ct->invI[c][0], w.y * a.ct->invI[c][1], w.z * a.ct->invI[c][2]};
			Iw2 = a.ct->I[c][0] * w.x * w.x + a.ct->I[c][1] * w.y * w.y + a.ct->I[c][2] * w.z * w.z;
		}
	}
	__shared__ double red[ITPB / 64][MAXC][4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int k = 0; k < ncomp; ++k) {  // fixed order of the sums: deterministic
		const bool mine = c == k;
		const double s0 = wave_sum_i(mine ? mv2 : 0.), s1 = wave_sum_i(mine ? Iw2 : 0.), s2 = wave_sum_i(mine ? one : 0.),
					 s3 = wave_sum_i(mine ? rdof : 0.);
		if (lane == 0) {
			red[wv][k][0] = s0;
			red[wv][k][1] = s1;
			red[wv][k][2] = s2;
			red[wv][k][3] = s3;
		}
	}
	__syncthreads();
	if ((int)threadIdx.x < ncomp * 4) {
		const int k = threadIdx.x >> 2, q = threadIdx.x & 3;
		double sum = 0.;
		for (int i = 0; i < ITPB / 64; ++i) sum += red[i][k][q];
		part[((size_t)blockIdx.x * MAXC + k) * 4 + q] = sum;
	}
}
__global__ void __launch_bounds__(256) k_kin_by_component_reduce(const double* __restrict__ part, uint32_t nb, int ncomp, double* __restrict__ out) {
	__shared__ double red[4];
	for (int kq = 0; kq < ncomp * 4; ++kq) {
		const int k = kq >> 2, q = kq & 3;
		double v = 0.;
		for (uint32_t b = threadIdx.x; b < nb; b += 256) v += part[((size_t)b * MAXC + k) * 4 + q];
		v = wave_sum_i(v);
		if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
		__syncthreads();
		if (threadIdx.x == 0) out[kq] = red[0] + red[1] + red[2] + red[3];
		__syncthreads();
	}
}
void launch_kin_by_component(const IntegArgs& a, int ncomp, double* scratch, double* out, hipStream_t s) {
	const uint32_t nb = (a.n_cap + ITPB - 1) / ITPB;
	if (nb == 0) {
		hipMemsetAsync(out, 0, (size_t)ncomp * 4 * sizeof(double), s);
		return;
	}
	if (a.has_rot) hipLaunchKernelGGL(k_kin_by_component<true>, dim3(nb), dim3(ITPB), 0, s, a, ncomp, scratch);
	else hipLaunchKernelGGL(k_kin_by_component<false>, dim3(nb), dim3(ITPB), 0, s, a, ncomp, scratch);
	hipLaunchKernelGGL(k_kin_by_component_reduce, dim3(1), dim3(256), 0, s, scratch, nb, ncomp, out);
}

void launch_kick_drift(const IntegArgs& a, hipStream_t s) {
	if (a.n_cap == 0) return;
	const dim3 grid((a.n_cap + ITPB - 1) / ITPB);
	if (a.has_rot) hipLaunchKernelGGL(k_kick_drift<true>, grid, dim3(ITPB), 0, s, a);
	else hipLaunchKernelGGL(k_kick_drift<false>, grid, dim3(ITPB), 0, s, a);
}

void launch_kick(const IntegArgs& a, hipStream_t s, uint32_t* nblocks) {
	const uint32_t nb = (a.n_cap + ITPB - 1) / ITPB;
	*nblocks = nb;
	if (nb == 0) return;
	if (a.has_rot) hipLaunchKernelGGL(k_kick<true>, dim3(nb), dim3(ITPB), 0, s, a);
	else hipLaunchKernelGGL(k_kick<false>, dim3(nb), dim3(ITPB), 0, s, a);
}

// One workgroup of RTPB threads over the per-block partials, four independent partial rows in flight per thread (at 10^8
// molecules there are 4 * 10^5 rows: with 256 threads and one row per trip this kernel took 0.7 ms).
constexpr int RTPB = 1024;
__global__ void __launch_bounds__(RTPB) k_kin_reduce(DevCounters* cnt, const double* partials, uint32_t nblocks, double target_T,
													 double* log) {
	double v[3] = {0., 0., 0.};
	uint32_t b = threadIdx.x;
	for (; b + 3u * RTPB < nblocks; b += 4u * RTPB) {
		double t[4][3];
#pragma unroll
		for (int u = 0; u < 4; ++u)
			for (int k = 0; k < 3; ++k) t[u][k] = partials[(size_t)(b + (uint32_t)u * RTPB) * 4 + k];
		for (int k = 0; k < 3; ++k) v[k] += (t[0][k] + t[1][k]) + (t[2][k] + t[3][k]);
	}
	for (; b < nblocks; b += RTPB)
		for (int k = 0; k < 3; ++k) v[k] += partials[(size_t)b * 4 + k];
	__shared__ double red[RTPB / 64][3];
	for (int k = 0; k < 3; ++k) v[k] = wave_sum_i(v[k]);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0)
		for (int k = 0; k < 3; ++k) red[w][k] = v[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int i = 1; i < RTPB / 64; ++i)
			for (int k = 0; k < 3; ++k) red[0][k] += red[i][k];
		for (int i = 1; i < 4; ++i)
			for (int k = 0; k < 3; ++k) red[i][k] = 0.;
		cnt->kin[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
		cnt->kin[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
		cnt->kin_n = cnt->n_real;
		cnt->kin_rotdof = (unsigned long long)(red[0][2] + red[1][2] + red[2][2] + red[3][2] + 0.5);
		// thermostat 0 of Domain::calculateGlobalValues (Domain.cpp:225-240)
		double bt = 1., br = 1.;
		if (target_T > 0. && cnt->kin_n > 0) {
			bt = pow(3.0 * (double)cnt->kin_n * target_T / cnt->kin[0], 0.4);
			br = (cnt->kin[1] == 0.) ? 1.0 : pow((double)cnt->kin_rotdof * target_T / cnt->kin[1], 0.4);
		}
		cnt->beta[0] = bt;
		cnt->beta[1] = br;
		if (log) {
			log[2] = cnt->kin[0];
			log[3] = cnt->kin[1];
			log[4] = (double)cnt->kin_n;
			log[5] = (double)cnt->kin_rotdof;
		}
	}
}

// VelocityScalingThermostat::apply (global betas): v *= beta_trans, D *= beta_rot
template <bool HAS_ROT>
__global__ void __launch_bounds__(ITPB) k_scale(IntegArgs a, double bt, double br, bool from_device) {
	const uint32_t p = blockIdx.x * ITPB + threadIdx.x;
	if (p >= a.cnt->n_real) return;
	if (from_device) {
		bt = a.cnt->beta[0];
		br = a.cnt->beta[1];
	}
	a.mol.vx[p] *= bt;
	a.mol.vy[p] *= bt;
	a.mol.vz[p] *= bt;
	if (HAS_ROT) {
		a.mol.Dx[p] *= br;
		a.mol.Dy[p] *= br;
		a.mol.Dz[p] *= br;
	}
}

void launch_scale(const IntegArgs& a, double beta_trans, double beta_rot, bool from_device, hipStream_t s) {
	if (a.n_cap == 0) return;
	const dim3 grid((a.n_cap + ITPB - 1) / ITPB);
	if (a.has_rot) hipLaunchKernelGGL(k_scale<true>, grid, dim3(ITPB), 0, s, a, beta_trans, beta_rot, from_device);
	else hipLaunchKernelGGL(k_scale<false>, grid, dim3(ITPB), 0, s, a, beta_trans, beta_rot, from_device);
}

// List mode, unfused drifts: advance the displacement bound by dt * max |v| of this pass and publish the rebuild flag
// (the fused force passes do the same inside k_force_reduce2).
__global__ void __launch_bounds__(RTPB) k_bound_update(DevCounters* cnt, const double* vmax_part, uint32_t nblocks, double dt, double limit,
												  int fresh, uint32_t seq, volatile uint32_t* flag, int local_criterion) {
	double m = 0.;
	uint32_t b = threadIdx.x;
	for (; b + 3u * RTPB < nblocks; b += 4u * RTPB) {
		const double t0 = vmax_part[b], t1 = vmax_part[b + RTPB], t2 = vmax_part[b + 2u * RTPB], t3 = vmax_part[b + 3u * RTPB];
		m = fmax(m, fmax(fmax(t0, t1), fmax(t2, t3)));
	}
	for (; b < nblocks; b += RTPB) m = fmax(m, vmax_part[b]);
	__shared__ double red[RTPB / 64];
	for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		m = 0.;
		for (int i = 0; i < RTPB / 64; ++i) m = fmax(m, red[i]);
		const double b = (fresh ? 0. : cnt->vl_bound) + dt * sqrt(m);
		cnt->vl_bound = b;
		// a drift WITHOUT per-brick bookkeeping counts with the global speed for every brick; with it (k_bound_local ran just before
		// this kernel) a brick neighbourhood's pair bound decides — never earlier than the global one
		bool rebuild = b > limit;
		if (local_criterion) {
			if (fresh) cnt->vl_base = 0.;
			rebuild = rebuild && cnt->vl_local_excess != 0u;
			cnt->vl_local_excess = 0u;
		} else {
			cnt->vl_base = (fresh ? 0. : cnt->vl_base) + dt * sqrt(m);
		}
		if (flag) {
			__threadfence_system();
			*flag = (seq << 1) | (rebuild ? 1u : 0u);
			__threadfence_system();
		}
	}
}
void launch_bound_update(DevCounters* cnt, const double* vmax_part, uint32_t nblocks, double dt, double limit, bool fresh, uint32_t seq,
						 volatile uint32_t* flag, hipStream_t s, bool local_criterion) {
	hipLaunchKernelGGL(k_bound_update, dim3(1), dim3(RTPB), 0, s, cnt, vmax_part, nblocks, dt, limit, fresh ? 1 : 0, seq, flag,
					   local_criterion ? 1 : 0);
}

void launch_kin_reduce(DevCounters* cnt, const double* partials, uint32_t nblocks, hipStream_t s, double target_T, double* log) {
	if (nblocks == 0) return;
	hipLaunchKernelGGL(k_kin_reduce, dim3(1), dim3(RTPB), 0, s, cnt, partials, nblocks, target_T, log);
}

}  // namespace ls1
