// leapfrog_body.hpp — upd_postF of step n followed by upd_preF of step n + 1 for ONE rigid molecule, on values (no memory access):
// the body of k_kick_then_kick_drift (kernels_integrate.hip) and of the fused epilogue of the pair-stream force pass
// (kernels_force_mslist.hip).  Both passes must advance a molecule to the same bits — a run may switch between them at any step —
// so FMA contraction is OFF inside these functions whatever the including translation unit says, and the quaternion helpers are
// restated here instead of taken from pairphys.hpp (whose functions follow the translation unit's setting).
//
// Reference: FullMolecule::upd_postF (/root/reference/src/molecules/FullMolecule.cpp:366-389), upd_preF (:334-364),
// Quaternion::rotateinv / differentiate (/root/reference/src/molecules/Quaternion.cpp:63-81, 93-98).
#pragma once
#include "pairphys.hpp"

namespace ls1 {

struct LeapState {
	double x, y, z, vx, vy, vz;
	double q[4];
	V3 D;
};

// angular velocity in the body frame: I^-1 (R(q)^T D)
__device__ __forceinline__ V3 leap_body_omega(const double q[4], V3 D, V3 invI) {
#pragma clang fp contract(off)
	const double w = q[0], x = q[1], y = q[2], z = q[3];
	const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
	const double wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
	const double m0 = ww + xx - yy - zz, m1 = 2. * (xy - wz), m2 = 2. * (wy + xz);
	const double m3 = 2. * (wz + xy), m4 = ww - xx + yy - zz, m5 = 2. * (yz - wx);
	const double m6 = 2. * (xz - wy), m7 = 2. * (wx + yz), m8 = ww - xx - yy + zz;
	const V3 o = {m0 * D.x + m3 * D.y + m6 * D.z, m1 * D.x + m4 * D.y + m7 * D.z, m2 * D.x + m5 * D.y + m8 * D.z};
	return {o.x * invI.x, o.y * invI.y, o.z * invI.z};
}
__device__ __forceinline__ void leap_q_diff(const double q[4], V3 w, double dq[4]) {
#pragma clang fp contract(off)
	dq[0] = .5 * (-q[1] * w.x - q[2] * w.y - q[3] * w.z);
	dq[1] = .5 * (q[0] * w.x - q[3] * w.y + q[2] * w.z);
	dq[2] = .5 * (q[3] * w.x + q[0] * w.y - q[1] * w.z);
	dq[3] = .5 * (-q[2] * w.x + q[1] * w.y + q[0] * w.z);
}

// upd_postF alone (FullMolecule.cpp:366-389): v += dt/2m F, D += dt/2 M; returns m v^2 and I w^2 of the kicked state (q unchanged).
// The body of k_kick and of the post-kick epilogue of the pair-stream list pass.
template <bool HAS_ROT>
__device__ __forceinline__ void leap_post(double dt_halve, double mass, V3 invI, V3 I, V3 F, V3 M, LeapState& s, double& mv2, double& Iw2) {
#pragma clang fp contract(off)
	const double dtInv2m = dt_halve / mass;
	s.vx = s.vx + dtInv2m * F.x;
	s.vy = s.vy + dtInv2m * F.y;
	s.vz = s.vz + dtInv2m * F.z;
	mv2 = mass * (s.vx * s.vx + s.vy * s.vy + s.vz * s.vz);
	Iw2 = 0.;
	if (HAS_ROT) {
		s.D = {s.D.x + dt_halve * M.x, s.D.y + dt_halve * M.y, s.D.z + dt_halve * M.z};
		const V3 w = leap_body_omega(s.q, s.D, invI);
		Iw2 = I.x * w.x * w.x + I.y * w.y * w.y + I.z * w.z * w.z;
	}
}

// s: state at the force evaluation (v and D half a step behind) -> position / orientation one step on, v and D half a step on.
// Returns |v|^2 of the drift (the displacement bound of the neighbour lists).
template <bool HAS_ROT>
__device__ __forceinline__ double leap_post_pre(double dt, double mass, V3 invI, V3 F, V3 M, LeapState& s) {
#pragma clang fp contract(off)
	const double dt_halve = .5 * dt;
	const double dtInv2m = dt_halve / mass;
	double vx = s.vx + dtInv2m * F.x;  // upd_postF
	double vy = s.vy + dtInv2m * F.y;
	double vz = s.vz + dtInv2m * F.z;
	vx += dtInv2m * F.x;  // upd_preF
	vy += dtInv2m * F.y;
	vz += dtInv2m * F.z;
	s.vx = vx;
	s.vy = vy;
	s.vz = vz;
	s.x = s.x + dt * vx;
	s.y = s.y + dt * vy;
	s.z = s.z + dt * vz;
	if (HAS_ROT) {
		double* const q = s.q;
		V3 D = {s.D.x + dt_halve * M.x, s.D.y + dt_halve * M.y, s.D.z + dt_halve * M.z};  // upd_postF
		V3 w = leap_body_omega(q, D, invI);
		double dq[4], qh[4];
		leap_q_diff(q, w, dq);
		for (int k = 0; k < 4; ++k) qh[k] = dq[k] * dt_halve + q[k];
		double qcorr = 1. / sqrt(qh[0] * qh[0] + qh[1] * qh[1] + qh[2] * qh[2] + qh[3] * qh[3]);
		for (int k = 0; k < 4; ++k) qh[k] *= qcorr;
		D.x += dt_halve * M.x;
		D.y += dt_halve * M.y;
		D.z += dt_halve * M.z;
		w = leap_body_omega(qh, D, invI);
		leap_q_diff(qh, w, dq);
		for (int k = 0; k < 4; ++k) q[k] += dq[k] * dt;
		qcorr = 1. / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
		for (int k = 0; k < 4; ++k) q[k] *= qcorr;
		s.D = D;
	}
	return vx * vx + vy * vy + vz * vz;
}

}  // namespace ls1
