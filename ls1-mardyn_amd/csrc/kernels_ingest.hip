// kernels_ingest.hip — device-side ingest / egress of molecule data: the host hands over raw chunks (AoS arrays as the
// reference's Molecule fields, or the records of the reference's binary checkpoint) and the transposition into the
// device SoA happens here, so a 10^8-molecule phase space never exists as host-side Molecule objects or SoA copies.
//
// Record layouts (little endian, packed, no padding) = FullMolecule::writeBinary / BinaryReader::readPhaseSpace:
//   ICRVQD  116 B  id u64 | cid u32 (1-based) | r 3 f64 | v 3 f64 | q 4 f64 | D 3 f64   molecules/FullMolecule.cpp:451-473
//   ICRV     60 B  id u64 | cid u32           | r 3 f64 | v 3 f64                       io/BinaryReader.cpp:179-213
//   IRV      56 B  id u64 |                     r 3 f64 | v 3 f64                       (same reader, cid = 1)
// All record sizes are multiples of 4, so every field is read with aligned 32-bit loads.
#include "common.hpp"

namespace ls1 {

constexpr int GTPB = 256;

__device__ __forceinline__ double load_f64_a4(const uint32_t* w) {
	const unsigned long long lo = w[0], hi = w[1];
	return __longlong_as_double((long long)(lo | (hi << 32)));
}
__device__ __forceinline__ void store_f64_a4(uint32_t* w, double v) {
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	w[0] = (uint32_t)b;
	w[1] = (uint32_t)(b >> 32);
}

__device__ __forceinline__ void ingest_store(const IngestArgs& a, uint32_t p, uint64_t id, int32_t cid, const double r[3],
											 const double v[3], const double q[4], const double D[3]) {
	bool bad = cid < 0 || cid >= a.ncomp;
	for (int d = 0; d < 3; ++d) bad |= !(r[d] >= a.bmin[d] && r[d] < a.bmax[d]);
	if (bad) {
		if (atomicAdd(&a.cnt->err_ingest, 1u) == 0u) a.cnt->err_ingest_first = a.first + p;
		cid = 0;
	}
	const uint32_t o = a.at + p;
	a.dst.x[o] = r[0];
	a.dst.y[o] = r[1];
	a.dst.z[o] = r[2];
	a.dst.vx[o] = v[0];
	a.dst.vy[o] = v[1];
	a.dst.vz[o] = v[2];
	a.dst.id[o] = id;
	a.dst.cid[o] = cid;
	if (a.has_rot) {
		a.dst.q0[o] = q[0];
		a.dst.q1[o] = q[1];
		a.dst.q2[o] = q[2];
		a.dst.q3[o] = q[3];
		a.dst.Dx[o] = D[0];
		a.dst.Dy[o] = D[1];
		a.dst.Dz[o] = D[2];
	}
}

// AoS arrays of one chunk (device staging copies of the caller's id / cid / r / v / q / D) -> SoA at [at, at + n)
__global__ void __launch_bounds__(GTPB) k_ingest_aos(IngestArgs a, const uint64_t* id, const int32_t* cid, const double* r,
													 const double* v, const double* q, const double* D) {
	const uint32_t p = blockIdx.x * GTPB + threadIdx.x;
	if (p >= a.n) return;
	const double rr[3] = {r[3 * (size_t)p], r[3 * (size_t)p + 1], r[3 * (size_t)p + 2]};
	const double vv[3] = {v[3 * (size_t)p], v[3 * (size_t)p + 1], v[3 * (size_t)p + 2]};
	double qq[4] = {1., 0., 0., 0.}, dd[3] = {0., 0., 0.};
	if (q)
		for (int k = 0; k < 4; ++k) qq[k] = q[4 * (size_t)p + k];
	if (D)
		for (int k = 0; k < 3; ++k) dd[k] = D[3 * (size_t)p + k];
	ingest_store(a, p, id[p], cid ? cid[p] : 0, rr, vv, qq, dd);
}

// packed checkpoint records -> SoA.  fmt: 0 ICRVQD, 1 ICRV, 2 IRV.  The component id on disk is 1-based.
__global__ void __launch_bounds__(GTPB) k_ingest_records(IngestArgs a, const uint32_t* rec, int fmt) {
	const uint32_t p = blockIdx.x * GTPB + threadIdx.x;
	if (p >= a.n) return;
	const int words = fmt == 0 ? 29 : (fmt == 1 ? 15 : 14);
	const uint32_t* w = rec + (size_t)p * words;
	const uint64_t id = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
	int32_t cid = 0;
	int o = 2;
	if (fmt != 2) {
		cid = (int32_t)w[2] - 1;
		o = 3;
	}
	double r[3], v[3], q[4] = {1., 0., 0., 0.}, D[3] = {0., 0., 0.};
	for (int k = 0; k < 3; ++k) r[k] = load_f64_a4(w + o + 2 * k);
	for (int k = 0; k < 3; ++k) v[k] = load_f64_a4(w + o + 6 + 2 * k);
	if (fmt == 0) {
		for (int k = 0; k < 4; ++k) q[k] = load_f64_a4(w + o + 12 + 2 * k);
		for (int k = 0; k < 3; ++k) D[k] = load_f64_a4(w + o + 20 + 2 * k);
	}
	ingest_store(a, p, id, cid, r, v, q, D);
}

// SoA [first, first + n) -> packed ICRVQD records (the checkpoint writer's payload), positions wrapped into the box
__global__ void __launch_bounds__(GTPB) k_egress_records(EgressArgs a, uint32_t* rec) {
	const uint32_t p = blockIdx.x * GTPB + threadIdx.x;
	if (p >= a.n) return;
	const uint32_t i = a.first + p;
	uint32_t* w = rec + (size_t)p * 29;
	const uint64_t id = a.src.id[i];
	w[0] = (uint32_t)id;
	w[1] = (uint32_t)(id >> 32);
	w[2] = (uint32_t)(a.src.cid[i] + 1);
	const double* px[3] = {a.x, a.y, a.z};
	for (int k = 0; k < 3; ++k) {
		double r = px[k][i];
		// a molecule may sit up to half a skin outside the box between two re-binning passes: same wrap as k_classify
		if (a.periodic[k]) {
			if (r < a.bmin[k]) {
				r += a.len[k];
				if (r >= a.bmax[k]) r = nextafter(a.bmax[k], a.bmin[k]);
			} else if (r >= a.bmax[k]) {
				r -= a.len[k];
				if (r <= a.bmin[k]) r = a.bmin[k];
			}
		}
		store_f64_a4(w + 3 + 2 * k, r);
	}
	store_f64_a4(w + 9, a.src.vx[i]);
	store_f64_a4(w + 11, a.src.vy[i]);
	store_f64_a4(w + 13, a.src.vz[i]);
	double q[4] = {1., 0., 0., 0.}, D[3] = {0., 0., 0.};
	if (a.has_rot) {
		q[0] = a.src.q0[i];
		q[1] = a.src.q1[i];
		q[2] = a.src.q2[i];
		q[3] = a.src.q3[i];
		D[0] = a.src.Dx[i];
		D[1] = a.src.Dy[i];
		D[2] = a.src.Dz[i];
	}
	for (int k = 0; k < 4; ++k) store_f64_a4(w + 15 + 2 * k, q[k]);
	for (int k = 0; k < 3; ++k) store_f64_a4(w + 23 + 2 * k, D[k]);
}

void launch_ingest_aos(const IngestArgs& a, const uint64_t* id, const int32_t* cid, const double* r, const double* v,
					   const double* q, const double* D, hipStream_t s) {
	if (a.n == 0) return;
	hipLaunchKernelGGL(k_ingest_aos, dim3((a.n + GTPB - 1) / GTPB), dim3(GTPB), 0, s, a, id, cid, r, v, q, D);
}
void launch_ingest_records(const IngestArgs& a, const void* rec, int fmt, hipStream_t s) {
	if (a.n == 0) return;
	hipLaunchKernelGGL(k_ingest_records, dim3((a.n + GTPB - 1) / GTPB), dim3(GTPB), 0, s, a, (const uint32_t*)rec, fmt);
}
void launch_egress_records(const EgressArgs& a, void* rec, hipStream_t s) {
	if (a.n == 0) return;
	hipLaunchKernelGGL(k_egress_records, dim3((a.n + GTPB - 1) / GTPB), dim3(GTPB), 0, s, a, (uint32_t*)rec);
}

}  // namespace ls1
