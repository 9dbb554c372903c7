/*
 * ls1_oracle.c — CPU restatement of the ls1-MarDyn linked-cell pair-force hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (ls1-mardyn_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against golden vectors produced
 * by the real reference (oracle/_ref/refdump, built from /root/reference sources by oracle/ref_build/Makefile)
 * for every fixture of the reference's own VectorizedCellProcessorTest / ForceCalculationTest plus periodic
 * and multi-step cases, and against the analytic known answers F=(+-24,+-24,0), U=0, virial=96 / F=0, U=-4.
 *
 * Plain scalar C99, FP64 throughout (the reference's MARDYN_DPDP build).  Each function cites the reference
 * file:line it restates (paths relative to /root/reference/src).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define LJ_STRIDE 7 /* x y z m eps sigma shift6(self) */
#define CH_STRIDE 5 /* x y z m q */
#define DP_STRIDE 7 /* x y z ex ey ez absMy */
#define QP_STRIDE 7 /* x y z ex ey ez absQ */

typedef struct ls1o_sys {
	int ncomp;
	int *nlj, *nc, *nd, *nq; /* per component site counts */
	int *olj, *oc, *od, *oq; /* per component offsets into the flat site tables */
	double *lj, *ch, *dp, *qp;
	double *mass, *I, *invI; /* [ncomp], [ncomp][3], [ncomp][3] */
	int ncenters;            /* total LJ centres over all components (VCP "lookup id" space) */
	double *eps24, *sig2, *shift6; /* [ncenters][ncenters] */
	double epsRF, rc, rcLJ, epsRFInvrc3;
} ls1o_sys;

/* ------------------------------------------------------------------------------------------------------------------
 * Parameter tables.  Restates Comp2Param::initialize (molecules/Comp2Param.cpp:10-97) for the LJ part and the table
 * layout of VectorizedCellProcessor::VectorizedCellProcessor (particleContainer/adapter/VectorizedCellProcessor.cpp:
 * 41-83): global centre index = running count of LJ centres over components (ensemble/EnsembleBase.cpp:91-101).
 * mix = (xi, eta) per unordered component pair i<j in reader order (io/ASCIIReader.cpp:221-230).
 * ----------------------------------------------------------------------------------------------------------------*/
ls1o_sys *ls1o_create(int ncomp, const int *nlj, const int *nc, const int *nd, const int *nq, const double *lj,
					  const double *ch, const double *dp, const double *qp, const double *mass, const double *I,
					  const double *mix, double epsRF, double rc, double rcLJ) {
	ls1o_sys *s = (ls1o_sys *)calloc(1, sizeof(ls1o_sys));
	s->ncomp = ncomp;
	s->nlj = (int *)malloc(sizeof(int) * ncomp);
	s->nc = (int *)malloc(sizeof(int) * ncomp);
	s->nd = (int *)malloc(sizeof(int) * ncomp);
	s->nq = (int *)malloc(sizeof(int) * ncomp);
	s->olj = (int *)malloc(sizeof(int) * (ncomp + 1));
	s->oc = (int *)malloc(sizeof(int) * (ncomp + 1));
	s->od = (int *)malloc(sizeof(int) * (ncomp + 1));
	s->oq = (int *)malloc(sizeof(int) * (ncomp + 1));
	s->olj[0] = s->oc[0] = s->od[0] = s->oq[0] = 0;
	for (int c = 0; c < ncomp; ++c) {
		s->nlj[c] = nlj[c];
		s->nc[c] = nc[c];
		s->nd[c] = nd[c];
		s->nq[c] = nq[c];
		s->olj[c + 1] = s->olj[c] + nlj[c];
		s->oc[c + 1] = s->oc[c] + nc[c];
		s->od[c + 1] = s->od[c] + nd[c];
		s->oq[c + 1] = s->oq[c] + nq[c];
	}
	size_t tl = s->olj[ncomp], tc = s->oc[ncomp], td = s->od[ncomp], tq = s->oq[ncomp];
	s->lj = (double *)malloc(sizeof(double) * (tl * LJ_STRIDE + 1));
	s->ch = (double *)malloc(sizeof(double) * (tc * CH_STRIDE + 1));
	s->dp = (double *)malloc(sizeof(double) * (td * DP_STRIDE + 1));
	s->qp = (double *)malloc(sizeof(double) * (tq * QP_STRIDE + 1));
	memcpy(s->lj, lj, sizeof(double) * tl * LJ_STRIDE);
	memcpy(s->ch, ch, sizeof(double) * tc * CH_STRIDE);
	memcpy(s->dp, dp, sizeof(double) * td * DP_STRIDE);
	memcpy(s->qp, qp, sizeof(double) * tq * QP_STRIDE);
	s->mass = (double *)malloc(sizeof(double) * ncomp);
	s->I = (double *)malloc(sizeof(double) * ncomp * 3);
	s->invI = (double *)malloc(sizeof(double) * ncomp * 3);
	for (int c = 0; c < ncomp; ++c) {
		s->mass[c] = mass[c];
		for (int d = 0; d < 3; ++d) {
			/* molecules/FullMolecule.h:88-101 */
			s->I[3 * c + d] = I[3 * c + d];
			s->invI[3 * c + d] = (I[3 * c + d] != 0.) ? 1. / I[3 * c + d] : 0.;
		}
	}
	s->epsRF = epsRF;
	s->rc = rc;
	s->rcLJ = rcLJ;
	/* VectorizedCellProcessor.cpp:24 */
	s->epsRFInvrc3 = 2. * (epsRF - 1.) / ((rc * rc * rc) * (2. * epsRF + 1.));
	s->ncenters = (int)tl;
	size_t n2 = tl * tl;
	s->eps24 = (double *)calloc(n2 + 1, sizeof(double));
	s->sig2 = (double *)calloc(n2 + 1, sizeof(double));
	s->shift6 = (double *)calloc(n2 + 1, sizeof(double));
	int mixpos = 0;
	for (int ci = 0; ci < ncomp; ++ci) {
		/* same component: Comp2Param.cpp:24-40 */
		for (int a = 0; a < nlj[ci]; ++a) {
			const double *sa = s->lj + (size_t)(s->olj[ci] + a) * LJ_STRIDE;
			for (int b = 0; b < nlj[ci]; ++b) {
				const double *sb = s->lj + (size_t)(s->olj[ci] + b) * LJ_STRIDE;
				size_t k = (size_t)(s->olj[ci] + a) * tl + (s->olj[ci] + b);
				double sg = .5 * (sa[5] + sb[5]);
				s->eps24[k] = 24. * sqrt(sa[4] * sb[4]);
				s->sig2[k] = sg * sg;
				s->shift6[k] = sa[6];
			}
		}
		/* unlike components: Comp2Param.cpp:42-95 */
		for (int cj = ci + 1; cj < ncomp; ++cj) {
			double xi = mix[2 * mixpos], eta = mix[2 * mixpos + 1];
			++mixpos;
			for (int a = 0; a < nlj[ci]; ++a) {
				const double *sa = s->lj + (size_t)(s->olj[ci] + a) * LJ_STRIDE;
				for (int b = 0; b < nlj[cj]; ++b) {
					const double *sb = s->lj + (size_t)(s->olj[cj] + b) * LJ_STRIDE;
					double e24 = 24. * xi * sqrt(sa[4] * sb[4]);
					double sg = eta * .5 * (sa[5] + sb[5]);
					double sg2 = sg * sg;
					double p2 = sg2 / (rcLJ * rcLJ);
					double p6 = p2 * p2 * p2;
					double sh = e24 * (p6 - p6 * p6);
					size_t kij = (size_t)(s->olj[ci] + a) * tl + (s->olj[cj] + b);
					size_t kji = (size_t)(s->olj[cj] + b) * tl + (s->olj[ci] + a);
					s->eps24[kij] = s->eps24[kji] = e24;
					s->sig2[kij] = s->sig2[kji] = sg2;
					s->shift6[kij] = s->shift6[kji] = sh;
				}
			}
		}
	}
	return s;
}

void ls1o_destroy(ls1o_sys *s) {
	if (!s) return;
	free(s->nlj); free(s->nc); free(s->nd); free(s->nq);
	free(s->olj); free(s->oc); free(s->od); free(s->oq);
	free(s->lj); free(s->ch); free(s->dp); free(s->qp);
	free(s->mass); free(s->I); free(s->invI);
	free(s->eps24); free(s->sig2); free(s->shift6);
	free(s);
}

/* LJ table export so tests can compare the device library's table with the oracle's. */
void ls1o_lj_table(const ls1o_sys *s, double *eps24, double *sig2, double *shift6) {
	size_t n2 = (size_t)s->ncenters * s->ncenters;
	memcpy(eps24, s->eps24, n2 * sizeof(double));
	memcpy(sig2, s->sig2, n2 * sizeof(double));
	memcpy(shift6, s->shift6, n2 * sizeof(double));
}

/* ------------------------------------------------------------------------------------------------------------------
 * Quaternion helpers: molecules/Quaternion.cpp:45-61 (rotate), :63-81 (rotateinv), :93-98 (differentiate).
 * q = (w, x, y, z).
 * ----------------------------------------------------------------------------------------------------------------*/
static void q_rotate(const double q[4], const double d[3], double o[3]) {
	double ww = q[0] * q[0], xx = q[1] * q[1], yy = q[2] * q[2], zz = q[3] * q[3];
	double wx = q[0] * q[1], wy = q[0] * q[2], wz = q[0] * q[3];
	double xy = q[1] * q[2], xz = q[1] * q[3], yz = q[2] * q[3];
	o[0] = (ww + xx - yy - zz) * d[0] + 2. * (xy - wz) * d[1] + 2. * (wy + xz) * d[2];
	o[1] = 2. * (wz + xy) * d[0] + (ww - xx + yy - zz) * d[1] + 2. * (yz - wx) * d[2];
	o[2] = 2. * (xz - wy) * d[0] + 2. * (wx + yz) * d[1] + (ww - xx - yy + zz) * d[2];
}
static void q_rotateinv(const double q[4], const double d[3], double o[3]) {
	double ww = q[0] * q[0], xx = q[1] * q[1], yy = q[2] * q[2], zz = q[3] * q[3];
	double wx = q[0] * q[1], wy = q[0] * q[2], wz = q[0] * q[3];
	double xy = q[1] * q[2], xz = q[1] * q[3], yz = q[2] * q[3];
	o[0] = (ww + xx - yy - zz) * d[0] + 2. * (xy + wz) * d[1] + 2. * (xz - wy) * d[2];
	o[1] = 2. * (xy - wz) * d[0] + (ww - xx + yy - zz) * d[1] + 2. * (yz + wx) * d[2];
	o[2] = 2. * (xz + wy) * d[0] + 2. * (yz - wx) * d[1] + (ww - xx - yy + zz) * d[2];
}
static void q_diff(const double q[4], const double w[3], double dq[4]) {
	dq[0] = .5 * (-q[1] * w[0] - q[2] * w[1] - q[3] * w[2]);
	dq[1] = .5 * (q[0] * w[0] - q[3] * w[1] + q[2] * w[2]);
	dq[2] = .5 * (q[3] * w[0] + q[0] * w[1] - q[1] * w[2]);
	dq[3] = .5 * (-q[2] * w[0] + q[1] * w[1] + q[0] * w[2]);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Pair bodies: restatement of molecules/potforce.h (scalar reference physics, the oracle of the reference's own
 * differential tests) — PotForceLJ :18-30, PotForce2Dipole :36-80, PotForce2Quadrupole :86-133,
 * PotForceDiQuadrupole :139-184, PotForce2Charge :190-199, PotForceChargeQuadrupole :205-231,
 * PotForceChargeDipole :237-263.
 * ----------------------------------------------------------------------------------------------------------------*/
static void pf_lj(const double dr[3], double dr2, double eps24, double sig2, double f[3], double *u6) {
	double invdr2 = 1. / dr2;
	double lj6 = sig2 * invdr2;
	lj6 = lj6 * lj6 * lj6;
	double lj12 = lj6 * lj6;
	double lj12m6 = lj12 - lj6;
	*u6 = eps24 * lj12m6;
	double fac = eps24 * (lj12 + lj12m6) * invdr2;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d];
}

static void cross(const double a[3], const double b[3], double o[3]) {
	o[0] = a[1] * b[2] - a[2] * b[1];
	o[1] = a[2] * b[0] - a[0] * b[2];
	o[2] = a[0] * b[1] - a[1] * b[0];
}

static void pf_2dipole(const double dr[3], double dr2, const double *eii, const double *ejj, double my2, double rffac,
					   double f[3], double m1[3], double m2[3], double *u, double *MyRF) {
	double invdr2 = 1. / dr2, invdr1 = sqrt(invdr2);
	double myfac = my2 * invdr2 * invdr1;
	double costi = 0., costj = 0., cosgij = 0.;
	for (int d = 0; d < 3; ++d) {
		costi += eii[d] * dr[d];
		costj += ejj[d] * dr[d];
		cosgij += eii[d] * ejj[d];
	}
	costi *= invdr1;
	costj *= invdr1;
	*u = myfac * (cosgij - 3. * costi * costj);
	*MyRF -= rffac * cosgij;
	double partialRijInvdr1 = -3. * (*u) * invdr2;
	double partialTiInvdr1 = -myfac * 3. * costj * invdr1;
	double partialTjInvdr1 = -myfac * 3. * costi * invdr1;
	double partialGij = myfac;
	double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d] - partialTiInvdr1 * eii[d] - partialTjInvdr1 * ejj[d];
	double eiXej[3], eXrij[3];
	cross(eii, ejj, eiXej);
	cross(eii, dr, eXrij);
	for (int d = 0; d < 3; ++d) m1[d] = -partialTiInvdr1 * eXrij[d] + (-partialGij + rffac) * eiXej[d];
	cross(ejj, dr, eXrij);
	for (int d = 0; d < 3; ++d) m2[d] = -partialTjInvdr1 * eXrij[d] + (partialGij - rffac) * eiXej[d];
}

static void pf_2quadrupole(const double dr[3], double dr2, const double *eii, const double *ejj, double q2075,
						   double f[3], double m1[3], double m2[3], double *u) {
	double invdr2 = 1. / dr2, invdr1 = sqrt(invdr2);
	double qfac = q2075 * invdr2 * invdr2 * invdr1;
	double costi = 0., costj = 0., cosgij = 0.;
	for (int d = 0; d < 3; ++d) {
		costi += eii[d] * dr[d];
		costj += ejj[d] * dr[d];
		cosgij += eii[d] * ejj[d];
	}
	costi *= invdr1;
	costj *= invdr1;
	double cos2ti = costi * costi, cos2tj = costj * costj;
	double term = (cosgij - 5. * costi * costj);
	*u = qfac * (1. - 5. * (cos2ti + cos2tj) - 15. * cos2ti * cos2tj + 2. * term * term);
	double partialRijInvdr1 = -5. * (*u) * invdr2;
	double partialTiInvdr1 = -qfac * 10. * (costi + 3. * costi * cos2tj + 2. * costj * term) * invdr1;
	double partialTjInvdr1 = -qfac * 10. * (costj + 3. * cos2ti * costj + 2. * costi * term) * invdr1;
	double partialGij = qfac * 4. * term;
	double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d] - partialTiInvdr1 * eii[d] - partialTjInvdr1 * ejj[d];
	double eiXej[3], eXrij[3];
	cross(eii, ejj, eiXej);
	cross(eii, dr, eXrij);
	for (int d = 0; d < 3; ++d) m1[d] = -partialTiInvdr1 * eXrij[d] - partialGij * eiXej[d];
	cross(ejj, dr, eXrij);
	for (int d = 0; d < 3; ++d) m2[d] = -partialTjInvdr1 * eXrij[d] + partialGij * eiXej[d];
}

static void pf_diquadrupole(const double dr[3], double dr2, const double *eii, const double *ejj, double myq15,
							double f[3], double m1[3], double m2[3], double *u) {
	double invdr2 = 1. / dr2, invdr1 = sqrt(invdr2);
	double myqfac = myq15 * invdr2 * invdr2;
	double costi = 0., costj = 0., cosgij = 0.;
	for (int d = 0; d < 3; ++d) {
		costi += eii[d] * dr[d];
		costj += ejj[d] * dr[d];
		cosgij += eii[d] * ejj[d];
	}
	costi *= invdr1;
	costj *= invdr1;
	double cos2tj = costj * costj;
	*u = myqfac * (-costi * (5. * cos2tj - 1.) + 2. * cosgij * costj);
	double partialRijInvdr1 = -4. * (*u) * invdr2;
	double partialTiInvdr1 = myqfac * (-5. * cos2tj + 1.) * invdr1;
	double partialTjInvdr1 = myqfac * 2. * (-5. * costi * costj + cosgij) * invdr1;
	double partialGij = myqfac * 2. * costj;
	double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d] - partialTiInvdr1 * eii[d] - partialTjInvdr1 * ejj[d];
	double eiXej[3], eXrij[3];
	cross(eii, ejj, eiXej);
	cross(eii, dr, eXrij);
	for (int d = 0; d < 3; ++d) m1[d] = -partialTiInvdr1 * eXrij[d] - partialGij * eiXej[d];
	cross(ejj, dr, eXrij);
	for (int d = 0; d < 3; ++d) m2[d] = -partialTjInvdr1 * eXrij[d] + partialGij * eiXej[d];
}

static void pf_2charge(const double dr[3], double dr2, double q1q2, double f[3], double *u) {
	double invdr2 = 1.0 / dr2, invdr = sqrt(invdr2);
	*u = q1q2 * invdr;
	double fac = (*u) * invdr2;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d];
}

static void pf_chargequadrupole(const double dr[3], double dr2, const double *ejj, double qQ05, double f[3],
								double m2[3], double *u) {
	double invdr2 = 1.0 / dr2, invdr = sqrt(invdr2);
	double costj = 0;
	for (int d = 0; d < 3; ++d) costj += ejj[d] * dr[d];
	costj *= invdr;
	double qQinv4dr3 = qQ05 * invdr * invdr2;
	*u = qQinv4dr3 * (3.0 * costj * costj - 1);
	double partialRijInvdr1 = -3.0 * (*u) * invdr2;
	double partialTjInvdr1 = 6.0 * costj * qQinv4dr3 * invdr;
	double fac = costj * partialTjInvdr1 * invdr - partialRijInvdr1;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d] - partialTjInvdr1 * ejj[d];
	double minuseXrij[3];
	cross(dr, ejj, minuseXrij); /* = -(e x r) */
	for (int d = 0; d < 3; ++d) m2[d] = partialTjInvdr1 * minuseXrij[d];
}

static void pf_chargedipole(const double dr[3], double dr2, const double *ejj, double minusqmy, double f[3],
							double m2[3], double *u) {
	double invdr2 = 1.0 / dr2, invdr = sqrt(invdr2);
	double costj = 0;
	for (int d = 0; d < 3; ++d) costj += ejj[d] * dr[d];
	costj *= invdr;
	double uInvcostj = minusqmy * invdr2;
	*u = uInvcostj * costj;
	double partialTjInvdr1 = uInvcostj * invdr;
	double fac = 3.0 * (*u) * invdr2;
	for (int d = 0; d < 3; ++d) f[d] = fac * dr[d] - partialTjInvdr1 * ejj[d];
	double minuseXrij[3];
	cross(dr, ejj, minuseXrij);
	for (int d = 0; d < 3; ++d) m2[d] = partialTjInvdr1 * minuseXrij[d];
}

/* ------------------------------------------------------------------------------------------------------------------
 * Working set: molecules (real + halo copies) with rotated site geometry, the per-site force accumulators of
 * CellDataSoA (adapter/CellDataSoA.h:41-74) and per-molecule M / Vi accumulators.
 * ----------------------------------------------------------------------------------------------------------------*/
typedef struct {
	size_t n, cap;
	double *r;      /* [n][3] molecule centre */
	int *cid;       /* component */
	long *src;      /* index of the real molecule this entry is / is a copy of */
	size_t *soff;   /* offset of the molecule's first site in the site arrays (order LJ, C, D, Q) */
	size_t nsites, scap;
	double *sd;     /* [nsites][3] rotated site offset d (FullMolecule.h:217-232)  */
	double *se;     /* [nsites][3] rotated orientation e (zero for LJ/charge)       */
	double *sF;     /* [nsites][3] site force accumulators                          */
	double *M;      /* [n][3] sum of site torques (dipole/quadrupole M)             */
	double *Vi;     /* [n][3] molecule virial (already halved as in calcFM)         */
} work_t;

static void work_reserve(work_t *w, size_t n, size_t ns) {
	if (n > w->cap) {
		size_t c = w->cap ? w->cap : 64;
		while (c < n) c *= 2;
		w->r = (double *)realloc(w->r, c * 3 * sizeof(double));
		w->cid = (int *)realloc(w->cid, c * sizeof(int));
		w->src = (long *)realloc(w->src, c * sizeof(long));
		w->soff = (size_t *)realloc(w->soff, (c + 1) * sizeof(size_t));
		w->M = (double *)realloc(w->M, c * 3 * sizeof(double));
		w->Vi = (double *)realloc(w->Vi, c * 3 * sizeof(double));
		w->cap = c;
	}
	if (ns > w->scap) {
		size_t c = w->scap ? w->scap : 64;
		while (c < ns) c *= 2;
		w->sd = (double *)realloc(w->sd, c * 3 * sizeof(double));
		w->se = (double *)realloc(w->se, c * 3 * sizeof(double));
		w->sF = (double *)realloc(w->sF, c * 3 * sizeof(double));
		w->scap = c;
	}
}

static int comp_nsites(const ls1o_sys *s, int c) { return s->nlj[c] + s->nc[c] + s->nd[c] + s->nq[c]; }

/* FullMolecule::setupSoACache (molecules/FullMolecule.cpp:714-770): normalise q, rotate offsets and axes. */
static void work_add(const ls1o_sys *s, work_t *w, const double r[3], const double qin[4], int cid, long src) {
	int ns = comp_nsites(s, cid);
	work_reserve(w, w->n + 1, w->nsites + ns);
	size_t i = w->n++;
	for (int d = 0; d < 3; ++d) {
		w->r[3 * i + d] = r[d];
		w->M[3 * i + d] = 0.;
		w->Vi[3 * i + d] = 0.;
	}
	w->cid[i] = cid;
	w->src[i] = src;
	w->soff[i] = w->nsites;
	double q[4];
	double mag = sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
	for (int k = 0; k < 4; ++k) q[k] = qin[k] / mag; /* Quaternion::normalize, Quaternion.h:40-42 */
	size_t k = w->nsites;
	for (int a = 0; a < s->nlj[cid]; ++a, ++k) {
		q_rotate(q, s->lj + (size_t)(s->olj[cid] + a) * LJ_STRIDE, w->sd + 3 * k);
		w->se[3 * k] = w->se[3 * k + 1] = w->se[3 * k + 2] = 0.;
	}
	for (int a = 0; a < s->nc[cid]; ++a, ++k) {
		q_rotate(q, s->ch + (size_t)(s->oc[cid] + a) * CH_STRIDE, w->sd + 3 * k);
		w->se[3 * k] = w->se[3 * k + 1] = w->se[3 * k + 2] = 0.;
	}
	for (int a = 0; a < s->nd[cid]; ++a, ++k) {
		const double *p = s->dp + (size_t)(s->od[cid] + a) * DP_STRIDE;
		q_rotate(q, p, w->sd + 3 * k);
		q_rotate(q, p + 3, w->se + 3 * k);
	}
	for (int a = 0; a < s->nq[cid]; ++a, ++k) {
		const double *p = s->qp + (size_t)(s->oq[cid] + a) * QP_STRIDE;
		q_rotate(q, p, w->sd + 3 * k);
		q_rotate(q, p + 3, w->se + 3 * k);
	}
	for (size_t t = w->nsites; t < k; ++t) w->sF[3 * t] = w->sF[3 * t + 1] = w->sF[3 * t + 2] = 0.;
	w->nsites = k;
	w->soff[w->n] = k;
}

static void work_free(work_t *w) {
	free(w->r); free(w->cid); free(w->src); free(w->soff);
	free(w->sd); free(w->se); free(w->sF); free(w->M); free(w->Vi);
}

typedef struct {
	double upot6lj, upotXpoles, virial, myRF;
} macro_t;

/* ------------------------------------------------------------------------------------------------------------------
 * Molecule pair: restatement of PotForce (molecules/potforce.h:282-503) with the bookkeeping of
 * ParticlePairs2PotForceAdapter::processPair (adapter/ParticlePairs2PotForceAdapter.h:150-181):
 * `macro` != 0 -> MOLECULE_MOLECULE (sum U, virial, MyRF), 0 -> MOLECULE_HALOMOLECULE (forces only).
 * drm = r_i - r_j.  Site forces add on i / subtract on j (or the swapped roles of :385-459).
 * ----------------------------------------------------------------------------------------------------------------*/
static void potforce(const ls1o_sys *s, work_t *w, size_t mi, size_t mj, const double drm[3], int calcLJ, int macro,
					 macro_t *acc) {
	const int ci = w->cid[mi], cj = w->cid[mj];
	const double *ri = w->r + 3 * mi, *rj = w->r + 3 * mj;
	double Virial[3] = {0., 0., 0.};
	double f[3], m1[3], m2[3], u, drs[3], dr2;
	double upot6 = 0., upotX = 0., myRF = 0.;
	size_t bi = w->soff[mi], bj = w->soff[mj];
	const int nc1 = s->nlj[ci], nc2 = s->nlj[cj];
	const int ne1 = s->nc[ci], ne2 = s->nc[cj];
	const int nd1 = s->nd[ci], nd2 = s->nd[cj];
	const int nq1 = s->nq[ci], nq2 = s->nq[cj];
	/* site index bases inside the molecule: LJ, charge, dipole, quadrupole */
	size_t iL = bi, iC = bi + nc1, iD = iC + ne1, iQ = iD + nd1;
	size_t jL = bj, jC = bj + nc2, jD = jC + ne2, jQ = jD + nd2;
#define ABS_I(k, out) for (int d_ = 0; d_ < 3; ++d_) out[d_] = ri[d_] + w->sd[3 * (k) + d_]
#define ABS_J(k, out) for (int d_ = 0; d_ < 3; ++d_) out[d_] = rj[d_] + w->sd[3 * (k) + d_]
#define DIST(a, b) do { for (int d_ = 0; d_ < 3; ++d_) drs[d_] = a[d_] - b[d_]; dr2 = drs[0]*drs[0] + drs[1]*drs[1] + drs[2]*drs[2]; } while (0)
#define FADD(k) for (int d_ = 0; d_ < 3; ++d_) w->sF[3 * (k) + d_] += f[d_]
#define FSUB(k) for (int d_ = 0; d_ < 3; ++d_) w->sF[3 * (k) + d_] -= f[d_]
#define MADD(m, v) for (int d_ = 0; d_ < 3; ++d_) w->M[3 * (m) + d_] += v[d_]
#define VADD for (int d_ = 0; d_ < 3; ++d_) Virial[d_] += 0.5 * drm[d_] * f[d_]
#define VSUB for (int d_ = 0; d_ < 3; ++d_) Virial[d_] -= 0.5 * drm[d_] * f[d_]
	double dii[3], djj[3];
	/* LJ-LJ: potforce.h:295-320 */
	if (calcLJ) {
		for (int si = 0; si < nc1; ++si) {
			ABS_I(iL + si, dii);
			for (int sj = 0; sj < nc2; ++sj) {
				ABS_J(jL + sj, djj);
				DIST(dii, djj);
				size_t k = (size_t)(s->olj[ci] + si) * s->ncenters + (s->olj[cj] + sj);
				pf_lj(drs, dr2, s->eps24[k], s->sig2[k], f, &u);
				u += s->shift6[k];
				FADD(iL + si);
				FSUB(jL + sj);
				upot6 += u;
				VADD;
			}
		}
	}
	for (int si = 0; si < ne1; ++si) {
		ABS_I(iC + si, dii);
		double qi = s->ch[(size_t)(s->oc[ci] + si) * CH_STRIDE + 4];
		/* charge-charge :332-346 */
		for (int sj = 0; sj < ne2; ++sj) {
			ABS_J(jC + sj, djj);
			double qj = s->ch[(size_t)(s->oc[cj] + sj) * CH_STRIDE + 4];
			DIST(dii, djj);
			pf_2charge(drs, dr2, qi * qj, f, &u);
			FADD(iC + si);
			FSUB(jC + sj);
			upotX += u;
			VADD;
		}
		/* charge-quadrupole :347-363 */
		for (int sj = 0; sj < nq2; ++sj) {
			ABS_J(jQ + sj, djj);
			double Qj = s->qp[(size_t)(s->oq[cj] + sj) * QP_STRIDE + 6];
			DIST(dii, djj);
			pf_chargequadrupole(drs, dr2, w->se + 3 * (jQ + sj), 0.5 * qi * Qj, f, m2, &u);
			FADD(iC + si);
			FSUB(jQ + sj);
			MADD(mj, m2);
			upotX += u;
			VADD;
		}
		/* charge-dipole :364-380 */
		for (int sj = 0; sj < nd2; ++sj) {
			ABS_J(jD + sj, djj);
			double myj = s->dp[(size_t)(s->od[cj] + sj) * DP_STRIDE + 6];
			DIST(dii, djj);
			pf_chargedipole(drs, dr2, w->se + 3 * (jD + sj), -qi * myj, f, m2, &u);
			FADD(iC + si);
			FSUB(jD + sj);
			MADD(mj, m2);
			upotX += u;
			VADD;
		}
	}
	for (int si = 0; si < nq1; ++si) {
		ABS_I(iQ + si, dii);
		const double *eii = w->se + 3 * (iQ + si);
		double Qi = s->qp[(size_t)(s->oq[ci] + si) * QP_STRIDE + 6];
		/* quadrupole-charge :387-402 (roles swapped: distance j-i, force subtracts on i) */
		for (int sj = 0; sj < ne2; ++sj) {
			ABS_J(jC + sj, djj);
			double qj = s->ch[(size_t)(s->oc[cj] + sj) * CH_STRIDE + 4];
			DIST(djj, dii);
			pf_chargequadrupole(drs, dr2, eii, 0.5 * qj * Qi, f, m1, &u);
			FSUB(iQ + si);
			FADD(jC + sj);
			MADD(mi, m1);
			upotX += u;
			VSUB;
		}
		/* quadrupole-quadrupole :403-421 */
		for (int sj = 0; sj < nq2; ++sj) {
			ABS_J(jQ + sj, djj);
			double Qj = s->qp[(size_t)(s->oq[cj] + sj) * QP_STRIDE + 6];
			DIST(dii, djj);
			pf_2quadrupole(drs, dr2, eii, w->se + 3 * (jQ + sj), .75 * Qi * Qj, f, m1, m2, &u);
			FADD(iQ + si);
			FSUB(jQ + sj);
			MADD(mi, m1);
			MADD(mj, m2);
			upotX += u;
			VADD;
		}
		/* quadrupole-dipole :422-440 */
		for (int sj = 0; sj < nd2; ++sj) {
			ABS_J(jD + sj, djj);
			double myj = s->dp[(size_t)(s->od[cj] + sj) * DP_STRIDE + 6];
			DIST(djj, dii);
			pf_diquadrupole(drs, dr2, w->se + 3 * (jD + sj), eii, 1.5 * Qi * myj, f, m2, m1, &u);
			FSUB(iQ + si);
			FADD(jD + sj);
			MADD(mi, m1);
			MADD(mj, m2);
			upotX += u;
			VSUB;
		}
	}
	for (int si = 0; si < nd1; ++si) {
		ABS_I(iD + si, dii);
		const double *eii = w->se + 3 * (iD + si);
		double myi = s->dp[(size_t)(s->od[ci] + si) * DP_STRIDE + 6];
		/* dipole-charge :445-460 */
		for (int sj = 0; sj < ne2; ++sj) {
			ABS_J(jC + sj, djj);
			double qj = s->ch[(size_t)(s->oc[cj] + sj) * CH_STRIDE + 4];
			DIST(djj, dii);
			pf_chargedipole(drs, dr2, eii, -qj * myi, f, m1, &u);
			FSUB(iD + si);
			FADD(jC + sj);
			MADD(mi, m1);
			upotX += u;
			VSUB;
		}
		/* dipole-quadrupole :461-478 */
		for (int sj = 0; sj < nq2; ++sj) {
			ABS_J(jQ + sj, djj);
			double Qj = s->qp[(size_t)(s->oq[cj] + sj) * QP_STRIDE + 6];
			DIST(dii, djj);
			pf_diquadrupole(drs, dr2, eii, w->se + 3 * (jQ + sj), 1.5 * myi * Qj, f, m1, m2, &u);
			FADD(iD + si);
			FSUB(jQ + sj);
			MADD(mi, m1);
			MADD(mj, m2);
			upotX += u;
			VADD;
		}
		/* dipole-dipole :479-497 */
		for (int sj = 0; sj < nd2; ++sj) {
			ABS_J(jD + sj, djj);
			double myj = s->dp[(size_t)(s->od[cj] + sj) * DP_STRIDE + 6];
			double my2 = myi * myj;
			double rffac = my2 * s->epsRFInvrc3;
			DIST(dii, djj);
			pf_2dipole(drs, dr2, eii, w->se + 3 * (jD + sj), my2, rffac, f, m1, m2, &u, &myRF);
			FADD(iD + si);
			FSUB(jD + sj);
			MADD(mi, m1);
			MADD(mj, m2);
			upotX += u;
			VADD;
		}
	}
	/* mi.Viadd(Virial); mj.Viadd(Virial) :499-500 */
	for (int d = 0; d < 3; ++d) {
		w->Vi[3 * mi + d] += Virial[d];
		w->Vi[3 * mj + d] += Virial[d];
	}
	if (macro) {
		acc->upot6lj += upot6;
		acc->upotXpoles += upotX;
		acc->myRF += myRF;
		acc->virial += 2 * (Virial[0] + Virial[1] + Virial[2]); /* ParticlePairs2PotForceAdapter.h:164 */
	}
#undef ABS_I
#undef ABS_J
#undef DIST
#undef FADD
#undef FSUB
#undef MADD
#undef VADD
#undef VSUB
}

/* ------------------------------------------------------------------------------------------------------------------
 * Linked-cell geometry: LinkedCells::rebuild (particleContainer/LinkedCells.cpp:136-204; note the float-rounded
 * cutoff at :152) and getCellIndexOfPoint (:830-886) for cellsInCutoff = 1.
 * ----------------------------------------------------------------------------------------------------------------*/
typedef struct {
	int box[3], dims[3];
	double bmin[3], bmax[3], clen[3], crec[3], hmin[3], hmax[3];
} grid_t;

static void grid_init(grid_t *g, const double bmin[3], const double bmax[3], double cutoff) {
	float rc = (float)(cutoff / 1);
	for (int d = 0; d < 3; ++d) {
		g->bmin[d] = bmin[d];
		g->bmax[d] = bmax[d];
		g->box[d] = (int)floor((bmax[d] - bmin[d]) / rc);
		if (g->box[d] < 1) g->box[d] = 1; /* reference exits ("region too small"); keep the oracle total */
		g->dims[d] = g->box[d] + 2;
		double diff = bmax[d] - bmin[d];
		g->clen[d] = diff / g->box[d];
		g->crec[d] = g->box[d] / diff;
		g->hmin[d] = bmin[d] - g->clen[d];
		g->hmax[d] = bmax[d] + g->clen[d];
	}
}

static long grid_cell(const grid_t *g, const double p[3]) {
	int ci[3];
	for (int d = 0; d < 3; ++d) {
		double x = p[d];
		if (x <= g->hmin[d]) x += g->clen[d] * 0.5;
		else if (x >= g->hmax[d]) x -= g->clen[d] * 0.5;
		int c = (int)floor((x - g->bmin[d]) * g->crec[d]) + 1;
		if (c < 0) c = 0;
		if (c > g->dims[d] - 1) c = g->dims[d] - 1;
		/* CellBorderAndFlagManager.h:114-130: halo/boundary interfaces snap to the bounding box, so a point inside
		 * the box can never land in a halo cell and vice versa (the reference's testPointInCell post-fix). */
		if (p[d] >= g->bmin[d] && c < 1) c = 1;
		if (p[d] < g->bmax[d] && c > g->dims[d] - 2) c = g->dims[d] - 2;
		if (p[d] < g->bmin[d]) c = 0;
		if (p[d] >= g->bmax[d]) c = g->dims[d] - 1;
		ci[d] = c;
	}
	return ((long)ci[2] * g->dims[1] + ci[1]) * g->dims[0] + ci[0];
}

static int grid_is_halo(const grid_t *g, long c) {
	int x = (int)(c % g->dims[0]);
	int y = (int)((c / g->dims[0]) % g->dims[1]);
	int z = (int)(c / ((long)g->dims[0] * g->dims[1]));
	return x == 0 || y == 0 || z == 0 || x == g->dims[0] - 1 || y == g->dims[1] - 1 || z == g->dims[2] - 1;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Periodic wrap of molecules that left the box: DomainDecompBase::handleDomainLeavingParticles
 * (parallel/DomainDecompBase.cpp:174-225) incl. its rounding clamps.
 * ----------------------------------------------------------------------------------------------------------------*/
void ls1o_wrap(size_t n, double *r, const double L[3]) {
	for (size_t i = 0; i < n; ++i) {
		for (int d = 0; d < 3; ++d) {
			double x = r[3 * i + d];
			if (x < 0.) {
				x += L[d];
				if (x >= L[d]) x = nexttoward(L[d], L[d] - 1.f);
			} else if (x >= L[d]) {
				x -= L[d];
				if (x <= 0.) x = 0.;
			}
			r[3 * i + d] = x;
		}
	}
}

/* ------------------------------------------------------------------------------------------------------------------
 * ls1o_forces: one traverseCells of the reference on a single (sequential) domain [0,L)^3.
 *   halo population   DomainDecompBase::populateHaloLayerWithCopies (parallel/DomainDecompBase.cpp:293-348), x then y
 *                     then z, each pass also copying earlier passes' copies -> edge and corner images
 *   pair set          every cell with itself + every unordered neighbouring cell pair once
 *                     (C08BasedTraversals::processBaseCell, LinkedCellTraversals/C08BasedTraversals.h:48-99,122-136)
 *   halo/macro policy VectorizedCellProcessor::processCell / processCellPair
 *                     (adapter/VectorizedCellProcessor.cpp:2734-2821)
 *   masks             strict r^2 < rc^2 on molecule centres, LJ uses rcLJ (VectorizedCellProcessor.cpp:967-968,
 *                     1013-1024; vectorization/SIMD_VectorizedCellProcessorHelpers.h:344-422)
 *   site reduction    FullMolecule::calcFM (molecules/FullMolecule.cpp:526-629)
 *   macroscopic       VectorizedCellProcessor::endTraversal (VectorizedCellProcessor.cpp:124-157)
 * Outputs F, M, Vi are [n][3] for the n real molecules; out4 = {upot, virial, upot6lj/6, upotXpoles + myRF}.
 * ----------------------------------------------------------------------------------------------------------------*/
int ls1o_forces(const ls1o_sys *s, size_t n, const double *r, const double *q, const int *cid, int periodic,
				const double L[3], double *F, double *M, double *Vi, double *out4) {
	work_t w;
	memset(&w, 0, sizeof(w));
	double bmin[3] = {0., 0., 0.}, bmax[3] = {L[0], L[1], L[2]};
	grid_t g;
	grid_init(&g, bmin, bmax, s->rc);
	const double unitq[4] = {1., 0., 0., 0.};
	for (size_t i = 0; i < n; ++i) work_add(s, &w, r + 3 * i, q ? q + 4 * i : unitq, cid[i], (long)i);

	if (periodic) {
		const double il = s->rc; /* interactionLength = container cutoff */
		for (int dim = 0; dim < 3; ++dim) {
			size_t cur = w.n; /* copies made in this pass are not re-copied in the same pass/direction logic below */
			for (int dir = -1; dir <= 1; dir += 2) {
				double shift = (dir < 0) ? L[dim] : -L[dim];
				double lo[3], hi[3];
				for (int d = 0; d < 3; ++d) {
					lo[d] = bmin[d] - il;
					hi[d] = bmax[d] + il;
				}
				if (dir < 0) {
					lo[dim] = bmin[dim];
					hi[dim] = bmin[dim] + il;
				} else {
					lo[dim] = bmax[dim] - il;
					hi[dim] = bmax[dim];
				}
				for (size_t i = 0; i < cur; ++i) {
					const double *p = w.r + 3 * i;
					/* region iterator: lo <= p < hi (ParticleCellBase / RegionParticleIterator in-box test) */
					if (!(p[0] >= lo[0] && p[0] < hi[0] && p[1] >= lo[1] && p[1] < hi[1] && p[2] >= lo[2] && p[2] < hi[2]))
						continue;
					double pn[3] = {p[0], p[1], p[2]};
					pn[dim] = p[dim] + shift;
					if (shift < 0) {
						if (pn[dim] >= bmin[dim]) pn[dim] = nexttoward(bmin[dim], bmin[dim] - 1.f);
					} else {
						if (pn[dim] < bmax[dim]) pn[dim] = nexttoward(bmax[dim], bmax[dim] + 1.f);
					}
					long src = w.src[i];
					work_add(s, &w, pn, q ? q + 4 * src : unitq, w.cid[i], src);
				}
			}
		}
	}

	/* bin molecules into cells (counting sort) */
	long ncells = (long)g.dims[0] * g.dims[1] * g.dims[2];
	long *cell = (long *)malloc(sizeof(long) * w.n);
	size_t *start = (size_t *)calloc((size_t)ncells + 1, sizeof(size_t));
	size_t *order = (size_t *)malloc(sizeof(size_t) * w.n);
	for (size_t i = 0; i < w.n; ++i) {
		cell[i] = grid_cell(&g, w.r + 3 * i);
		start[cell[i] + 1]++;
	}
	for (long c = 0; c < ncells; ++c) start[c + 1] += start[c];
	{
		size_t *fill = (size_t *)malloc(sizeof(size_t) * (size_t)ncells);
		memcpy(fill, start, sizeof(size_t) * (size_t)ncells);
		for (size_t i = 0; i < w.n; ++i) order[fill[cell[i]]++] = i;
		free(fill);
	}

	macro_t acc = {0., 0., 0., 0.};
	const double rc2 = s->rc * s->rc, rcLJ2 = s->rcLJ * s->rcLJ;
	for (long c1 = 0; c1 < ncells; ++c1) {
		if (start[c1 + 1] == start[c1]) continue;
		int x1 = (int)(c1 % g.dims[0]);
		int y1 = (int)((c1 / g.dims[0]) % g.dims[1]);
		int z1 = (int)(c1 / ((long)g.dims[0] * g.dims[1]));
		int h1 = grid_is_halo(&g, c1);
		/* processCell: VectorizedCellProcessor.cpp:2734-2744 */
		if (!h1) {
			for (size_t a = start[c1]; a < start[c1 + 1]; ++a) {
				for (size_t b = a + 1; b < start[c1 + 1]; ++b) {
					size_t mi = order[a], mj = order[b];
					double drm[3], dd = 0.;
					for (int d = 0; d < 3; ++d) {
						drm[d] = w.r[3 * mi + d] - w.r[3 * mj + d];
						dd += drm[d] * drm[d];
					}
					if (dd < rc2 && dd != 0.) potforce(s, &w, mi, mj, drm, dd < rcLJ2, 1, &acc);
				}
			}
		}
		/* 13 forward neighbours: each unordered neighbouring pair exactly once */
		for (int dz = 0; dz <= 1; ++dz)
			for (int dy = (dz ? -1 : 0); dy <= 1; ++dy)
				for (int dx = ((dz || dy) ? -1 : 1); dx <= 1; ++dx) {
					int x2 = x1 + dx, y2 = y1 + dy, z2 = z1 + dz;
					if (x2 < 0 || y2 < 0 || z2 < 0 || x2 >= g.dims[0] || y2 >= g.dims[1] || z2 >= g.dims[2]) continue;
					long c2 = ((long)z2 * g.dims[1] + y2) * g.dims[0] + x2;
					if (start[c2 + 1] == start[c2]) continue;
					int h2 = grid_is_halo(&g, c2);
					if (h1 && h2) continue;
					/* non-halo cell first (C08BasedTraversals.h:87-92); macroscopic iff no halo involved or
					 * index(first) < index(second) (VectorizedCellProcessor.cpp:2792-2818) */
					long ca = c1, cb = c2;
					if (h1) { ca = c2; cb = c1; }
					int macro = (!h1 && !h2) || (ca < cb);
					for (size_t a = start[ca]; a < start[ca + 1]; ++a)
						for (size_t b = start[cb]; b < start[cb + 1]; ++b) {
							size_t mi = order[a], mj = order[b];
							double drm[3], dd = 0.;
							for (int d = 0; d < 3; ++d) {
								drm[d] = w.r[3 * mi + d] - w.r[3 * mj + d];
								dd += drm[d] * drm[d];
							}
							if (dd < rc2) potforce(s, &w, mi, mj, drm, dd < rcLJ2, macro, &acc);
						}
				}
	}

	/* calcFM for the real molecules */
	for (size_t i = 0; i < n; ++i) {
		double Fm[3] = {0., 0., 0.}, Mm[3] = {0., 0., 0.};
		for (size_t k = w.soff[i]; k < w.soff[i + 1]; ++k) {
			const double *d = w.sd + 3 * k, *fs = w.sF + 3 * k;
			Fm[0] += fs[0];
			Fm[1] += fs[1];
			Fm[2] += fs[2];
			Mm[0] += d[1] * fs[2] - d[2] * fs[1];
			Mm[1] += d[2] * fs[0] - d[0] * fs[2];
			Mm[2] += d[0] * fs[1] - d[1] * fs[0];
		}
		for (int d = 0; d < 3; ++d) {
			F[3 * i + d] = Fm[d];
			M[3 * i + d] = Mm[d] + w.M[3 * i + d];
			Vi[3 * i + d] = w.Vi[3 * i + d];
		}
	}
	out4[0] = acc.upot6lj / 6.0 + acc.upotXpoles + acc.myRF;
	out4[1] = acc.virial + 3.0 * acc.myRF;
	out4[2] = acc.upot6lj / 6.0;
	out4[3] = acc.upotXpoles + acc.myRF;
	free(cell);
	free(start);
	free(order);
	work_free(&w);
	return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Leapfrog: FullMolecule::upd_preF (molecules/FullMolecule.cpp:334-364) and upd_postF (:366-389), driven as
 * Leapfrog::transition1to2 / transition2to3 (integrators/Leapfrog.cpp:48-64,66-150).  D = angular momentum.
 * ----------------------------------------------------------------------------------------------------------------*/
void ls1o_upd_preF(const ls1o_sys *s, size_t n, double dt, const int *cid, double *r, double *v, double *q, double *D,
				   const double *F, const double *M) {
	const double dt_halve = .5 * dt;
	for (size_t i = 0; i < n; ++i) {
		const int c = cid[i];
		const double dtInv2m = dt_halve / s->mass[c];
		for (int d = 0; d < 3; ++d) {
			v[3 * i + d] += dtInv2m * F[3 * i + d];
			r[3 * i + d] += dt * v[3 * i + d];
		}
		double *qq = q + 4 * i, *DD = D + 3 * i;
		double wv[3], qh[4], dq[4];
		q_rotateinv(qq, DD, wv);
		for (int d = 0; d < 3; ++d) wv[d] *= s->invI[3 * c + d];
		q_diff(qq, wv, dq);
		for (int k = 0; k < 4; ++k) qh[k] = dq[k] * dt_halve + qq[k];
		double qcorr = 1. / sqrt(qh[0] * qh[0] + qh[1] * qh[1] + qh[2] * qh[2] + qh[3] * qh[3]);
		for (int k = 0; k < 4; ++k) qh[k] *= qcorr;
		for (int d = 0; d < 3; ++d) DD[d] += dt_halve * M[3 * i + d];
		q_rotateinv(qh, DD, wv);
		for (int d = 0; d < 3; ++d) wv[d] *= s->invI[3 * c + d];
		q_diff(qh, wv, dq);
		for (int k = 0; k < 4; ++k) qq[k] += dq[k] * dt;
		qcorr = 1. / sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
		for (int k = 0; k < 4; ++k) qq[k] *= qcorr;
	}
}

void ls1o_upd_postF(const ls1o_sys *s, size_t n, double dt_halve, const int *cid, double *v, const double *q,
					double *D, const double *F, const double *M, double *sums2) {
	double summv2 = 0., sumIw2 = 0.;
	for (size_t i = 0; i < n; ++i) {
		const int c = cid[i];
		const double dtInv2m = dt_halve / s->mass[c];
		double v2 = 0.;
		for (int d = 0; d < 3; ++d) {
			v[3 * i + d] += dtInv2m * F[3 * i + d];
			v2 += v[3 * i + d] * v[3 * i + d];
			D[3 * i + d] += dt_halve * M[3 * i + d];
		}
		summv2 += s->mass[c] * v2;
		double wv[3], Iw2 = 0.;
		q_rotateinv(q + 4 * i, D + 3 * i, wv);
		for (int d = 0; d < 3; ++d) {
			wv[d] *= s->invI[3 * c + d];
			Iw2 += s->I[3 * c + d] * wv[d] * wv[d];
		}
		sumIw2 += Iw2;
	}
	sums2[0] = summv2;
	sums2[1] = sumIw2;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Homogeneous long-range correction: longRange/Homogeneous.cpp:21-135 (init + calculateLongRange),
 * integrals _TICCu/_TICSu/_TISSu/_TICCv/_TICSv/_TISSv :137-180 (Lustig 1988).  nmol[c] = molecules of component c,
 * rho = N/V.  out2 = {UpotCorr, VirialCorr} as Domain::setUpotCorr / setVirialCorr receive them.
 * ----------------------------------------------------------------------------------------------------------------*/
static double ticcu(int n, double rc, double s2) { return -pow(rc, 2 * n + 3) / (pow(s2, n) * (2 * n + 3)); }
static double ticsu(int n, double rc, double s2, double tau) {
	return -(pow(rc + tau, 2 * n + 3) - pow(rc - tau, 2 * n + 3)) * rc / (4 * pow(s2, n) * tau * (n + 1) * (2 * n + 3)) +
		   (pow(rc + tau, 2 * n + 4) - pow(rc - tau, 2 * n + 4)) / (4 * pow(s2, n) * tau * (n + 1) * (2 * n + 3) * (2 * n + 4));
}
static double tissu(int n, double rc, double s2, double t1, double t2) {
	const double tp = t1 + t2, tm = t1 - t2;
	return -(pow(rc + tp, 2 * n + 4) - pow(rc + tm, 2 * n + 4) - pow(rc - tm, 2 * n + 4) + pow(rc - tp, 2 * n + 4)) * rc /
			   (8 * pow(s2, n) * t1 * t2 * (n + 1) * (2 * n + 3) * (2 * n + 4)) +
		   (pow(rc + tp, 2 * n + 5) - pow(rc + tm, 2 * n + 5) - pow(rc - tm, 2 * n + 5) + pow(rc - tp, 2 * n + 5)) /
			   (8 * pow(s2, n) * t1 * t2 * (n + 1) * (2 * n + 3) * (2 * n + 4) * (2 * n + 5));
}
static double ticcv(int n, double rc, double s2) { return 2 * n * ticcu(n, rc, s2); }
static double ticsv(int n, double rc, double s2, double tau) {
	return -(pow(rc + tau, 2 * n + 2) - pow(rc - tau, 2 * n + 2)) * rc * rc / (4 * pow(s2, n) * tau * (n + 1)) -
		   3 * ticsu(n, rc, s2, tau);
}
static double tissv(int n, double rc, double s2, double t1, double t2) {
	const double tp = t1 + t2, tm = t1 - t2;
	return -(pow(rc + tp, 2 * n + 3) - pow(rc + tm, 2 * n + 3) - pow(rc - tm, 2 * n + 3) + pow(rc - tp, 2 * n + 3)) * rc * rc /
			   (8 * pow(s2, n) * t1 * t2 * (n + 1) * (2 * n + 3)) -
		   3 * tissu(n, rc, s2, t1, t2);
}

int ls1o_lrc_homogeneous(const ls1o_sys *s, const unsigned long *nmol, double *out2) {
	double UpotCorrLJ = 0., VirialCorrLJ = 0., MySelbstTerm = 0.;
	unsigned long N = 0;
	for (int i = 0; i < s->ncomp; ++i) N += nmol[i];
	for (int i = 0; i < s->ncomp; ++i) {
		double cb[3] = {0., 0., 0.};
		for (int a = 0; a < s->nc[i]; ++a) {
			const double *c = s->ch + (size_t)(s->oc[i] + a) * CH_STRIDE;
			for (int d = 0; d < 3; ++d) cb[d] += c[4] * c[d];
		}
		for (int a = 0; a < s->nd[i]; ++a) {
			const double *p = s->dp + (size_t)(s->od[i] + a) * DP_STRIDE;
			const double norm = 1.0 / sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
			for (int d = 0; d < 3; ++d) cb[d] += p[6] * p[3 + d] * norm;
		}
		MySelbstTerm += (cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2]) * (double)nmol[i];
		for (int j = 0; j < s->ncomp; ++j)
			for (int a = 0; a < s->nlj[i]; ++a) {
				const double *sa = s->lj + (size_t)(s->olj[i] + a) * LJ_STRIDE;
				const double tau1 = sqrt(sa[0] * sa[0] + sa[1] * sa[1] + sa[2] * sa[2]);
				for (int b = 0; b < s->nlj[j]; ++b) {
					const double *sb = s->lj + (size_t)(s->olj[j] + b) * LJ_STRIDE;
					double tau2 = sqrt(sb[0] * sb[0] + sb[1] * sb[1] + sb[2] * sb[2]);
					if (tau1 + tau2 >= s->rcLJ) return -1;
					const size_t k = (size_t)(s->olj[i] + a) * s->ncenters + (s->olj[j] + b);
					if (s->shift6[k] != 0.0) continue;
					const double fac = (double)nmol[i] * (double)nmol[j] * s->eps24[k], s2 = s->sig2[k], rc = s->rcLJ;
					if (tau1 == 0. && tau2 == 0.) {
						UpotCorrLJ += fac * (ticcu(-6, rc, s2) - ticcu(-3, rc, s2));
						VirialCorrLJ += fac * (ticcv(-6, rc, s2) - ticcv(-3, rc, s2));
					} else if (tau1 != 0. && tau2 != 0.) {
						UpotCorrLJ += fac * (tissu(-6, rc, s2, tau1, tau2) - tissu(-3, rc, s2, tau1, tau2));
						VirialCorrLJ += fac * (tissv(-6, rc, s2, tau1, tau2) - tissv(-3, rc, s2, tau1, tau2));
					} else {
						if (tau2 == 0.) tau2 = tau1;
						UpotCorrLJ += fac * (ticsu(-6, rc, s2, tau2) - ticsu(-3, rc, s2, tau2));
						VirialCorrLJ += fac * (ticsv(-6, rc, s2, tau2) - ticsv(-3, rc, s2, tau2));
					}
				}
			}
	}
	(void)N;
	out2[0] = UpotCorrLJ;
	out2[1] = VirialCorrLJ;
	out2[2] = MySelbstTerm;
	return 0;
}

/* second half: Homogeneous::calculateLongRange (Homogeneous.cpp:113-135); in3 = the three sums of the first half */
void ls1o_lrc_finish(const ls1o_sys *s, const double *in3, double rho, unsigned long N, double *out2) {
	const double fac = 3.14159265358979323846 * rho / (3. * (double)N);
	const double UpotCorrLJ = fac * in3[0], VirialCorrLJ = -fac * in3[1];
	const double MySelbstTerm = -0.5 * s->epsRFInvrc3 * in3[2];
	out2[0] = UpotCorrLJ + MySelbstTerm;
	out2[1] = VirialCorrLJ + 3. * MySelbstTerm;
}
