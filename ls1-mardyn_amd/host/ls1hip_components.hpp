// ls1hip_components.hpp — the reference's component set (ensemble/EnsembleBase.h, molecules/Component.h) as the flat
// site tables of ls1hip_set_components (same columns as the .inp component block, io/ASCIIReader.cpp:178-206).
// Shared by the seam-A and seam-B adapters; compiled against the reference's headers.
#pragma once
#include <vector>

#include "Domain.h"
#include "molecules/Component.h"

#include "ls1hip.h"

inline int ls1hip_set_components_from(ls1hip_ctx* ctx, const std::vector<Component>& comps, Domain& domain, double cutoffRadius,
									  double LJcutoffRadius) {
	const int nc = (int)comps.size();
	std::vector<int> nlj(nc), nch(nc), nd(nc), nq(nc);
	std::vector<double> lj, ch, dp, qp, mass(nc), I(3 * nc);
	for (int k = 0; k < nc; ++k) {
		const Component& c = comps[k];
		nlj[k] = c.numLJcenters(); nch[k] = c.numCharges(); nd[k] = c.numDipoles(); nq[k] = c.numQuadrupoles();
		for (unsigned s = 0; s < c.numLJcenters(); ++s) {
			const LJcenter& a = c.ljcenter(s);
			for (double v : {a.rx(), a.ry(), a.rz(), a.m(), a.eps(), a.sigma(), a.shift6()}) lj.push_back(v);
		}
		for (unsigned s = 0; s < c.numCharges(); ++s) {
			const Charge& a = c.charge(s);
			for (double v : {a.rx(), a.ry(), a.rz(), a.m(), a.q()}) ch.push_back(v);
		}
		for (unsigned s = 0; s < c.numDipoles(); ++s) {
			const Dipole& a = c.dipole(s);
			for (double v : {a.rx(), a.ry(), a.rz(), a.ex(), a.ey(), a.ez(), a.absMy()}) dp.push_back(v);
		}
		for (unsigned s = 0; s < c.numQuadrupoles(); ++s) {
			const Quadrupole& a = c.quadrupole(s);
			for (double v : {a.rx(), a.ry(), a.rz(), a.ex(), a.ey(), a.ez(), a.absQ()}) qp.push_back(v);
		}
		mass[k] = c.m();
		I[3 * k] = c.I11(); I[3 * k + 1] = c.I22(); I[3 * k + 2] = c.I33();
	}
	std::vector<double> mix = domain.getmixcoeff();
	mix.resize((size_t)nc * (nc - 1), 0.0);  // (xi, eta) per unordered pair
	int rc = ls1hip_set_components(ctx, nc, nlj.data(), nch.data(), nd.data(), nq.data(), lj.data(), ch.data(), dp.data(), qp.data(),
								   mass.data(), I.data(), mix.data(), domain.getepsilonRF(), cutoffRadius, LJcutoffRadius);
	if (rc) return rc;
	// the reference's own count of rotational degrees of freedom (from the site masses; an I override does not change it)
	std::vector<int> rdof(nc);
	for (int k = 0; k < nc; ++k) rdof[k] = (int)comps[k].getRotationalDegreesOfFreedom();
	return ls1hip_set_rot_dof(ctx, nc, rdof.data());
}
