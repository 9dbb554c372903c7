// refdump — golden-vector generator. TEST INFRASTRUCTURE ONLY (never shipped, never timed as product).
//
// This is OUR harness; it is linked against object files compiled from the reference sources where they
// lie under /root/reference (see Makefile in this directory) and drives the reference's own public API:
//   ASCIIReader            (/root/reference/src/io/ASCIIReader.cpp:47-460)
//   LinkedCells            (/root/reference/src/particleContainer/LinkedCells.cpp)
//   DomainDecompBase       (/root/reference/src/parallel/DomainDecompBase.cpp:51-86, sequential PBC halo)
//   VectorizedCellProcessor(/root/reference/src/particleContainer/adapter/VectorizedCellProcessor.cpp)
//   LegacyCellProcessor    (/root/reference/src/particleContainer/adapter/LegacyCellProcessor.cpp:56-152)
//   Leapfrog               (/root/reference/src/integrators/Leapfrog.cpp:35-150)
// following the call sequence of LinkedCellsTest::doForceComparisonTest
// (/root/reference/src/particleContainer/tests/LinkedCellsTest.cpp:511-600) and of the time step in
// Simulation::simulate (/root/reference/src/Simulation.cpp:995-1099).
//
// usage: refdump <file.inp> <cutoff> <periodic 0|1> <out.bin> [--legacy] [--steps N --dt DT] [--nvt]
//   --nvt: global velocity-scaling thermostat after every step, exactly the sequence of Simulation::simulate
//          (Simulation.cpp:1099-1131): calculateGlobalValues -> VelocityScalingThermostat::apply (global betas; the
//          component-wise branch when the .inp header assigns thermostats to components: ThermostatTemperature / ComponentThermostat)
// output (little-endian): magic "LS1GOLD1", u64 N, u64 nsteps, f64 cutoff, f64 dt, f64 L[3],
//   f64 upot, f64 virial, f64 summv2, f64 sumIw2, then N records sorted by molecule id:
//   u64 id, u64 cid, f64 r[3], v[3], q[4], D[3], F[3], M[3], Vi[3]
//   then a trailer: magic "LS1LRC01", f64 upot_corr, f64 virial_corr = Homogeneous long-range correction of the INITIAL
//   configuration (longRange/Homogeneous.cpp:21-135) as Domain::calculateGlobalValues adds it (Domain.cpp:176-181)
//   (r,v,q,D = state at which F,M,Vi were evaluated, i.e. after nsteps full steps; v,D after the post-force kick
//    when nsteps>0).
#include "Simulation.h"
#include "Domain.h"
#include "ensemble/EnsembleBase.h"
#include "integrators/Leapfrog.h"
#include "longRange/Homogeneous.h"
#include "io/ASCIIReader.h"
#include "molecules/Molecule.h"
#include "parallel/DomainDecompBase.h"
#include "particleContainer/LinkedCells.h"
#include "particleContainer/adapter/LegacyCellProcessor.h"
#include "particleContainer/adapter/ParticlePairs2PotForceAdapter.h"
#include "particleContainer/adapter/VectorizedCellProcessor.h"
#include "thermostats/VelocityScalingThermostat.h"
#include "utils/Logger.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Rec {
	uint64_t id, cid;
	double r[3], v[3], q[4], D[3], F[3], M[3], Vi[3];
};

static void forces(ParticleContainer* c, DomainDecompBase* dd, Domain* domain, CellProcessor* cp, bool periodic) {
	c->update();
	if (periodic) dd->balanceAndExchange(0., false, c, domain);
	c->updateMoleculeCaches();
	c->traverseCells(*cp);
	for (auto m = c->iterator(ParticleIterator::ALL_CELLS); m.isValid(); ++m) m->calcFM();
	c->deleteOuterParticles();
}

int main(int argc, char** argv) {
	if (argc < 5) {
		fprintf(stderr, "usage: %s file.inp cutoff periodic out.bin [--legacy] [--steps N --dt DT]\n", argv[0]);
		return 2;
	}
	Log::global_log = new Log::Logger(Log::Error);
	const std::string file = argv[1];
	const double rc = atof(argv[2]);
	const bool periodic = atoi(argv[3]) != 0;
	const char* out = argv[4];
	bool legacy = false;
	unsigned long nsteps = 0;
	double dt = 0.0;
	bool nvt = false;
	for (int a = 5; a < argc; ++a) {
		if (!strcmp(argv[a], "--legacy")) legacy = true;
		else if (!strcmp(argv[a], "--steps")) nsteps = strtoul(argv[++a], nullptr, 10);
		else if (!strcmp(argv[a], "--dt")) dt = atof(argv[++a]);
		else if (!strcmp(argv[a], "--nvt")) nvt = true;
	}

	new Simulation();  // assigns global_simulation
	Domain* domain = global_simulation->getDomain();
	DomainDecompBase* dd = &global_simulation->domainDecomposition();
	global_simulation->setcutoffRadius(rc);
	global_simulation->setLJCutoff(rc);

	ASCIIReader reader;
	reader.setPhaseSpaceHeaderFile(file);
	reader.setPhaseSpaceFile(file);
	reader.readPhaseSpaceHeader(domain, 1.0);
	double bmin[3], bmax[3];
	for (int d = 0; d < 3; ++d) {
		bmin[d] = dd->getBoundingBoxMin(d, domain);
		bmax[d] = dd->getBoundingBoxMax(d, domain);
	}
	LinkedCells* c = new LinkedCells(bmin, bmax, rc);
	reader.readPhaseSpace(c, domain, dd);
	c->deleteOuterParticles();
	c->update();
	c->updateMoleculeCaches();
	domain->initParameterStreams(rc, rc);

	CellProcessor* cp;
	ParticlePairs2PotForceAdapter* pph = nullptr;
	if (legacy) {
		pph = new ParticlePairs2PotForceAdapter(*domain);
		cp = new LegacyCellProcessor(rc, rc, pph);
	} else {
		cp = new VectorizedCellProcessor(*domain, rc, rc);
	}

	forces(c, dd, domain, cp, periodic);
	double upotCorr = 0., virialCorr = 0.;
	{
		Homogeneous lrc(rc, rc, domain, global_simulation);
		lrc.init();
		lrc.calculateLongRange();
		const double ul = domain->getLocalUpot(), vl = domain->getLocalVirial();
		domain->calculateGlobalValues(dd, c, true, 1.0);
		upotCorr = domain->getGlobalUpot() - ul;
		virialCorr = domain->getAverageGlobalVirial() * (double)domain->getglobalNumMolecules() - vl;
		domain->setUpotCorr(0.);
		domain->setVirialCorr(0.);
	}
	double summv2 = 0., sumIw2 = 0.;
	if (nsteps > 0) {
		Leapfrog integ(dt);
		integ.init();
		// prepare_start leaves the integrator in POST_FORCE state after the initial force evaluation
		// (/root/reference/src/Simulation.cpp:829-892); Leapfrog::init sets that state.
		for (unsigned long s = 0; s < nsteps; ++s) {
			integ.eventNewTimestep(c, domain);
			forces(c, dd, domain, cp, periodic);
			integ.eventForcesCalculated(c, domain);
			if (nvt) {
				domain->calculateGlobalValues(dd, c, true, 1.0);
				VelocityScalingThermostat vst;
				if (domain->severalThermostats()) {
					// the component-wise branch of the driver (Simulation.cpp:1112-1126): one (beta_trans, beta_rot) per
					// thermostat id of the legacy .inp header (ASCIIReader.cpp:104-124), directed velocity 0
					vst.enableComponentwise();
					const size_t ncomp = global_simulation->getEnsemble()->getComponents()->size();
					for (unsigned cid = 0; cid < ncomp; ++cid) {
						const int th = domain->getThermostat(cid);
						vst.setBetaTrans(th, domain->getGlobalBetaTrans(th));
						vst.setBetaRot(th, domain->getGlobalBetaRot(th));
						double v0[3];
						for (int d = 0; d < 3; ++d) v0[d] = domain->getThermostatDirectedVelocity(th, d);
						vst.setVelocity(th, v0);
					}
				} else {
					vst.setGlobalBetaTrans(domain->getGlobalBetaTrans());
					vst.setGlobalBetaRot(domain->getGlobalBetaRot());
				}
				vst.apply(c);
			}
		}
	}
	// kinetic sums exactly as Leapfrog::transition2to3 accumulates them (Leapfrog.cpp:120-128 -> FullMolecule.cpp:366-389)
	for (auto m = c->iterator(ParticleIterator::ONLY_INNER_AND_BOUNDARY); m.isValid(); ++m)
		m->calculate_mv2_Iw2(summv2, sumIw2);

	std::vector<Rec> recs;
	for (auto m = c->iterator(ParticleIterator::ONLY_INNER_AND_BOUNDARY); m.isValid(); ++m) {
		Rec r;
		r.id = m->getID();
		r.cid = m->componentid();
		for (int d = 0; d < 3; ++d) {
			r.r[d] = m->r(d);
			r.v[d] = m->v(d);
			r.D[d] = m->D(d);
			r.F[d] = m->F(d);
			r.M[d] = m->M(d);
			r.Vi[d] = m->Vi(d);
		}
		r.q[0] = m->q().qw();
		r.q[1] = m->q().qx();
		r.q[2] = m->q().qy();
		r.q[3] = m->q().qz();
		recs.push_back(r);
	}
	std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.id < b.id; });

	FILE* f = fopen(out, "wb");
	if (!f) { perror(out); return 1; }
	const char magic[8] = {'L', 'S', '1', 'G', 'O', 'L', 'D', '1'};
	fwrite(magic, 1, 8, f);
	uint64_t n = recs.size(), ns = nsteps;
	fwrite(&n, 8, 1, f);
	fwrite(&ns, 8, 1, f);
	fwrite(&rc, 8, 1, f);
	fwrite(&dt, 8, 1, f);
	double L[3] = {domain->getGlobalLength(0), domain->getGlobalLength(1), domain->getGlobalLength(2)};
	fwrite(L, 8, 3, f);
	double upot = domain->getLocalUpot(), virial = domain->getLocalVirial();
	fwrite(&upot, 8, 1, f);
	fwrite(&virial, 8, 1, f);
	fwrite(&summv2, 8, 1, f);
	fwrite(&sumIw2, 8, 1, f);
	fwrite(recs.data(), sizeof(Rec), recs.size(), f);
	const char lmagic[8] = {'L', 'S', '1', 'L', 'R', 'C', '0', '1'};
	fwrite(lmagic, 1, 8, f);
	fwrite(&upotCorr, 8, 1, f);
	fwrite(&virialCorr, 8, 1, f);
	fclose(f);
	printf("%s rc=%g periodic=%d legacy=%d steps=%lu N=%lu upot=%.17g virial=%.17g lrc=(%.12g, %.12g)\n", file.c_str(), rc,
		   (int)periodic, (int)legacy, nsteps, (unsigned long)n, upot, virial, upotCorr, virialCorr);
	return 0;
}
