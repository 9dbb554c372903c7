// host_sim.cpp — drives the C++17 host layer (ls1-mardyn_amd/host/ls1hip_host.hpp) the way Simulation::simulate drives
// the reference's plug-ins, on a case file written by tests/test_gpu_hostcpp.py, and dumps the final state.
//   case file (little endian): "LS1CASE1", i32 ncomp, i32 nsteps, f64 rc, f64 dt, f64 L[3], f64 eps_rf,
//     i32 nlj[ncomp], nc[ncomp], nd[ncomp], nq[ncomp], then u64-counted f64 arrays lj, ch, dp, qp, mass, I, mix,
//     u64 N, u64 id[N], i32 cid[N], f64 r[3N], v[3N], q[4N], D[3N]
//   result file: "LS1RES01", u64 N, f64 upot, virial, summv2, sumIw2, u64 id[N], f64 r[3N], v[3N], q[4N], D[3N], F[3N], M[3N]
#include <cstdio>
#include <cstring>
#include <iostream>

#include "ls1hip_host.hpp"

namespace {
template <class T>
void rd(FILE* f, T* p, size_t n) {
	if (n && fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read");
}
std::vector<double> rdv(FILE* f) {
	uint64_t n = 0;
	rd(f, &n, 1);
	std::vector<double> v(n);
	rd(f, v.data(), n);
	return v;
}
template <class T>
void wr(FILE* f, const T* p, size_t n) {
	if (n && fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write");
}
}  // namespace

int main(int argc, char** argv) {
	if (argc != 3) {
		std::cerr << "usage: host_sim case.bin result.bin\n";
		return 2;
	}
	try {
		FILE* f = fopen(argv[1], "rb");
		if (!f) throw std::runtime_error("cannot open case file");
		char magic[8];
		rd(f, magic, 8);
		if (memcmp(magic, "LS1CASE1", 8)) throw std::runtime_error("bad magic");
		int32_t ncomp = 0, nsteps = 0;
		double rc = 0., dt = 0., L[3], eps_rf = 0.;
		rd(f, &ncomp, 1); rd(f, &nsteps, 1); rd(f, &rc, 1); rd(f, &dt, 1); rd(f, L, 3); rd(f, &eps_rf, 1);
		ls1hip::ComponentTables t;
		t.ncomp = ncomp;
		t.eps_rf = eps_rf;
		for (auto* v : {&t.nlj, &t.nc, &t.nd, &t.nq}) {
			v->resize(ncomp);
			rd(f, v->data(), ncomp);
		}
		t.lj = rdv(f); t.ch = rdv(f); t.dp = rdv(f); t.qp = rdv(f); t.mass = rdv(f); t.I = rdv(f); t.mix = rdv(f);
		uint64_t N = 0;
		rd(f, &N, 1);
		std::vector<uint64_t> id(N);
		std::vector<int32_t> cid(N);
		std::vector<double> r(3 * N), v(3 * N), q(4 * N), D(3 * N);
		rd(f, id.data(), N); rd(f, cid.data(), N); rd(f, r.data(), 3 * N); rd(f, v.data(), 3 * N); rd(f, q.data(), 4 * N);
		rd(f, D.data(), 3 * N);
		fclose(f);

		// ---- the reference's object graph for this path (Simulation.cpp:411-455,168-186,771) ----
		ls1hip::Domain domain({L[0], L[1], L[2]});
		ls1hip::LinkedCells container({0., 0., 0.}, {L[0], L[1], L[2]}, rc, t);
		ls1hip::VectorizedCellProcessor cellProcessor(domain, rc, rc);
		ls1hip::DomainDecompBase decomp;
		ls1hip::Leapfrog integrator(dt);
		container.addParticles(N, id.data(), cid.data(), r.data(), v.data(), q.data(), D.data());
		ls1hip::simulate(container, decomp, cellProcessor, integrator, domain, (unsigned long)nsteps);

		auto m = container.molecules();
		FILE* o = fopen(argv[2], "wb");
		if (!o) throw std::runtime_error("cannot open result file");
		const uint64_t n = m.id.size();
		const double sums[4] = {domain.getLocalUpot(), domain.getLocalVirial(), domain.getLocalSummv2(), domain.getLocalSumIw2()};
		wr(o, "LS1RES01", 8);
		wr(o, &n, 1); wr(o, sums, 4);
		wr(o, m.id.data(), n); wr(o, m.r.data(), 3 * n); wr(o, m.v.data(), 3 * n); wr(o, m.q.data(), 4 * n); wr(o, m.D.data(), 3 * n);
		wr(o, m.F.data(), 3 * n); wr(o, m.M.data(), 3 * n);
		fclose(o);
		std::cout << "host_sim: N=" << n << " steps=" << nsteps << " upot=" << sums[0] << "\n";
	} catch (const ls1hip::Error& e) {
		std::cerr << "ls1hip error " << e.code << ": " << e.what() << "\n";
		return 1;
	} catch (const std::exception& e) {
		std::cerr << "error: " << e.what() << "\n";
		return 1;
	}
	return 0;
}
