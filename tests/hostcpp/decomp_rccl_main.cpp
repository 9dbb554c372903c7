// decomp_rccl_main.cpp — one rank of the C++ / RCCL decomposed loop (ls1-mardyn_amd/host/DomainDecompRccl.hpp) on a 1CLJ case
// file written by tests/test_gpu_decomp_rccl.py.  Ranks: RANK / WORLD_SIZE / LOCAL_RANK from the environment (default 0 / 1 / 0);
// the ncclUniqueId travels through LS1HIP_RCCL_ID_FILE.
//   case file (little endian): "LS1DCMP1", f64 rc, dt, skin, L[3], i32 nsteps, i32 loopback, u64 N, u64 id[N], f64 r[3N], v[3N]
//   result file <out>.<rank>:  "LS1DRES1", u64 n, f64 upot0, virial0, upot, virial, summv2, f64 nglobal, u64 id[n], f64 r[3n], v[3n], F[3n]
#include <cstdlib>
#include <iostream>

#include "DomainDecompRccl.hpp"

namespace {
template <class T>
void rd(FILE* f, T* p, size_t n) {
	if (n && fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read");
}
template <class T>
void wr(FILE* f, const T* p, size_t n) {
	if (n && fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write");
}
int env_int(const char* name, int dflt) {
	const char* e = getenv(name);
	return e ? atoi(e) : dflt;
}
void chk(ls1hip_ctx* c, int rc, const char* what) {
	if (rc != LS1HIP_OK) throw std::runtime_error(std::string(what) + ": " + (c ? ls1hip_last_error(c) : "?"));
}
}  // namespace

// --geometry WORLD LOOPBACK: the rank grid as text (no GPU needed), one line per rank:
//   rank gx gy gz cx cy cz lo[3] hi[3] nbr[27] | peers...
static int print_geometry(int world, int loopback) {
	const double L[3] = {10., 12., 14.};
	for (int r = 0; r < world; ++r) {
		ls1hip::CartDecomp dc(world, r, L, loopback != 0);
		double lo[3], hi[3];
		int nbr[27];
		dc.bounding_box(lo, hi);
		dc.neighbor_table(nbr);
		printf("%d %d %d %d %d %d %d", r, dc.grid[0], dc.grid[1], dc.grid[2], dc.coords[0], dc.coords[1], dc.coords[2]);
		for (int d = 0; d < 3; ++d) printf(" %.17g", lo[d]);
		for (int d = 0; d < 3; ++d) printf(" %.17g", hi[d]);
		for (int k = 0; k < 27; ++k) printf(" %d", nbr[k]);
		printf(" |");
		for (int p : dc.peers()) printf(" %d", p);
		printf("\n");
	}
	return 0;
}

int main(int argc, char** argv) {
	if (argc == 4 && std::string(argv[1]) == "--geometry") return print_geometry(atoi(argv[2]), atoi(argv[3]));
	if (argc != 3) {
		std::cerr << "usage: decomp_rccl_main case.bin result_prefix\n";
		return 2;
	}
	const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", 0);
	try {
		FILE* f = fopen(argv[1], "rb");
		if (!f) throw std::runtime_error("cannot open case file");
		char magic[8];
		rd(f, magic, 8);
		if (memcmp(magic, "LS1DCMP1", 8)) throw std::runtime_error("bad magic");
		double rc, dt, skin, L[3];
		int32_t nsteps, loopback;
		uint64_t N;
		rd(f, &rc, 1); rd(f, &dt, 1); rd(f, &skin, 1); rd(f, L, 3); rd(f, &nsteps, 1); rd(f, &loopback, 1); rd(f, &N, 1);
		std::vector<uint64_t> id(N);
		std::vector<double> r(3 * N), v(3 * N);
		rd(f, id.data(), N); rd(f, r.data(), 3 * N); rd(f, v.data(), 3 * N);
		fclose(f);

		ls1hip::CartDecomp dc(world, rank, L, loopback != 0);
		double lo[3], hi[3];
		int nbr[27];
		dc.bounding_box(lo, hi);
		dc.neighbor_table(nbr);

		ls1hip_ctx* ctx = nullptr;
		chk(nullptr, ls1hip_create(local, &ctx), "ls1hip_create");
		const int one = 1, zero = 0;
		const double lj[LS1HIP_LJ_STRIDE] = {0., 0., 0., 1., 1., 1., 0.}, mass = 1., I[3] = {0., 0., 0.};
		chk(ctx, ls1hip_set_components(ctx, 1, &one, &zero, &zero, &zero, lj, nullptr, nullptr, nullptr, &mass, I, nullptr, 1e10, rc, rc),
			"ls1hip_set_components");
		if (skin > 0.) chk(ctx, ls1hip_set_verlet(ctx, 2, skin), "ls1hip_set_verlet");
		chk(ctx, ls1hip_set_domain(ctx, L, lo, hi, rank, nbr), "ls1hip_set_domain");
		std::vector<uint64_t> mid;
		std::vector<int32_t> mcid;
		std::vector<double> mr, mv, mq, mD;
		for (uint64_t i = 0; i < N; ++i) {
			bool in = true;
			for (int d = 0; d < 3; ++d) in = in && r[3 * i + d] >= lo[d] && r[3 * i + d] < hi[d];
			if (!in) continue;
			mid.push_back(id[i]);
			mcid.push_back(0);
			for (int d = 0; d < 3; ++d) {
				mr.push_back(r[3 * i + d]);
				mv.push_back(v[3 * i + d]);
				mD.push_back(0.);
			}
			mq.push_back(1.); mq.push_back(0.); mq.push_back(0.); mq.push_back(0.);
		}
		chk(ctx, ls1hip_upload(ctx, mid.size(), mid.data(), mcid.data(), mr.data(), mv.data(), mq.data(), mD.data()), "ls1hip_upload");

		ls1hip::RcclTransport tr(world, rank, local, world > 1 ? ls1hip::rccl_id_source_from_env() : std::string());
		ls1hip::DecomposedLoop loop(dc, ctx, tr);
		const ls1hip::GlobalValues g0 = loop.initial_forces();
		ls1hip::GlobalValues g = g0;
		if (nsteps > 0) g = loop.run(dt, (unsigned long)nsteps);

		size_t n = 0, nh = 0;
		chk(ctx, ls1hip_count(ctx, &n, &nh), "ls1hip_count");
		std::vector<uint64_t> oid(n);
		std::vector<int32_t> ocid(n);
		std::vector<double> orr(3 * n), ov(3 * n), oq(4 * n), oD(3 * n), oF(3 * n);
		chk(ctx, ls1hip_download_state(ctx, n, oid.data(), ocid.data(), orr.data(), ov.data(), oq.data(), oD.data()), "ls1hip_download_state");
		chk(ctx, ls1hip_download_forces(ctx, n, oF.data(), nullptr, nullptr), "ls1hip_download_forces");
		long builds = 0, lsteps = 0;
		ls1hip_get_option(ctx, "verlet_builds", &builds);
		ls1hip_get_option(ctx, "verlet_steps", &lsteps);
		const std::string out = std::string(argv[2]) + "." + std::to_string(rank);
		FILE* o = fopen(out.c_str(), "wb");
		if (!o) throw std::runtime_error("cannot open result file");
		const uint64_t n64 = n;
		const double sums[8] = {g0.upot, g0.virial, g.upot, g.virial, g.summv2, (double)g.n, (double)builds, (double)lsteps};
		wr(o, "LS1DRES1", 8);
		wr(o, &n64, 1); wr(o, sums, 8);
		wr(o, oid.data(), n); wr(o, orr.data(), 3 * n); wr(o, ov.data(), 3 * n); wr(o, oF.data(), 3 * n);
		fclose(o);
		ls1hip_destroy(ctx);
	} catch (const std::exception& e) {
		std::cerr << "decomp_rccl_main rank " << rank << ": " << e.what() << "\n";
		return 1;
	}
	return 0;
}
