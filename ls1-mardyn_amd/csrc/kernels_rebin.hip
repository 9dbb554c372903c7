// kernels_rebin.hip — device re-binning (counting sort by cell), periodic wrap / leaver packing and halo-copy
// generation.  All streaming, HBM-bound integer/byte work: coalesced SoA reads, one atomic per molecule on a
// per-cell counter (u32, ~12 hits per counter), canonical in-cell order by molecule id so that results do not
// depend on atomic arrival order.
//
// Reference behaviour restated here:
//   LinkedCells::update / update_via_copies        particleContainer/LinkedCells.cpp:243-356
//   handleDomainLeavingParticles                   parallel/DomainDecompBase.cpp:174-225
//   populateHaloLayerWithCopies                    parallel/DomainDecompBase.cpp:293-348
//   CommunicationBuffer record fields              parallel/CommunicationBuffer.cpp:131-145,167-176
#include "common.hpp"

namespace ls1 {

constexpr uint32_t KEY_INVALID = 0xffffffffu;
constexpr int TPB = 256;

__device__ __forceinline__ double next_toward_up(double x) { return nextafter(x, x + 1.0); }
__device__ __forceinline__ double next_toward_down(double x) { return nextafter(x, x - 1.0); }

// One atomic per RUN of equal keys inside a wave instead of one per molecule: the input is (nearly) cell-sorted from
// the previous step, so a wave of 64 molecules touches ~5 cells.  Must be called by converged code paths only for
// the lanes that take part (others pass through the ballot as inactive).
__device__ __forceinline__ uint32_t cell_counter_add(uint32_t* count, uint32_t key) {
	const unsigned long long act = __ballot(1);
	const int lane = threadIdx.x & 63;
	const uint32_t prev = __shfl_up(key, 1);
	const bool prev_active = lane > 0 && ((act >> (lane - 1)) & 1ull);
	const bool head = !prev_active || prev != key;
	const unsigned long long heads = __ballot(head);
	// run of this lane: [start, end) in lane numbers, all active lanes in it are contiguous and share `key`
	const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
	const int start = 63 - __clzll(below);
	const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
	const unsigned long long act_above = (lane == 63) ? 0ull : (~act >> (lane + 1));
	int len_after = 63 - lane;  // lanes after me up to the wave end
	if (above) len_after = min(len_after, __ffsll((long long)above) - 1);
	if (act_above) len_after = min(len_after, __ffsll((long long)act_above) - 1);
	const int end = lane + 1 + len_after;
	uint32_t base = 0;
	if (head) base = atomicAdd(&count[key], (uint32_t)(end - start));
	base = __shfl(base, start);
	return base + (uint32_t)(lane - start);
}

// ---- stage A: wrap / classify / key / in-cell rank --------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_classify(RebinArgs a) {
	const uint32_t p = blockIdx.x * TPB + threadIdx.x;
	if (p >= a.n_in) return;
	double r[3] = {a.src.x[p], a.src.y[p], a.src.z[p]};
	int s[3];
	bool out = false;
	for (int d = 0; d < 3; ++d) {
		s[d] = (r[d] < a.g.bmin[d]) ? -1 : ((r[d] >= a.g.bmax[d]) ? 1 : 0);
		out |= (s[d] != 0);
	}
	uint32_t key = KEY_INVALID;
	if (out) {
		const int dir = (s[2] + 1) * 9 + (s[1] + 1) * 3 + (s[0] + 1);
		const int dest = a.nbr[dir];
		if (dest == a.my_rank) {
			// periodic wrap handled on this rank, incl. the reference's rounding clamps (DomainDecompBase.cpp:206-219)
			for (int d = 0; d < 3; ++d) {
				const double sh = a.shift[dir][d];
				if (sh == 0.) continue;
				r[d] += sh;
				if (sh < 0.) {
					if (r[d] <= a.g.bmin[d]) r[d] = a.g.bmin[d];
				} else {
					if (r[d] >= a.g.bmax[d]) r[d] = next_toward_down(a.g.bmax[d]);
				}
			}
			a.src.x[p] = r[0];
			a.src.y[p] = r[1];
			a.src.z[p] = r[2];
			out = false;
			for (int d = 0; d < 3; ++d) out |= (r[d] < a.g.bmin[d]) || (r[d] >= a.g.bmax[d]);
			if (out) {
				atomicAdd(&a.cnt->err_lost, 1u);  // moved by more than a box length
				a.key[p] = KEY_INVALID;
				return;
			}
		} else if (dest < 0) {
			atomicAdd(&a.cnt->err_lost, 1u);  // left through an open boundary
			a.key[p] = KEY_INVALID;
			return;
		} else {
			// leaves towards another rank: pack the reference's "leaving molecule" record, receiver frame
			const uint32_t slot = atomicAdd(&a.cnt->exp_leave[dir], 1u);
			const uint32_t cap = a.exp_off[dir + 1] - a.exp_off[dir];
			if (slot >= cap) {
				atomicAdd(&a.cnt->err_overflow, 1u);
			} else {
				double* rec = a.exp_leave + (size_t)(a.exp_off[dir] + slot) * LS1HIP_LEAVING_DOUBLES;
				rec[0] = __longlong_as_double((long long)a.src.id[p]);
				rec[1] = __longlong_as_double((long long)a.src.cid[p]);
				rec[2] = r[0] + a.shift[dir][0];
				rec[3] = r[1] + a.shift[dir][1];
				rec[4] = r[2] + a.shift[dir][2];
				rec[5] = a.src.vx[p];
				rec[6] = a.src.vy[p];
				rec[7] = a.src.vz[p];
				if (a.has_rot) {
					rec[8] = a.src.q0[p];
					rec[9] = a.src.q1[p];
					rec[10] = a.src.q2[p];
					rec[11] = a.src.q3[p];
					rec[12] = a.src.Dx[p];
					rec[13] = a.src.Dy[p];
					rec[14] = a.src.Dz[p];
				} else {
					rec[8] = 1.;
					rec[9] = rec[10] = rec[11] = rec[12] = rec[13] = rec[14] = 0.;
				}
			}
			a.key[p] = KEY_INVALID;
			return;
		}
	}
	const int cx = cell_coord_owned(a.g, 0, r[0]);
	const int cy = cell_coord_owned(a.g, 1, r[1]);
	const int cz = cell_coord_owned(a.g, 2, r[2]);
	key = (uint32_t)cell_index(a.g, cx, cy, cz);
	a.key[p] = key;
	a.rank[p] = cell_counter_add(a.count, key);
}

// ---- exclusive scan over the cell grid, restricted to one cell class --------------------------------------------
// mode 0: inner/boundary cells (halo cells contribute 0), result base 0,      total -> cnt->n_real
// mode 1: halo cells,                                     result base n_real, total -> cnt->n_halo
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = TPB * SCAN_ITEMS;

__device__ __forceinline__ uint32_t scan_val(const Grid& g, const uint32_t* count, int c, int mode) {
	if (c >= g.ncells) return 0;
	int cx, cy, cz;
	cell_coords(g, c, cx, cy, cz);
	const bool h = cell_is_halo(g, cx, cy, cz);
	return (h == (mode == 1)) ? count[c] : 0u;
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
	__shared__ uint32_t wsum[TPB / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t inc = v;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o);
		if (lane >= o) inc += t;
	}
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (int i = 0; i < TPB / 64; ++i) {
		if (i < w) base += wsum[i];
		tot += wsum[i];
	}
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

__global__ void __launch_bounds__(TPB) k_scan_blocksums(Grid g, const uint32_t* count, uint32_t* blocksum, int mode) {
	const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
	uint32_t v = 0;
	for (int i = 0; i < SCAN_ITEMS; ++i) v += scan_val(g, count, base + i, mode);
	uint32_t tot;
	block_exclusive_scan(v, &tot);
	if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// Second and last pass of the scan: every block sums the block sums in front of it (a few hundred values: cheaper than
// a separate single-block "top" launch), scans its own cells and writes cell_begin / cell_end; the last block publishes
// the total (n_real or n_halo).
__global__ void __launch_bounds__(TPB) k_scan_apply(Grid g, const uint32_t* count, const uint32_t* blocksum, int nblocks,
													uint32_t* cell_begin, uint32_t* cell_end, DevCounters* cnt, int mode,
													uint32_t cap_halo) {
	__shared__ uint32_t s_base;
	{
		uint32_t part = 0;
		for (int i = threadIdx.x; i < (int)blockIdx.x; i += TPB) part += blocksum[i];
		uint32_t tot;
		block_exclusive_scan(part, &tot);
		if (threadIdx.x == 0) s_base = tot + ((mode == 1) ? cnt->n_real : 0u);
		__syncthreads();
	}
	const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
	uint32_t vals[SCAN_ITEMS], v = 0;
	for (int i = 0; i < SCAN_ITEMS; ++i) {
		vals[i] = scan_val(g, count, base + i, mode);
		v += vals[i];
	}
	uint32_t tot;
	uint32_t ex = block_exclusive_scan(v, &tot) + s_base;
	for (int i = 0; i < SCAN_ITEMS; ++i) {
		const int c = base + i;
		if (c < g.ncells) {
			int cx, cy, cz;
			cell_coords(g, c, cx, cy, cz);
			if (cell_is_halo(g, cx, cy, cz) == (mode == 1)) {
				cell_begin[c] = ex;
				cell_end[c] = ex + vals[i];
			}
		}
		ex += vals[i];
	}
	if ((int)blockIdx.x == nblocks - 1 && threadIdx.x == 0) {
		const uint32_t total = s_base + tot;
		if (mode == 0) {
			cnt->n_real = total;
		} else {
			cnt->n_halo = total - cnt->n_real;
			// clamp the staged count to the capacity (after an overflow the error flag is set; keeps indices in range)
			if (cnt->n_halo_staged > cap_halo) cnt->n_halo_staged = cap_halo;
		}
	}
}

static void run_scan(const Grid& g, const uint32_t* count, uint32_t* blocksum, uint32_t* cell_begin, uint32_t* cell_end,
					 DevCounters* cnt, int mode, hipStream_t s, uint32_t cap_halo = 0) {
	const int nblocks = (g.ncells + SCAN_BLOCK - 1) / SCAN_BLOCK;
	hipLaunchKernelGGL(k_scan_blocksums, dim3(nblocks), dim3(TPB), 0, s, g, count, blocksum, mode);
	hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(TPB), 0, s, g, count, blocksum, nblocks, cell_begin, cell_end, cnt, mode,
					   cap_halo);
}

// ---- stage B: scatter -> canonical in-cell order -> gather ------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_scatter(const uint32_t* key, const uint32_t* rank, const uint32_t* cell_begin,
												 uint32_t* perm, const uint64_t* id, uint64_t* idk, uint32_t n, uint32_t sub) {
	const uint32_t p = blockIdx.x * TPB + threadIdx.x;
	if (p >= n) return;
	const uint32_t k = key[p];
	if (k == KEY_INVALID) return;
	const uint32_t slot = cell_begin[k] + rank[p] - sub;
	perm[slot] = p;
	if (idk) idk[slot] = id[p];  // ids in slot order: the canonical-order pass reads them contiguously per cell
}

// Number of entries of idk[cb, ce) that precede (myid, k) in (id, slot) order.  Eight loads are issued per trip before any
// is used: with one load per trip the loop was a chain of ~12 L1 latencies per molecule (0.1 ms of the re-bin).
__device__ __forceinline__ uint32_t rank_by_id(const uint64_t* idk, uint32_t cb, uint32_t ce, uint32_t k, uint64_t myid) {
	uint32_t r = 0;
	constexpr int U = 8;  // 16 measured slower (0.43 vs 0.41 ms re-bin): wasted loads on 12-molecule cells
	for (uint32_t q0 = cb; q0 < ce; q0 += U) {
		uint64_t o[U];
#pragma unroll
		for (int u = 0; u < U; ++u) o[u] = idk[min(q0 + (uint32_t)u, ce - 1u)];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint32_t q = q0 + (uint32_t)u;
			r += (q < ce) & ((o[u] < myid) | ((o[u] == myid) & (q < k)));
		}
	}
	return r;
}

// Gather into the new set.  Slot k (arrival order inside its cell) is moved to its canonical position: the cell's
// molecules ordered by id, found by counting the smaller ids in the cell's slice (a dozen contiguous, L1-resident
// u64) — no serial per-cell sort, every thread stays busy.
__global__ void __launch_bounds__(TPB) k_gather(RebinArgs a) {
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= a.cnt->n_real) return;
	const uint32_t i = a.perm[k];
	// the payload loads depend on i only: issue them before the rank computation so that their latency overlaps it
	const double x = a.src.x[i], y = a.src.y[i], z = a.src.z[i];
	const double vx = a.src.vx[i], vy = a.src.vy[i], vz = a.src.vz[i];
	const uint64_t id = a.src.id[i];
	const int32_t cid = a.src.cid[i];
	const uint32_t key = a.key[i];
	uint32_t p = k;
	if (a.deterministic) {
		const uint32_t cb = a.cell_begin[key], ce = a.cell_end[key];
		const uint64_t myid = a.idk[k];
		p = cb + rank_by_id(a.idk, cb, ce, k, myid);
	}
	a.dst.x[p] = x;
	a.dst.y[p] = y;
	a.dst.z[p] = z;
	a.dst.vx[p] = vx;
	a.dst.vy[p] = vy;
	a.dst.vz[p] = vz;
	a.dst.id[p] = id;
	a.dst.cid[p] = cid;
	a.ckey[p] = key;
	if (a.has_rot) {
		a.dst.q0[p] = a.src.q0[i];
		a.dst.q1[p] = a.src.q1[i];
		a.dst.q2[p] = a.src.q2[i];
		a.dst.q3[p] = a.src.q3[i];
		a.dst.Dx[p] = a.src.Dx[i];
		a.dst.Dy[p] = a.src.Dy[i];
		a.dst.Dz[p] = a.src.Dz[i];
	}
}

// cell counters and leaving-export counters of a new re-bin (one launch; two hipMemsetAsync were four fill kernels)
__global__ void __launch_bounds__(TPB) k_rebin_reset(uint32_t* count, int ncells, DevCounters* cnt) {
	const int c = blockIdx.x * TPB + threadIdx.x;
	if (c < ncells) count[c] = 0;
	if (c < 27) cnt->exp_leave[c] = 0;
}

void launch_rebin_classify(const RebinArgs& a, hipStream_t s) {
	hipLaunchKernelGGL(k_rebin_reset, dim3((a.g.ncells + TPB - 1) / TPB), dim3(TPB), 0, s, a.count, a.g.ncells, a.cnt);
	if (a.n_in == 0) return;
	hipLaunchKernelGGL(k_classify, dim3((a.n_in + TPB - 1) / TPB), dim3(TPB), 0, s, a);
}

void launch_rebin_sort_gather(const RebinArgs& a, hipStream_t s) {
	run_scan(a.g, a.count, a.blocksum, a.cell_begin, a.cell_end, a.cnt, 0, s);
	if (a.n_in == 0) return;
	const dim3 grid((a.n_in + TPB - 1) / TPB);
	hipLaunchKernelGGL(k_scatter, grid, dim3(TPB), 0, s, a.key, a.rank, a.cell_begin, a.perm, a.src.id,
					   a.deterministic ? a.idk : nullptr, a.n_in, 0u);
	hipLaunchKernelGGL(k_gather, grid, dim3(TPB), 0, s, a);
}

// append received "leaving molecule" records to the source set at [at, at+n)
__global__ void __launch_bounds__(TPB) k_leave_import(RebinArgs a, const double* rec, uint32_t n, uint32_t at) {
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= n) return;
	const double* r = rec + (size_t)i * LS1HIP_LEAVING_DOUBLES;
	const uint32_t p = at + i;
	a.src.id[p] = (uint64_t)__double_as_longlong(r[0]);
	a.src.cid[p] = (int32_t)__double_as_longlong(r[1]);
	// The sender shifted the position into this rank's frame (r + shift across a periodic face): x = -tiny + L rounds to
	// exactly L = bmax, outside [bmin, bmax).  Same rounding clamps as the local wrap (DomainDecompBase.cpp:206-219): an
	// immigrant is by construction owned by this rank, so it is pulled onto the box.
	double rin[3] = {r[2], r[3], r[4]};
	for (int d = 0; d < 3; ++d) {
		if (rin[d] < a.g.bmin[d]) rin[d] = a.g.bmin[d];
		if (rin[d] >= a.g.bmax[d]) rin[d] = next_toward_down(a.g.bmax[d]);
	}
	a.src.x[p] = rin[0];
	a.src.y[p] = rin[1];
	a.src.z[p] = rin[2];
	a.src.vx[p] = r[5];
	a.src.vy[p] = r[6];
	a.src.vz[p] = r[7];
	if (a.has_rot) {
		a.src.q0[p] = r[8];
		a.src.q1[p] = r[9];
		a.src.q2[p] = r[10];
		a.src.q3[p] = r[11];
		a.src.Dx[p] = r[12];
		a.src.Dy[p] = r[13];
		a.src.Dz[p] = r[14];
	}
	// bin on arrival (same rule as k_classify for a molecule inside the box)
	const int cx = cell_coord_owned(a.g, 0, rin[0]);
	const int cy = cell_coord_owned(a.g, 1, rin[1]);
	const int cz = cell_coord_owned(a.g, 2, rin[2]);
	const uint32_t key = (uint32_t)cell_index(a.g, cx, cy, cz);
	a.key[p] = key;
	a.rank[p] = cell_counter_add(a.count, key);
}

void launch_leave_import(const RebinArgs& a, const double* dev_records, uint32_t n, uint32_t at, hipStream_t s) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_leave_import, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, a, dev_records, n, at);
}

__global__ void __launch_bounds__(TPB) k_copy(double* dst, const double* src, uint32_t n) {
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i < n) dst[i] = src[i];
}
// all segments of a message set in ONE launch (a launch per direction was 26 x 2.6 us per exchange)
__global__ void __launch_bounds__(TPB) k_copy_segments(PackSegments seg, const double* src, double* dst) {
	const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
	if (i >= seg.total) return;
	int k = 0;
	while (k + 1 < seg.n && i >= seg.dst_off[k + 1]) ++k;  // scalar-ish walk over <= 27 prefix offsets
	dst[i] = src[seg.src_off[k] + (i - seg.dst_off[k])];
}
void launch_pack_segments(const PackSegments& seg, const double* src, double* dst, hipStream_t s) {
	if (seg.total == 0 || seg.n == 0) return;
	hipLaunchKernelGGL(k_copy_segments, dim3((uint32_t)((seg.total + TPB - 1) / TPB)), dim3(TPB), 0, s, seg, src, dst);
}

void launch_pack_copy(double* dst, const double* src, uint32_t ndoubles, hipStream_t s) {
	if (ndoubles == 0) return;
	hipLaunchKernelGGL(k_copy, dim3((ndoubles + TPB - 1) / TPB), dim3(TPB), 0, s, dst, src, ndoubles);
}

// ---- halo copies --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void halo_stage_write(const HaloArgs& a, uint32_t slot, const double r[3], uint64_t id,
												  int32_t cid, double q0, double q1, double q2, double q3, uint32_t src, int dir) {
	a.hs.src[slot] = src;
	a.hs.dir[slot] = (uint8_t)dir;
	a.hs.x[slot] = r[0];
	a.hs.y[slot] = r[1];
	a.hs.z[slot] = r[2];
	a.hs.id[slot] = id;
	a.hs.cid[slot] = cid;
	if (a.has_rot) {
		a.hs.q0[slot] = q0;
		a.hs.q1[slot] = q1;
		a.hs.q2[slot] = q2;
		a.hs.q3[slot] = q3;
	}
	const int cx = cell_coord_any(a.g, 0, r[0]);
	const int cy = cell_coord_any(a.g, 1, r[1]);
	const int cz = cell_coord_any(a.g, 2, r[2]);
	uint32_t key = KEY_INVALID;
	if (cell_is_halo(a.g, cx, cy, cz)) {
		key = (uint32_t)cell_index(a.g, cx, cy, cz);
		// one atomic per run of equal keys in the wave (the 16 lanes of a shell cell map to one halo cell per direction):
		// a returning atomic per image, ~12 per counter, cost 85 of the 100 us of k_halo_gen
		a.hs.rank[slot] = cell_counter_add(a.count, key);
	} else {
		atomicAdd(&a.cnt->err_lost, 1u);  // a "halo copy" that lies inside the box
	}
	a.hs.key[slot] = key;
}

__device__ __forceinline__ bool halo_flags(const HaloArgs& a, const double r[3], bool lo[3], bool hi[3]) {
	bool any = false;
	for (int d = 0; d < 3; ++d) {
		lo[d] = r[d] < a.g.bmin[d] + a.rc;   // region [min, min+rc): DomainDecompBase.cpp:309-311
		hi[d] = r[d] >= a.g.bmax[d] - a.rc;  // region [max-rc, max):  DomainDecompBase.cpp:312-315
		any |= lo[d] | hi[d];
	}
	return any;
}

// calls f(dir) for every direction in which a molecule with near-face flags lo/hi has an image (open sides skipped)
template <class F>
__device__ __forceinline__ void halo_for_each_image(const HaloArgs& a, const bool lo[3], const bool hi[3], F&& f) {
	for (int sz = -1; sz <= 1; ++sz) {
		if (!(sz == 0 || (sz < 0 ? lo[2] : hi[2]))) continue;
		for (int sy = -1; sy <= 1; ++sy) {
			if (!(sy == 0 || (sy < 0 ? lo[1] : hi[1]))) continue;
			for (int sx = -1; sx <= 1; ++sx) {
				if (!(sx == 0 || (sx < 0 ? lo[0] : hi[0]))) continue;
				if (sx == 0 && sy == 0 && sz == 0) continue;
				const int dir = (sz + 1) * 9 + (sy + 1) * 3 + (sx + 1);
				if (a.nbr[dir] < 0) continue;  // open boundary: no image from that side
				f(dir);
			}
		}
	}
}

// write the image of molecule p in direction dir to slot `slot` of its destination (local staging or export buffer)
__device__ __forceinline__ void halo_emit(const HaloArgs& a, uint32_t p, const double r[3], int dir, uint32_t slot) {
	const uint64_t id = a.mol.id[p];
	const int32_t cid = a.mol.cid[p];
	double q0 = 1., q1 = 0., q2 = 0., q3 = 0.;
	if (a.has_rot) {
		q0 = a.mol.q0[p];
		q1 = a.mol.q1[p];
		q2 = a.mol.q2[p];
		q3 = a.mol.q3[p];
	}
	double rn[3];
	for (int d = 0; d < 3; ++d) rn[d] = r[d] + a.shift[dir][d];
	if (a.nbr[dir] == a.my_rank) {
		// rounding guards of populateHaloLayerWithCopies (DomainDecompBase.cpp:330-343)
		for (int d = 0; d < 3; ++d) {
			const double sh = a.shift[dir][d];
			if (sh < 0.) {
				if (rn[d] >= a.g.bmin[d]) rn[d] = next_toward_down(a.g.bmin[d]);
			} else if (sh > 0.) {
				if (rn[d] < a.g.bmax[d]) rn[d] = next_toward_up(a.g.bmax[d]);
			}
		}
		if (slot >= a.cap_halo) {
			atomicAdd(&a.cnt->err_overflow, 1u);
			return;
		}
		halo_stage_write(a, slot, rn, id, cid, q0, q1, q2, q3, p, dir);
	} else {
		const uint32_t cap = a.exp_off[dir + 1] - a.exp_off[dir];
		if (slot >= cap) {
			atomicAdd(&a.cnt->err_overflow, 1u);
			return;
		}
		double* rec = a.exp_halo + (size_t)(a.exp_off[dir] + slot) * LS1HIP_HALO_DOUBLES;
		a.exp_src[a.exp_off[dir] + slot] = p;  // list-reuse mode re-sends only the position of this molecule
		rec[0] = __longlong_as_double((long long)id);
		rec[1] = __longlong_as_double((long long)cid);
		rec[2] = rn[0];
		rec[3] = rn[1];
		rec[4] = rn[2];
		rec[5] = q0;
		rec[6] = q1;
		rec[7] = q2;
		rec[8] = q3;
	}
}

// 16 lanes per SHELL cell, HG_ITER cells per lane group, 64 cells per workgroup.  Only cells of the outermost hw layers
// of owned cells can hold molecules within rc of a face (cell edge >= rc/hw); their indices are listed once per domain
// (ls1hip_set_domain), so the pass reads ~6 % of the molecules instead of all.
// Slot allocation: the images of a workgroup are counted per direction in LDS, ONE global atomic per workgroup and
// destination reserves the range (all local images share one; measured: one same-address global atomic per wave
// serialised the whole kernel at ~15 ns each = 0.19 ms at 10^7 molecules), then LDS atomics hand out the slots.
// The order inside a destination buffer is arbitrary; the halo sort (key, rank, id) canonicalises it.
constexpr int HG_LANES = 16;
constexpr int HG_ITER = 4;
__global__ void __launch_bounds__(TPB) k_halo_gen(HaloArgs a) {
	__shared__ uint32_t s_cnt[27], s_base[27];
	const int tid = threadIdx.x;
	if (tid < 27) s_cnt[tid] = 0;
	__syncthreads();
	constexpr uint32_t GROUPS = TPB / HG_LANES;
	const uint32_t k0 = blockIdx.x * (GROUPS * HG_ITER) + (uint32_t)tid / HG_LANES;
	const uint32_t sub = (uint32_t)tid % HG_LANES;
	uint32_t pb[HG_ITER], pe[HG_ITER];
#pragma unroll
	for (int it = 0; it < HG_ITER; ++it) {
		const uint32_t k = k0 + (uint32_t)it * GROUPS;
		pb[it] = pe[it] = 0;
		if (k < a.nshell) {
			const uint32_t c = a.shell[k];
			pb[it] = a.cell_begin[c];
			pe[it] = a.cell_end[c];
		}
	}
	// pass 1: count images per direction
#pragma unroll
	for (int it = 0; it < HG_ITER; ++it)
		for (uint32_t p = pb[it] + sub; p < pe[it]; p += HG_LANES) {
			const double r[3] = {a.mol.x[p], a.mol.y[p], a.mol.z[p]};
			bool lo[3], hi[3];
			if (halo_flags(a, r, lo, hi)) halo_for_each_image(a, lo, hi, [&](int dir) { atomicAdd(&s_cnt[dir], 1u); });
		}
	__syncthreads();
	// reserve: remote directions one atomic each (distinct counters), all local images one atomic
	if (tid < 27) {
		const int dest = a.nbr[tid];
		if (tid != 13 && dest >= 0 && dest != a.my_rank && s_cnt[tid]) s_base[tid] = atomicAdd(&a.cnt->exp_halo[tid], s_cnt[tid]);
	} else if (tid == 64) {
		uint32_t tot = 0;
		for (int dir = 0; dir < 27; ++dir)
			if (dir != 13 && a.nbr[dir] == a.my_rank) tot += s_cnt[dir];
		uint32_t base = tot ? atomicAdd(&a.cnt->n_halo_staged, tot) : 0u;
		for (int dir = 0; dir < 27; ++dir)
			if (dir != 13 && a.nbr[dir] == a.my_rank) {
				s_base[dir] = base;
				base += s_cnt[dir];
			}
	}
	__syncthreads();
	if (tid < 27) s_cnt[tid] = 0;
	__syncthreads();
	// pass 2: emit
#pragma unroll
	for (int it = 0; it < HG_ITER; ++it)
		for (uint32_t p = pb[it] + sub; p < pe[it]; p += HG_LANES) {
			const double r[3] = {a.mol.x[p], a.mol.y[p], a.mol.z[p]};
			bool lo[3], hi[3];
			if (halo_flags(a, r, lo, hi))
				halo_for_each_image(a, lo, hi, [&](int dir) {
					const uint32_t slot = s_base[dir] + atomicAdd(&s_cnt[dir], 1u);
					halo_emit(a, p, r, dir, slot);
				});
		}
}

// one global atomic per 1024-record workgroup reserves the staging slots (a per-record or per-wave atomic on the single
// counter serialises at ~15 ns each)
constexpr int HI_TPB = 1024;
__global__ void __launch_bounds__(HI_TPB) k_halo_import(HaloArgs a, const double* rec, uint32_t n) {
	__shared__ uint32_t s_base;
	const uint32_t first = blockIdx.x * HI_TPB;
	if (threadIdx.x == 0) s_base = atomicAdd(&a.cnt->n_halo_staged, min((uint32_t)HI_TPB, n - first));
	__syncthreads();
	const uint32_t i = first + threadIdx.x;
	if (i >= n) return;
	const double* r = rec + (size_t)i * LS1HIP_HALO_DOUBLES;
	const uint32_t slot = s_base + threadIdx.x;
	if (slot >= a.cap_halo) {
		atomicAdd(&a.cnt->err_overflow, 1u);
		return;
	}
	const double rr[3] = {r[2], r[3], r[4]};
	a.imp_slot[a.imp_at + i] = slot;
	halo_stage_write(a, slot, rr, (uint64_t)__double_as_longlong(r[0]), (int32_t)__double_as_longlong(r[1]), r[5], r[6],
					 r[7], r[8], 0xffffffffu, 13);
}

__global__ void __launch_bounds__(TPB) k_halo_gather(HaloArgs a) {
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= a.cnt->n_halo) return;
	const uint32_t i = a.perm[k];
	const uint32_t n_real = a.cnt->n_real;
	uint32_t p = n_real + k;
	if (a.deterministic) {
		// canonical position inside the halo cell: its copies ordered by molecule id, by counting the smaller ids of the
		// cell's slice (as k_gather does for the owned cells; replaces a serial per-cell insertion sort pass)
		const uint32_t key = a.hs.key[i];
		const uint32_t cb = a.cell_begin[key] - n_real, ce = a.cell_end[key] - n_real;
		const uint64_t myid = a.idk[k];
		p = n_real + cb + rank_by_id(a.idk, cb, ce, k, myid);
	}
	a.hsrc[p - n_real] = a.hs.src[i];
	a.hdir[p - n_real] = a.hs.dir[i];
	a.s2s[i] = p - n_real;
	a.mol.x[p] = a.hs.x[i];
	a.mol.y[p] = a.hs.y[i];
	a.mol.z[p] = a.hs.z[i];
	a.mol.id[p] = a.hs.id[i];
	a.mol.cid[p] = a.hs.cid[i];
	if (a.has_rot) {
		a.mol.q0[p] = a.hs.q0[i];
		a.mol.q1[p] = a.hs.q1[i];
		a.mol.q2[p] = a.hs.q2[i];
		a.mol.q3[p] = a.hs.q3[i];
	}
}

// zero the per-cell counters of halo cells only (real cells keep their counts from the rebin)
__global__ void __launch_bounds__(TPB) k_zero_halo_counts(Grid g, uint32_t* count, DevCounters* cnt) {
	const int c = blockIdx.x * TPB + threadIdx.x;
	if (c == 0) {  // the halo counters of the new exchange (was a separate one-thread launch)
		cnt->n_halo = 0;
		cnt->n_halo_staged = 0;
	}
	if (c < 27) cnt->exp_halo[c] = 0;
	if (c >= g.ncells) return;
	int cx, cy, cz;
	cell_coords(g, c, cx, cy, cz);
	if (cell_is_halo(g, cx, cy, cz)) count[c] = 0;
}

void launch_halo_generate(const HaloArgs& a, hipStream_t s) {
	hipLaunchKernelGGL(k_zero_halo_counts, dim3((a.g.ncells + TPB - 1) / TPB), dim3(TPB), 0, s, a.g, a.count, a.cnt);
	if (a.n_real_cap == 0) return;
	if (a.nshell == 0) return;
	hipLaunchKernelGGL(k_halo_gen, dim3(((size_t)a.nshell * HG_LANES + TPB * HG_ITER - 1) / (TPB * HG_ITER)), dim3(TPB), 0, s, a);
}

void launch_halo_import(const HaloArgs& a, const double* dev_records, uint32_t n, hipStream_t s) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_halo_import, dim3((n + HI_TPB - 1) / HI_TPB), dim3(HI_TPB), 0, s, a, dev_records, n);
}

__global__ void __launch_bounds__(TPB) k_halo_scatter(HaloArgs a) {
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= a.cnt->n_halo_staged) return;
	const uint32_t k = a.hs.key[i];
	if (k == KEY_INVALID) return;
	const uint32_t slot = a.cell_begin[k] + a.hs.rank[i] - a.cnt->n_real;
	a.perm[slot] = i;
	a.idk[slot] = a.hs.id[i];
}

// List-reuse mode: between two rebuilds the halo copies keep their slots; only their positions follow the source
// molecules (image = source + shift of its direction; the rounding guards of the generation only matter for binning).
__global__ void __launch_bounds__(TPB) k_halo_refresh(HaloArgs a, const double* sx, const double* sy, const double* sz, double* dx,
													  double* dy, double* dz) {
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= a.cnt->n_halo) return;
	const uint32_t src = a.hsrc[k];
	if (src == 0xffffffffu) return;  // imported copy: refreshed by its owner's message
	const int dir = a.hdir[k];
	const uint32_t p = a.cnt->n_real + k;
	dx[p] = sx[src] + a.shift[dir][0];
	dy[p] = sy[src] + a.shift[dir][1];
	dz[p] = sz[src] + a.shift[dir][2];
}
void launch_halo_refresh(const HaloArgs& a, const double* sx, const double* sy, const double* sz, double* dx, double* dy,
						 double* dz, hipStream_t s) {
	if (a.cap_halo == 0) return;
	hipLaunchKernelGGL(k_halo_refresh, dim3((a.cap_halo + TPB - 1) / TPB), dim3(TPB), 0, s, a, sx, sy, sz, dx, dy, dz);
}

// Multi-rank list-reuse: the molecules exported as halo copies at build time re-send only their current position
// (receiver frame), in the build-time slot layout; the receiver routes record j of the (identically ordered) message to
// the slot its build-time twin was sorted into.
__global__ void __launch_bounds__(TPB) k_refresh_pack(HaloArgs a, const double* sx, const double* sy, const double* sz, double* out) {
	const uint32_t g = blockIdx.x * TPB + threadIdx.x;
	if (g >= a.exp_off[27]) return;
	int dir = 0;
	while (dir < 26 && g >= a.exp_off[dir + 1]) ++dir;
	if (g - a.exp_off[dir] >= a.cnt->exp_halo[dir]) return;
	const uint32_t p = a.exp_src[g];
	out[(size_t)g * 3] = sx[p] + a.shift[dir][0];
	out[(size_t)g * 3 + 1] = sy[p] + a.shift[dir][1];
	out[(size_t)g * 3 + 2] = sz[p] + a.shift[dir][2];
}
void launch_refresh_pack(const HaloArgs& a, const double* sx, const double* sy, const double* sz, double* out, hipStream_t s) {
	if (a.exp_off[27] == 0) return;
	hipLaunchKernelGGL(k_refresh_pack, dim3((a.exp_off[27] + TPB - 1) / TPB), dim3(TPB), 0, s, a, sx, sy, sz, out);
}
__global__ void __launch_bounds__(TPB) k_refresh_import(HaloArgs a, const double* rec, uint32_t n, double* dx, double* dy, double* dz) {
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= n) return;
	const uint32_t k = a.s2s[a.imp_slot[a.imp_at + i]];
	if (k == 0xffffffffu) return;  // its build-time twin was rejected (error already flagged)
	const uint32_t p = a.cnt->n_real + k;
	dx[p] = rec[(size_t)i * 3];
	dy[p] = rec[(size_t)i * 3 + 1];
	dz[p] = rec[(size_t)i * 3 + 2];
}
void launch_refresh_import(const HaloArgs& a, const double* rec, uint32_t n, double* dx, double* dy, double* dz, hipStream_t s) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_refresh_import, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, a, rec, n, dx, dy, dz);
}

void launch_halo_finalize(const HaloArgs& a, hipStream_t s) {
	run_scan(a.g, a.count, a.blocksum, a.cell_begin, a.cell_end, a.cnt, 1, s, a.cap_halo);
	if (a.cap_halo == 0) return;
	const dim3 grid((a.cap_halo + TPB - 1) / TPB);
	hipLaunchKernelGGL(k_halo_scatter, grid, dim3(TPB), 0, s, a);
	hipLaunchKernelGGL(k_halo_gather, grid, dim3(TPB), 0, s, a);
}

}  // namespace ls1
