// Semantics probe of the gfx950 LDS-DMA load used for the region staging of the list force pass (global_load_lds_dword):
// per-lane global source, wave-uniform LDS base + lane * 4, EXEC-masked lanes write nothing, 8-byte aligned (not 16) runs of
// doubles copied dword-wise land bit-exactly.   hipcc --offload-arch=gfx950 -O2 tools/probes/glds_probe.hip -o tools/probes/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__global__ void __launch_bounds__(256) k(const double* src, double* out, int n, int src_off, int dst_off) {
	__shared__ double s[1024];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	for (int i = tid; i < 1024; i += 256) s[i] = -1.0;
	__syncthreads();
	if (wv == 1) {  // one wave copies n doubles from src + src_off to s + dst_off, dword by dword
		const uint32_t nd = 2u * (uint32_t)n;
		for (uint32_t k0 = 0; k0 < nd; k0 += 64u) {
			const uint32_t d = k0 + (uint32_t)lane;
			if (d < nd)
				__builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const uint32_t*>(src + src_off) + d),
												 (lds_ptr)(reinterpret_cast<uint32_t*>(s + dst_off) + k0), 4, 0, 0);
		}
	}
	__syncthreads();
	for (int i = tid; i < 1024; i += 256) out[i] = s[i];
}

int main() {
	const int N = 4096;
	std::vector<double> h(N), o(1024);
	for (int i = 0; i < N; ++i) h[i] = 1000.0 + i + 1.0 / (i + 3);
	double *d, *dout;
	hipMalloc(&d, N * sizeof(double));
	hipMalloc(&dout, 1024 * sizeof(double));
	hipMemcpy(d, h.data(), N * sizeof(double), hipMemcpyHostToDevice);
	int bad = 0;
	const int cases[][3] = {{100, 7, 3}, {97, 1, 5}, {33, 12, 1}, {64, 0, 0}, {1, 3, 9}, {200, 5, 11}};
	for (auto& c : cases) {
		hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dout, c[0], c[1], c[2]);
		hipMemcpy(o.data(), dout, 1024 * sizeof(double), hipMemcpyDeviceToHost);
		for (int i = 0; i < 1024; ++i) {
			const double want = (i >= c[2] && i < c[2] + c[0]) ? h[c[1] + i - c[2]] : -1.0;
			if (o[i] != want) {
				if (bad < 10) printf("case n=%d src_off=%d dst_off=%d: s[%d] = %.17g, want %.17g\n", c[0], c[1], c[2], i, o[i], want);
				++bad;
			}
		}
	}
	printf(bad ? "glds probe: %d mismatches\n" : "glds probe ok (%d)\n", bad);
	return bad != 0;
}
