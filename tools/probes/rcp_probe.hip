// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the Newton-refined forms used by the force kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/rcp_probe.hip -o tools/probes/rcp_probe && tools/probes/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double d = x[i];
	double r0 = __builtin_amdgcn_rcp(d);
	double e = fma(-d, r0, 1.0);
	double r1 = fma(r0, e, r0);
	e = fma(-d, r1, 1.0);
	double r2 = fma(r1, e, r1);
	double q0 = __builtin_amdgcn_rsq(d);
	double f = fma(-d * q0, q0, 1.0);
	double q1 = fma(0.5 * q0, f, q0);
	f = fma(-d * q1, q1, 1.0);
	double q2 = fma(0.5 * q1, f, q1);
	out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
	out[6 * i + 3] = q0; out[6 * i + 4] = q1; out[6 * i + 5] = q2;
}

int main() {
	const int n = 1 << 22;
	std::vector<double> x(n), o(6 * (size_t)n);
	unsigned long long s = 88172645463325252ull;
	for (int i = 0; i < n; ++i) {
		s ^= s << 13; s ^= s >> 7; s ^= s << 17;
		const double u = (double)(s >> 11) / 9007199254740992.0;
		x[i] = std::exp(std::log(1e-3) + u * std::log(1e9));  // 1e-3 .. 1e6, log-uniform
	}
	double *dx, *dout;
	hipMalloc(&dx, n * sizeof(double)); hipMalloc(&dout, 6 * (size_t)n * sizeof(double));
	hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
	hipMemcpy(o.data(), dout, 6 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
	double m[6] = {0, 0, 0, 0, 0, 0};
	for (int i = 0; i < n; ++i) {
		const long double ex = 1.0L / (long double)x[i], eq = 1.0L / sqrtl((long double)x[i]);
		for (int k2 = 0; k2 < 3; ++k2) m[k2] = std::fmax(m[k2], (double)fabsl(((long double)o[6 * (size_t)i + k2] - ex) / ex));
		for (int k2 = 3; k2 < 6; ++k2) m[k2] = std::fmax(m[k2], (double)fabsl(((long double)o[6 * (size_t)i + k2] - eq) / eq));
	}
	printf("max relative error over %d log-uniform inputs in [1e-3, 1e6]\n", n);
	printf("v_rcp_f64 raw %.3e   + 1 Newton %.3e   + 2 Newton %.3e\n", m[0], m[1], m[2]);
	printf("v_rsq_f64 raw %.3e   + 1 Newton %.3e   + 2 Newton %.3e\n", m[3], m[4], m[5]);
	return 0;
}
