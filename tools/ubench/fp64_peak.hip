// fp64 vector FMA throughput of the device ("measure a v_fma_f64 microbenchmark on the box and use the
// measured value"): 16 independent fma chains per lane, 8 waves per SIMD.  Prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_fma(double a, int n, double *out)
{
	double x[16];
#pragma unroll
	for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i;
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) x[i] = __builtin_fma(x[i], a, 1e-9);
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += x[i];
	if (s == 12345.678) out[blockIdx.x] = s;
}
int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int blocks = p.multiProcessorCount * 8, n = 20000;
	double *out;
	hipMalloc(&out, blocks * sizeof(double));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	k_fma<<<blocks, 256>>>(0.999999, 100, out);
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int r = 0; r < 5; r++) {
		hipEventRecord(e0);
		k_fma<<<blocks, 256>>>(0.999999, n, out);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	const double flops = 2.0 * 16 * (double)n * 256.0 * blocks;
	printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"fp64_fma_tflops\": %.2f, \"ms\": %.3f}\n", p.name,
	       p.multiProcessorCount, p.clockRate / 1000, flops / (best * 1e-3) / 1e12, best);
	return 0;
}
