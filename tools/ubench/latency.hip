// Dependent-chain latency microbenchmark: scalar loads (K$), LDS reads, vector global loads, fp64 fma, fp64 div.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define AS4 __attribute__((address_space(4)))
__global__ void k_sload(const int AS4 *tab, int n, int *out, unsigned long long *cyc) {
  int idx = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) idx = tab[idx];
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = idx; cyc[blockIdx.x] = t1 - t0; }
}
__global__ void k_lds(const int *tab, int n, int *out, unsigned long long *cyc) {
  __shared__ int s[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = tab[i];
  __syncthreads();
  int idx = threadIdx.x & 1;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) idx = s[idx];
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = idx; cyc[blockIdx.x] = t1 - t0; }
}
__global__ void k_lds_rfl(const int *tab, int n, int *out, unsigned long long *cyc) {
  __shared__ int s[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = tab[i];
  __syncthreads();
  int idx = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) idx = __builtin_amdgcn_readfirstlane(s[idx]);
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = idx; cyc[blockIdx.x] = t1 - t0; }
}
__global__ void k_vload(const int *tab, int n, int *out, unsigned long long *cyc) {
  int idx = threadIdx.x & 1;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) idx = tab[idx];
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = idx; cyc[blockIdx.x] = t1 - t0; }
}
__global__ void k_fma(double a, int n, double *out, unsigned long long *cyc) {
  double x = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) x = x * a + 1.0;
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_div(double a, int n, double *out, unsigned long long *cyc) {
  double x = threadIdx.x + 1.5;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) x = a / x + 1.0;
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_ldsrw(int n, double *out, unsigned long long *cyc) {
  __shared__ double s[1024];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) { double v = s[(threadIdx.x + 1) & 63]; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); s[threadIdx.x] = v + 1.0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = s[threadIdx.x]; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int N = 2048, n = 4096;
  std::vector<int> h(N);
  for (int i = 0; i < N; i++) h[i] = (i * 37 + 11) % N;
  int *tab, *out; unsigned long long *cyc; double *dout;
  hipMalloc(&tab, N * 4); hipMalloc(&out, 4096); hipMalloc(&cyc, 8 * 1024); hipMalloc(&dout, 8 * 64 * 1024);
  hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice);
  unsigned long long c[1024];
  auto rep = [&](const char *name, int blocks) {
    hipDeviceSynchronize(); hipMemcpy(c, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; i++) s += c[i];
    printf("%-28s blocks=%4d  %.1f ticks/iter\n", name, blocks, s / blocks / n);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1, 256, 1024}) {
    for (int rpt = 0; rpt < 2; rpt++) {
    hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(64), 0, 0, (const int AS4 *)tab, n, out, cyc); if (rpt) rep("s_load chain", blocks);
    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, tab, n, out, cyc); if (rpt) rep("ds_read chain", blocks);
    hipLaunchKernelGGL(k_lds_rfl, dim3(blocks), dim3(64), 0, 0, tab, n, out, cyc); if (rpt) rep("ds_read+readfirstlane chain", blocks);
    hipLaunchKernelGGL(k_vload, dim3(blocks), dim3(64), 0, 0, tab, n, out, cyc); if (rpt) rep("global_load chain", blocks);
    hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, 1.0000001, n, dout, cyc); if (rpt) rep("fp64 fma chain", blocks);
    hipLaunchKernelGGL(k_div, dim3(blocks), dim3(64), 0, 0, 1.0000001, n, dout, cyc); if (rpt) rep("fp64 div+add chain", blocks);
    hipLaunchKernelGGL(k_ldsrw, dim3(blocks), dim3(64), 0, 0, n, dout, cyc); if (rpt) rep("lds read->write round", blocks);
    }
  }
  // wall-clock calibration of the tick
  hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, 1.0000001, 4000000, dout, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);
  printf("tick calibration: %llu ticks in %.3f ms -> %.1f MHz\n", c[0], ms, c[0] / ms / 1e3);
  return 0;
}
