// Micro-benchmark: do fp32 MFMA and ordinary VALU instructions of the SAME SIMD overlap on gfx950, or do they take turns?
// Each wave loops over {K x v_mfma_f32_16x16x4_f32 on independent accumulators, V x v_add_f32 on independent registers}.
// Prints cycles per iteration per SIMD for several (K, V, waves per SIMD).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int V>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x + i;
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < V; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(b));
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int V>
void run(int wgs_per_cu) {
  float *out;
  long long *cyc, h = 0;
  const int iters = 20000, grid = 256 * wgs_per_cu;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMalloc(&cyc, 8);
  hipLaunchKernelGGL((k<K, V>), dim3(grid), dim3(256), 0, 0, out, 200, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, V>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // clock64 = s_memtime at 100 MHz on this part; report the wave-0 view in ns and the kernel time per iteration
  printf("K=%d V=%2d waves/SIMD=%d : %.1f ns per iteration (kernel %.3f ms)  [MFMA alone would be %d x 32 cyc, VALU alone %d x 4 cyc per wave]\n", K, V,
         wgs_per_cu, ms * 1e6 / iters, ms, K, V);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    if (w == 1) { run<3, 0>(1); run<0, 8>(1); run<3, 8>(1); run<3, 16>(1); run<3, 24>(1); run<0, 24>(1); }
    if (w == 2) { run<3, 0>(2); run<0, 8>(2); run<3, 8>(2); run<3, 16>(2); run<3, 24>(2); run<0, 24>(2); }
    if (w == 4) { run<3, 0>(4); run<0, 8>(4); run<3, 8>(4); run<3, 16>(4); run<3, 24>(4); run<0, 24>(4); }
  }
  return 0;
}
