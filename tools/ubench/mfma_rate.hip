// Micro-benchmark: issue rate of the two fp32 MFMA shapes on gfx950, alone, at 1/2/4 waves per SIMD.
//   v_mfma_f32_16x16x4_f32 : 1024 MACs per instruction, documented 8 passes
//   v_mfma_f32_32x32x2_f32 : 2048 MACs per instruction, documented 16 passes
// Prints ns per instruction per SIMD and the MAC rate; the ratio between the shapes does not depend on the clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  float s = 0;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % NACC], 0, 0, 0);
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
  } else {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(int w) {
  float *out;
  const int iters = 4000, grid = 256 * w;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_instr = (double)iters * 12 * w;                  // per SIMD
  const double macs = SHAPE == 16 ? 1024.0 : 2048.0;
  printf("%dx%d  accumulators=%d waves/SIMD=%d : %6.2f ns per instruction per SIMD, %5.1f MAC/ns/SIMD  -> %.0f TFLOP/s chip\n", SHAPE, SHAPE,
         NACC, w, ms * 1e6 / n_instr, macs / (ms * 1e6 / n_instr), 2 * macs / (ms * 1e6 / n_instr) * 1024 / 1e3);
  hipFree(out);
}

int main() {
  run<16, 3>(1); run<16, 3>(2); run<16, 3>(4); run<16, 6>(4);
  run<32, 2>(1); run<32, 2>(2); run<32, 2>(4); run<32, 4>(4);
  return 0;
}
