// Micro-benchmark: fp32 MFMA issue rate with k waves per SIMD (experiments only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = a0 * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    float4 av = make_float4(a, a + 1, a + 2, a + 3);
    if (LDS) av = *reinterpret_cast<const float4*>(sm + ((threadIdx.x * 4 + it * 36) & 4092));
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k32<NACC, LDS>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k32<NACC, LDS>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 256.0 * blocks_per_cu * 4 * (double)iters * NACC * 4 * 4096;
  printf("%s nacc %d lds %d blocks/CU %d : %.2f ms %.1f TF\n", name, NACC, (int)LDS, blocks_per_cu, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int k = 1; k <= 4; ++k) run<4, false>("32x32x2", k);
  for (int k = 1; k <= 4; ++k) run<4, true>("32x32x2", k);
  for (int k = 1; k <= 3; ++k) run<1, false>("32x32x2", k);
  run<2, false>("32x32x2", 1);
  return 0;
}
