// What the matrix pipe of THIS chip sustains in loops with no memory traffic at all (experiments only, not part of the
// library): the practical ceiling the convolution kernels are compared with in DESIGN.md beside the datasheet peak.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
// Every wave runs ITERS x 8 independent MFMAs on register operands (8 accumulator tiles, so issue is back to back), one or
// two waves per SIMD, all 256 CUs.  The LDS variants re-read both operands with ds_read_b128 before every MFMA pair, which
// is what a 2x2-tile GEMM inner loop does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 8192;
__device__ unsigned long long g_ticks[2];   // s_memtime at the start / end of workgroup 0's first wave
#define TICK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ticks[i] = __builtin_amdgcn_s_memtime(); } while (0)

__device__ inline float rnd(unsigned x) {   // hash -> (-1, 1): operand bits toggle like real activations / weights
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int)x * (1.0f / 2147483648.0f);
}

__global__ __launch_bounds__(256) void k_bf16_32(float* out, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd(threadIdx.x * 16 + i + (unsigned)seed); b[i] = (__bf16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 4096); }
  f32x16 acc[8] = {};
  TICK(0);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_bf16_16(float* out, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd(threadIdx.x * 16 + i + (unsigned)seed); b[i] = (__bf16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 4096); }
  f32x4 acc[16] = {};
  TICK(0);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_f32_32(float* out, float seed) {
  const float a = rnd(threadIdx.x + (unsigned)seed), b = rnd(threadIdx.x * 3 + 1 + blockIdx.x * 4096);
  f32x16 acc[8] = {};
  TICK(0);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 2x2 tiles of 32x32x16: per k-step two A and two B fragments come from LDS (conflict-free 16-byte rows), four MFMAs
__global__ __launch_bounds__(256) void k_bf16_32_lds(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][64 * 8 * 4], sb[2][64 * 8 * 4];
  for (int i = threadIdx.x; i < 64 * 8 * 4; i += 256) {
    sa[0][i] = (__bf16)rnd(i + (unsigned)seed); sa[1][i] = (__bf16)rnd(i + 77777);
    sb[0][i] = (__bf16)rnd(i + 33333 + blockIdx.x * 4096); sb[1][i] = (__bf16)rnd(i + 55555);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4] = {};
  TICK(0);
  for (int it = 0; it < ITERS * 2; ++it) {
    const int o = (wave * 64 + lane) * 8 + 0 * (it & 1);
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&sa[0][o]), a1 = *reinterpret_cast<const bf16x8*>(&sa[1][o]);
    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&sb[0][o]), b1 = *reinterpret_cast<const bf16x8*>(&sb[1][o]);
    asm volatile("" ::: "memory");
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// dependent-chain variants: NACC accumulator tiles only, so an MFMA's accumulator was written NACC MFMAs earlier
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16_32_acc(float* out, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd(threadIdx.x * 16 + i + (unsigned)seed); b[i] = (__bf16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 4096); }
  f32x16 acc[NACC] = {};
  TICK(0);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t % NACC], 0, 0, 0);
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 1x2 tiles: per k-step one A and two B fragments from LDS, two MFMAs on two accumulators, fragments read one step ahead
__global__ __launch_bounds__(256) void k_bf16_32_lds12(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][64 * 8 * 4], sb[2][64 * 8 * 4];
  for (int i = threadIdx.x; i < 64 * 8 * 4; i += 256) {
    sa[0][i] = (__bf16)rnd(i + (unsigned)seed); sa[1][i] = (__bf16)rnd(i + 77777);
    sb[0][i] = (__bf16)rnd(i + 33333 + blockIdx.x * 4096); sb[1][i] = (__bf16)rnd(i + 55555);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2] = {};
  const int stride = (int)seed >> 20;      // 0 at run time, unknown at compile time: the reads stay in the loop
  int o = (wave * 64 + lane) * 8;
  bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&sa[0][o]), b0 = *reinterpret_cast<const bf16x8*>(&sb[0][o]),
         b1 = *reinterpret_cast<const bf16x8*>(&sb[1][o]);
  TICK(0);
  for (int it = 0; it < ITERS * 4; ++it) {
    o = (o + stride) & 2047;
    const bf16x8 na = *reinterpret_cast<const bf16x8*>(&sa[0][o]), nb0 = *reinterpret_cast<const bf16x8*>(&sb[0][o]),
                 nb1 = *reinterpret_cast<const bf16x8*>(&sb[1][o]);
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = na; b0 = nb0; b1 = nb1;
  }
  TICK(1);
  float s = 0.f;
  for (int t = 0; t < 2; ++t) s += acc[t][0] + acc[t][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Does vector arithmetic of one wave overlap the MFMAs of another wave on the same SIMD?  512-thread workgroups: waves
// 0-3 run MFMAs back to back (when mode & 1), waves 4-7 packed fp32 FMAs (when mode & 2).  g_mix[0..1] = ticks of wave 0,
// g_mix[2..3] = ticks of wave 4.
__device__ unsigned long long g_mix[4];
template <int mode>
__global__ __launch_bounds__(512) void k_mix(float* out, int) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int wave = threadIdx.x >> 6;
  constexpr int mfma_first = !(mode & 4);          // mode & 4: the FMA waves are the older ones (0-3); mode & 8: they raise their priority
  const int t0 = mfma_first ? 0 : 256, t1 = 256 - t0;
  if ((mode & 512) || (wave < 4) == (mfma_first != 0)) {      // mode & 512: all eight waves take the MFMA role
    if (!(mode & 1)) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd(threadIdx.x * 16 + i); b[i] = (__bf16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 4096); }
    f32x16 acc[4] = {};
    if (blockIdx.x == 0 && threadIdx.x == t0) g_mix[0] = __builtin_amdgcn_s_memtime();
    constexpr int nops = (mode >> 4) & 3;       // mode bits 4-5: s_nop cycles after every MFMA: 0, 8, 16 or 24
    f32x2 w[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
    const f32x2 wm = {1.0001f, 0.9999f}, wc = {rnd(threadIdx.x), rnd(threadIdx.x + 99)};
    constexpr int own = (mode >> 6) & 7;        // mode bits 6-8: packed FMAs of this same wave after every MFMA
    for (int it = 0; it < 2048; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t & 3], 0, 0, 0);
        if (nops == 1) asm volatile("s_nop 7");
        if (nops == 2) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
        if (nops == 3) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
        if (own >= 2) { w[0] = __builtin_elementwise_fma(w[0], wm, wc); w[1] = __builtin_elementwise_fma(w[1], wm, wc); }
        if (own >= 4) { w[2] = __builtin_elementwise_fma(w[2], wm, wc); w[3] = __builtin_elementwise_fma(w[3], wm, wc); }
        if (own >= 6) { w[0] = __builtin_elementwise_fma(w[0], wm, wc); w[1] = __builtin_elementwise_fma(w[1], wm, wc); }
      }
    }
    acc[0][1] += w[0].x + w[1].y + w[2].x + w[3].y;
    if (blockIdx.x == 0 && threadIdx.x == t0) g_mix[1] = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    f32x2 v[8], m = {1.0001f, 0.9999f}, c = {rnd(threadIdx.x), rnd(threadIdx.x + 99)};
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){rnd(threadIdx.x + i), rnd(threadIdx.x * 3 + i)};
    if (mode & 8) __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && threadIdx.x == t1) g_mix[2] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 8192; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
    }
    if (blockIdx.x == 0 && threadIdx.x == t1) g_mix[3] = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

static void run_mix(float* out) {
  const int modes[] = {1, 2, 3, 3 | 8, 3 | 4, 3 | 4 | 8, 3 | 16, 3 | 32, 3 | 48, 1 | (2 << 6), 1 | (4 << 6), 1 | (6 << 6), 1 | 512, 1 | 512 | (4 << 6), 1 | 512 | (6 << 6)};
  const char* names[] = {"MFMA waves only", "FMA waves only", "both, MFMA waves older", "both, MFMA older, FMA prio 3", "both, FMA waves older",
                         "both, FMA older, FMA prio 3", "both, s_nop 7 after each MFMA", "both, 2 x s_nop 7", "both, 3 x s_nop 7",
                         "MFMA waves only, +2 own FMAs", "MFMA waves only, +4 own FMAs", "MFMA waves only, +6 own FMAs",
                         "8 MFMA waves", "8 MFMA waves, +4 own FMAs", "8 MFMA waves, +6 own FMAs"};
  typedef void (*kfn)(float*, int);
  const kfn fns[] = {k_mix<1>, k_mix<2>, k_mix<3>, k_mix<3 | 8>, k_mix<3 | 4>, k_mix<3 | 4 | 8>, k_mix<3 | 16>, k_mix<3 | 32>, k_mix<3 | 48>,
                     k_mix<1 | (2 << 6)>, k_mix<1 | (4 << 6)>, k_mix<1 | (6 << 6)>, k_mix<1 | 512>, k_mix<1 | 512 | (4 << 6)>, k_mix<1 | 512 | (6 << 6)>};
  for (int k = 0; k < 15; ++k) {
    const int mode = modes[k];
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fns[k], dim3(256), dim3(512), 0, 0, out, mode);
    (void)hipDeviceSynchronize();
    unsigned long long t[4];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_mix), sizeof(t));
    printf("mix, %-30s: ticks per MFMA %.1f (16384 per wave), ticks per v_pk_fma_f32 %.2f (65536 per wave)\n", names[k],
           (mode & 1) ? (double)(t[1] - t[0]) / 16384.0 : 0.0, (mode & 2) ? (double)(t[3] - t[2]) / 65536.0 : 0.0);
  }
}

template <typename K>
static void run(const char* name, K kern, double flop_per_wave, double mfma_per_wave, int waves_per_simd, float* out) {
  const int blocks = 256 * waves_per_simd;          // 256-thread workgroups: one wave per SIMD each
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);   // clocks settle
  (void)hipDeviceSynchronize();
  const int reps = 100;
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f + i);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_wave * 4.0 * blocks * reps;
  unsigned long long t[2];
  (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof(t));
  const double ticks = (double)(t[1] - t[0]);
  printf("%-44s %d wave(s)/SIMD: %8.1f TFLOP/s  (%.3f ms per launch; s_memtime ticks per MFMA of one wave %.1f, ticks per us %.0f)\n", name,
         waves_per_simd, flops / (ms * 1e-3) / 1e12, ms / reps, ticks / mfma_per_wave, ticks / (ms / reps * 1e3));
}

int main(int argc, char** argv) {
  float* out;
  if (hipMalloc(&out, 256 * 2 * 256 * sizeof(float)) != hipSuccess) return 1;
  if (argc > 1 && !strcmp(argv[1], "brief")) {     // what bench.py reads for frac_vs_sustained: two lines, under a second
    run("bf16 32x32x16, 2x2 tile, operands from LDS", k_bf16_32_lds, (double)ITERS * 2 * 4 * 2.0 * 32 * 32 * 16, ITERS * 8.0, 2, out);
    run("fp32 32x32x2, register operands", k_f32_32, (double)ITERS * 8 * 2.0 * 32 * 32 * 2, ITERS * 8.0, 2, out);
    (void)hipFree(out);
    return 0;
  }
  for (int w = 1; w <= 2; ++w) {
    run("bf16 32x32x16, register operands", k_bf16_32, (double)ITERS * 8 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("bf16 16x16x32, register operands", k_bf16_16, (double)ITERS * 16 * 2.0 * 16 * 16 * 32, ITERS * 16.0, w, out);
    run("bf16 32x32x16, 2x2 tile, operands from LDS", k_bf16_32_lds, (double)ITERS * 2 * 4 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("bf16 32x32x16, register operands, 4 acc tiles", k_bf16_32_acc<4>, (double)ITERS * 8 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("bf16 32x32x16, register operands, 2 acc tiles", k_bf16_32_acc<2>, (double)ITERS * 8 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("bf16 32x32x16, register operands, 1 acc tile", k_bf16_32_acc<1>, (double)ITERS * 8 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("bf16 32x32x16, 1x2 tile, operands from LDS", k_bf16_32_lds12, (double)ITERS * 4 * 2 * 2.0 * 32 * 32 * 16, ITERS * 8.0, w, out);
    run("fp32 32x32x2, register operands", k_f32_32, (double)ITERS * 8 * 2.0 * 32 * 32 * 2, ITERS * 8.0, w, out);
  }
  run_mix(out);
  (void)hipFree(out);
  return 0;
}
