// fp32 convolutions on the bf16 matrix pipe by operand splitting (UMT_PREC_FP32_BF16X3).
//
// An fp32 number is the exact sum of three bf16 numbers, x = x0 + x1 + x2 with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (round to nearest even; the two subtractions are exact and
// |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|).  A product a*b is then the sum of nine bf16 x bf16 products, each
// exact in fp32; the kernel keeps the six largest,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// and drops a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b| - below the 2^-24-of-the-running-sum rounding every
// fp32 accumulation step makes anyway.  The sums run in the fp32 accumulators of
// v_mfma_f32_32x32x16_bf16, whose rate is 16x that of v_mfma_f32_32x32x2_f32: six of them per fp32
// product leave 2.67x the fp32 pipe's arithmetic rate.  Results are NOT bit-identical to the fmaf-chain
// kernel (conv_igemm.hip), they agree with it to fp32 rounding noise; tests/test_gpu_x3.py states the bound.
// (The split is exact for |x| >= 2^-110; below that the low pieces underflow and the sum is off by < 2^-133.)
// Finite operands are assumed: an infinite activation splits into (inf, NaN, NaN) and poisons its outputs
// with NaN where the fp32 kernel would give inf; NaN propagates as NaN in both.
//
// Interface and structure are those of conv3x3s1_halo_f32: fp32 channels-last rows in and out, fp32
// epilogue (conv bias -> BN -> residual -> ReLU), one flattened halo range of pixel rows staged per
// 32-channel chunk, weights streamed from L2 in fragment order.  The activations are split ONCE, when a
// row is written to LDS (three bf16 planes per row), not per operand fetch: a staged element feeds 9 taps
// x cout/32 fragments.  The weights are split on the host at umt_model_finalize:
//     packed[tap][cin/16][plane][h][cout_w][8].
#include "umt_internal.h"
#include "conv_f32_common.h"
#include <stdlib.h>

// workgroup tilings (waves along pixels, waves along cout, 32x32 tiles per wave along pixels / cout).  A wave's B
// fragments are reused for TM pixel tiles: with TM = 1 the four SIMDs of a CU ask the vector L1 for 64 B per cycle.
#ifndef UMT_X3_CFG64
#define UMT_X3_CFG64 2, 2, 2, 1
#endif
#ifndef UMT_X3_CFG32
#define UMT_X3_CFG32 4, 1, 2, 1
#endif

namespace umt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_PLANE = 64;                // bytes of one plane of a row: 32 bf16
constexpr int X3_ROW = 3 * X3_PLANE + 16;   // 208 B = 13 sixteen-byte slots: 16 consecutive rows hit 16 distinct slots

// two fp32 -> their packed bf16 roundings, and the exact remainders
__device__ __forceinline__ unsigned x3_split2(float& a, float& b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  a -= __builtin_bit_cast(float, p << 16);
  b -= __builtin_bit_cast(float, p & 0xffff0000u);
  return p;
}

// float4 -> three planes of 4 bf16 (8 bytes each)
__device__ __forceinline__ void x3_split4(float4 v, uint2& p0, uint2& p1, uint2& p2) {
  p0.x = x3_split2(v.x, v.y); p0.y = x3_split2(v.z, v.w);
  p1.x = x3_split2(v.x, v.y); p1.y = x3_split2(v.z, v.w);
  p2.x = x3_split2(v.x, v.y); p2.y = x3_split2(v.z, v.w);
}

#ifndef UMT_X3_MINB_SMALL
#define UMT_X3_MINB_SMALL 3
#endif
template <int WAVES_M, int WAVES_N, int TM, int TN, bool RES>
__global__ __launch_bounds__(256, (WAVES_N == 2 && TM * TN <= 2 ? UMT_X3_MINB_SMALL : 2)) void conv3x3s1_halo_x3(ConvArgs a) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char sX[];   // (BM + 2W + 2) rows x 208 B + one zero row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int g0 = m0 - W - 1;
  const int c16_total = a.cin_s >> 4;
  const int nchunks = (a.cin_s + KC - 1) / KC;

  // byte offset of the lane's operand row per tap (the all-zero row where the neighbour does not exist)
  int arow[TM][9];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + li;
    const bool live = m < a.M;
    const int mm = live ? m : 0;
    const int rem = mm - div_magic32(mm, H * W, a.mg_hw) * (H * W);
    const int y = div_magic32(rem, W, a.mg_w), x = rem - y * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        arow[i][ky * 3 + kx] = (ok ? (wm * TM + i) * 32 + li + ky * W + kx : rows) * X3_ROW + 16 * lh;
      }
  }
  if (tid < X3_ROW / 16) *reinterpret_cast<uint4*>(sX + rows * X3_ROW + 16 * tid) = make_uint4(0, 0, 0, 0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // B fragments: buffer loads with a scalar (tap, channel group, plane) offset and one 32-bit lane offset.
  // A k-step is one 16-channel group of one tap (18 per chunk); its 3 x TN fragments are requested two
  // k-steps (2 x 6 TM TN MFMAs) ahead into a ring of three slots - 18 % 3 == 0, so a slot index is static.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int w_plane = 2 * a.cout_w * 16;                  // bytes per (tap, group, plane)
  const int wvoff = (n0 + lh * a.cout_w + wn * TN * 32 + li) * 16;
  uint4 bring[3][3][TN];
  auto load_b = [&](uint4 (&dst)[3][TN], int tap, int c16) {
    const int g = min(c16, c16_total - 1);                // a group past the end meets zero-filled A
    const int soff = (tap * c16_total + g) * 3 * w_plane;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + j * 512, soff + p * w_plane, 0);
        dst[p][j] = make_uint4(v.x, v.y, v.z, v.w);
      }
  };
  load_b(bring[0], 0, 0);
  load_b(bring[1], 0, 1);

  // A staging as in conv3x3s1_halo_f32 (thread t owns slots t + 256 k: row idx>>3, channel quad idx&7); the
  // next chunk's range is fetched into registers while this chunk's MFMAs run
  constexpr int NLD = (BM + 2 * 48 + 2 + 31) / 32;     // W <= 48
  float4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int chunk) {
    const int c0 = chunk * KC;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 3, q = idx & 7;
      const int g = g0 + r;
      const bool ok = r < rows && g >= 0 && g < a.M && (c0 + 4 * q) < a.cin_s;
      const size_t off = ok ? (size_t)g * a.cin_s + c0 + 4 * q : 0;
      areg[k] = *reinterpret_cast<const float4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(0);

  auto read_a = [&](bf16x8 (&dst)[TM][3], int tap, int c) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sX + arow[i][tap] + p * X3_PLANE + c * 32));
  };
  // the six products of one k-step, smallest terms first: plane pairs (pa, pb) with pa + pb = 2, 1, 0
  auto mfma6 = [&](const bf16x8 (&av)[TM][3], const uint4 (&b)[3][TN]) {
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int pa = 0; pa <= s; ++pa)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][pa], __builtin_bit_cast(bf16x8, b[s - pa][j]), acc[i][j], 0, 0, 0);
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) {
        uint2 p0, p1, p2;
        x3_split4(v, p0, p1, p2);
        unsigned char* d = sX + (idx >> 3) * X3_ROW + 8 * (idx & 7);
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + X3_PLANE) = p1;
        *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = p2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int tap = ks >> 1, c = ks & 1;
      if (ks < 16) load_b(bring[(ks + 2) % 3], (ks + 2) >> 1, chunk * 2 + (ks & 1));
      else if (chunk + 1 < nchunks) load_b(bring[(ks + 2) % 3], 0, (chunk + 1) * 2 + (ks & 1));
      if (ks == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 av[TM][3];
      read_a(av, tap, c);
      mfma6(av, bring[ks % 3]);
    }
    __syncthreads();
  }

  float* patch = reinterpret_cast<float*>(sX) + wave * 32 * LDA;
  epilogue<TM, TN, RES>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, a);
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_halo_x3(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  dim3 grid((a.M + BM - 1) / BM, a.cout_w / BN);
  size_t lds = (size_t)(BM + 2 * a.W + 2 + 1) * X3_ROW;
  const size_t patch = (size_t)4 * 32 * LDA * sizeof(float);
  if (lds < patch) lds = patch;
  void (*kern)(ConvArgs) = a.residual ? conv3x3s1_halo_x3<WAVES_M, WAVES_N, TM, TN, true>
                                      : conv3x3s1_halo_x3<WAVES_M, WAVES_N, TM, TN, false>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);      // > 64 KB of LDS: see conv_x3_prepare_device
}

// ------------------------------------------------------------------------------ stride 2 + down-sample
// Split-operand form of conv3x3s2_ds_f32 (conv_igemm.hip): conv1 (3x3, stride 2) of a stage's first block and its 1x1
// stride-2 down-sample branch in one kernel.  The input range of a 64-pixel output tile (about four input pixels per
// output pixel) is staged once per 32-channel chunk, split into planes on the way; the centre tap feeds both
// accumulator sets.  WAVES_M x TM = 2 pixel tiles: either two waves along pixels and two along cout (64-wide cout
// tile) or one along pixels with both tiles and four along cout (128-wide) - the latter halves the weight fetches per MFMA.
constexpr int X3S2_BM = 64;

struct ConvS2ArgsX3 {
  ConvArgs c1;              // c1.w: conv1 weights, three-way split
  const uint16_t* ds_w;     // down-sample weights, three-way split [1][cin/16][plane][2][cout_w][8], or null
  const float* ds_bias;
  const float* ds_scale;
  const float* ds_shift;
  float* ds_out;
  int rows_cap;             // staged rows the host sized LDS for; the zero row is row rows_cap
};

template <int WAVES_M, int TM, bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3s2_ds_x3(ConvS2ArgsX3 g) {
  const ConvArgs& a = g.c1;
  constexpr int WAVES_N = 4 / WAVES_M;
  static_assert(WAVES_M * TM * 32 == X3S2_BM, "64 output pixels per workgroup");
  constexpr int BM = X3S2_BM, BN = WAVES_N * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char sX[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c16_total = a.cin_s >> 4;
  const int nchunks = (a.cin_s + KC - 1) / KC;

  // centre-tap input pixel of output pixel m: 2W * (global output row) + 2 * xo
  auto centre = [&](int m) {
    const int R = div_magic32(m, Wo, a.mg_w), xo = m - R * Wo;      // for this kernel mg_w / mg_hw divide by Wo / Ho
    return 2 * W * R + 2 * xo;
  };
  const int m_last = min(m0 + BM, a.M) - 1;
  const int g0 = centre(m0) - W - 1;                     // global input pixel row of LDS row 0
  const int rows = centre(m_last) + W + 1 - g0 + 1;      // <= rows_cap (checked on the host)
  const int zrow = g.rows_cap;

  int arow[TM][9];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + li;
    const bool live = m < a.M;
    const int mm = live ? m : m0;
    const int R = div_magic32(mm, Wo, a.mg_w), xo = mm - R * Wo, yo = R - div_magic32(R, Ho, a.mg_hw) * Ho;
    const int base = 2 * W * R + 2 * xo - g0;            // LDS row of the centre tap
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
        const bool ok = live && yy >= 0 && yy < a.H && xx >= 0 && xx < W;
        arow[i][ky * 3 + kx] = (ok ? base + (ky - 1) * W + (kx - 1) : zrow) * X3_ROW + 16 * lh;
      }
  }
  if (tid < X3_ROW / 16) *reinterpret_cast<uint4*>(sX + zrow * X3_ROW + 16 * tid) = make_uint4(0, 0, 0, 0);

  f32x16 acc[TM][1], accd[TM][1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; accd[i][0][r] = 0.f; }

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t dblock =
      __builtin_amdgcn_make_buffer_rsrc(DS ? const_cast<uint16_t*>(g.ds_w) : reinterpret_cast<uint16_t*>(const_cast<float*>(a.w)), 0, 0x7fffffff, 0x00020000);
  const int w_plane = 2 * a.cout_w * 16;
  const int wvoff = (n0 + lh * a.cout_w + wn * 32 + li) * 16;
  uint4 bring[3][3], bds[2][3];
  auto load_b = [&](uint4 (&dst)[3], __amdgpu_buffer_rsrc_t blk, int tap, int c16) {
    const int gch = min(c16, c16_total - 1);
    const int soff = (tap * c16_total + gch) * 3 * w_plane;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(blk, wvoff, soff + p * w_plane, 0);
      dst[p] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  load_b(bring[0], wblock, 0, 0);
  load_b(bring[1], wblock, 0, 1);

  constexpr int NLD = 14;                                // float4 slots per thread: 448 rows x 8 quads / 256
  float4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int chunk) {
    const int c0 = chunk * KC;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 3, q = idx & 7;
      const int gp = g0 + r;
      const bool ok = r < rows && gp >= 0 && (c0 + 4 * q) < a.cin_s;      // gp < n*H*W follows from r < rows
      const size_t off = ok ? (size_t)gp * a.cin_s + c0 + 4 * q : 0;
      areg[k] = *reinterpret_cast<const float4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(0);

  auto mfma6 = [&](f32x16 (&dst)[TM][1], const bf16x8 (&av)[TM][3], const uint4 (&b)[3]) {
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
      for (int pa = 0; pa <= s; ++pa)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          dst[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][pa], __builtin_bit_cast(bf16x8, b[s - pa]), dst[i][0], 0, 0, 0);
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) {
        uint2 p0, p1, p2;
        x3_split4(v, p0, p1, p2);
        unsigned char* d = sX + (idx >> 3) * X3_ROW + 8 * (idx & 7);
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + X3_PLANE) = p1;
        *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = p2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int tap = ks >> 1, c = ks & 1;
      if (ks < 16) load_b(bring[(ks + 2) % 3], wblock, (ks + 2) >> 1, chunk * 2 + (ks & 1));
      else if (chunk + 1 < nchunks) load_b(bring[(ks + 2) % 3], wblock, 0, (chunk + 1) * 2 + (ks & 1));
      if (DS && (ks == 5 || ks == 6)) load_b(bds[ks - 5], dblock, 0, chunk * 2 + (ks - 5));
      if (ks == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 av[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          av[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sX + arow[i][tap] + p * X3_PLANE + c * 32));
      mfma6(acc, av, bring[ks % 3]);
      if (DS && tap == 4) mfma6(accd, av, bds[c]);
    }
    __syncthreads();
  }

  float* patch = reinterpret_cast<float*>(sX) + wave * 32 * LDA;
  epilogue<TM, 1, false>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * 32, a);
  if (DS) {
    ConvArgs d = a;
    d.bias = g.ds_bias; d.scale = g.ds_scale; d.shift = g.ds_shift; d.out = g.ds_out; d.relu = 0; d.residual = nullptr;
    epilogue<TM, 1, false>(accd, patch, lane, m0 + wm * TM * 32, n0 + wn * 32, d);
  }
}

bool conv_s2_x3_applicable(const ConvLayer& c1, const ConvLayer* ds, int h, int w) {
  if ((c1.flags & CONV_X3_NO_S2) || !c1.wx3 || (ds && !ds->wx3) || c1.cin_s % 16 != 0 || c1.cout_w % 128 != 0) return false;   // see below for 64
  return conv_s2_fused_applicable(c1, ds, h, w);
}

int launch_conv_s2_fused_x3(const ConvLayer& c1, const ConvLayer* ds, const float* in, int n, int h, int w, float* out,
                            float* ds_out, hipStream_t stream) {
  UMT_REQUIRE(conv_s2_x3_applicable(c1, ds, h, w), UMT_ERR_INVALID, "conv_s2_x3: layer shapes not supported");
  UMT_REQUIRE(n > 0 && (!ds || ds_out), UMT_ERR_INVALID, "conv_s2_x3: bad arguments");
  ConvS2ArgsX3 g;
  ConvArgs& a = g.c1;
  a.in = in; a.w = reinterpret_cast<const float*>(c1.wx3); a.bias = c1.bias; a.scale = c1.scale; a.shift = c1.shift;
  a.residual = nullptr; a.out = out;
  a.H = h; a.W = w; a.Ho = h / 2; a.Wo = w / 2;
  a.cin_s = c1.cin_s; a.cout_s = c1.cout_s; a.cout_w = c1.cout_w; a.stride = 2; a.taps = 9; a.relu = c1.relu;
  UMT_REQUIRE((int64_t)n * h * w < ((int64_t)1 << 31) - 65536, UMT_ERR_INVALID, "conv_s2_x3: too many pixels");
  a.M = n * a.Ho * a.Wo;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Wo) + 1u;      // this kernel divides by Wo and Ho
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Ho) + 1u;
  g.ds_w = ds ? ds->wx3 : nullptr; g.ds_bias = ds ? ds->bias : nullptr; g.ds_scale = ds ? ds->scale : nullptr;
  g.ds_shift = ds ? ds->shift : nullptr; g.ds_out = ds_out;
  g.rows_cap = 2 * w * ((X3S2_BM - 1 + a.Wo - 1) / a.Wo) + 3 * w + 1;  // <= 448 (conv_s2_fused_applicable)
  size_t lds = (size_t)(g.rows_cap + 1) * X3_ROW;
  const size_t patch = (size_t)4 * 32 * LDA * sizeof(float);
  if (lds < patch) lds = patch;
  // 128-wide cout tiles only.  The <2, 1> arrangement (64-wide tiles) was measured on the 32->64 layer at 48x48: its
  // 433 staged rows need 90 KB of LDS, one workgroup per CU, 1.17 ms against 1.09 ms for conv3x3s2_ds_f32.
  dim3 grid((a.M + X3S2_BM - 1) / X3S2_BM, c1.cout_w / 128);
  void (*kern)(ConvS2ArgsX3) = ds ? conv3x3s2_ds_x3<1, 2, true> : conv3x3s2_ds_x3<1, 2, false>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

bool conv_x3_applicable(const ConvLayer& L, int w) {
  return L.wx3 && L.taps == 9 && L.stride == 1 && w <= 48 && L.cin_s % 16 == 0 &&
         (L.cout_w == 32 || L.cout_w == 64 || L.cout_w == 96 || L.cout_w % 128 == 0);
}

int launch_conv_x3(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual, float* out,
                   hipStream_t stream) {
  UMT_REQUIRE(conv_x3_applicable(L, w), UMT_ERR_INVALID, "conv_x3: layer not supported (taps %d stride %d w %d cout %d)",
              L.taps, L.stride, w, L.cout_w);
  UMT_REQUIRE(L.cout_s % 8 == 0 && L.cout_s <= L.cout_w && n > 0 && h > 0, UMT_ERR_INVALID, "conv_x3: bad shapes");
  ConvArgs a;
  a.in = in; a.w = reinterpret_cast<const float*>(L.wx3); a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
  a.residual = residual; a.out = out;
  a.H = h; a.W = w; a.Ho = h; a.Wo = w;
  a.cin_s = L.cin_s; a.cout_s = L.cout_s; a.cout_w = L.cout_w;
  a.stride = 1; a.taps = 9; a.relu = L.relu;
  const int64_t M = (int64_t)n * h * w;
  UMT_REQUIRE(M < (int64_t)1 << 31, UMT_ERR_INVALID, "conv_x3: too many pixels");
  a.M = (int)M;
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
  if (L.cout_w % 128 == 0) launch_halo_x3<2, 2, 2, 2>(a, stream);
  else if (L.cout_w == 64) launch_halo_x3<UMT_X3_CFG64>(a, stream);
  else launch_halo_x3<UMT_X3_CFG32>(a, stream);      // 32 and 96 (the regressor's 76 -> 76 layers: three 32-wide tiles; one
                                                     // 96-wide tile per workgroup, <4, 1, 1, 3>, measured 250 us against 239)
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// every instance that can ask for more than 64 KB of dynamic LDS (called by conv_prepare_device at umt_model_finalize)
template <int WAVES_M, int WAVES_N, int TM, int TN>
static int prepare_halo_x3() {
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s1_halo_x3<WAVES_M, WAVES_N, TM, TN, true>), 160 * 1024 - 256));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s1_halo_x3<WAVES_M, WAVES_N, TM, TN, false>), 160 * 1024 - 256));
  return 0;
}
int conv_x3_prepare_device() {
  UMT_TRY((prepare_halo_x3<2, 2, 2, 2>()));
  UMT_TRY((prepare_halo_x3<UMT_X3_CFG64>()));
  UMT_TRY((prepare_halo_x3<UMT_X3_CFG32>()));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_ds_x3<1, 2, true>), 160 * 1024 - 256));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_ds_x3<1, 2, false>), 160 * 1024 - 256));
  return 0;
}

}  // namespace umt
