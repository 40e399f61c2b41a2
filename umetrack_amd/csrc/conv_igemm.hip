// Implicit-GEMM convolution on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces every Conv2d(+BatchNorm2d)(+residual)(+ReLU) of the reference network
// (lib/models/backbone_resnet.py:56-72, model_utils.py:134,141-163,195-208,
// temporal.py:29-38).  GEMM view:  D[pixel][cout] = sum_k A[pixel][k] * B[k][cout],
// k = (tap, cin).  Pixels are rows of the channels-last activation matrix, so an
// A fragment is one 16-byte load of 4 consecutive input channels and a D row is
// one contiguous 128-byte store of 32 output channels.
//
// MFMA operand map (32x32x2, wave64): lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5),
// column l&31.  One ds_read_b128 per operand feeds four k-steps: lane (i,h) reads
// channels c+4h..c+4h+3, so k-step t contracts channels {c+t, c+4+t}; the packed
// weights use the same order.  fp32 MFMA is an exact k-ordered fmaf chain, so the
// result differs from the reference's CPU convolution only by summation order.
#include "umt_internal.h"
#include "conv_f32_common.h"
#include <stdlib.h>

// experiment switches for tools/conv_bench.hip (all off in the library build)
#ifndef UMT_CFG64
#define UMT_CFG64 2, 2, 2, 1
#endif
// padded cout 96 (fusion 108->72, temporal 90, regressor 76): three 32-wide cout tiles in grid.y rather than
// one 96-wide workgroup tile - 3x the workgroups for the 6x6 maps and 158 instead of 280 registers
#ifndef UMT_CFG96
#define UMT_CFG96 4, 1, 1, 1
#endif
#ifndef UMT_CFG96G
#define UMT_CFG96G 4, 1, 1, 1
#endif
#ifndef UMT_CFG32
#define UMT_CFG32 4, 1, 1, 1
#endif
#ifndef UMT_X_NOBLOAD
#define UMT_X_NOBLOAD 0
#endif
#ifndef UMT_X_NOALOAD
#define UMT_X_NOALOAD 0
#endif
#ifndef UMT_X_NOSTORE
#define UMT_X_NOSTORE 0
#endif

namespace umt {


template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvArgs a) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int A_ROWS = BM / 32;  // float4 loads per thread per step
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroup");
  __shared__ __attribute__((aligned(16))) float smem[BM * LDA + KC * BN];
  float* sA = smem;
  float* sB = smem + BM * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int pad = a.taps == 9 ? 1 : 0;
  const int c8_total = a.cin_s >> 3;

  // A staging: thread -> (pixel row ar + 32*r, channel quad ag)
  const int ar = tid >> 3, ag = tid & 7;
  int yi0[A_ROWS], xi0[A_ROWS];
  int pixbase[A_ROWS];
  bool rowok[A_ROWS];
  const int hw_out = a.Ho * a.Wo;
#pragma unroll
  for (int r = 0; r < A_ROWS; ++r) {
    const int m = m0 + ar + 32 * r;
    rowok[r] = m < a.M;
    const int mm = rowok[r] ? m : 0;
    const int img = mm / hw_out;
    const int rem = mm - img * hw_out;
    const int yo = rem / a.Wo;
    const int xo = rem - yo * a.Wo;
    yi0[r] = yo * a.stride - pad;
    xi0[r] = xo * a.stride - pad;
    pixbase[r] = img * a.H * a.W;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int tap = 0; tap < a.taps; ++tap) {
    const int ky = a.taps == 9 ? tap / 3 : 0;
    const int kx = a.taps == 9 ? tap - 3 * ky : 0;
    for (int c0 = 0; c0 < a.cin_s; c0 += KC) {
      const int kc8 = min(KC / 8, c8_total - (c0 >> 3));
      // ---- stage A: BM pixels x 32 channels of this tap ----
#pragma unroll
      for (int r = 0; r < A_ROWS; ++r) {
        const int yi = yi0[r] + ky, xi = xi0[r] + kx;
        const bool ok = rowok[r] && yi >= 0 && yi < a.H && xi >= 0 && xi < a.W &&
                        (c0 + 4 * ag) < a.cin_s;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok)
          v = *reinterpret_cast<const float4*>(
              a.in + (size_t)(pixbase[r] + yi * a.W + xi) * a.cin_s + c0 + 4 * ag);
        *reinterpret_cast<float4*>(sA + (ar + 32 * r) * LDA + 4 * ag) = v;
      }
      // ---- stage B: kc8 x 2 x BN float4 of packed weights ----
      {
        const float* wsrc = a.w + ((size_t)tap * c8_total + (c0 >> 3)) * 2 * a.cout_w * 4;
        for (int idx = tid; idx < kc8 * 2 * BN; idx += 256) {
          const int row = idx / BN;  // c8l*2 + h
          const int j = idx - row * BN;
          *reinterpret_cast<float4*>(sB + (size_t)idx * 4) =
              *reinterpret_cast<const float4*>(wsrc + ((size_t)row * a.cout_w + n0 + j) * 4);
        }
      }
      __syncthreads();
      for (int c8l = 0; c8l < kc8; ++c8l) {
        float4 av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[i] = *reinterpret_cast<const float4*>(sA + ((wm * TM + i) * 32 + li) * LDA + c8l * 8 + 4 * lh);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[j] = *reinterpret_cast<const float4*>(sB + ((size_t)(c8l * 2 + lh) * BN + (wn * TN + j) * 32 + li) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }

  // ---- epilogue (sA is free: every wave passed the barrier that ended the last MFMA phase) ----
  float* patch = sA + wave * 32 * LDA;
  epilogue<TM, TN, false>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, a);
}


// ---------------------------------------------------------------------------------------------
// Stride-1 3x3 convolutions (21 of the 24 backbone 3x3 layers and the regressor blocks):
// "flattened halo" variant.  With stride 1 the input and output pixel rows share one index
// space, and tap (ky,kx) of output pixel m reads input pixel m + (ky-1)*W + (kx-1) whenever that
// neighbour exists.  So per 32-channel chunk the workgroup stages ONE contiguous range of
// BM + 2W + 2 pixel rows into LDS and all nine taps read it at a shifted row; the zero padding
// at image borders becomes a per-lane 9-bit mask.  That is 9x fewer global loads and barriers for
// the activations than the gather kernel above.  Weights do not go through LDS at all: each wave
// streams its B fragments (already packed in fragment order) from L2 into registers, one tap
// ahead of the MFMAs that consume them.
#ifndef UMT_F32_MINB_1TILE
#define UMT_F32_MINB_1TILE 2
#endif
template <int WAVES_M, int WAVES_N, int TM, int TN, bool RES>
__global__ __launch_bounds__(256, (TM * TN == 1 ? UMT_F32_MINB_1TILE : 2)) void conv3x3s1_halo_f32(ConvArgs a) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroup");
  extern __shared__ __attribute__((aligned(16))) float sA[];   // (BM + 2W + 2) rows x LDA + one zero row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int g0 = m0 - W - 1;             // global pixel row of LDS row 0
  const int c8_total = a.cin_s >> 3;
  const int nchunks = (a.cin_s + KC - 1) / KC;

  // Zero padding at image borders: tap (ky,kx) of a pixel whose neighbour does not exist reads the
  // all-zero LDS row `rows` instead of a shifted row.  The choice is made once per tile, on the
  // address, so the tap loop applies no mask to the operands (no VALU between ds_read and MFMA).
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
        arow[i][ky * 3 + kx] = (ok ? (wm * TM + i) * 32 + li + ky * W + kx : rows) * LDA + 4 * lh;
      }
  }
  if (tid < LDA / 4) *reinterpret_cast<float4*>(sA + rows * LDA + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // B fragment stream: lane (li, lh) reads packed[tap][c8][lh][n0 + col][0..3].  The wave-uniform
  // part of the address stays scalar and the loads are unconditional (a channel group past the
  // end is clamped to the last one: its A operand is zero-filled, so the product vanishes) - a
  // predicated load would make hipcc branch around every load and serialise the prefetch.
  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 4 * sizeof(float));
  const char* wblock = reinterpret_cast<const char*>(a.w + (size_t)n0 * 4);
  const size_t w_c8 = (size_t)2 * a.cout_w * 4 * sizeof(float);   // bytes per c8 group
  const size_t w_tap = (size_t)c8_total * w_c8;                   // bytes per tap
  // buffer loads: resource (base of the packed weights) and the (tap, channel group) offset are scalar, the
  // lane offset is one 32-bit VGPR computed once - no per-load 64-bit VALU address arithmetic
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)(n0 * 16 + wlane_off);
  (void)wblock;
  float4 bcur[4][TN], bnext[4][TN];
  auto load_b = [&](float4 (&dst)[4][TN], int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c8 = min(chunk * 4 + c, c8_total - 1);
      const int soff = (int)(tap * w_tap + c8 * w_c8);            // wave-uniform
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4v v = __builtin_bit_cast(f32x4v, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + j * 512, soff, 0));
        dst[c][j] = make_float4(v.x, v.y, v.z, v.w);
      }
    }
  };
  load_b(bcur, 0, 0);
  // epilogue constants requested now: a one-tile workgroup has nowhere else to hide their latency.  Only the 64 x 64 wave
  // tiles take it (+0.6-1.1 % on the 128/256-channel layers; the 64-channel configuration lost 1.4 % to the registers)
  constexpr bool KPRE = TM * TN == 4;
  float4 kconst[TN][3];
  if (KPRE) load_epilogue_constants<TN>(kconst, lane, n0 + wn * TN * 32, a);

  // A staging: thread t owns LDS slots idx = t + 256*k (row idx>>3, channel quad idx&7).  Loads are
  // unconditional at clamped addresses and zeroed afterwards, so they issue back to back; the range
  // of the NEXT chunk is fetched into registers while this chunk's MFMAs run and written to LDS
  // after the barrier that retires this chunk's reads.
  constexpr int NLD = (BM + 2 * 48 + 2 + 31) / 32;     // W <= 48
  float4 areg[NLD];
  unsigned aok = 0;                                     // bit k: slot k holds real data (else zero)
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
      if (UMT_X_NOALOAD) areg[k] = make_float4(1.f, 2.f, 3.f, 4.f);
      else areg[k] = *reinterpret_cast<const float4*>(a.in + off);   // not touched again until the LDS write
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(0);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + (idx >> 3) * LDA + 4 * (idx & 7)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (!UMT_X_NOBLOAD) {
        if (tap < 8) load_b(bnext, tap + 1, chunk);
        else if (chunk + 1 < nchunks) load_b(bnext, 0, chunk + 1);
      }
      if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads ahead of this tap's MFMAs
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // channel groups past the end (cin_s = 80 of the regressor layers: 2 of the last chunk's 4) hold zeros: the
        // single-tile configuration of the head skips their MFMAs, the backbone layers have none
        if (TM * TN == 1 && chunk * 4 + c >= c8_total) continue;
        float4 av[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(sA + arow[i][tap] + c * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bcur[c][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bcur[c][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bcur[c][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bcur[c][j].w, acc[i][j], 0, 0, 0);
          }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j) bcur[c][j] = bnext[c][j];
    }
    __syncthreads();
  }

  // ---- epilogue ----
  float* patch = sA + wave * 32 * LDA;
  epilogue<TM, TN, RES, KPRE>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, a, kconst);
}

// ---------------------------------------------------------------------------------------------
// First block of a stage (backbone_resnet.py:56-72 with stride 2 and a down-sample branch): the
// stride-2 3x3 conv1 and the stride-2 1x1 down-sample convolution read the same input, and the 1x1's
// operand is exactly the centre tap of the 3x3.  One kernel computes both.  Like the stride-1 halo
// kernel it stages, per 32-channel chunk, ONE contiguous range of input pixel rows - here the rows
// from the first output pixel's top-left tap to the last output pixel's bottom-right tap, i.e. about
// 4 input pixels per output pixel - and every tap reads it through a per-lane row address fixed once
// per tile (the all-zero row where the tap falls outside the image).  The centre tap additionally
// feeds a second set of accumulators with the down-sample weights.  Against the per-tap gather
// kernel this stages A once instead of nine times per chunk, needs 2 instead of 18 barriers per
// chunk, and drops the separate down-sample launch with its re-read of the input.
constexpr int S2_BM = 64;            // output pixels per workgroup: 2 waves along pixels x 2 along cout
constexpr int S2_ROWS_MAX = 448;     // staged input rows (host checks 2W*ceil(63/Wo) + 3W + 1 <= this)


template <int TN, bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3s2_ds_f32(ConvS2Args g) {
  const ConvArgs& a = g.c1;
  constexpr int BM = S2_BM, BN = 2 * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float sA[];   // rows x LDA + one zero row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c8_total = a.cin_s >> 3;
  const int nchunks = (a.cin_s + KC - 1) / KC;

  // centre-tap input pixel of output pixel m: 2W * (global output row) + 2 * xo
  auto centre = [&](int m) {
    const int R = div_magic32(m, Wo, a.mg_w), xo = m - R * Wo;      // for this kernel mg_w / mg_hw divide by Wo / Ho
    return 2 * W * R + 2 * xo;
  };
  const int m_last = min(m0 + BM, a.M) - 1;
  const int g0 = centre(m0) - W - 1;                     // global input pixel row of LDS row 0
  const int rows = centre(m_last) + W + 1 - g0 + 1;      // <= S2_ROWS_MAX (checked on the host)
  const int zrow = S2_ROWS_MAX;                          // the all-zero LDS row

  int arow[9];
  {
    const int m = m0 + wm * 32 + li;
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
        arow[ky * 3 + kx] = (ok ? base + (ky - 1) * W + (kx - 1) : zrow) * LDA + 4 * lh;
      }
  }
  if (tid < LDA / 4) *reinterpret_cast<float4*>(sA + zrow * LDA + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16 acc[1][TN], accd[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][j][r] = 0.f; accd[0][j][r] = 0.f; }

  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 4 * sizeof(float));
  const size_t w_c8 = (size_t)2 * a.cout_w * 4 * sizeof(float);
  const size_t w_tap = (size_t)c8_total * w_c8;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t dblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DS ? g.ds_w : a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)(n0 * 16 + wlane_off);
  float4 bcur[4][TN], bnext[4][TN], bds[4][TN];
  auto load_b = [&](float4 (&dst)[4][TN], __amdgpu_buffer_rsrc_t blk, int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c8 = min(chunk * 4 + c, c8_total - 1);
      const int soff = (int)(tap * w_tap + c8 * w_c8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4v v = __builtin_bit_cast(f32x4v, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(blk, wvoff + j * 512, soff, 0));
        dst[c][j] = make_float4(v.x, v.y, v.z, v.w);
      }
    }
  };
  load_b(bcur, wblock, 0, 0);

  constexpr int NLD = S2_ROWS_MAX / 32;                  // float4 slots per thread
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

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + (idx >> 3) * LDA + 4 * (idx & 7)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) load_b(bnext, wblock, tap + 1, chunk);
      else if (chunk + 1 < nchunks) load_b(bnext, wblock, 0, chunk + 1);
      if (DS && tap == 3) load_b(bds, dblock, 0, chunk);
      if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 av = *reinterpret_cast<const float4*>(sA + arow[tap] + c * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bcur[c][j].x, acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bcur[c][j].y, acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bcur[c][j].z, acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bcur[c][j].w, acc[0][j], 0, 0, 0);
        }
        if (DS && tap == 4) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bds[c][j].x, accd[0][j], 0, 0, 0);
            accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bds[c][j].y, accd[0][j], 0, 0, 0);
            accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bds[c][j].z, accd[0][j], 0, 0, 0);
            accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bds[c][j].w, accd[0][j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j) bcur[c][j] = bnext[c][j];
    }
    __syncthreads();
  }

  float* patch = sA + wave * 32 * LDA;
  epilogue<1, TN, false>(acc, patch, lane, m0 + wm * 32, n0 + wn * TN * 32, a);
  if (DS) {
    ConvArgs d = a;
    d.bias = g.ds_bias; d.scale = g.ds_scale; d.shift = g.ds_shift; d.out = g.ds_out; d.relu = 0; d.residual = nullptr;
    epilogue<1, TN, false>(accd, patch, lane, m0 + wm * 32, n0 + wn * TN * 32, d);
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of conv3x3s2_ds_f32 (round 2).  Same tile (64 output pixels), same staged input range, same k order
// (32-channel chunk, tap, 8-channel group, element; the down-sample product of the centre tap in its own accumulators),
// same epilogue: bit-identical results.  What changes is what happens BETWEEN tiles and chunks:
//   * a workgroup walks a contiguous run of tiles of its XCD (neighbouring workgroups share their overlapping input
//     rows through that XCD's L2) and fetches the NEXT work item's input range - the next chunk or the next tile's first
//     chunk - into registers while the current chunk's MFMAs run.  The 32 -> 64 layer has a single chunk per tile: in
//     the one-tile-per-workgroup kernel every workgroup exposed its own 57 KB fetch (89 TFLOP/s);
//   * the weight stream keeps the whole-tap double buffer (a ring of single (tap, channel group) entries three ahead was
//     1.2-2x SLOWER: a fragment requested after the input-range fetch waits for all of it in the in-order load counter,
//     and three entries of MFMAs do not cover an HBM round trip; a whole tap's 16 MFMAs nearly do).
#ifndef UMT_S2P_ATAP
#define UMT_S2P_ATAP 0
#endif
#ifdef UMT_S2P_STAMP       // diagnostic builds only (tools/s2_stamps.py): s_memtime at the phase boundaries of one tile
__device__ unsigned long long g_s2p_stamp[64];
#define S2PSTAMP(k) do { if (blockIdx.x == 5 && tid == 0 && iter == 1 && a.cin_s == UMT_S2P_STAMP) g_s2p_stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" void umt_s2_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2p_stamp), sizeof(g_s2p_stamp)); }
#else
#define S2PSTAMP(k) do {} while (0)
#endif
template <int TN, bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3s2p_ds_f32(ConvS2Args g, int tiles_m, int tiles_n) {
  const ConvArgs& a = g.c1;
  constexpr int BM = S2_BM, BN = 2 * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float sA[];   // rows x LDA + one zero row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c8_total = a.cin_s >> 3;
  const int nchunks = (a.cin_s + KC - 1) / KC;
  const int zrow = S2_ROWS_MAX;

  // tiles [t_lo, t_hi) of this workgroup's XCD, dealt round-robin to the XCD's workgroups
  const int ntiles = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x + 7 - xcd) >> 3;
  const int t_lo = (int)((int64_t)ntiles * xcd / 8), t_hi = (int)((int64_t)ntiles * (xcd + 1) / 8);

  auto centre = [&](int m) {
    const int Rr = div_magic32(m, Wo, a.mg_w), xo = m - Rr * Wo;      // for this kernel mg_w / mg_hw divide by Wo / Ho
    return 2 * W * Rr + 2 * xo;
  };
  struct Tile { int m0, n0, g0, rows; };
  auto tile_of = [&](int t) {
    Tile T;
    const int tm = t / tiles_n;
    T.m0 = tm * BM;
    T.n0 = (t - tm * tiles_n) * BN;
    const int m_last = min(T.m0 + BM, a.M) - 1;
    T.g0 = centre(T.m0) - W - 1;
    T.rows = centre(m_last) + W + 1 - T.g0 + 1;
    return T;
  };

  if (tid < LDA / 4) *reinterpret_cast<float4*>(sA + zrow * LDA + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
  float* sC = sA + (S2_ROWS_MAX + 1) * LDA;              // [conv1, down-sample][bias, scale, shift][cout_w]
  for (int i = tid; i < a.cout_w; i += 256) {
    const bool in = i < a.cout_s;
    sC[i] = in ? a.bias[i] : 0.f;
    sC[a.cout_w + i] = in ? a.scale[i] : 0.f;
    sC[2 * a.cout_w + i] = in ? a.shift[i] : 0.f;
    if (DS) {
      sC[3 * a.cout_w + i] = in ? g.ds_bias[i] : 0.f;
      sC[4 * a.cout_w + i] = in ? g.ds_scale[i] : 0.f;
      sC[5 * a.cout_w + i] = in ? g.ds_shift[i] : 0.f;
    }
  }

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t dblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DS ? g.ds_w : a.w), 0, 0x7fffffff, 0x00020000);
  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 4 * sizeof(float));
  const size_t w_c8 = (size_t)2 * a.cout_w * 4 * sizeof(float);
  const size_t w_tap = (size_t)c8_total * w_c8;

  // weight fragments: whole-tap double buffer as in conv3x3s2_ds_f32 (a tap's 16 x TN MFMAs cover the next tap's loads
  // even when they queue behind the input-range fetch in the in-order load counter; a finer ring does not)
  float4 bcur[4][TN], bnext[4][TN], bds[4][TN];
  auto load_b = [&](float4 (&dst)[4][TN], __amdgpu_buffer_rsrc_t blk, int tap, int chunk, int voff) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c8 = min(chunk * 4 + c, c8_total - 1);
      const int soff = (int)(tap * w_tap + c8 * w_c8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4v v = __builtin_bit_cast(f32x4v, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(blk, voff + j * 512, soff, 0));
        dst[c][j] = make_float4(v.x, v.y, v.z, v.w);
      }
    }
  };

  constexpr int NLD = S2_ROWS_MAX / 32;                  // float4 slots per thread
  float4 areg[NLD];
  unsigned aok = 0;
  int arows = 0;                                         // rows of the range held in areg
  auto load_a = [&](const Tile& T, int chunk) {
    const int c0 = chunk * KC;
    aok = 0;
    arows = T.rows;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 3, q = idx & 7;
      const int gp = T.g0 + r;
      const bool ok = r < T.rows && gp >= 0 && (c0 + 4 * q) < a.cin_s;      // gp < n*H*W follows from r < rows
      const size_t off = ok ? (size_t)gp * a.cin_s + c0 + 4 * q : 0;
      areg[k] = *reinterpret_cast<const float4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };

  int t = t_lo + wg_in_xcd;
  if (t >= t_hi) return;
  Tile cur = tile_of(t);
  load_a(cur, 0);
  int voff = (int)(cur.n0 * 16 + wlane_off);
  load_b(bcur, wblock, 0, 0, voff);

  for (int iter = 0; t < t_hi; t += wgs_per_xcd, ++iter) {
    S2PSTAMP(0);
#ifdef UMT_S2P_STAMP
    if (blockIdx.x == 5 && tid == 0 && iter < 31 && a.cin_s == UMT_S2P_STAMP) { g_s2p_stamp[32 + iter] = __builtin_amdgcn_s_memtime(); g_s2p_stamp[33 + iter] = 0; }
#endif
    const bool has_next = t + wgs_per_xcd < t_hi;
    Tile nxt = cur;
    if (has_next) nxt = tile_of(t + wgs_per_xcd);
    const int voff_next = (int)(nxt.n0 * 16 + wlane_off);

    int arow[9];
    {
      const int m = cur.m0 + wm * 32 + li;
      const bool live = m < a.M;
      const int mm = live ? m : cur.m0;
      const int Rr = div_magic32(mm, Wo, a.mg_w), xo = mm - Rr * Wo, yo = Rr - div_magic32(Rr, Ho, a.mg_hw) * Ho;
      const int base = 2 * W * Rr + 2 * xo - cur.g0;            // LDS row of the centre tap
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
          const bool ok = live && yy >= 0 && yy < a.H && xx >= 0 && xx < W;
          arow[ky * 3 + kx] = (ok ? base + (ky - 1) * W + (kx - 1) : zrow) * LDA + 4 * lh;
        }
    }
    f32x16 acc[1][TN], accd[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][j][r] = 0.f; accd[0][j][r] = 0.f; }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k;
        float4 v = areg[k];
        if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((idx >> 3) < arows) *reinterpret_cast<float4*>(sA + (idx >> 3) * LDA + 4 * (idx & 7)) = v;
      }
      S2PSTAMP(1 + 4 * chunk);
      __syncthreads();
      S2PSTAMP(2 + 4 * chunk);
      const bool last_chunk = chunk + 1 == nchunks;
      // the A fragment of a step (tap, 8-channel group) is read one step ahead of its MFMAs
      float4 avr[2];
      avr[0] = *reinterpret_cast<const float4*>(sA + arow[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) load_b(bnext, wblock, tap + 1, chunk, voff);
        else if (!last_chunk) load_b(bnext, wblock, 0, chunk + 1, voff);
        else if (has_next) load_b(bnext, wblock, 0, 0, voff_next);
        if (DS && tap == 3) load_b(bds, dblock, 0, chunk, voff);
        if (tap == UMT_S2P_ATAP && !last_chunk) load_a(cur, chunk + 1);      // the next chunk's input range
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int st = tap * 4 + c;
          if (st + 1 < 36) avr[(st + 1) & 1] = *reinterpret_cast<const float4*>(sA + arow[(st + 1) >> 2] + ((st + 1) & 3) * 8);
          const float4 av = avr[st & 1];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bcur[c][j].x, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bcur[c][j].y, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bcur[c][j].z, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bcur[c][j].w, acc[0][j], 0, 0, 0);
          }
          if (DS && tap == 4) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bds[c][j].x, accd[0][j], 0, 0, 0);
              accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bds[c][j].y, accd[0][j], 0, 0, 0);
              accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bds[c][j].z, accd[0][j], 0, 0, 0);
              accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bds[c][j].w, accd[0][j], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < TN; ++j) bcur[c][j] = bnext[c][j];
      }
      S2PSTAMP(3 + 4 * chunk);
      __syncthreads();
      S2PSTAMP(4 + 4 * chunk);
    }
    // the next tile's input range is requested here, behind the last weight load of this tile (a weight fragment requested
    // after it would wait for all of it in the in-order load counter), and arrives under the two epilogues
    if (has_next) load_a(nxt, 0);
    // (phase stamps of a 32 -> 64 tile, tools/s2_stamps.py on a -DUMT_S2P_STAMP=32 build: 30.5 k ticks of which rows
    // written 3.8 k, the nine taps 11.1 k = the MFMA time, next rows requested 2.8 k, the two epilogues 3.8 + 8 k - the
    // stores of 32 KB per tile wait on a memory system that all 512 workgroups load at 3 TB/s; starting the second
    // workgroup of every CU half a tile late changed nothing)
    S2PSTAMP(17);

    // the BN constants come from LDS: a global load here would wait for the input range just requested (a wave's loads
    // return in order) and put an HBM round trip between the last MFMA and the first store of every tile
    float* patch = sA + wave * 32 * LDA;
    const float4 no_res[4] = {};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cobase = cur.n0 + wn * TN * 32 + j * 32;
      const int co = min(cobase + 4 * (lane & 7), a.cout_w - 4);
      const float4 kc[3] = {*reinterpret_cast<const float4*>(sC + co), *reinterpret_cast<const float4*>(sC + a.cout_w + co),
                            *reinterpret_cast<const float4*>(sC + 2 * a.cout_w + co)};
      store_tile<false, true>(acc[0][j], no_res, patch, lane, cur.m0 + wm * 32, cobase, a, kc);
    }
    S2PSTAMP(18);
    if (DS) {
      ConvArgs d = a;
      d.out = g.ds_out; d.relu = 0; d.residual = nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cobase = cur.n0 + wn * TN * 32 + j * 32;
        const int co = min(cobase + 4 * (lane & 7), a.cout_w - 4);
        const float4 kc[3] = {*reinterpret_cast<const float4*>(sC + 3 * a.cout_w + co), *reinterpret_cast<const float4*>(sC + 4 * a.cout_w + co),
                              *reinterpret_cast<const float4*>(sC + 5 * a.cout_w + co)};
        store_tile<false, true>(accd[0][j], no_res, patch, lane, cur.m0 + wm * 32, cobase, d, kc);
      }
    }
    S2PSTAMP(20);
    __syncthreads();          // the patches alias the staging area the next tile is about to fill
    S2PSTAMP(21);
    cur = nxt;
    voff = voff_next;
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_halo(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  dim3 grid((a.M + BM - 1) / BM, a.cout_w / BN);
  const size_t lds = (size_t)(BM + 2 * a.W + 2 + 1) * LDA * sizeof(float);     // + the all-zero row
  if (a.residual)
    hipLaunchKernelGGL((conv3x3s1_halo_f32<WAVES_M, WAVES_N, TM, TN, true>), grid, dim3(256), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3x3s1_halo_f32<WAVES_M, WAVES_N, TM, TN, false>), grid, dim3(256), lds, stream, a);
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_cfg(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  dim3 grid((a.M + BM - 1) / BM, a.cout_w / BN);
  hipLaunchKernelGGL((conv_igemm_f32<WAVES_M, WAVES_N, TM, TN>), grid, dim3(256), 0, stream, a);
}

bool conv_small_applicable(const ConvArgs& a, bool force_off);      // conv_small.hip
void launch_conv_small(const ConvArgs& a, hipStream_t stream);
bool conv_s2_small_applicable(const ConvS2Args& g, bool force_off);
void launch_conv_s2_small(const ConvS2Args& g, hipStream_t stream);

int launch_conv(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual,
                float* out, hipStream_t stream) {
  UMT_REQUIRE(L.taps == 1 || L.taps == 9, UMT_ERR_INVALID, "conv: taps must be 1 or 9, got %d", L.taps);
  UMT_REQUIRE(L.stride == 1 || L.stride == 2, UMT_ERR_INVALID, "conv: stride %d", L.stride);
  UMT_REQUIRE(L.cin_s % 8 == 0 && L.cout_s % 8 == 0 && L.cout_w % 32 == 0 && L.cout_s <= L.cout_w,
              UMT_ERR_INVALID, "conv: channel strides %d/%d/%d", L.cin_s, L.cout_s, L.cout_w);
  UMT_REQUIRE(n > 0 && h > 0 && w > 0, UMT_ERR_INVALID, "conv: empty input");
  if (conv_thin_f32_applicable(L, w) && (int64_t)n * h * w >= 4 * 256 * 128)   // one tile per wave group and CU at least
    return launch_conv_thin_f32(L, in, n, h, w, residual, out, stream);
  ConvArgs a;
  a.in = in; a.w = L.w; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
  a.residual = residual; a.out = out;
  a.H = h; a.W = w;
  const int pad = L.taps == 9 ? 1 : 0, k = L.taps == 9 ? 3 : 1;
  a.Ho = (h + 2 * pad - k) / L.stride + 1;
  a.Wo = (w + 2 * pad - k) / L.stride + 1;
  a.cin_s = L.cin_s; a.cout_s = L.cout_s; a.cout_w = L.cout_w;
  a.stride = L.stride; a.taps = L.taps; a.relu = L.relu;
  const int64_t M = (int64_t)n * a.Ho * a.Wo;
  UMT_REQUIRE(M < (int64_t)1 << 31 && (int64_t)n * h * w < (int64_t)1 << 31, UMT_ERR_INVALID,
              "conv: %lld output pixels exceed the 32-bit pixel index", (long long)M);
  a.M = (int)M;
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
  const bool force_gather = (L.flags & CONV_GATHER) != 0;
  if (L.taps == 9 && L.stride == 1 && w <= 48 && !force_gather) {
    // Few pixels (the tracker's per-frame call: 4 crops -> 144 pixels at 6x6): the wide tiles would leave 4-18
    // workgroups on 256 CUs, each walking its whole K loop with 2-4 MFMAs per k-step.  conv_small.hip runs these
    // launches on 16x16x4 tiles (16 pixels x 16 couts per wave: a sixteenth of the K-loop latency of the wide
    // configuration); with UMT_CONV_NOSMALL one 32x32 tile per wave.  The k order per output is the same in all
    // three, so results are bit-identical.
    const int64_t wide_wgs = ((M + 127) / 128) * (L.cout_w % 128 == 0 ? L.cout_w / 128 : 1);
    const bool few = wide_wgs < 64 && (L.cout_w % 128 == 0 || L.cout_w == 64 || L.cout_w == 96);
    if (few && conv_small_applicable(a, (L.flags & CONV_NO_SMALL) != 0)) launch_conv_small(a, stream);   // 16x16x4 tiles (conv_small.hip)
    else if (few) launch_halo<4, 1, 1, 1>(a, stream);
    else if (L.cout_w % 128 == 0) launch_halo<2, 2, 2, 2>(a, stream);
    else if (L.cout_w == 96) launch_halo<UMT_CFG96>(a, stream);
    else if (L.cout_w == 64) launch_halo<UMT_CFG64>(a, stream);
    else if (L.cout_w == 32) launch_halo<UMT_CFG32>(a, stream);
    else return set_error(UMT_ERR_INVALID, "conv: unsupported padded cout %d", L.cout_w);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (L.taps == 1 && L.stride == 1 && !force_gather && M <= 2048 && conv_small_applicable(a, (L.flags & CONV_NO_SMALL) != 0)) {
    launch_conv_small(a, stream);      // few pixels (the per-frame call's projection and last regressor layer): 16x16x4 tiles
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (L.cout_w % 128 == 0) launch_cfg<2, 2, 2, 2>(a, stream);
  else if (L.cout_w == 96) launch_cfg<UMT_CFG96G>(a, stream);
  else if (L.cout_w == 64) launch_cfg<2, 2, 2, 1>(a, stream);
  else if (L.cout_w == 32) launch_cfg<4, 1, 1, 1>(a, stream);
  else return set_error(UMT_ERR_INVALID, "conv: unsupported padded cout %d", L.cout_w);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv_thin_prepare_device();
int conv_x3_prepare_device();
int conv_chain_prepare_device();

unsigned conv_flags_from_env() {
  static const struct { const char* name; unsigned bit; } kSwitches[] = {
      {"UMT_CONV_NOTHIN", CONV_NO_THIN},         {"UMT_CONV_NOFUSE", CONV_NO_FUSE}, {"UMT_CONV_NOSTREAM128", CONV_NO_STREAM128},
      {"UMT_CONV_NOTHIN_F32", CONV_NO_THIN_F32}, {"UMT_CONV_GATHER", CONV_GATHER},  {"UMT_CONV_NOS2", CONV_NO_S2},
      {"UMT_X3_NOS2", CONV_X3_NO_S2},            {"UMT_CONV_NOCHAIN", CONV_NO_CHAIN}, {"UMT_CONV_NOS2P", CONV_NO_S2P},
      {"UMT_CONV_NOPIPE32", CONV_NO_PIPE32},      {"UMT_CONV_NOPIPE64", CONV_NO_PIPE64},
      {"UMT_CONV_NODEPTH", CONV_NO_DEPTH},        {"UMT_CONV_NOSMALL", CONV_NO_SMALL},
      {"UMT_STEM_VALU", CONV_STEM_VALU},          {"UMT_CONV_NOWALK", CONV_NO_WALK},
      {"UMT_SEQ_ONEKERNEL", SEQ_ONE_KERNEL},      {"UMT_HEAD_NOX3", HEAD_NO_X3},
      {"UMT_STEM_NOSPLIT", STEM_NO_SPLIT}};
  unsigned f = 0;
  for (const auto& s : kSwitches)
    if (getenv(s.name)) f |= s.bit;
  return f;
}

constexpr size_t S2P_CONST_BYTES_MAX = 6 * 512 * sizeof(float);      // BN constants of both convolutions, cout_w <= 512

int conv_prepare_device() {
  const size_t lds = (size_t)(S2_ROWS_MAX + 1) * LDA * sizeof(float);
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_ds_f32<1, false>), lds));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_ds_f32<1, true>), lds));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2p_ds_f32<1, false>), lds + S2P_CONST_BYTES_MAX));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2p_ds_f32<1, true>), lds + S2P_CONST_BYTES_MAX));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2p_ds_f32<2, false>), lds + S2P_CONST_BYTES_MAX));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2p_ds_f32<2, true>), lds + S2P_CONST_BYTES_MAX));
  UMT_TRY(conv_x3_prepare_device());
  UMT_TRY(conv_chain_prepare_device());
  UMT_TRY(stage1_depth_prepare_device());
  return conv_thin_prepare_device();
}

// conv1 (3x3, stride 2) of a stage's first block, fused with its down-sample 1x1 (ds may be null).
bool conv_s2_fused_applicable(const ConvLayer& c1, const ConvLayer* ds, int h, int w) {
  if ((c1.flags & CONV_NO_S2) || c1.taps != 9 || c1.stride != 2 || (h & 1) || (w & 1) || c1.cout_w % 64 != 0 || c1.cin_s % 8 != 0) return false;
  const int wo = w / 2;
  if (2 * w * ((S2_BM - 1 + wo - 1) / wo) + 3 * w + 1 > S2_ROWS_MAX) return false;
  if (ds && (ds->taps != 1 || ds->stride != 2 || ds->cin_s != c1.cin_s || ds->cout_w != c1.cout_w || ds->cout_s != c1.cout_s))
    return false;
  return true;
}

int launch_conv_s2_fused(const ConvLayer& c1, const ConvLayer* ds, const float* in, int n, int h, int w, float* out,
                         float* ds_out, hipStream_t stream) {
  UMT_REQUIRE(conv_s2_fused_applicable(c1, ds, h, w), UMT_ERR_INVALID, "conv_s2: layer shapes not supported");
  UMT_REQUIRE(n > 0 && (!ds || ds_out), UMT_ERR_INVALID, "conv_s2: bad arguments");
  ConvS2Args g;
  ConvArgs& a = g.c1;
  a.in = in; a.w = c1.w; a.bias = c1.bias; a.scale = c1.scale; a.shift = c1.shift; a.residual = nullptr; a.out = out;
  a.H = h; a.W = w; a.Ho = h / 2; a.Wo = w / 2;
  a.cin_s = c1.cin_s; a.cout_s = c1.cout_s; a.cout_w = c1.cout_w; a.stride = 2; a.taps = 9; a.relu = c1.relu;
  const int64_t M = (int64_t)n * a.Ho * a.Wo;
  UMT_REQUIRE((int64_t)n * h * w < ((int64_t)1 << 31) - 65536, UMT_ERR_INVALID, "conv_s2: too many pixels");
  a.M = (int)M;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Wo) + 1u;      // this kernel divides by Wo and Ho
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Ho) + 1u;
  g.ds_w = ds ? ds->w : nullptr; g.ds_bias = ds ? ds->bias : nullptr; g.ds_scale = ds ? ds->scale : nullptr;
  g.ds_shift = ds ? ds->shift : nullptr; g.ds_out = ds_out;
  const size_t lds = (size_t)(S2_ROWS_MAX + 1) * LDA * sizeof(float);
  // 64 x 64 workgroup tiles (one 32x32 tile per wave and convolution): wider cout tiles would need the
  // weight double buffers of both convolutions for two tiles and spill
  if (conv_s2_small_applicable(g, (c1.flags & CONV_NO_SMALL) != 0)) {      // few pixels: 16x16x4 tiles (conv_small.hip)
    launch_conv_s2_small(g, stream);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (!(c1.flags & CONV_NO_S2P) && c1.cout_w <= 512) {
    // persistent kernel: 64 x 128 tiles where cout allows, two workgroups per CU walking their XCD's tiles
    const bool wide = false;      // TN = 2 needs both whole-tap buffers for two tiles: spills (measured 2x slower)
    const int tiles_m = (a.M + S2_BM - 1) / S2_BM, tiles_n = c1.cout_w / (wide ? 128 : 64);
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    const unsigned wgs = (unsigned)(ntiles < 512 ? (ntiles + 7) / 8 * 8 : 512);
    void (*pk)(ConvS2Args, int, int) = wide ? (ds ? conv3x3s2p_ds_f32<2, true> : conv3x3s2p_ds_f32<2, false>)
                                            : (ds ? conv3x3s2p_ds_f32<1, true> : conv3x3s2p_ds_f32<1, false>);
    hipLaunchKernelGGL(pk, dim3(wgs), dim3(256), lds + (size_t)6 * c1.cout_w * sizeof(float), stream, g, tiles_m, tiles_n);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 grid((a.M + S2_BM - 1) / S2_BM, c1.cout_w / 64);
  void (*kern)(ConvS2Args) = ds ? conv3x3s2_ds_f32<1, true> : conv3x3s2_ds_f32<1, false>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace umt
