// bf16 variant of the backbone convolutions (UMT_PREC_BF16): bf16 activations and weights,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 epilogue (conv bias -> BN -> residual ->
// ReLU) rounded once to bf16.  Same two kernel structures as conv_igemm.hip:
//   * conv_gather_bf16      - any 1x1 / 3x3, stride 1 / 2 (operands staged per tap),
//   * conv3x3s1_halo_bf16   - stride-1 3x3 with one flattened halo range of pixel rows in LDS
//                             per 32-channel chunk and weights streamed through registers.
// Operand map of the 32x32x16 instruction: lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j]
// and B[k = 8h+j][col r], j = 0..7, i.e. one 16-byte load of 8 consecutive channels per operand;
// the packed weights are [tap][cin/16][h][cout32][8] so a B fragment is one 16-byte load too.
// This path trades the reference's fp32 arithmetic for ~8 bits of mantissa per activation; it is
// reported separately from the fp32 parity mode (tests/test_gpu_bf16.py states the tolerance).
#include "umt_internal.h"

#ifndef UMT_CFG128H
#define UMT_CFG128H 2, 2, 2, 2
#endif
#ifndef UMT_CFG64H
#define UMT_CFG64H 2, 2, 2, 1
#endif

namespace umt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgsH {
  const uint16_t* in;
  const uint16_t* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const uint16_t* residual;
  void* out;
  int out_f32;
  int H, W, Ho, Wo;
  int cin_s, cout_s, cout_w;
  int stride, taps, relu;
  int M;
  unsigned mg_hw, mg_w;   // floor(2^32 / (H*W)) + 1 and floor(2^32 / W) + 1: division by multiplication in the kernels
};

// floor(m / d) for 0 <= m < 2^31 and 1 <= d <= 65536 with mg = floor(2^32 / d) + 1: the estimate is exact or one too high
__device__ __forceinline__ int div_magic(int m, int d, unsigned mg) {
  int q = (int)__umulhi((unsigned)m, mg);
  if (q * d > m) --q;
  return q;
}

constexpr int KCH = 32;          // channels per staged chunk
constexpr int LDH = 40;          // bf16 per LDS row (80 B): 16 consecutive rows hit 16 distinct 16-B slots
constexpr int LDP = 36;          // floats per row of the fp32 epilogue patch

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// Epilogue.  The residual rows of ALL the wave's tiles are requested first (one exposed memory latency per
// workgroup instead of one per tile: the residual is the block input, long evicted from L2 when conv2
// ends), then each accumulator tile goes through the wave's LDS patch (transpose) so that a lane owns 8
// consecutive channels of a pixel row - 16 bytes of bf16, the widest store: conv bias -> BN scale/shift ->
// residual -> ReLU -> one rounding.  Four lanes complete a 64-byte row, 16 rows per pass, two passes per tile.
template <int TM, int TN>
__device__ __forceinline__ void load_residual_h(uint4 (&res)[TN][TM][2], int lane, int mwave, int cowave,
                                                const ConvArgsH& a) {
  const int q = lane & 3, rsub = lane >> 2;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int m = mwave + i * 32 + 16 * k + rsub;
        const int co = cowave + j * 32 + 8 * q;
        const bool ok = a.residual && m < a.M && co < a.cout_s;
        const size_t o = ok ? (size_t)m * a.cout_s + co : 0;
        res[j][i][k] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) res[j][i][k] = *reinterpret_cast<const uint4*>(a.residual + o);
      }
}

// max(v, 0) as one integer max on the bits (negative floats are negative integers; -0 -> +0)
__device__ __forceinline__ float relu1(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

__device__ __forceinline__ void affine4(float4& v, const float4& cb, const float4& cs, const float4& ch) {
  v.x = (v.x + cb.x) * cs.x + ch.x;
  v.y = (v.y + cb.y) * cs.y + ch.y;
  v.z = (v.z + cb.z) * cs.z + ch.z;
  v.w = (v.w + cb.w) * cs.w + ch.w;
}

template <bool PRE>
__device__ __forceinline__ void store_tile_h(const f32x16& acc, const uint4 (&res)[2], float* patch, int lane, int mbase,
                                             int cobase, const ConvArgsH& a) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDP + li] = acc[r];
  __builtin_amdgcn_wave_barrier();
  const int q = lane & 3, rsub = lane >> 2;
  const int co = cobase + 8 * q;
  if (co < a.cout_s) {
    const float4 cb0 = *reinterpret_cast<const float4*>(a.bias + co), cb1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    const float4 cs0 = *reinterpret_cast<const float4*>(a.scale + co), cs1 = *reinterpret_cast<const float4*>(a.scale + co + 4);
    const float4 ch0 = *reinterpret_cast<const float4*>(a.shift + co), ch1 = *reinterpret_cast<const float4*>(a.shift + co + 4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = 16 * k + rsub;
      const int m = mbase + row;
      if (m < a.M) {
        float4 v = *reinterpret_cast<const float4*>(patch + row * LDP + 8 * q);
        float4 u = *reinterpret_cast<const float4*>(patch + row * LDP + 8 * q + 4);
        const size_t o = (size_t)m * a.cout_s + co;
        affine4(v, cb0, cs0, ch0);
        affine4(u, cb1, cs1, ch1);
        if (PRE || a.residual) {
          const uint4 rr = PRE ? res[k] : *reinterpret_cast<const uint4*>(a.residual + o);
          v.x += bf16_lo(rr.x); v.y += bf16_hi(rr.x); v.z += bf16_lo(rr.y); v.w += bf16_hi(rr.y);
          u.x += bf16_lo(rr.z); u.y += bf16_hi(rr.z); u.z += bf16_lo(rr.w); u.w += bf16_hi(rr.w);
        }
        if (a.relu) {
          v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w);
          u.x = relu1(u.x); u.y = relu1(u.y); u.z = relu1(u.z); u.w = relu1(u.w);
        }
        if (a.out_f32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) = v;
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o + 4) = u;
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + o) =
              make_uint4(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w), pack_bf16x2(u.x, u.y), pack_bf16x2(u.z, u.w));
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// PRE: the kernel instance is the one launched for layers WITH a residual (prefetch into registers);
// otherwise a residual, if any, is read tile by tile.
template <int TM, int TN, bool PRE>
__device__ __forceinline__ void epilogue_h(const f32x16 (&acc)[TM][TN], float* patch, int lane, int mwave, int cowave,
                                           const ConvArgsH& a) {
  uint4 res[TN][TM][2];
  if (PRE) load_residual_h<TM, TN>(res, lane, mwave, cowave, a);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
      store_tile_h<PRE>(acc[i][j], res[j][i], patch, lane, mwave + i * 32, cowave + j * 32, a);
}

// ------------------------------------------------------------------------------------- gather
template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(256) void conv_gather_bf16(ConvArgsH a) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int A_SLOTS = BM * 4 / 256;       // 16-B slots per thread per step
  constexpr int A_BYTES = BM * LDH * 2;
  constexpr int B_BYTES = 4 * BN * 16;        // 2 k16 groups x 2 halves x BN x 16 B
  constexpr int P_BYTES = 4 * 32 * LDP * 4;
  constexpr int SMEM = (A_BYTES + B_BYTES) > P_BYTES ? (A_BYTES + B_BYTES) : P_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* sA = smem;
  unsigned char* sB = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int pad = a.taps == 9 ? 1 : 0;
  const int c16_total = a.cin_s >> 4;

  const int ar = tid >> 2, aq = tid & 3;
  int yi0[A_SLOTS], xi0[A_SLOTS], pixbase[A_SLOTS];
  bool rowok[A_SLOTS];
  const int hw_out = a.Ho * a.Wo;
#pragma unroll
  for (int r = 0; r < A_SLOTS; ++r) {
    const int m = m0 + ar + 64 * r;
    rowok[r] = m < a.M;
    const int mm = rowok[r] ? m : 0;
    const int img = mm / hw_out;
    const int rem = mm - img * hw_out;
    const int yo = rem / a.Wo, xo = rem - yo * a.Wo;
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
    for (int c0 = 0; c0 < a.cin_s; c0 += KCH) {
      const int kc16 = min(2, c16_total - (c0 >> 4));
#pragma unroll
      for (int r = 0; r < A_SLOTS; ++r) {
        const int yi = yi0[r] + ky, xi = xi0[r] + kx;
        const bool ok = rowok[r] && yi >= 0 && yi < a.H && xi >= 0 && xi < a.W && (c0 + 8 * aq) < a.cin_s;
        const size_t off = ok ? (size_t)(pixbase[r] + yi * a.W + xi) * a.cin_s + c0 + 8 * aq : 0;
        uint4 v = *reinterpret_cast<const uint4*>(a.in + off);
        if (!ok) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(sA + ((ar + 64 * r) * LDH + 8 * aq) * 2) = v;
      }
      {
        const uint16_t* wsrc = a.w + ((size_t)tap * c16_total + (c0 >> 4)) * 2 * a.cout_w * 8;
        for (int idx = tid; idx < kc16 * 2 * BN; idx += 256) {
          const int row = idx / BN;  // k16l*2 + h
          const int j = idx - row * BN;
          *reinterpret_cast<uint4*>(sB + (size_t)idx * 16) =
              *reinterpret_cast<const uint4*>(wsrc + ((size_t)row * a.cout_w + n0 + j) * 8);
        }
      }
      __syncthreads();
      for (int k = 0; k < kc16; ++k) {
        bf16x8 av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[i] = *reinterpret_cast<const bf16x8*>(sA + (((wm * TM + i) * 32 + li) * LDH + k * 16 + 8 * lh) * 2);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[j] = *reinterpret_cast<const bf16x8*>(sB + ((size_t)(k * 2 + lh) * BN + (wn * TN + j) * 32 + li) * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float* patch = reinterpret_cast<float*>(smem) + wave * 32 * LDP;
  epilogue_h<TM, TN, false>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, a);
}

// ------------------------------------------------------------------------------------- halo
// Two-tile waves (the 64-channel layers) fit 4 waves per SIMD when asked to (126 instead of 132-136 registers, no
// spills: -6 % on those layers); the four-tile waves of the wider layers spill below 184 registers and are left alone.
#ifndef UMT_HALO_MINB_SMALL
#define UMT_HALO_MINB_SMALL 3
#endif
template <int WAVES_M, int WAVES_N, int TM, int TN, bool RES>
__global__ __launch_bounds__(256, (TM * TN <= 2 ? UMT_HALO_MINB_SMALL : 1)) void conv3x3s1_halo_bf16(ConvArgsH a) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char sAh[];   // max(rows*80, 4*32*36*4) bytes

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
  const int nchunks = (a.cin_s + KCH - 1) / KCH;

  // Zero padding at image borders: a tap whose neighbour does not exist reads the all-zero LDS row
  // `rows`; the choice is made once per tile on the address, so the tap loop applies no mask.
  int arow[TM][9];        // byte offset of the lane's operand row per tap
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + li;
    const bool live = m < a.M;
    const int mm = live ? m : 0;
    const int rem = mm - div_magic(mm, H * W, a.mg_hw) * (H * W);
    const int y = div_magic(rem, W, a.mg_w), x = rem - y * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        arow[i][ky * 3 + kx] = ((ok ? (wm * TM + i) * 32 + li + ky * W + kx : rows) * LDH + 8 * lh) * 2;
      }
  }
  if (tid < LDH * 2 / 16) *reinterpret_cast<uint4*>(sAh + rows * LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 16);
  const char* wblock = reinterpret_cast<const char*>(a.w + (size_t)n0 * 8);
  const size_t w_c16 = (size_t)2 * a.cout_w * 16;       // bytes per 16-channel group
  const size_t w_tap = (size_t)c16_total * w_c16;
  // Weight fragments run ahead of their MFMAs through a register ring of three slots.  Two-tile waves (64-channel
  // layers): a slot is a TAP (9 per chunk, so a tap's slot is static), the fragments of tap t+2 are requested while
  // tap t runs.  Four-tile waves (128/256-channel layers): a slot is a K-STEP (one 16-channel group of one tap, 18
  // per chunk), requested two k-steps = 8 MFMAs ahead: half the ring registers, 164 instead of 184 in all, which
  // is a third wave per SIMD (-7 % on those layers; the two-tile kernel measured no gain from it).
  constexpr bool KRING = TM * TN >= 4;
  uint4 bring[KRING ? 1 : 3][2][TN];
  uint4 bk[3][TN];
  // buffer loads: resource (base of the packed weights) and the (tap, channel group) offset are scalar, the
  // lane offset is one 32-bit VGPR computed once - no per-load 64-bit VALU address arithmetic
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)(n0 * 16 + wlane_off);
  auto load_b = [&](uint4 (&dst)[2][TN], int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int c16 = min(chunk * 2 + c, c16_total - 1);
      const int soff = (int)(tap * w_tap + c16 * w_c16);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + j * 512, soff, 0);
        dst[c][j] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  };
  auto load_bk = [&](uint4 (&dst)[TN], int tap, int c16raw) {
    const int c16 = min(c16raw, c16_total - 1);
    const int soff = (int)(tap * w_tap + c16 * w_c16);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + j * 512, soff, 0);
      dst[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  if constexpr (KRING) {
    load_bk(bk[0], 0, 0);
    load_bk(bk[1], 0, 1);
  } else {
    load_b(bring[0], 0, 0);
    load_b(bring[1], 1, 0);
  }

  constexpr int NLD = ((BM + 2 * 48 + 2) * 4 + 255) / 256;
  uint4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int chunk) {
    const int c0 = chunk * KCH;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 2, q = idx & 3;
      const int g = g0 + r;
      const bool ok = r < rows && g >= 0 && g < a.M && (c0 + 8 * q) < a.cin_s;
      const size_t off = ok ? (size_t)g * a.cin_s + c0 + 8 * q : 0;
      areg[k] = *reinterpret_cast<const uint4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(0);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      uint4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
      if ((idx >> 2) < rows) *reinterpret_cast<uint4*>(sAh + ((idx >> 2) * LDH + 8 * (idx & 3)) * 2) = v;
    }
    __syncthreads();
    if constexpr (KRING) {
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int nk = ks + 2;
      if (nk < 18) load_bk(bk[nk % 3], nk >> 1, chunk * 2 + (nk & 1));
      else if (chunk + 1 < nchunks) load_bk(bk[nk % 3], (nk - 18) >> 1, (chunk + 1) * 2 + (nk & 1));
      if (ks == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 av[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sAh + arow[i][ks >> 1] + (ks & 1) * 32));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], __builtin_bit_cast(bf16x8, bk[ks % 3][j]), acc[i][j], 0, 0, 0);
    }
    } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 7) load_b(bring[(tap + 2) % (KRING ? 1 : 3)], tap + 2, chunk);
      else if (chunk + 1 < nchunks) load_b(bring[(tap + 2) % (KRING ? 1 : 3)], tap - 7, chunk + 1);
      if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 av[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sAh + arow[i][tap] + c * 32));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], __builtin_bit_cast(bf16x8, bring[tap % (KRING ? 1 : 3)][c][j]), acc[i][j], 0, 0, 0);
      }
    }
    }
    __syncthreads();
  }
  float* patch = reinterpret_cast<float*>(sAh) + wave * 32 * LDP;
  epilogue_h<TM, TN, RES>(acc, patch, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, a);
}

// ------------------------------------------------------------------------------ stride 2 + down-sample
// bf16 form of conv3x3s2_ds_f32 (conv_igemm.hip): conv1 (3x3, stride 2) of a stage's first block and its
// 1x1 stride-2 down-sample branch in one kernel; the input range of a 64-pixel output tile is staged once
// per 32-channel chunk and the centre tap feeds both accumulator sets.
constexpr int S2H_BM = 64;
constexpr int S2H_ROWS_MAX = 448;

struct ConvS2ArgsH {
  ConvArgsH c1;
  const uint16_t* ds_w;
  const float* ds_bias;
  const float* ds_scale;
  const float* ds_shift;
  uint16_t* ds_out;
};

template <int TN, bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3s2_ds_bf16(ConvS2ArgsH g) {
  const ConvArgsH& a = g.c1;
  constexpr int BM = S2H_BM, BN = 2 * TN * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char sAh[];   // rows x 80 B + one zero row (or the patches)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c16_total = a.cin_s >> 4;
  const int nchunks = (a.cin_s + KCH - 1) / KCH;

  auto centre = [&](int m) {
    const int R = div_magic(m, Wo, a.mg_w), xo = m - R * Wo;        // for this kernel mg_w / mg_hw divide by Wo / Ho
    return 2 * W * R + 2 * xo;
  };
  const int m_last = min(m0 + BM, a.M) - 1;
  const int g0 = centre(m0) - W - 1;
  const int rows = centre(m_last) + W + 1 - g0 + 1;
  const int zrow = S2H_ROWS_MAX;

  int arow[9];
  {
    const int m = m0 + wm * 32 + li;
    const bool live = m < a.M;
    const int mm = live ? m : m0;
    const int R = div_magic(mm, Wo, a.mg_w), xo = mm - R * Wo, yo = R - div_magic(R, Ho, a.mg_hw) * Ho;
    const int base = 2 * W * R + 2 * xo - g0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
        const bool ok = live && yy >= 0 && yy < a.H && xx >= 0 && xx < W;
        arow[ky * 3 + kx] = ((ok ? base + (ky - 1) * W + (kx - 1) : zrow) * LDH + 8 * lh) * 2;
      }
  }
  if (tid < LDH * 2 / 16) *reinterpret_cast<uint4*>(sAh + zrow * LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);

  f32x16 acc[1][TN], accd[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][j][r] = 0.f; accd[0][j][r] = 0.f; }

  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 16);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t dblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(DS ? g.ds_w : a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)(n0 * 16 + wlane_off);
  const size_t w_c16 = (size_t)2 * a.cout_w * 16;
  const size_t w_tap = (size_t)c16_total * w_c16;
  uint4 bring[3][2][TN], bds[2][TN];
  auto load_b = [&](uint4 (&dst)[2][TN], __amdgpu_buffer_rsrc_t blk, int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int c16 = min(chunk * 2 + c, c16_total - 1);
      const int soff = (int)(tap * w_tap + c16 * w_c16);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(blk, wvoff + j * 512, soff, 0);
        dst[c][j] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  };
  load_b(bring[0], wblock, 0, 0);
  load_b(bring[1], wblock, 1, 0);

  constexpr int NLD = S2H_ROWS_MAX * 4 / 256;
  uint4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int chunk) {
    const int c0 = chunk * KCH;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 2, q = idx & 3;
      const int gp = g0 + r;
      const bool ok = r < rows && gp >= 0 && (c0 + 8 * q) < a.cin_s;
      const size_t off = ok ? (size_t)gp * a.cin_s + c0 + 8 * q : 0;
      areg[k] = *reinterpret_cast<const uint4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(0);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      uint4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
      if ((idx >> 2) < rows) *reinterpret_cast<uint4*>(sAh + ((idx >> 2) * LDH + 8 * (idx & 3)) * 2) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 7) load_b(bring[(tap + 2) % 3], wblock, tap + 2, chunk);
      else if (chunk + 1 < nchunks) load_b(bring[(tap + 2) % 3], wblock, tap - 7, chunk + 1);
      if (DS && tap == 2) load_b(bds, dblock, 0, chunk);
      if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sAh + arow[tap] + c * 32));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bring[tap % 3][c][j]), acc[0][j], 0, 0, 0);
        if (DS && tap == 4) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            accd[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bds[c][j]), accd[0][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  float* patch = reinterpret_cast<float*>(sAh) + wave * 32 * LDP;
  epilogue_h<1, TN, false>(acc, patch, lane, m0 + wm * 32, n0 + wn * TN * 32, a);
  if (DS) {
    ConvArgsH d = a;
    d.bias = g.ds_bias; d.scale = g.ds_scale; d.shift = g.ds_shift; d.out = g.ds_out; d.relu = 0; d.residual = nullptr;
    epilogue_h<1, TN, false>(accd, patch, lane, m0 + wm * 32, n0 + wn * TN * 32, d);
  }
}

// The 64 -> 128 layer as a WALK: two workgroups per CU, each taking tiles t, t + grid, ... with the next tile's input
// rows already in flight.  conv3x3s2_ds_bf16 gives every 64-pixel tile a workgroup of its own, whose 80 MFMAs per wave
// (2.6 k cycles) follow an HBM round trip for the first chunk and most of another for the second, and whose epilogue's
// constant loads queue behind them: 25 k ticks per tile, the matrix pipe 21 % busy (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES;
// phase stamps: tools/s2_stamps.py on a -DUMT_S2_STAMP=64 build).  What the stamps led to, in order:
//  * a deeper weight ring (9 k-step slots, 8 ahead) changed nothing: the waits were not for weights;
//  * a wave's loads return in order, so a weight fragment requested after a row load waits for HBM (chunk 0's nine taps:
//    7.3 k ticks against 1.3 k of MFMA) and so do BN constants loaded in the epilogue (10.6 k ticks): the constants sit
//    in LDS, and the rows of the next tile are requested after this tile's last weight load, in front of the epilogue
//    (which loads nothing): taps 3.3 + 2.7 k, epilogue 5 k;
//  * A fragments read one k-step ahead of their MFMAs: taps 2.6 + 2.5 k.
// Measured per 4096 crops: 245 -> 225 us for this layer.  The same walk with four chunks per tile measured 216 us
// against 202 us for the 128 -> 256 layer (chunks 2 and 3 have no slot in front of an epilogue) and is not dispatched.
// Both layers move 0.6 / 0.3 GB of activations through HBM per launch (input once, two outputs): ~110 / 55 us at the
// rate this chip sustains, so what is left here is overlap of the epilogue (6.7 k of a tile's 16.7 k ticks) with the
// other workgroup's MFMAs, not arithmetic.  Same k order and epilogue arithmetic: bit-identical results.
#ifdef UMT_S2_STAMP        // diagnostic builds only (tools/s2_stamps.py): s_memtime at the phase boundaries of one tile
__device__ unsigned long long g_s2_stamp[64];     // [0..31] phases of one tile, [32..63] start of every tile of the workgroup
#define S2STAMP_TILE() do { if (blockIdx.x == 5 && tid == 0 && iter < 31 && a.cin_s == UMT_S2_STAMP) { g_s2_stamp[32 + iter] = __builtin_amdgcn_s_memtime(); g_s2_stamp[33 + iter] = 0; } } while (0)
#define S2STAMP(k) do { if (blockIdx.x == 5 && tid == 0 && iter == 1 && a.cin_s == UMT_S2_STAMP) g_s2_stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2STAMP(k) do {} while (0)
#define S2STAMP_TILE() do {} while (0)
#endif
// epilogue of one 32 x 32 accumulator tile with the BN constants in LDS (no global load: a wave's loads return in order,
// and one issued here would wait for the next tile's input rows already in flight)
__device__ __forceinline__ void walk_store_tile(const f32x16& acc, float* patch, int lane, int mbase, int cobase, int M, int cout_s,
                                                const float* cB, const float* cS, const float* cH, int relu, uint16_t* out) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDP + li] = acc[r];
  __builtin_amdgcn_wave_barrier();
  const int q = lane & 3, rsub = lane >> 2;
  const int co = cobase + 8 * q;
  if (co < cout_s) {
    const float4 cb0 = *reinterpret_cast<const float4*>(cB + co), cb1 = *reinterpret_cast<const float4*>(cB + co + 4);
    const float4 cs0 = *reinterpret_cast<const float4*>(cS + co), cs1 = *reinterpret_cast<const float4*>(cS + co + 4);
    const float4 ch0 = *reinterpret_cast<const float4*>(cH + co), ch1 = *reinterpret_cast<const float4*>(cH + co + 4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = 16 * k + rsub;
      const int m = mbase + row;
      if (m < M) {
        float4 v = *reinterpret_cast<const float4*>(patch + row * LDP + 8 * q);
        float4 u = *reinterpret_cast<const float4*>(patch + row * LDP + 8 * q + 4);
        affine4(v, cb0, cs0, ch0);
        affine4(u, cb1, cs1, ch1);
        if (relu) {
          v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w);
          u.x = relu1(u.x); u.y = relu1(u.y); u.z = relu1(u.z); u.w = relu1(u.w);
        }
        *reinterpret_cast<uint4*>(out + (size_t)m * cout_s + co) =
            make_uint4(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w), pack_bf16x2(u.x, u.y), pack_bf16x2(u.z, u.w));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

constexpr int S2W_ROW_BYTES = (S2H_ROWS_MAX + 1) * LDH * 2;      // staged rows + the zero row; the BN constants follow

template <int TN, bool DS, int NCH>
__global__ __launch_bounds__(256, 2) void conv3x3s2_walk_bf16(ConvS2ArgsH g, int total, int ntn_shift) {
  const ConvArgsH& a = g.c1;
  constexpr int BM = S2H_BM, BN = 2 * TN * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char sAh[];   // rows x 80 B + one zero row (or the patches), constants
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c16_total = a.cin_s >> 4;
  constexpr int nchunks = NCH;                                          // cin_s / 32 (host-checked)
  const int zrow = S2H_ROWS_MAX;
  int t = blockIdx.x;
  if (t >= total) return;

  auto centre = [&](int m) {
    const int R = div_magic(m, Wo, a.mg_w), xo = m - R * Wo;
    return 2 * W * R + 2 * xo;
  };
  auto geom = [&](int tt, int& m0, int& n0, int& g0, int& rows) {
    const int tile_m = tt >> ntn_shift, tile_n = tt - (tile_m << ntn_shift);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
    const int m_last = min(m0 + BM, a.M) - 1;
    g0 = centre(m0) - W - 1;
    rows = centre(m_last) + W + 1 - g0 + 1;
  };
  if (tid < LDH * 2 / 16) *reinterpret_cast<uint4*>(sAh + zrow * LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
  float* sC = reinterpret_cast<float*>(sAh + S2W_ROW_BYTES);            // [c1, ds][bias, scale, shift][cout_w]
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

  const unsigned wlane_off = (unsigned)(((size_t)lh * a.cout_w + wn * TN * 32 + li) * 16);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t dblock = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(DS ? g.ds_w : a.w), 0, 0x7fffffff, 0x00020000);
  const size_t w_c16 = (size_t)2 * a.cout_w * 16;
  const size_t w_tap = (size_t)c16_total * w_c16;
  uint4 bring[3][2][TN], bds[2][TN];
  auto load_b = [&](uint4 (&dst)[2][TN], __amdgpu_buffer_rsrc_t blk, int tap, int chunk, int voff) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int c16 = min(chunk * 2 + c, c16_total - 1);
      const int soff = (int)(tap * w_tap + c16 * w_c16);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(blk, voff + j * 512, soff, 0);
        dst[c][j] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  };

  constexpr int NLD = S2H_ROWS_MAX * 4 / 256;
  uint4 ar0[NLD], ar1[NLD];
  unsigned ok0 = 0, ok1 = 0;
  auto load_a = [&](uint4 (&dst)[NLD], unsigned& okm, int g0, int rows, int chunk) {   // rows = 0: nothing to fetch
    const int c0 = chunk * KCH;
    okm = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 2, q = idx & 3;
      const int gp = g0 + r;
      const bool ok = r < rows && gp >= 0 && (c0 + 8 * q) < a.cin_s;
      const size_t off = ok ? (size_t)gp * a.cin_s + c0 + 8 * q : 0;
      dst[k] = *reinterpret_cast<const uint4*>(a.in + off);
      okm |= (ok ? 1u : 0u) << k;
    }
  };

  int m0, n0, g0, rows;
  geom(t, m0, n0, g0, rows);
  int wvoff = (int)(n0 * 16 + wlane_off);
  load_b(bring[0], wblock, 0, 0, wvoff);
  load_b(bring[1], wblock, 1, 0, wvoff);
  load_a(ar0, ok0, g0, rows, 0);
  load_a(ar1, ok1, g0, rows, 1);

  for (int iter = 0; t < total; t += gridDim.x, ++iter) {
    S2STAMP(0);
    S2STAMP_TILE();
    int arow[9];
    {
      const int m = m0 + wm * 32 + li;
      const bool live = m < a.M;
      const int mm = live ? m : m0;
      const int R = div_magic(mm, Wo, a.mg_w), xo = mm - R * Wo, yo = R - div_magic(R, Ho, a.mg_hw) * Ho;
      const int base = 2 * W * R + 2 * xo - g0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
          const bool ok = live && yy >= 0 && yy < a.H && xx >= 0 && xx < W;
          arow[ky * 3 + kx] = ((ok ? base + (ky - 1) * W + (kx - 1) : zrow) * LDH + 8 * lh) * 2;
        }
    }
    f32x16 acc[TN], accd[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accd[j][r] = 0.f; }

    const int tn = t + (int)gridDim.x;
    const bool has_next = tn < total;
    int m0n = m0, n0n = n0, g0n = g0, rowsn = 0;
    if (has_next) geom(tn, m0n, n0n, g0n, rowsn);
    const int wvoff_n = (int)(n0n * 16 + wlane_off);

    // Where the input rows are requested: the NEXT tile's two chunks after this tile's last weight load, in front of
    // the epilogue.  (NCH = 4, measured and not dispatched: chunks 2 and 3 are requested as soon as their register set
    // is free, and the weights behind them wait once.)
    auto step = [&](uint4 (&ar)[NLD], unsigned& okm, int chunk) {
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k;
        uint4 v = ar[k];
        if (!((okm >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
        if ((idx >> 2) < rows) *reinterpret_cast<uint4*>(sAh + ((idx >> 2) * LDH + 8 * (idx & 3)) * 2) = v;
      }
      S2STAMP(1 + 4 * chunk);
      __syncthreads();
      S2STAMP(2 + 4 * chunk);
      if (chunk + 2 < nchunks) load_a(ar, okm, g0, rows, chunk + 2);       // (compile-time condition)
      const bool last = chunk + 1 >= nchunks;
      // the A fragment of a k-step (tap, 16-channel group) is read one step ahead of its MFMAs
      uint4 av[2];
      av[0] = *reinterpret_cast<const uint4*>(sAh + arow[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < 7) load_b(bring[(tap + 2) % 3], wblock, tap + 2, chunk, wvoff);
        else load_b(bring[(tap + 2) % 3], wblock, tap - 7, last ? 0 : chunk + 1, last ? wvoff_n : wvoff);
        if (DS && tap == 2) load_b(bds, dblock, 0, chunk, wvoff);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ks = tap * 2 + c;
          if (ks + 1 < 18) av[(ks + 1) & 1] = *reinterpret_cast<const uint4*>(sAh + arow[(ks + 1) >> 1] + ((ks + 1) & 1) * 32);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 a8 = __builtin_bit_cast(bf16x8, av[ks & 1]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, bring[tap % 3][c][j]), acc[j], 0, 0, 0);
          if (DS && tap == 4) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
              accd[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, bds[c][j]), accd[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      S2STAMP(3 + 4 * chunk);
      __syncthreads();
      S2STAMP(4 + 4 * chunk);
    };
#pragma unroll
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      step(ar0, ok0, chunk);
      step(ar1, ok1, chunk + 1);
    }
    load_a(ar0, ok0, g0n, rowsn, 0);       // rowsn = 0 after the last tile: nothing is fetched
    load_a(ar1, ok1, g0n, rowsn, 1);
    float* patch = reinterpret_cast<float*>(sAh) + wave * 32 * LDP;
    const int cw = a.cout_w;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      walk_store_tile(acc[j], patch, lane, m0 + wm * 32, n0 + wn * TN * 32 + j * 32, a.M, a.cout_s, sC, sC + cw, sC + 2 * cw, a.relu,
                      reinterpret_cast<uint16_t*>(a.out));
    if (DS) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        walk_store_tile(accd[j], patch, lane, m0 + wm * 32, n0 + wn * TN * 32 + j * 32, a.M, a.cout_s, sC + 3 * cw, sC + 4 * cw,
                        sC + 5 * cw, 0, g.ds_out);
    }
    S2STAMP(20);
    __syncthreads();                 // the patches alias the staged rows
    S2STAMP(21);
    m0 = m0n; n0 = n0n; g0 = g0n; rows = rowsn; wvoff = wvoff_n;
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_gather_h(const ConvArgsH& a, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  dim3 grid((a.M + BM - 1) / BM, a.cout_w / BN);
  hipLaunchKernelGGL((conv_gather_bf16<WAVES_M, WAVES_N, TM, TN>), grid, dim3(256), 0, stream, a);
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_halo_h(const ConvArgsH& a, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  dim3 grid((a.M + BM - 1) / BM, a.cout_w / BN);
  size_t lds = (size_t)(BM + 2 * a.W + 2 + 1) * LDH * 2;      // + the all-zero row
  const size_t patch = (size_t)4 * 32 * LDP * sizeof(float);
  if (lds < patch) lds = patch;
  if (a.residual)
    hipLaunchKernelGGL((conv3x3s1_halo_bf16<WAVES_M, WAVES_N, TM, TN, true>), grid, dim3(256), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3x3s1_halo_bf16<WAVES_M, WAVES_N, TM, TN, false>), grid, dim3(256), lds, stream, a);
}

int launch_conv_bf16(const ConvLayer& L, const uint16_t* in, int n, int h, int w, const uint16_t* residual, void* out,
                     int out_f32, hipStream_t stream) {
  UMT_REQUIRE(L.wh, UMT_ERR_STATE, "conv_bf16: layer has no bf16 weights");
  UMT_REQUIRE(L.taps == 1 || L.taps == 9, UMT_ERR_INVALID, "conv_bf16: taps %d", L.taps);
  UMT_REQUIRE(L.cin_s % 16 == 0 && L.cout_s % 8 == 0 && L.cout_w % 32 == 0, UMT_ERR_INVALID,
              "conv_bf16: channel strides %d/%d/%d", L.cin_s, L.cout_s, L.cout_w);
  if (!out_f32 && conv_thin_applicable(L, w))
    return launch_conv_thin_bf16(L, in, n, h, w, residual, reinterpret_cast<uint16_t*>(out), stream);
  ConvArgsH a;
  a.in = in; a.w = L.wh; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
  a.residual = residual; a.out = out; a.out_f32 = out_f32;
  a.H = h; a.W = w;
  const int pad = L.taps == 9 ? 1 : 0, k = L.taps == 9 ? 3 : 1;
  a.Ho = (h + 2 * pad - k) / L.stride + 1;
  a.Wo = (w + 2 * pad - k) / L.stride + 1;
  a.cin_s = L.cin_s; a.cout_s = L.cout_s; a.cout_w = L.cout_w;
  a.stride = L.stride; a.taps = L.taps; a.relu = L.relu;
  const int64_t M = (int64_t)n * a.Ho * a.Wo;
  UMT_REQUIRE(M < (int64_t)1 << 31 && (int64_t)n * h * w < (int64_t)1 << 31, UMT_ERR_INVALID, "conv_bf16: too many pixels");
  a.M = (int)M;
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
  const bool halo = L.taps == 9 && L.stride == 1 && w <= 48;
  if (L.cout_w % 128 == 0) halo ? launch_halo_h<UMT_CFG128H>(a, stream) : launch_gather_h<2, 2, 2, 2>(a, stream);
  else if (L.cout_w == 96) halo ? launch_halo_h<4, 1, 1, 3>(a, stream) : launch_gather_h<4, 1, 1, 3>(a, stream);
  else if (L.cout_w == 64) halo ? launch_halo_h<UMT_CFG64H>(a, stream) : launch_gather_h<2, 2, 2, 1>(a, stream);
  else if (L.cout_w == 32) halo ? launch_halo_h<4, 1, 1, 1>(a, stream) : launch_gather_h<4, 1, 1, 1>(a, stream);
  else return set_error(UMT_ERR_INVALID, "conv_bf16: unsupported padded cout %d", L.cout_w);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef UMT_S2_STAMP
extern "C" void umt_s2_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2_stamp), sizeof(g_s2_stamp)); }
#endif

int launch_conv_s2_fused_bf16(const ConvLayer& c1, const ConvLayer* ds, const uint16_t* in, int n, int h, int w,
                              uint16_t* out, uint16_t* ds_out, hipStream_t stream) {
  UMT_REQUIRE(conv_s2_fused_applicable(c1, ds, h, w) && c1.cin_s % 16 == 0, UMT_ERR_INVALID, "conv_s2_bf16: layer shapes not supported");
  UMT_REQUIRE(c1.wh && (!ds || (ds->wh && ds_out)) && n > 0, UMT_ERR_INVALID, "conv_s2_bf16: bad arguments");
  ConvS2ArgsH g;
  ConvArgsH& a = g.c1;
  a.in = in; a.w = c1.wh; a.bias = c1.bias; a.scale = c1.scale; a.shift = c1.shift; a.residual = nullptr; a.out = out;
  a.out_f32 = 0;
  a.H = h; a.W = w; a.Ho = h / 2; a.Wo = w / 2;
  a.cin_s = c1.cin_s; a.cout_s = c1.cout_s; a.cout_w = c1.cout_w; a.stride = 2; a.taps = 9; a.relu = c1.relu;
  UMT_REQUIRE((int64_t)n * h * w < ((int64_t)1 << 31) - 65536, UMT_ERR_INVALID, "conv_s2_bf16: too many pixels");
  a.M = n * a.Ho * a.Wo;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Wo) + 1u;      // this kernel divides by Wo and Ho
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.Ho) + 1u;
  g.ds_w = ds ? ds->wh : nullptr; g.ds_bias = ds ? ds->bias : nullptr; g.ds_scale = ds ? ds->scale : nullptr;
  g.ds_shift = ds ? ds->shift : nullptr; g.ds_out = ds_out;
  size_t lds = (size_t)(S2H_ROWS_MAX + 1) * LDH * 2;
  const size_t patch = (size_t)4 * 32 * LDP * sizeof(float);
  if (lds < patch) lds = patch;
  const bool wide = c1.cout_w % 128 == 0;
  dim3 grid((a.M + S2H_BM - 1) / S2H_BM, c1.cout_w / (wide ? 128 : 64));
  const bool no_walk = (c1.flags & CONV_NO_WALK) != 0;
  if (wide && !no_walk && !a.out_f32 && a.cin_s == 64 && (grid.y & (grid.y - 1)) == 0 && (int64_t)grid.x * grid.y > 512) {
    int shift = 0;
    while ((1u << shift) < grid.y) ++shift;
    const int total = (int)(grid.x * grid.y);
    const size_t lds_w = (size_t)S2W_ROW_BYTES + (size_t)6 * a.cout_w * sizeof(float);
    void (*kw)(ConvS2ArgsH, int, int) = ds ? conv3x3s2_walk_bf16<2, true, 2> : conv3x3s2_walk_bf16<2, false, 2>;
    hipLaunchKernelGGL(kw, dim3(512), dim3(256), lds_w, stream, g, total, shift);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  void (*kern)(ConvS2ArgsH) = wide ? (ds ? conv3x3s2_ds_bf16<2, true> : conv3x3s2_ds_bf16<2, false>)
                                   : (ds ? conv3x3s2_ds_bf16<1, true> : conv3x3s2_ds_bf16<1, false>);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace umt
