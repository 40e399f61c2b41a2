// Shared by the kernels that write fp32 pixel rows (conv_igemm.hip, conv_x3.hip): the argument block, the
// multiply-high division and the transposing epilogue.
#pragma once
#include "umt_internal.h"

#ifndef UMT_X_NOSTORE
#define UMT_X_NOSTORE 0
#endif

namespace umt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* in;
  const float* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  float* out;
  int H, W, Ho, Wo;
  int cin_s, cout_s, cout_w;
  int stride, taps, relu;
  int M;  // n * Ho * Wo output pixels
  unsigned mg_hw, mg_w;   // floor(2^32 / (H*W)) + 1 and floor(2^32 / W) + 1 (division by multiplication)
};

// fused stride-2 conv1 + down-sample launches (conv3x3s2_ds_f32, conv3x3s2p_ds_f32, conv3x3s2_small_f32)
struct ConvS2Args {
  ConvArgs c1;        // conv1: in, w, epilogue constants, out; H, W, Ho, Wo, cin_s, cout_s, cout_w, relu, M
  const float* ds_w;  // down-sample: packed [1][cin_s/8][2][cout_w][4], or null
  const float* ds_bias;
  const float* ds_scale;
  const float* ds_shift;
  float* ds_out;
};

// floor(m / d) for 0 <= m < 2^31 and 1 <= d <= 65536 with mg = floor(2^32 / d) + 1: the estimate is exact or one too high
__device__ __forceinline__ int div_magic32(int m, int d, unsigned mg) {
  int q = (int)__umulhi((unsigned)m, mg);
  if (q * d > m) --q;
  return q;
}

constexpr int KC = 32;        // input channels staged per step
constexpr int LDA = KC + 4;   // padded A row (floats): 16 consecutive pixels hit 16 distinct 16-B slots


// Epilogue shared by both kernels.  An accumulator tile has its 32 output channels on the lanes,
// so storing it directly costs sixteen 4-byte-per-lane instructions per tile (and as many loads
// for the residual).  Instead each wave transposes the tile through its private 32x36-float LDS
// patch and then moves whole 128-B rows with 16 bytes per lane: conv bias -> BN scale/shift ->
// residual -> ReLU are applied on float4s.  Caller guarantees all waves are past their last LDS read.
// PRE (kernel instances launched for layers WITH a residual): the residual rows of all the wave's
// tiles are requested before the first transpose, so a workgroup exposes one memory latency, not one
// per tile (the residual is the block input, long evicted from L2 when conv2 ends).
// KPRE: the tile's bias / scale / shift quads were read at kernel start (kc[0..2]) instead of here, where their global-memory
// latency would sit between the last MFMA and the first store of a one-tile workgroup.
template <bool PRE, bool KPRE = false>
__device__ __forceinline__ void store_tile(const f32x16& acc, const float4 (&res)[4], float* patch, int lane, int mbase,
                                           int cobase, const ConvArgs& a, const float4* kc = nullptr) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDA + li] = acc[r];
  __builtin_amdgcn_wave_barrier();
  const int q = lane & 7, rsub = lane >> 3;
  const int co = cobase + 4 * q;
  if (co < a.cout_s) {
    const float4 cb = KPRE ? kc[0] : *reinterpret_cast<const float4*>(a.bias + co);
    const float4 cs = KPRE ? kc[1] : *reinterpret_cast<const float4*>(a.scale + co);
    const float4 ch = KPRE ? kc[2] : *reinterpret_cast<const float4*>(a.shift + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 8 * k + rsub;
      const int m = mbase + row;
      if (m < a.M) {
        float4 v = *reinterpret_cast<const float4*>(patch + row * LDA + 4 * q);
        const size_t o = (size_t)m * a.cout_s + co;
        v.x = (v.x + cb.x) * cs.x + ch.x;
        v.y = (v.y + cb.y) * cs.y + ch.y;
        v.z = (v.z + cb.z) * cs.z + ch.z;
        v.w = (v.w + cb.w) * cs.w + ch.w;
        if (PRE || a.residual) {
          const float4 rr = PRE ? res[k] : *reinterpret_cast<const float4*>(a.residual + o);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (a.relu) {      // max(v, 0) as one integer max on the bits (negative floats are negative integers; -0 -> +0)
          v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
          v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
          v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
          v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
        }
        if (!UMT_X_NOSTORE || v.x == 123.456f) *reinterpret_cast<float4*>(a.out + o) = v;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// constants of the wave's TN cout tiles for store_tile<., true>: lane (q = lane & 7) owns channels cowave + 32 j + 4 q
template <int TN>
__device__ __forceinline__ void load_epilogue_constants(float4 (&kc)[TN][3], int lane, int cowave, const ConvArgs& a) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = min(cowave + j * 32 + 4 * (lane & 7), a.cout_w - 4);     // in range; unused where co >= cout_s
    kc[j][0] = *reinterpret_cast<const float4*>(a.bias + co);
    kc[j][1] = *reinterpret_cast<const float4*>(a.scale + co);
    kc[j][2] = *reinterpret_cast<const float4*>(a.shift + co);
  }
}

template <int TM, int TN, bool PRE, bool KPRE = false>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TM][TN], float* patch, int lane, int mwave, int cowave,
                                         const ConvArgs& a, const float4 (*kc)[3] = nullptr) {
  float4 res[TN][TM][4];
  if (PRE) {
    const int q = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = mwave + i * 32 + 8 * k + rsub;
          const int co = cowave + j * 32 + 4 * q;
          res[j][i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < a.M && co < a.cout_s) res[j][i][k] = *reinterpret_cast<const float4*>(a.residual + (size_t)m * a.cout_s + co);
        }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
      store_tile<PRE, KPRE>(acc[i][j], res[j][i], patch, lane, mwave + i * 32, cowave + j * 32, a, KPRE ? kc[j] : nullptr);
}

}  // namespace umt
