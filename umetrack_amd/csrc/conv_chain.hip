// Three consecutive 1x1 convolutions (+ bias, BatchNorm, ReLU) in one kernel, fp32 matrix pipe.
//
// The head has two such chains on its 6x6 maps: view fusion 144 -> 108 -> 72 -> 72 (lib/models/model_utils.py:195-208)
// and the temporal module 90 -> 90 -> 90 -> 90 (lib/models/temporal.py:29-38).  As three launches of conv_igemm_f32 each
// layer reads its input rows and writes its output rows through HBM with two barriers per 32-channel chunk, and runs
// at 1.7 TB/s and a quarter of the matrix pipe.  Here a wave owns 32 pixels and keeps them in registers through all
// three layers:
//   * the MFMA operands are SWAPPED (weights as the row operand, pixels as the column operand), so the accumulator of a
//     layer is its output TRANSPOSED: lane (pixel j, half h) holds, in registers 4g'..4g'+3 of cout tile t, the
//     channels 32t + 8g' + 4h + {0..3} of pixel j - which is exactly the column operand of the NEXT layer's k-steps
//     for channel group g = 4t + g' (lane (j, h) supplies channels 8g + 4h + e).  After bias/BN/ReLU in place, the
//     accumulator registers ARE the next layer's operand: no LDS transposition, no HBM round trip;
//   * the three packed weight tensors (as model.hip packs them for conv_igemm_f32: [group][half][cout][4]) and the
//     epilogue constants sit in LDS for the lifetime of the persistent workgroup; a weight fragment is one
//     conflict-free ds_read_b128 per four MFMAs;
//   * the input rows are read once (one 16-byte load per lane and channel group, all issued up front), the output rows
//     written once.
// k order (8-channel group, then element; the two halves of a group inside one MFMA) and the epilogue arithmetic are
// conv_igemm_f32's, and fp32 MFMA products commute, so the results are bit-identical to the three launches
// (tests/test_gpu_edges.py holds them so through UMT_CONV_NOCHAIN).
#include "umt_internal.h"
#include "conv_f32_common.h"

namespace umt {

struct Chain3Args {
  const float* in;           // rows [M][cin_s0]
  float* out;                // rows [M][cout_s2]
  const float* w[3];         // packed [cin_s/8][2][cout_w][4]
  const float* bias[3];
  const float* scale[3];
  const float* shift[3];
  int relu[3];
  int M, n_wave_tiles;       // pixels, ceil(M / 32)
  int cin_s0, cout_s2;
};

constexpr int CHAIN_THREADS = 512;

template <int KG, int NT>
constexpr int chain_w_floats() { return KG * 2 * NT * 32 * 4; }

// one layer on a wave's 32 pixels: acc[t] (cout tile t) += W * operand; operand group g is opnd(g) -> 4 values of this lane
template <int KG, int NT, typename Opnd>
__device__ __forceinline__ void chain_layer(const float* sW, int li, int lh, Opnd opnd, f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // weight fragments one channel group ahead of their MFMAs, pinned there: left alone, the scheduler hoists all KG * NT
  // LDS reads to the top of the layer and spills 400 registers
  float4 wcur[NT], wnext[NT];
  auto read_w = [&](float4 (&dst)[NT], int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      dst[t] = *reinterpret_cast<const float4*>(sW + (((size_t)g * 2 + lh) * (NT * 32) + t * 32 + li) * 4);
  };
  read_w(wcur, 0);
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (g + 1 < KG) read_w(wnext, g + 1);
    const float4 x = opnd(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[t].x, x.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[t].y, x.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[t].z, x.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[t].w, x.w, acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) wcur[t] = wnext[t];
  }
}

// conv bias -> BN scale/shift -> ReLU on the transposed accumulator (the arithmetic of store_tile in
// conv_f32_common.h); sC = [bias | scale | shift], each NT*32 floats
template <int NT>
__device__ __forceinline__ void chain_epilogue(const float* sC, int lh, int relu, const f32x16 (&acc)[NT], float4 (&out)[NT * 4]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co = t * 32 + 8 * gq + 4 * lh;
      const float4 cb = *reinterpret_cast<const float4*>(sC + co);
      const float4 cs = *reinterpret_cast<const float4*>(sC + NT * 32 + co);
      const float4 ch = *reinterpret_cast<const float4*>(sC + 2 * NT * 32 + co);
      float4 v = make_float4(acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]);
      v.x = (v.x + cb.x) * cs.x + ch.x;
      v.y = (v.y + cb.y) * cs.y + ch.y;
      v.z = (v.z + cb.z) * cs.z + ch.z;
      v.w = (v.w + cb.w) * cs.w + ch.w;
      if (relu) {      // max(v, 0) as one integer max on the bits, as store_tile
        v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
        v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
        v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
        v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
      }
      out[t * 4 + gq] = v;                    // channels 32t + 8gq + 4lh + {0..3}: operand group 4t + gq of the next layer
      __builtin_amdgcn_sched_barrier(0);      // the constants of one channel quad at a time (else 36 float4 are read ahead)
    }
}

// K0G: input channel groups of 8 read from memory; NiT: 32-wide cout tiles of layer i; KiG (i > 0): channel groups of
// layer i's input (<= 4 * N(i-1)T)
template <int K0G, int N0T, int K1G, int N1T, int K2G, int N2T>
__global__ __launch_bounds__(CHAIN_THREADS, 1) void conv1x1_chain3_f32(Chain3Args a) {
  static_assert(K1G <= 4 * N0T && K2G <= 4 * N1T, "a layer cannot read more channels than its predecessor produced");
  extern __shared__ __attribute__((aligned(16))) float smem_c[];
  constexpr int W0 = chain_w_floats<K0G, N0T>(), W1 = chain_w_floats<K1G, N1T>(), W2 = chain_w_floats<K2G, N2T>();
  float* sW0 = smem_c;
  float* sW1 = sW0 + W0;
  float* sW2 = sW1 + W1;
  float* sC0 = sW2 + W2;                 // [3][N0T * 32]
  float* sC1 = sC0 + 3 * N0T * 32;
  float* sC2 = sC1 + 3 * N1T * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  for (int i = tid; i < W0 / 4; i += CHAIN_THREADS) reinterpret_cast<float4*>(sW0)[i] = reinterpret_cast<const float4*>(a.w[0])[i];
  for (int i = tid; i < W1 / 4; i += CHAIN_THREADS) reinterpret_cast<float4*>(sW1)[i] = reinterpret_cast<const float4*>(a.w[1])[i];
  for (int i = tid; i < W2 / 4; i += CHAIN_THREADS) reinterpret_cast<float4*>(sW2)[i] = reinterpret_cast<const float4*>(a.w[2])[i];
  for (int i = tid; i < N0T * 32; i += CHAIN_THREADS) { sC0[i] = a.bias[0][i]; sC0[N0T * 32 + i] = a.scale[0][i]; sC0[2 * N0T * 32 + i] = a.shift[0][i]; }
  for (int i = tid; i < N1T * 32; i += CHAIN_THREADS) { sC1[i] = a.bias[1][i]; sC1[N1T * 32 + i] = a.scale[1][i]; sC1[2 * N1T * 32 + i] = a.shift[1][i]; }
  for (int i = tid; i < N2T * 32; i += CHAIN_THREADS) { sC2[i] = a.bias[2][i]; sC2[N2T * 32 + i] = a.scale[2][i]; sC2[2 * N2T * 32 + i] = a.shift[2][i]; }
  __syncthreads();

  const int waves_total = gridDim.x * (CHAIN_THREADS / 64);
  for (int wt = blockIdx.x * (CHAIN_THREADS / 64) + wave; wt < a.n_wave_tiles; wt += waves_total) {
    const int m = wt * 32 + li;
    const bool live = m < a.M;
    const float* row = a.in + (size_t)(live ? m : a.M - 1) * a.cin_s0 + 4 * lh;
    float4 xin[K0G];
#pragma unroll
    for (int g = 0; g < K0G; ++g) xin[g] = *reinterpret_cast<const float4*>(row + 8 * g);

    float4 h0[N0T * 4], h1[N1T * 4], h2[N2T * 4];
    {
      f32x16 acc[N0T];
      chain_layer<K0G, N0T>(sW0, li, lh, [&](int g) { return xin[g]; }, acc);
      chain_epilogue<N0T>(sC0, lh, a.relu[0], acc, h0);
    }
    {
      f32x16 acc[N1T];
      chain_layer<K1G, N1T>(sW1, li, lh, [&](int g) { return h0[g]; }, acc);
      chain_epilogue<N1T>(sC1, lh, a.relu[1], acc, h1);
    }
    {
      f32x16 acc[N2T];
      chain_layer<K2G, N2T>(sW2, li, lh, [&](int g) { return h1[g]; }, acc);
      chain_epilogue<N2T>(sC2, lh, a.relu[2], acc, h2);
    }
    if (live) {
      float* orow = a.out + (size_t)m * a.cout_s2 + 4 * lh;
#pragma unroll
      for (int t = 0; t < N2T; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (t * 32 + 8 * q + 4 * lh < a.cout_s2) *reinterpret_cast<float4*>(orow + t * 32 + 8 * q) = h2[t * 4 + q];
    }
  }
}

// ---------------------------------------------------------------------------------------- few rows
// The tracker's per-frame call has 72 rows (2 hands x 6 x 6).  In the kernel above one wave then walks its 32 rows
// through all three layers and all of a layer's cout tiles in turn: 288 + 168 + 108 dependent MFMAs (17 us) after 114 KB
// of weights went to LDS for three waves to use once - 32 us per launch.  Here a workgroup of four waves takes ONE
// 32-row tile, wave t computes cout tile t of every layer (72 + 56 + 36 MFMAs), the weight fragments come straight from
// L2 two channel groups ahead, and a layer's output crosses to the other waves through a 16 KB LDS tile (two barriers).
// Per output the same k-ordered chain and the same epilogue expression as above: bit-identical (tests/test_gpu_edges.py).
constexpr int CHAIN_SMALL_LD = 132;          // floats per LDS row: up to 128 channels + 4 (rows 4 banks apart)

template <int KG, int NT, typename Opnd>
__device__ __forceinline__ void chain_small_layer(const float* w, const float* bias, const float* scale, const float* shift, int relu,
                                                  int t, int li, int lh, Opnd opnd, float4 (&out)[4]) {
  // packed weights [g][half][NT * 32][4]: lane (li, lh) of cout tile t
  const float* wl = w + ((size_t)lh * (NT * 32) + t * 32 + li) * 4;
  constexpr int GS = 2 * NT * 32 * 4;        // floats per channel group
  float4 wr[3];
  wr[0] = *reinterpret_cast<const float4*>(wl);
  if (KG > 1) wr[1] = *reinterpret_cast<const float4*>(wl + GS);
  float4 cb[4], cs[4], ch[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int co = t * 32 + 8 * gq + 4 * lh;
    cb[gq] = *reinterpret_cast<const float4*>(bias + co);
    cs[gq] = *reinterpret_cast<const float4*>(scale + co);
    ch[gq] = *reinterpret_cast<const float4*>(shift + co);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (g + 2 < KG) wr[(g + 2) % 3] = *reinterpret_cast<const float4*>(wl + (size_t)(g + 2) * GS);
    const float4 x = opnd(g);
    __builtin_amdgcn_sched_barrier(0);
    const float4 wv = wr[g % 3];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, x.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, x.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, x.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, x.w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {            // the arithmetic of chain_epilogue
    float4 v = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
    v.x = (v.x + cb[gq].x) * cs[gq].x + ch[gq].x;
    v.y = (v.y + cb[gq].y) * cs[gq].y + ch[gq].y;
    v.z = (v.z + cb[gq].z) * cs[gq].z + ch[gq].z;
    v.w = (v.w + cb[gq].w) * cs[gq].w + ch[gq].w;
    if (relu) {
      v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
      v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
      v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
      v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
    }
    out[gq] = v;
  }
}

template <int K0G, int N0T, int K1G, int N1T, int K2G, int N2T>
__global__ __launch_bounds__(256) void conv1x1_chain3_small_f32(Chain3Args a) {
  static_assert(N0T <= 4 && N1T <= 4 && N2T <= 4, "one wave per cout tile");
  __shared__ __attribute__((aligned(16))) float sH[2][32 * CHAIN_SMALL_LD];
  const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m = blockIdx.x * 32 + li;
  const bool live = m < a.M;
  float4 q[4];
  if (t < N0T) {
    const float* row = a.in + (size_t)(live ? m : a.M - 1) * a.cin_s0 + 4 * lh;
    float4 xin[K0G];
#pragma unroll
    for (int g = 0; g < K0G; ++g) xin[g] = *reinterpret_cast<const float4*>(row + 8 * g);
    chain_small_layer<K0G, N0T>(a.w[0], a.bias[0], a.scale[0], a.shift[0], a.relu[0], t, li, lh, [&](int g) { return xin[g]; }, q);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<float4*>(&sH[0][li * CHAIN_SMALL_LD + t * 32 + 8 * gq + 4 * lh]) = q[gq];
  }
  __syncthreads();
  if (t < N1T) {
    chain_small_layer<K1G, N1T>(a.w[1], a.bias[1], a.scale[1], a.shift[1], a.relu[1], t, li, lh,
                                [&](int g) { return *reinterpret_cast<const float4*>(&sH[0][li * CHAIN_SMALL_LD + 8 * g + 4 * lh]); }, q);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<float4*>(&sH[1][li * CHAIN_SMALL_LD + t * 32 + 8 * gq + 4 * lh]) = q[gq];
  }
  __syncthreads();
  if (t < N2T) {
    chain_small_layer<K2G, N2T>(a.w[2], a.bias[2], a.scale[2], a.shift[2], a.relu[2], t, li, lh,
                                [&](int g) { return *reinterpret_cast<const float4*>(&sH[1][li * CHAIN_SMALL_LD + 8 * g + 4 * lh]); }, q);
    if (live) {
      float* orow = a.out + (size_t)m * a.cout_s2 + 4 * lh;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        if (t * 32 + 8 * gq + 4 * lh < a.cout_s2) *reinterpret_cast<float4*>(orow + t * 32 + 8 * gq) = q[gq];
    }
  }
}

constexpr int CHAIN_SMALL_MAX_TILES = 256;   // up to 8192 rows (227 hands): one 32-row tile per workgroup

template <int K0G, int N0T, int K1G, int N1T, int K2G, int N2T>
static constexpr size_t chain_lds_bytes() {
  return (size_t)(chain_w_floats<K0G, N0T>() + chain_w_floats<K1G, N1T>() + chain_w_floats<K2G, N2T>() + 3 * 32 * (N0T + N1T + N2T)) *
         sizeof(float);
}

template <int K0G, int N0T, int K1G, int N1T, int K2G, int N2T>
static bool chain_shape(const ConvLayer& a, const ConvLayer& b, const ConvLayer& c) {
  return a.cin_s == 8 * K0G && a.cout_w == 32 * N0T && b.cin_s == 8 * K1G && b.cout_w == 32 * N1T && c.cin_s == 8 * K2G &&
         c.cout_w == 32 * N2T && a.cout_s == b.cin_s && b.cout_s == c.cin_s;
}

// the two chains of the head: temporal 90 -> 90 -> 90 -> 90 and fusion 144 -> 108 -> 72 -> 72
#define UMT_CHAIN_TEMPORAL 12, 3, 12, 3, 12, 3
#define UMT_CHAIN_FUSION 18, 4, 14, 3, 9, 3

bool conv_chain3_applicable(const ConvLayer& a, const ConvLayer& b, const ConvLayer& c) {
  if ((a.flags & CONV_NO_CHAIN) || a.taps != 1 || b.taps != 1 || c.taps != 1 || a.stride != 1 || b.stride != 1 || c.stride != 1)
    return false;
  return chain_shape<UMT_CHAIN_TEMPORAL>(a, b, c) || chain_shape<UMT_CHAIN_FUSION>(a, b, c);
}

int conv_chain_prepare_device() {
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv1x1_chain3_f32<UMT_CHAIN_TEMPORAL>), chain_lds_bytes<UMT_CHAIN_TEMPORAL>()));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv1x1_chain3_f32<UMT_CHAIN_FUSION>), chain_lds_bytes<UMT_CHAIN_FUSION>()));
  return 0;
}

int launch_conv_chain3(const ConvLayer& l0, const ConvLayer& l1, const ConvLayer& l2, const float* in, int64_t rows, float* out,
                       hipStream_t stream) {
  UMT_REQUIRE(conv_chain3_applicable(l0, l1, l2), UMT_ERR_INVALID, "conv_chain3: layer shapes not supported");
  UMT_REQUIRE(rows > 0 && rows < ((int64_t)1 << 31) - 64, UMT_ERR_INVALID, "conv_chain3: %lld rows", (long long)rows);
  Chain3Args a;
  const ConvLayer* L[3] = {&l0, &l1, &l2};
  for (int i = 0; i < 3; ++i) {
    a.w[i] = L[i]->w; a.bias[i] = L[i]->bias; a.scale[i] = L[i]->scale; a.shift[i] = L[i]->shift; a.relu[i] = L[i]->relu;
  }
  a.in = in; a.out = out;
  a.M = (int)rows;
  a.n_wave_tiles = (int)((rows + 31) / 32);
  a.cin_s0 = l0.cin_s;
  a.cout_s2 = l2.cout_s;
  const bool temporal = chain_shape<UMT_CHAIN_TEMPORAL>(l0, l1, l2);
  if (a.n_wave_tiles <= CHAIN_SMALL_MAX_TILES && !(l0.flags & CONV_NO_SMALL)) {
    void (*ks)(Chain3Args) = temporal ? conv1x1_chain3_small_f32<UMT_CHAIN_TEMPORAL> : conv1x1_chain3_small_f32<UMT_CHAIN_FUSION>;
    hipLaunchKernelGGL(ks, dim3((unsigned)a.n_wave_tiles), dim3(256), 0, stream, a);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const int wgs = (a.n_wave_tiles + CHAIN_THREADS / 64 - 1) / (CHAIN_THREADS / 64);
  const dim3 grid((unsigned)(wgs < 256 ? wgs : 256));     // one persistent workgroup per CU (gfx950: 256 CUs)
  void (*kern)(Chain3Args) = temporal ? conv1x1_chain3_f32<UMT_CHAIN_TEMPORAL> : conv1x1_chain3_f32<UMT_CHAIN_FUSION>;
  const size_t lds = temporal ? chain_lds_bytes<UMT_CHAIN_TEMPORAL>() : chain_lds_bytes<UMT_CHAIN_FUSION>();
  hipLaunchKernelGGL(kern, grid, dim3(CHAIN_THREADS), lds, stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace umt
