// Persistent bf16 kernel for the 32 -> 32 channel stride-1 3x3 convolutions of the first backbone stage
// (lib/models/backbone_resnet.py:56-72 with planes = 32: conv1/conv2 of the two stage-1 BasicBlocks) in
// UMT_PREC_BF16 mode.
//
// These layers have K = 9 * 32 only: 18 MFMAs per 32x32 output tile against 4 KB read and 2-4 KB
// written, so they are bound by memory and by per-tile latencies, not by the matrix pipe.  With one
// workgroup per 128-pixel tile (conv3x3s1_halo_bf16) they ran at 2.2-2.9 TB/s.  Here
//   * the whole weight tensor lives in REGISTERS for the lifetime of the wave (18 uint4 per lane hold
//     the packed B fragments of all 9 taps), so the tap loop issues no global loads;
//   * one persistent workgroup of 8 waves per CU = two groups of 4 waves, one wave of each group per
//     SIMD, working on alternate 256-pixel tiles half a tile out of phase: between two workgroup
//     barriers one group runs the MFMAs of its tile while the other runs the epilogue of its previous
//     tile and writes the halo range of its next one into LDS, then they swap;
//   * the halo range of a group's tile j+1 and the residual rows of tile j are requested into registers
//     before the MFMAs of tile j, so the memory latency is hidden inside the wave;
//   * tiles are dealt so that each XCD works on one contiguous run of pixel rows (halo rows shared by
//     neighbouring tiles are fetched into one L2 only).
// Arithmetic (k order: tap, 16-channel group) is identical to conv3x3s1_halo_bf16: results are bit-identical
// to that kernel's (tools/conv_bench.hip checks it).  The fp32 layers of the same shape stay on
// conv3x3s1_halo_f32: with 36 float4 of weights per lane the persistent form spills or loses its
// prefetch registers, and measured 7 % faster without and 3 % slower with a residual.
#include "umt_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace umt {

// max(v, 0) as one integer max on the bits, as conv_f32_common.h: negative floats are negative integers, -0 -> +0,
// and a (positive, canonical) NaN stays a NaN, so a poisoned activation is not laundered into a finite one
__device__ __forceinline__ float thin_relu(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ThinArgs {
  const void* in;        // rows [M][32] (fp32 or bf16)
  const void* w;         // packed weights (fp32 [9][4][2][32][4] or bf16 [9][2][2][32][8])
  const float* bias;
  const float* scale;
  const float* shift;
  const void* residual;  // rows [M][32] or null
  void* out;             // rows [M][32]
  int H, W, M, relu;
  int ntiles;
  unsigned mg_hw, mg_w;  // floor(2^32 / (H*W)) + 1 and floor(2^32 / W) + 1
};
#ifdef UMT_STAMP
// diagnostic build only (tools/conv_bench.hip): cycle stamps of workgroup 0, never part of the library
__device__ unsigned long long g_stamp[2][64][4];
#define STAMP(ph, k) do { if (blockIdx.x == 0 && tid == 0 && (ph) < 64) g_stamp[group][ph][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(ph, k) do {} while (0)
#endif
#ifdef UMT_STAMP
#define PSTAMP(j, k) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (j) < 64) g_stamp[threadIdx.x >> 8][j][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(j, k) do {} while (0)
#endif

__device__ __forceinline__ int thin_div(int m, int d, unsigned mg) {      // floor(m / d), 0 <= m < 2^31
  int q = (int)__umulhi((unsigned)m, mg);
  if (q * d > m) --q;
  return q;
}

constexpr int T_LDH = 40;   // bf16 per bf16 LDS row (80 B)
constexpr int T_LDP = 36;   // floats per row of the epilogue patch

// tiles of this workgroup: XCD x gets the contiguous run [x*tpx, (x+1)*tpx) and its workgroups
// (blockIdx % 8 == x, the round-robin deal) stride through it together
__device__ __forceinline__ void tile_walk(int ntiles, int& first, int& step, int& end) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  first = xcd * tpx + slot;
  step = nslots;
  end = min((xcd + 1) * tpx, ntiles);
}

__device__ __forceinline__ unsigned border_mask(int m, int M, int H, int W, unsigned mg_hw, unsigned mg_w) {
  unsigned mk = 0;
  if (m < M) {
    const int rem = m - thin_div(m, H * W, mg_hw) * (H * W);
    const int y = thin_div(rem, W, mg_w), x = rem - y * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) mk |= 1u << (ky * 3 + kx);
      }
  }
  return mk;
}

// Tiles of one 4-wave group: every other tile of the workgroup's walk.
struct GroupWalk {
  int first, step, count;     // tiles first + j*step, j < count, belong to this wave's group
};
__device__ __forceinline__ GroupWalk group_walk(int ntiles, int group) {
  int t, tstep, tend;
  tile_walk(ntiles, t, tstep, tend);
  const int n = t < tend ? (tend - t + tstep - 1) / tstep : 0;     // tiles of the workgroup
  GroupWalk g;
  g.first = t + group * tstep;
  g.step = 2 * tstep;
  g.count = (n + 1 - group) >> 1;
  return g;
}

// ------------------------------------------------------------------------------------------ bf16
__device__ __forceinline__ unsigned thin_pack_bf16x2(float x, float y) {
  const __bf16 p = (__bf16)x, r = (__bf16)y;
  return (unsigned)__builtin_bit_cast(unsigned short, p) | ((unsigned)__builtin_bit_cast(unsigned short, r) << 16);
}

// Batch norm, residual add, ReLU and the bf16 rounding of eight consecutive channels of one pixel, written on register
// PAIRS: 4 packed adds + 4 packed fused multiply-adds (+ 4 packed adds for the residual), 8 integer max and 4 packed
// conversions.  Each component rounds exactly as the scalar (x + bias) * scale + shift, += residual, max(x, 0), (bf16)
// sequence does.  The same arithmetic on float4 fields compiled to pairs of the wrong components with ~40 register moves
// per call around them -- and vector instructions are what an epilogue wave is short of while the other group's MFMAs
// run on its SIMD (tools/mfma_peak.hip, mix: a saturated matrix pipe leaves a second wave 2.5 vector instructions per
// 32-cycle MFMA, a third of the rate it has alone).
typedef float thin_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 thin_bf16x2 __attribute__((ext_vector_type(2)));
typedef int thin_i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int thin_u32x4 __attribute__((ext_vector_type(4)));
template <bool RES>
__device__ __forceinline__ uint4 thin_bn_res_relu_pack8(float4 v, float4 u, const float4& cb0, const float4& cb1, const float4& cs0,
                                                        const float4& cs1, const float4& ch0, const float4& ch1, thin_u32x4 rr,
                                                        int relu_floor) {
  thin_f32x2 p[4] = {{v.x, v.y}, {v.z, v.w}, {u.x, u.y}, {u.z, u.w}};
  const thin_f32x2 b[4] = {{cb0.x, cb0.y}, {cb0.z, cb0.w}, {cb1.x, cb1.y}, {cb1.z, cb1.w}};
  const thin_f32x2 sc[4] = {{cs0.x, cs0.y}, {cs0.z, cs0.w}, {cs1.x, cs1.y}, {cs1.z, cs1.w}};
  const thin_f32x2 sh[4] = {{ch0.x, ch0.y}, {ch0.z, ch0.w}, {ch1.x, ch1.y}, {ch1.z, ch1.w}};
  unsigned o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p[i] = __builtin_elementwise_fma(p[i] + b[i], sc[i], sh[i]);
    if (RES) p[i] += (thin_f32x2){__builtin_bit_cast(float, rr[i] << 16), __builtin_bit_cast(float, rr[i] & 0xffff0000u)};
    // ReLU as an integer max on the bit patterns (relu_floor = 0, or INT_MIN for none), on the PAIR: written per component
    // (p.x = max(..), p.y = max(..)) this compiler converts component x twice and drops y
    const thin_i32x2 pi = __builtin_elementwise_max(__builtin_bit_cast(thin_i32x2, p[i]), (thin_i32x2){relu_floor, relu_floor});
    o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_bit_cast(thin_f32x2, pi), thin_bf16x2));
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int TM, bool RES>
__global__ __launch_bounds__(512, 2) void conv3x3_thin32_bf16(ThinArgs a) {
  constexpr int BM = 128 * TM;
  constexpr int NLD = ((BM + 2 * 48 + 2) * 4 + 255) / 256;   // uint4 slots per thread for one halo range
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, group = threadIdx.x >> 8;
  const int a_bytes = ((rows + 1) * T_LDH * 2 + 15) & ~15;      // halo range + the all-zero row
  unsigned char* sA = smem_h + group * (a_bytes + 4 * 32 * T_LDP * 4);
  float* patch = reinterpret_cast<float*>(sA + a_bytes) + wave * 32 * T_LDP;
  const int li = lane & 31, lh = lane >> 5;
  const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);
  const uint16_t* resid = reinterpret_cast<const uint16_t*>(a.residual);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out);

  uint4 breg[9][2];
  {
    const uint16_t* wp = reinterpret_cast<const uint16_t*>(a.w);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        breg[tap][c] = *reinterpret_cast<const uint4*>(wp + ((size_t)((tap * 2 + c) * 2 + lh) * 32 + li) * 8);
  }
  // epilogue: a lane owns 8 consecutive channels (16 bytes of bf16) of a pixel row, 16 rows per pass
  const int q = lane & 3, rsub = lane >> 2;
  const float4 cb0 = *reinterpret_cast<const float4*>(a.bias + 8 * q), cb1 = *reinterpret_cast<const float4*>(a.bias + 8 * q + 4);
  const float4 cs0 = *reinterpret_cast<const float4*>(a.scale + 8 * q), cs1 = *reinterpret_cast<const float4*>(a.scale + 8 * q + 4);
  const float4 ch0 = *reinterpret_cast<const float4*>(a.shift + 8 * q), ch1 = *reinterpret_cast<const float4*>(a.shift + 8 * q + 4);

  const GroupWalk gw = group_walk(a.ntiles, group);
  const GroupWalk g0 = group_walk(a.ntiles, 0);
  const int nphase = 2 * g0.count + 1;

  uint4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int tile) {
    const int gbase = tile * BM - W - 1;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 2, qq = idx & 3;
      const int g = gbase + r;
      const bool ok = r < rows && g >= 0 && g < a.M;
      const size_t off = ok ? (size_t)g * 32 + 8 * qq : 0;
      areg[k] = *reinterpret_cast<const uint4*>(in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      uint4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
      if ((idx >> 2) < rows) *reinterpret_cast<uint4*>(sA + ((idx >> 2) * T_LDH + 8 * (idx & 3)) * 2) = v;
    }
  };

  if (tid < T_LDH * 2 / 16) *reinterpret_cast<uint4*>(sA + rows * T_LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
  if (gw.count > 0) {
    load_a(gw.first);
    write_a();
  }
  __syncthreads();

  f32x16 acc[TM];
  uint4 res[TM][2];
  for (int ph = 0; ph < nphase; ++ph) {
    const int j = (ph - group) >> 1;
    const bool mfma_phase = ((ph - group) & 1) == 0;
    if (ph >= group && j < gw.count) {
      const int t = gw.first + j * gw.step;
      const int mw = t * BM + wave * TM * 32;
      if (mfma_phase) {
        if (j + 1 < gw.count) load_a(t + gw.step);
        if (RES) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int m = mw + i * 32 + 16 * k + rsub;
              const size_t off = m < a.M ? (size_t)m * 32 + 8 * q : 0;
              res[i][k] = *reinterpret_cast<const uint4*>(resid + off);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        int arow[TM][9];      // byte offset of the lane's operand row per tap; the zero row where padded
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned mk = border_mask(mw + i * 32 + li, a.M, H, W, a.mg_hw, a.mg_w);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
            arow[i][tap] = ((((mk >> tap) & 1u) ? (wave * TM + i) * 32 + li + (tap / 3) * W + (tap % 3) : rows) * T_LDH + 8 * lh) * 2;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const uint4 raw = *reinterpret_cast<const uint4*>(sA + arow[i][tap] + c * 32);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, raw),
                                                               __builtin_bit_cast(bf16x8, breg[tap][c]), acc[i], 0, 0, 0);
            }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * T_LDP + li] = acc[i][r];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int row = 16 * k + rsub;
            const int m = mw + i * 32 + row;
            float4 v = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q);
            float4 u = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q + 4);
            v.x = (v.x + cb0.x) * cs0.x + ch0.x; v.y = (v.y + cb0.y) * cs0.y + ch0.y;
            v.z = (v.z + cb0.z) * cs0.z + ch0.z; v.w = (v.w + cb0.w) * cs0.w + ch0.w;
            u.x = (u.x + cb1.x) * cs1.x + ch1.x; u.y = (u.y + cb1.y) * cs1.y + ch1.y;
            u.z = (u.z + cb1.z) * cs1.z + ch1.z; u.w = (u.w + cb1.w) * cs1.w + ch1.w;
            if (RES) {
              const uint4 rr = res[i][k];
              v.x += __builtin_bit_cast(float, rr.x << 16); v.y += __builtin_bit_cast(float, rr.x & 0xffff0000u);
              v.z += __builtin_bit_cast(float, rr.y << 16); v.w += __builtin_bit_cast(float, rr.y & 0xffff0000u);
              u.x += __builtin_bit_cast(float, rr.z << 16); u.y += __builtin_bit_cast(float, rr.z & 0xffff0000u);
              u.z += __builtin_bit_cast(float, rr.w << 16); u.w += __builtin_bit_cast(float, rr.w & 0xffff0000u);
            }
            if (a.relu) {
              v.x = thin_relu(v.x); v.y = thin_relu(v.y); v.z = thin_relu(v.z); v.w = thin_relu(v.w);
              u.x = thin_relu(u.x); u.y = thin_relu(u.y); u.z = thin_relu(u.z); u.w = thin_relu(u.w);
            }
            if (m < a.M)
              *reinterpret_cast<uint4*>(out + (size_t)m * 32 + 8 * q) =
                  make_uint4(thin_pack_bf16x2(v.x, v.y), thin_pack_bf16x2(v.z, v.w), thin_pack_bf16x2(u.x, u.y),
                             thin_pack_bf16x2(u.z, u.w));
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (j + 1 < gw.count) write_a();
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ 64 -> 64, weights in LDS
// Same persistent ping-pong scheme for the 64 -> 64 channel stride-1 3x3 layers of stage 2 (W <= 24).  Their
// packed weights (73.7 KB of bf16) do not fit in registers but do fit in LDS beside two halo ranges, so the
// workgroup copies them in once and the tap loop reads BOTH operands from LDS: no global load sits between two
// MFMAs, and the next tile's halo range and this tile's residual rows are in flight during the whole MFMA
// phase.  One workgroup per CU (162.9 KB of LDS), 128-pixel tiles, a wave owns 32 pixels x 64 channels.
// k order (32-channel chunk, tap, 16-channel group) is conv3x3s1_halo_bf16's: results are bit-identical to it.
constexpr int W64_LDA = 72;                    // bf16 per LDS row: 64 channels + 16 B (144 B; 16 rows -> 16 distinct slots)
constexpr int W64_BM = 128;
constexpr int W64_WMAX = 24;
constexpr int W64_WBYTES = 9 * 64 * 64 * 2;    // packed weights [9][4][2][64][8] bf16
constexpr int W64_CBYTES = 3 * 64 * 4;         // bias, scale, shift
#ifndef W64_AHEAD
#define W64_AHEAD 2
#endif
#ifndef W64_EPI_PRIO
#define W64_EPI_PRIO 3
#endif

template <bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_wlds64_bf16(ThinArgs a) {
  constexpr int BM = W64_BM;
  constexpr int NLD = ((BM + 2 * W64_WMAX + 2) * 8 + 255) / 256;   // uint4 slots per thread for one halo range
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, group = threadIdx.x >> 8;
  const int a_bytes = ((rows + 1) * W64_LDA * 2 + 15) & ~15;      // halo range + the all-zero row
  unsigned char* sW = smem_w;
  const float* sC = reinterpret_cast<const float*>(smem_w + W64_WBYTES);
  unsigned char* sA = smem_w + W64_WBYTES + W64_CBYTES + group * (a_bytes + 4 * 32 * T_LDP * 4);
  float* patch = reinterpret_cast<float*>(sA + a_bytes) + wave * 32 * T_LDP;
  const int li = lane & 31, lh = lane >> 5;
  const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);
  const uint16_t* resid = reinterpret_cast<const uint16_t*>(a.residual);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out);

  // weights and the epilogue constants -> LDS, once per workgroup
  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    for (int i = threadIdx.x; i < W64_WBYTES / 16; i += 512) *reinterpret_cast<uint4*>(sW + 16 * i) = wp[i];
    float* c = reinterpret_cast<float*>(smem_w + W64_WBYTES);
    if (threadIdx.x < 64) {
      c[threadIdx.x] = a.bias[threadIdx.x];
      c[64 + threadIdx.x] = a.scale[threadIdx.x];
      c[128 + threadIdx.x] = a.shift[threadIdx.x];
    }
  }
  const int q = lane & 3, rsub = lane >> 2;

  const GroupWalk gw = group_walk(a.ntiles, group);
  const GroupWalk g0 = group_walk(a.ntiles, 0);
  const int nphase = 2 * g0.count + 1;

  // One halo range = rows [tile * BM - W - 1, + rows) of the activation tensor, clipped to [0, M): a contiguous run, read
  // through a buffer descriptor that covers exactly that run, so the rows outside it come back as zeros from the
  // hardware range check (offsets below the run wrap to > 2^31).  No validity mask is carried from the loads to the LDS
  // stores: with one, the register allocator parked the mask in a register of the load destinations and made every
  // MFMA phase wait for the HBM round trip of the loads it had just issued (2700 of its 5300 cycles).
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  uint4 areg[NLD];
  auto load_a = [&](int tile_v) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_v);
    const int gbase = tile * BM - W - 1;
    const int g_lo = gbase > 0 ? gbase : 0, g_hi = gbase + rows < a.M ? gbase + rows : a.M;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(in) + (size_t)g_lo * 64, 0, g_hi > g_lo ? (g_hi - g_lo) * 128 : 0, 0x00020000);
    const int shift = (g_lo - gbase) * 128;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16 - shift, 0, 0);
      areg[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      if ((idx >> 3) < rows) *reinterpret_cast<uint4*>(sA + ((idx >> 3) * W64_LDA + 8 * (idx & 7)) * 2) = areg[k];
    }
  };

  if (tid < W64_LDA * 2 / 16) *reinterpret_cast<uint4*>(sA + rows * W64_LDA * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
  if (gw.count > 0) {
    load_a(gw.first);
    write_a();
  }
  __syncthreads();

  const int relu_floor = a.relu ? 0 : (int)0x80000000;
  f32x16 acc[2];
  // The residual rows of a tile are requested in its MFMA phase and used in its epilogue phase.  They are loaded into
  // accumulation registers by hand: as an ordinary load into vector registers the allocator moved half of one destination
  // quad out of the way of an address temporary right after the request, which cost every MFMA phase an HBM round trip.
  // The compiler does not count these loads, so the epilogue waits for them itself (its own counted waits stay safe: the
  // load counter retires in order, and extra loads in it only make a counted wait stricter).
  u32x4 res[2][2];
  for (int ph = 0; ph < nphase; ++ph) {
    STAMP(ph, 0);
    const int j = (ph - group) >> 1;
    const bool mfma_phase = ((ph - group) & 1) == 0;
    if (ph >= group && j < gw.count) {
      const int t = gw.first + j * gw.step;
      const int mw = t * BM + wave * 32;
      if (mfma_phase) {
        int arow[9];      // byte offset of the lane's operand row per tap; the zero row where padded
        {
          const unsigned mk = border_mask(mw + li, a.M, H, W, a.mg_hw, a.mg_w);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
            arow[tap] = ((((mk >> tap) & 1u) ? wave * 32 + li + (tap / 3) * W + (tap % 3) : rows) * W64_LDA + 8 * lh) * 2;
        }
        const unsigned char* bl = sW + (lh * 64 + li) * 16;       // + ((tap * 4 + c16) * 2) * 64 * 16 + n * 512
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < gw.count) load_a(t + gw.step);
        if (RES) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int m = mw + 16 * k + rsub;
              const size_t off = m < a.M ? (size_t)m * 64 + n * 32 + 8 * q : 0;
              const uint16_t* rp = resid + off;
              asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(res[n][k]) : "v"(rp) : "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          // k-step st = (chunk, tap, c); its three fragments are read from LDS W64_AHEAD steps before its MFMAs and
          // pinned there (read in the step that uses them, every MFMA pair waited a full LDS round trip)
          STAMP(ph, 3);
          constexpr int NS = 36, D = W64_AHEAD;
          uint4 av[D + 1], bv0[D + 1], bv1[D + 1];
          auto rd = [&](int slot, int st) {
            const int chunk = st / 18, tap = (st % 18) >> 1, c16 = chunk * 2 + (st & 1);
            av[slot] = *reinterpret_cast<const uint4*>(sA + arow[tap] + c16 * 32);
            bv0[slot] = *reinterpret_cast<const uint4*>(bl + (tap * 4 + c16) * 2048);
            bv1[slot] = *reinterpret_cast<const uint4*>(bl + (tap * 4 + c16) * 2048 + 512);
          };
#pragma unroll
          for (int st = 0; st < D; ++st) rd(st, st);
#pragma unroll
          for (int st = 0; st < NS; ++st) {
            if (st + D < NS) rd((st + D) % (D + 1), st + D);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[st % (D + 1)]),
                                                             __builtin_bit_cast(bf16x8, bv0[st % (D + 1)]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[st % (D + 1)]),
                                                             __builtin_bit_cast(bf16x8, bv1[st % (D + 1)]), acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        __builtin_amdgcn_s_setprio(W64_EPI_PRIO);      // as in conv3x3_stream128_bf16: the epilogue phase issues first
        if (RES)
          asm volatile("s_waitcnt vmcnt(0)" : "+a"(res[0][0]), "+a"(res[0][1]), "+a"(res[1][0]), "+a"(res[1][1])::"memory");
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int co = n * 32 + 8 * q;
          const float4 cb0 = *reinterpret_cast<const float4*>(sC + co), cb1 = *reinterpret_cast<const float4*>(sC + co + 4);
          const float4 cs0 = *reinterpret_cast<const float4*>(sC + 64 + co), cs1 = *reinterpret_cast<const float4*>(sC + 64 + co + 4);
          const float4 ch0 = *reinterpret_cast<const float4*>(sC + 128 + co), ch1 = *reinterpret_cast<const float4*>(sC + 128 + co + 4);
#pragma unroll
          for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * T_LDP + li] = acc[n][r];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int row = 16 * k + rsub;
            const int m = mw + row;
            const float4 v = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q);
            const float4 u = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q + 4);
            const uint4 o = thin_bn_res_relu_pack8<RES>(v, u, cb0, cb1, cs0, cs1, ch0, ch1, RES ? res[n][k] : u32x4{0, 0, 0, 0}, relu_floor);
            if (m < a.M) *reinterpret_cast<uint4*>(out + (size_t)m * 64 + co) = o;
          }
          __builtin_amdgcn_wave_barrier();
          if (n == 0) STAMP(ph, 3);
        }
        STAMP(ph, 2);
        if (j + 1 < gw.count) write_a();
        __builtin_amdgcn_s_setprio(0);
      }
    }
    STAMP(ph, 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ 64 -> 64, one role per wave
// conv3x3_wlds64_bf16 restructured as conv3x3_pipe32_f32 is (see there for why): every wave carries the epilogue of its
// previous tile, the LDS stores of the next halo range, the loads of the range after next and of its residual rows, and
// the next tile's border arithmetic between its own MFMAs.  Four waves, one per SIMD, each 64 pixels x 64 channels (2 x 2
// MFMA tiles: one ds_read_b128 per MFMA; with 32 x 64 per wave and eight waves the LDS array, at 1.5 reads per MFMA,
// set the pace and the kernel was no faster than the two-group one), 256-pixel tiles, 36 k-steps of four MFMAs.  Halo
// ranges double-buffered in LDS beside the packed weights (162.9 KB in all), one barrier per tile, the accumulators of
// two consecutive tiles alternate between two register sets.  Operands swapped (weights as the row operand): a lane owns
// 2 x 16 channels of two pixels and stores 8-byte quads, no transposing patch.  k order (32-channel chunk, tap,
// 16-channel group) and the epilogue arithmetic are conv3x3s1_halo_bf16's: bit-identical.
// Measured (tools/conv_bench.hip, 4096 crops, random data): the MFMA + LDS loop alone (-DP64_NOSIDE: no side work, wrong
// results) runs at 33 cycles per MFMA, 112 us per layer = 1.56 PFLOP/s, the chip's sustained rate; with one kind of side work
// at a time (-DP64_SIDE_MASK) halo stores cost +14 us, halo loads +9, border arithmetic +18, the epilogue +50 of which 38 were
// its 8-byte stores (hence the quad trade below).  All of it: 172 us without and 190 us with a residual (906 MB: 4.8 TB/s, the
// HBM floor of that layer) against 185 / 208 for the two-group kernel; in the bf16 bench 0.62-0.67 / 0.76-0.84 ms per layer and
// 16 384 crops against 0.70-0.81 / 0.80-0.90.  Operand prefetch depth (W64_AHEAD 1, 2, 3) makes no difference.
// The side-work steps still take 170-300 cycles instead of 134: the parts between the MFMAs are short but are dependent
// chains, a wave issues in order, and one wave per SIMD has nobody to cover a chain's latency.
constexpr int P64_BM = 256;
constexpr int P64_THREADS = 256;

template <bool RES>
__global__ __launch_bounds__(P64_THREADS, 1) void conv3x3_pipe64_bf16(ThinArgs a) {
  constexpr int BM = P64_BM, NT = P64_THREADS;
  constexpr int NLD = ((BM + 2 * W64_WMAX + 2) * 8 + NT - 1) / NT;   // uint4 slots per thread for one halo range (10)
  static_assert(NLD == 10, "conv3x3_pipe64_bf16: the k-step schedule below places 10 halo slots");
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a_bytes = (rows + 2) * W64_LDA * 2;                    // halo range + the all-zero row + a scratch row (144-byte rows)
  unsigned char* sW = smem_q;
  const float* sC = reinterpret_cast<const float*>(smem_q + W64_WBYTES);
  unsigned char* sA0 = smem_q + W64_WBYTES + W64_CBYTES;
  const int li = lane & 31, lh = lane >> 5;
  const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);
  const uint16_t* resid = reinterpret_cast<const uint16_t*>(a.residual);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out);
  const int relu_floor = a.relu ? 0 : (int)0x80000000;

  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    for (int i = tid; i < W64_WBYTES / 16; i += NT) *reinterpret_cast<uint4*>(sW + 16 * i) = wp[i];
    float* c = reinterpret_cast<float*>(smem_q + W64_WBYTES);
    if (tid < 64) {
      c[tid] = a.bias[tid];
      c[64 + tid] = a.scale[tid];
      c[128 + tid] = a.shift[tid];
    }
    if (tid < 2 * (W64_LDA * 2 / 16))      // the all-zero row of both buffers
      *reinterpret_cast<uint4*>(sA0 + (tid / 9) * a_bytes + rows * W64_LDA * 2 + 16 * (tid % 9)) = make_uint4(0, 0, 0, 0);
  }

  int t0, tstep, tend;
  tile_walk(a.ntiles, t0, tstep, tend);
  const int count = t0 < tend ? (tend - t0 + tstep - 1) / tstep : 0;

  // halo range of a tile through a buffer descriptor that covers exactly its rows inside [0, M): zeros outside
  uint4 areg[NLD];
  __amdgpu_buffer_rsrc_t a_rs;
  int a_shift = 0;
  auto load_a_begin = [&](int tile, bool have) {      // both workgroup-uniform
    const int gbase = tile * BM - W - 1;
    const int g_lo = gbase > 0 ? gbase : 0, g_hi = gbase + rows < a.M ? gbase + rows : a.M;
    a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(in) + (size_t)g_lo * 64, 0,
                                             have && g_hi > g_lo ? (g_hi - g_lo) * 128 : 0, 0x00020000);
    a_shift = (g_lo - gbase) * 128;
  };
  auto load_a_one = [&](int k) {
    areg[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (tid + NT * k) * 16 - a_shift, 0, 0));
  };
  // (slots past the range's last row, and every slot when there is no next tile, go to the scratch row: no branch, no
  // exec mask change between two MFMAs)
  const int wa_base = ((tid >> 3) * W64_LDA + 8 * (tid & 7)) * 2, wa_scratch = ((rows + 1) * W64_LDA + 8 * (tid & 7)) * 2;
  auto write_a_one = [&](unsigned char* sA, int k, int nrows) {
    const int off = (tid >> 3) + (NT >> 3) * k < nrows ? wa_base + k * (NT >> 3) * W64_LDA * 2 : wa_scratch;
    *reinterpret_cast<uint4*>(sA + off) = areg[k];
  };
  // byte offset of the lane's operand row per tap (the zero row where padded), for the wave's two 32-pixel row tiles: a
  // 9-bit mask of the taps inside the image from the pixel's coordinates first, then one select per tap, spread over
  // the k-steps (the empty asm statements keep each result in a register where it is computed: left alone the compiler
  // holds the comparisons in scalar registers and does the selects at the top of the next tile, in front of its first reads)
  unsigned bmask[2] = {0, 0};
  const int zero_off = (rows * W64_LDA + 8 * lh) * 2;
  auto arow_begin = [&](int i, int m) {
    const int rem = m - thin_div(m, H * W, a.mg_hw) * (H * W);
    const int y = thin_div(rem, W, a.mg_w), x = rem - y * W;
    unsigned mk = m < a.M ? 0x1ffu : 0u;
    mk &= y == 0 ? ~0x007u : ~0u;
    mk &= y == H - 1 ? ~0x1c0u : ~0u;
    mk &= x == 0 ? ~0x049u : ~0u;
    mk &= x == W - 1 ? ~0x124u : ~0u;
    asm volatile("" : "+v"(mk));
    bmask[i] = mk;
  };
  auto arow_tap = [&](int i, int tap) {
    int r = (bmask[i] >> tap) & 1u ? ((wave * 64 + i * 32 + li + (tap / 3) * W + (tap % 3)) * W64_LDA + 8 * lh) * 2 : zero_off;
    asm volatile("" : "+v"(r));
    return r;
  };

  uint4 res4[8];
  int arow[2][9], arow_n[2][9];
  // piece p = 8i + 4n + g: channels 32n + 8g + 4lh .. + 3 (a "quad") of pixel row m of row tile i:
  // (acc + bias) * scale + shift (+ residual), ReLU, 4 x bf16.  In parts, so that each fits between two MFMAs, with the
  // constants read from LDS one piece ahead.  A lane's quads are 8 bytes every 16: stored as they are (8 bytes per lane, 16
  // per pixel row and instruction) the stores alone cost 38 us of a 186 us layer.  So an even quad waits for the odd one
  // after it, the two halves of the wave trade one quad each (v_permlane32_swap: lanes 32-63 of the first operand against
  // lanes 0-31 of the second), and every lane stores 16 bytes, 32 per pixel row and instruction.  The residual is loaded in
  // that traded layout and traded back by the same swap.
  float4 kq[2][3];
  auto piece_consts = [&](int p) {
    const float* kc = sC + 32 * ((p >> 2) & 1) + 8 * (p & 3) + 4 * lh;
    kq[p & 1][0] = *reinterpret_cast<const float4*>(kc);
    kq[p & 1][1] = *reinterpret_cast<const float4*>(kc + 64);
    kq[p & 1][2] = *reinterpret_cast<const float4*>(kc + 128);
  };
  f32x2 e_lo, e_hi;
  unsigned e_o0 = 0, e_o1 = 0, s_o0 = 0, s_o1 = 0, r_od0 = 0, r_od1 = 0;
  auto piece_math = [&](const f32x16 (&ac)[2][2], int p) {
    const int i = p >> 3, n = (p >> 2) & 1, g = p & 3;
    const float4 cb = kq[p & 1][0], cs = kq[p & 1][1], ch = kq[p & 1][2];
    e_lo = __builtin_elementwise_fma((f32x2){ac[i][n][4 * g], ac[i][n][4 * g + 1]} + (f32x2){cb.x, cb.y}, (f32x2){cs.x, cs.y},
                                     (f32x2){ch.x, ch.y});
    e_hi = __builtin_elementwise_fma((f32x2){ac[i][n][4 * g + 2], ac[i][n][4 * g + 3]} + (f32x2){cb.z, cb.w}, (f32x2){cs.z, cs.w},
                                     (f32x2){ch.z, ch.w});
    if (RES) {
      unsigned r0, r1;
      if (!(p & 1)) {
        const uint4 r = res4[p >> 1];
        const auto w0 = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
        r0 = w0[0]; r1 = w1[0];
        r_od0 = w0[1]; r_od1 = w1[1];
      } else {
        r0 = r_od0; r1 = r_od1;
      }
      e_lo += (f32x2){__builtin_bit_cast(float, r0 << 16), __builtin_bit_cast(float, r0 & 0xffff0000u)};
      e_hi += (f32x2){__builtin_bit_cast(float, r1 << 16), __builtin_bit_cast(float, r1 & 0xffff0000u)};
    }
  };
  auto piece_round = [&](int p) {
    const thin_i32x2 fl = {relu_floor, relu_floor};
    const thin_i32x2 li2 = __builtin_elementwise_max(__builtin_bit_cast(thin_i32x2, e_lo), fl);
    const thin_i32x2 hi2 = __builtin_elementwise_max(__builtin_bit_cast(thin_i32x2, e_hi), fl);
    e_o0 = __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_bit_cast(thin_f32x2, li2), thin_bf16x2));
    e_o1 = __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_bit_cast(thin_f32x2, hi2), thin_bf16x2));
    if (!(p & 1)) { s_o0 = e_o0; s_o1 = e_o1; }
  };
  auto piece_store = [&](int p, int m) {      // odd pieces only: the quad before it and this one
    if (!(p & 1)) return;
    const int c0 = 32 * ((p >> 2) & 1) + 8 * ((p & 2) + lh);
    const auto w0 = __builtin_amdgcn_permlane32_swap(s_o0, e_o0, false, false);
    const auto w1 = __builtin_amdgcn_permlane32_swap(s_o1, e_o1, false, false);
#ifdef P64_NOSTORE
    if (m < 0)
#else
    if (m < a.M)
#endif
      *reinterpret_cast<uint4*>(out + (size_t)m * 64 + c0) = make_uint4(w0[0], w1[0], w0[1], w1[1]);
  };
  auto epilogue_piece = [&](const f32x16 (&ac)[2][2], int p, int m) {      // all of it at once: the last tile of the walk
    piece_consts(p);
    piece_math(ac, p);
    piece_round(p);
    piece_store(p, m);
  };
  auto load_res_one = [&](int q, int m) {      // quads 2q and 2q + 1, in the traded layout
    const int c0 = 32 * ((q >> 1) & 1) + 16 * (q & 1) + 8 * lh;
    const size_t off = m < a.M ? (size_t)m * 64 + c0 : 0;
    res4[q] = *reinterpret_cast<const uint4*>(resid + off);
  };

  if (count > 0) {
    load_a_begin(t0, true);
#pragma unroll
    for (int k = 0; k < NLD; ++k) load_a_one(k);
#pragma unroll
    for (int k = 0; k < NLD; ++k) write_a_one(sA0, k, rows);
    load_a_begin(t0 + tstep, count > 1);
#pragma unroll
    for (int k = 0; k < NLD; ++k) load_a_one(k);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      arow_begin(i, t0 * BM + wave * 64 + i * 32 + li);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) arow[i][tap] = arow_tap(i, tap);
    }
  }
  __syncthreads();

  // one tile: MFMAs into ac, the previous tile's epilogue out of ap
  auto tile_body = [&](f32x16 (&ac)[2][2], const f32x16 (&ap)[2][2], int j) {
    const int t = t0 + j * tstep;
    const int mw = t * BM + wave * 64, mw_p = mw - tstep * BM;
    const unsigned char* sA = sA0 + (j & 1) * a_bytes;
    unsigned char* sAn = sA0 + ((j + 1) & 1) * a_bytes;
    const bool have_next = j + 1 < count, have_prev = j > 0;
    const int rows_next = have_next ? rows : 0;
    const int m_prev = have_prev ? mw_p + li : 0x7fffff00;      // no previous tile: a row past M, nothing is stored
    const unsigned char* bl = sW + (lh * 64 + li) * 16;       // + ((tap * 4 + c16) * 2) * 64 * 16 + n * 512
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[i][n][r] = 0.f;
    constexpr int NS = 36, D = W64_AHEAD;
    uint4 av0[D + 1], av1[D + 1], bv0[D + 1], bv1[D + 1];
    auto rd = [&](int slot, int st) {
      const int chunk = st / 18, tap = (st % 18) >> 1, c16 = chunk * 2 + (st & 1);
      av0[slot] = *reinterpret_cast<const uint4*>(sA + arow[0][tap] + c16 * 32);
      av1[slot] = *reinterpret_cast<const uint4*>(sA + arow[1][tap] + c16 * 32);
      bv0[slot] = *reinterpret_cast<const uint4*>(bl + (tap * 4 + c16) * 2048);
      bv1[slot] = *reinterpret_cast<const uint4*>(bl + (tap * 4 + c16) * 2048 + 512);
    };
#pragma unroll
    for (int st = 0; st < D; ++st) rd(st, st);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st == 0) PSTAMP(j, 0);
      if (st == 10) PSTAMP(j, 3);
      if (st == 26) PSTAMP(j, 2);
      if (st == 35) PSTAMP(j, 1);
      if (st + D < NS) rd((st + D) % (D + 1), st + D);
      // Side work goes BETWEEN the step's four MFMAs, a part after each: a wave issues in order, so vector instructions
      // placed after a group of MFMAs only start when the last of them has issued and have one MFMA's shadow, not four.
      // Nothing below branches (a piece that has nothing to do works on a row past M, an empty range, or unused registers).
      const int sl = st % (D + 1);
      auto side = [&](int part) {
#ifdef P64_NOSIDE      // diagnostic builds (wrong results): no side work at all, or with P64_SIDE_MASK only some of it
        return;
#endif
#ifdef P64_SIDE_MASK   // bit 0: halo stores, 1: epilogue, 2: halo loads, 3: border arithmetic (+ residual loads)
        if (!((P64_SIDE_MASK >> (st < 5 ? 0 : st < 10 ? 2 : st < 26 ? 1 : 3)) & 1)) return;
#endif
        if (st < 5) {                                              // halo range of tile j+1 -> the other buffer
          if (part == 0) write_a_one(sAn, 2 * st, rows_next);
          if (part == 2) write_a_one(sAn, 2 * st + 1, rows_next);
        } else if (st < 10) {                                      // halo range of tile j+2 -> registers, as early as the
          if (part == 0) {                                         // registers are free: it is stored at the top of the next
            if (st == 5) load_a_begin(t + 2 * tstep, j + 2 < count);   // tile, 30 k-steps from here (an HBM round trip under
            load_a_one(2 * (st - 5));                              // load is 10-15 of them)
          }
          if (part == 2) load_a_one(2 * (st - 5) + 1);
          if (st == 9 && part == 3) piece_consts(0);
        } else if (st < 10 + 16) {                                 // epilogue of tile j-1, one channel quad per step
          const int p = st - 10;
          if (part == 0) piece_math(ap, p);
          if (part == 1) piece_round(p);
          if (part == 2) piece_store(p, m_prev + (p >> 3) * 32);
          if (part == 3 && p + 1 < 16) piece_consts(p + 1);
        } else if (st == 26) {                                     // border arithmetic of tile j+1
          if (part == 0) arow_begin(0, mw + tstep * BM + li);
          if (part == 2) arow_begin(1, mw + tstep * BM + 32 + li);
        } else {
          if (part == 0) arow_n[0][st - 27] = arow_tap(0, st - 27);
          if (part == 2) arow_n[1][st - 27] = arow_tap(1, st - 27);
        }
        if (RES && st >= 26 && st < 34) {                          // residual rows of tile j
          if (part == 1) load_res_one(st - 26, mw + ((st - 26) >> 2) * 32 + li);
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      ac[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv0[sl]), __builtin_bit_cast(bf16x8, av0[sl]), ac[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      side(0);
      __builtin_amdgcn_sched_barrier(0);
      ac[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv1[sl]), __builtin_bit_cast(bf16x8, av0[sl]), ac[0][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      side(1);
      __builtin_amdgcn_sched_barrier(0);
      ac[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv0[sl]), __builtin_bit_cast(bf16x8, av1[sl]), ac[1][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      side(2);
      __builtin_amdgcn_sched_barrier(0);
      ac[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv1[sl]), __builtin_bit_cast(bf16x8, av1[sl]), ac[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      side(3);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) arow[i][tap] = arow_n[i][tap];
    __syncthreads();
  };

  f32x16 accA[2][2], accB[2][2];
  for (int j = 0; j < count; j += 2) {
    tile_body(accA, accB, j);
    if (j + 1 < count) tile_body(accB, accA, j + 1);
  }
  if (count > 0) {
    const int mw_l = (t0 + (count - 1) * tstep) * BM + wave * 64;
    if (count & 1) {
#pragma unroll
      for (int p = 0; p < 16; ++p) epilogue_piece(accA, p, mw_l + (p >> 3) * 32 + li);
    } else {
#pragma unroll
      for (int p = 0; p < 16; ++p) epilogue_piece(accB, p, mw_l + (p >> 3) * 32 + li);
    }
  }
}

// ------------------------------------------------------------------------------ 128 -> 128, weights streamed
// The 128 -> 128 channel stride-1 3x3 layers of stage 3 (W <= 12) in the same persistent two-group scheme.  Their
// weights (295 KB) fit neither registers nor LDS and stay where conv3x3s1_halo_bf16 has them: streamed from L2
// through a register ring of K-STEP slots (one 16-channel group of one tap; 72 per tile), here six slots deep
// because the workgroup is alone on its CU and has the registers.  What changes is everything around the
// MFMAs: the whole 128-channel halo range of a tile sits in LDS (272-byte rows), so a tile's 288 MFMAs per wave
// run without a barrier; the other group's epilogue, its residual rows and its next halo range overlap them;
// and the activation / residual loads of a group are issued AFTER the last weight load of its MFMA phase, so
// that no weight fragment queues behind an HBM access in the in-order load counter.  k order (32-channel chunk,
// tap, 16-channel group) is conv3x3s1_halo_bf16's: bit-identical results.
constexpr int S128_LDA = 136;                  // bf16 per LDS row: 128 channels + 16 B (272 B = 17 slots)
constexpr int S128_BM = 128;
constexpr int S128_WMAX = 12;
constexpr int S128_CBYTES = 3 * 128 * 4;
constexpr int S128_RING = 6;

#ifndef UMT_S128_MI
#define UMT_S128_MI 2     // 32-pixel tiles per wave: 2 -> waves 2 x 2 (64 px x 64 channels), 4 -> waves 1 x 4 (128 px x 32 channels)
#endif
template <bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_stream128_bf16(ThinArgs a) {
  constexpr int MI = UMT_S128_MI, NI = 4 / MI, WAVES_N = 4 / NI;
  constexpr int BM = S128_BM;
  constexpr int NLD = ((BM + 2 * S128_WMAX + 2) * 16 + 255) / 256;   // uint4 slots per thread for one halo range
  constexpr int R = S128_RING, NK = 72;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, group = threadIdx.x >> 8;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;              // a wave owns MI*32 pixels x NI*32 channels
  const int a_bytes = ((rows + 1) * S128_LDA * 2 + 15) & ~15;      // halo range + the all-zero row
  const float* sC = reinterpret_cast<const float*>(smem_s);
  unsigned char* sA = smem_s + S128_CBYTES + group * (a_bytes + 4 * 32 * T_LDP * 4);
  float* patch = reinterpret_cast<float*>(sA + a_bytes) + wave * 32 * T_LDP;
  const int li = lane & 31, lh = lane >> 5;
  const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);
  const uint16_t* resid = reinterpret_cast<const uint16_t*>(a.residual);
  uint16_t* out = reinterpret_cast<uint16_t*>(a.out);

  if (threadIdx.x < 128) {
    float* c = reinterpret_cast<float*>(smem_s);
    c[threadIdx.x] = a.bias[threadIdx.x];
    c[128 + threadIdx.x] = a.scale[threadIdx.x];
    c[256 + threadIdx.x] = a.shift[threadIdx.x];
  }
  const int q = lane & 3, rsub = lane >> 2;
  const int relu_floor = a.relu ? 0 : (int)0x80000000;

  const GroupWalk gw = group_walk(a.ntiles, group);
  const GroupWalk g0 = group_walk(a.ntiles, 0);
  const int nphase = 2 * g0.count + 1;

  // weight fragments: packed [tap][8 groups of 16 channels][2][128][8] bf16; scalar (tap, group) offset, one lane offset
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 9 * 128 * 128 * 2, 0x00020000);
  const int wvoff = ((lh * 128) + wn * NI * 32 + li) * 16;
  uint4 bk[R][NI];
  auto load_bk = [&](uint4 (&dst)[NI], int ks) {       // ks = chunk * 18 + tap * 2 + c
    const int chunk = ks / 18, tap = (ks % 18) >> 1, c16 = chunk * 2 + (ks & 1);
    const int soff = (tap * 8 + c16) * 4096;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + n * 512, soff, 0);
      dst[n] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };

  uint4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int tile) {
    const int gbase = tile * BM - W - 1;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 4, qq = idx & 15;
      const int g = gbase + r;
      const bool ok = r < rows && g >= 0 && g < a.M;
      const size_t off = ok ? (size_t)g * 128 + 8 * qq : 0;
      areg[k] = *reinterpret_cast<const uint4*>(in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      uint4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
      if ((idx >> 4) < rows) *reinterpret_cast<uint4*>(sA + ((idx >> 4) * S128_LDA + 8 * (idx & 15)) * 2) = v;
    }
  };

  if (tid < S128_LDA * 2 / 16) *reinterpret_cast<uint4*>(sA + rows * S128_LDA * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
  if (gw.count > 0) {
    load_a(gw.first);
    write_a();
  }
  __syncthreads();

  f32x16 acc[MI][NI];
  uint4 res[MI][NI][2];
  for (int ph = 0; ph < nphase; ++ph) {
    STAMP(ph, 0);
    const int j = (ph - group) >> 1;
    const bool mfma_phase = ((ph - group) & 1) == 0;
    if (ph >= group && j < gw.count) {
      const int t = gw.first + j * gw.step;
      const int mw = t * BM + wm * MI * 32;
      if (mfma_phase) {
#pragma unroll
        for (int s0 = 0; s0 < R - 1; ++s0) load_bk(bk[s0], s0);
        int arow[MI][9];      // byte offset of the lane's operand row per tap; the zero row where padded
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const unsigned mk = border_mask(mw + i * 32 + li, a.M, H, W, a.mg_hw, a.mg_w);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
            arow[i][tap] = ((((mk >> tap) & 1u) ? wm * MI * 32 + i * 32 + li + (tap / 3) * W + (tap % 3) : rows) * S128_LDA + 8 * lh) * 2;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
        // A fragments run one k-step ahead of their MFMAs: in an MFMA phase this wave is the only one on its SIMD that
        // issues matrix instructions (its partner is in the epilogue), so a read waited for in its own k-step leaves
        // the pipe idle for the whole LDS latency (measured: 236 cycles per k-step against 128 of MFMA)
        uint4 av[2][MI];
        auto read_a = [&](uint4 (&dst)[MI], int ks) {
          const int tap = (ks % 18) >> 1, c16 = (ks / 18) * 2 + (ks & 1);
#pragma unroll
          for (int i = 0; i < MI; ++i) dst[i] = *reinterpret_cast<const uint4*>(sA + arow[i][tap] + c16 * 32);
        };
        read_a(av[0], 0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          if (ks + R - 1 < NK) load_bk(bk[(ks + R - 1) % R], ks + R - 1);
          if (ks == NK - R + 1) {        // the last weight load of this tile has just been issued
            if (j + 1 < gw.count) load_a(t + gw.step);
            if (RES) {
#pragma unroll
              for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NI; ++n)
#pragma unroll
                  for (int k = 0; k < 2; ++k) {
                    const int m = mw + i * 32 + 16 * k + rsub;
                    const size_t off = m < a.M ? (size_t)m * 128 + wn * NI * 32 + n * 32 + 8 * q : 0;
                    res[i][n][k] = *reinterpret_cast<const uint4*>(resid + off);
                  }
            }
          }
          if (ks + 1 < NK) read_a(av[(ks + 1) & 1], ks + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NI; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ks & 1][i]),
                                                                  __builtin_bit_cast(bf16x8, bk[ks % R][n]), acc[i][n], 0, 0, 0);
        }
      } else {
        // the epilogue phase issues first (older waves win arbitration otherwise: group 1's epilogue, beside group 0's
        // MFMAs, took 17 k cycles against 12.7 k for group 0's): the MFMA group fills what it leaves
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const int co = wn * NI * 32 + n * 32 + 8 * q;
          const float4 cb0 = *reinterpret_cast<const float4*>(sC + co), cb1 = *reinterpret_cast<const float4*>(sC + co + 4);
          const float4 cs0 = *reinterpret_cast<const float4*>(sC + 128 + co), cs1 = *reinterpret_cast<const float4*>(sC + 128 + co + 4);
          const float4 ch0 = *reinterpret_cast<const float4*>(sC + 256 + co), ch1 = *reinterpret_cast<const float4*>(sC + 256 + co + 4);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * T_LDP + li] = acc[i][n][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int row = 16 * k + rsub;
              const int m = mw + i * 32 + row;
              const float4 v = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q);
              const float4 u = *reinterpret_cast<const float4*>(patch + row * T_LDP + 8 * q + 4);
              thin_u32x4 rr = {0, 0, 0, 0};
              if (RES) rr = thin_u32x4{res[i][n][k].x, res[i][n][k].y, res[i][n][k].z, res[i][n][k].w};
              const uint4 o = thin_bn_res_relu_pack8<RES>(v, u, cb0, cb1, cs0, cs1, ch0, ch1, rr, relu_floor);
              if (m < a.M) *reinterpret_cast<uint4*>(out + (size_t)m * 128 + co) = o;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
        STAMP(ph, 2);
        if (j + 1 < gw.count) write_a();
        __builtin_amdgcn_s_setprio(0);
      }
    }
    STAMP(ph, 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ fp32, 32 -> 32, weights in LDS
// The four 32 -> 32 layers of stage 1 in the fp32 (reference-parity) mode, same two-group scheme.  On
// conv3x3s1_halo_f32 they ran the matrix pipe 68 % busy (three waves per SIMD, each workgroup exposing its own halo
// fetch, residual fetch and epilogue).  Here the packed weights (36.9 KB of fp32) sit in LDS, both operands of
// every v_mfma_f32_32x32x2_f32 are ds_read_b128 (one A and one B fragment per four MFMAs), a tile's 144 MFMAs per
// wave run without a barrier or a global load, and the other group's epilogue, residual and next halo range
// overlap them.  k order (tap, 8-channel group, element) and the epilogue arithmetic are conv3x3s1_halo_f32's:
// bit-identical results (tools/conv_bench.hip and tests/test_gpu_edges.py hold them so).
constexpr int F32_LDA = 36;                    // floats per LDS row (144 B)
constexpr int F32_BM = 128;
constexpr int F32_WBYTES = 9 * 4 * 2 * 32 * 16;   // packed weights [9][4][2][32][4] fp32

template <bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_wlds32_f32(ThinArgs a) {
  constexpr int BM = F32_BM;
  constexpr int NLD = ((BM + 2 * 48 + 2) * 8 + 255) / 256;   // float4 slots per thread for one halo range
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, group = threadIdx.x >> 8;
  const int a_bytes = (rows + 1) * F32_LDA * 4;                  // halo range + the all-zero row (multiple of 16)
  unsigned char* sW = smem_f;
  unsigned char* sA = smem_f + F32_WBYTES + group * (a_bytes + 4 * 32 * F32_LDA * 4);
  float* patch = reinterpret_cast<float*>(sA + a_bytes) + wave * 32 * F32_LDA;
  const int li = lane & 31, lh = lane >> 5;
  const float* in = reinterpret_cast<const float*>(a.in);
  const float* resid = reinterpret_cast<const float*>(a.residual);
  float* out = reinterpret_cast<float*>(a.out);

  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    for (int i = threadIdx.x; i < F32_WBYTES / 16; i += 512) *reinterpret_cast<uint4*>(sW + 16 * i) = wp[i];
  }
  // epilogue: a lane owns 4 consecutive channels of a pixel row, 8 rows per pass
  const int q = lane & 7, rsub = lane >> 3;
  const float4 cb = *reinterpret_cast<const float4*>(a.bias + 4 * q);
  const float4 cs = *reinterpret_cast<const float4*>(a.scale + 4 * q);
  const float4 ch = *reinterpret_cast<const float4*>(a.shift + 4 * q);

  const GroupWalk gw = group_walk(a.ntiles, group);
  const GroupWalk g0 = group_walk(a.ntiles, 0);
  const int nphase = 2 * g0.count + 1;

  float4 areg[NLD];
  unsigned aok = 0;
  auto load_a = [&](int tile) {
    const int gbase = tile * BM - W - 1;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 3, qq = idx & 7;
      const int g = gbase + r;
      const bool ok = r < rows && g >= 0 && g < a.M;
      const size_t off = ok ? (size_t)g * 32 + 4 * qq : 0;
      areg[k] = *reinterpret_cast<const float4*>(in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + ((idx >> 3) * F32_LDA + 4 * (idx & 7)) * 4) = v;
    }
  };

  if (tid < F32_LDA / 4) *reinterpret_cast<float4*>(sA + (rows * F32_LDA + 4 * tid) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gw.count > 0) {
    load_a(gw.first);
    write_a();
  }
  __syncthreads();

  f32x16 acc;
  float4 res[4];
  for (int ph = 0; ph < nphase; ++ph) {
    const int j = (ph - group) >> 1;
    const bool mfma_phase = ((ph - group) & 1) == 0;
    if (ph >= group && j < gw.count) {
      const int t = gw.first + j * gw.step;
      const int mw = t * BM + wave * 32;
      if (mfma_phase) {
        if (j + 1 < gw.count) load_a(t + gw.step);
        if (RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int m = mw + 8 * k + rsub;
            const size_t off = m < a.M ? (size_t)m * 32 + 4 * q : 0;
            res[k] = *reinterpret_cast<const float4*>(resid + off);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        int arow[9];      // byte offset of the lane's operand row per tap; the zero row where padded
        {
          const unsigned mk = border_mask(mw + li, a.M, H, W, a.mg_hw, a.mg_w);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
            arow[tap] = ((((mk >> tap) & 1u) ? wave * 32 + li + (tap / 3) * W + (tap % 3) : rows) * F32_LDA + 4 * lh) * 4;
        }
        const unsigned char* bl = sW + (lh * 32 + li) * 16;      // + (tap * 4 + c8) * 1024
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 av = *reinterpret_cast<const float4*>(sA + arow[tap] + c * 32);
            const float4 bv = *reinterpret_cast<const float4*>(bl + (tap * 4 + c) * 1024);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * F32_LDA + li] = acc[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = 8 * k + rsub;
          const int m = mw + row;
          float4 v = *reinterpret_cast<const float4*>(patch + row * F32_LDA + 4 * q);
          v.x = (v.x + cb.x) * cs.x + ch.x;
          v.y = (v.y + cb.y) * cs.y + ch.y;
          v.z = (v.z + cb.z) * cs.z + ch.z;
          v.w = (v.w + cb.w) * cs.w + ch.w;
          if (RES) { v.x += res[k].x; v.y += res[k].y; v.z += res[k].z; v.w += res[k].w; }
          if (a.relu) {      // max(v, 0) as one integer max on the bits, as conv_f32_common.h
            v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
            v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
            v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
            v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
          }
          if (m < a.M) *reinterpret_cast<float4*>(out + (size_t)m * 32 + 4 * q) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (j + 1 < gw.count) write_a();
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ fp32, 32 -> 32, one role per wave
// conv3x3_wlds32_f32 with the two-group scheme taken out.  What tools/mfma_peak.hip (mix) showed about this chip: a wave
// that issues MFMAs back to back leaves the OTHER waves of its SIMD no vector-instruction slot at all (wave priority and
// age change nothing), while up to six vector instructions of the SAME wave per MFMA cost nothing.  So in the two-group
// kernel the epilogue group only ran while the MFMA group stalled, and a phase lasted the MFMA loop PLUS the other
// group's vector work (11.5 k cycles for 9.2 k of MFMAs).  Here every wave does everything for its own 32 pixels, in
// the shadow of its own MFMAs: the 36 k-steps (4 MFMAs = 256 matrix-pipe cycles each) of tile j carry, one piece per
// step, the LDS stores of the next tile's halo range, the epilogue of tile j-1 (accumulators kept in registers), the
// loads of the halo range after next and of tile j's residual rows, and the next tile's border arithmetic.  Eight
// waves = a 256-pixel tile, halo ranges double-buffered in LDS, one barrier per tile.  The MFMA operands are swapped
// (weights as the row operand): a lane owns 16 channels of ONE pixel, so the epilogue needs no transposing patch and
// stores four float4 per lane.  k order (tap, 8-channel group, element) and the epilogue arithmetic are
// conv3x3s1_halo_f32's; swapping the operands transposes the result tile and nothing else: bit-identical results.
constexpr int P32_BM = 256;
constexpr int P32_CBYTES = 3 * 32 * 4;

template <bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_pipe32_f32(ThinArgs a) {
  constexpr int BM = P32_BM;
  constexpr int NLD = ((BM + 2 * 48 + 2) * 8 + 511) / 512;   // float4 slots per thread for one halo range (6)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
  const int W = a.W, H = a.H;
  const int rows = BM + 2 * W + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a_bytes = (rows + 1) * F32_LDA * 4;                  // halo range + the all-zero row (multiple of 16)
  unsigned char* sW = smem_p;
  const float* sC = reinterpret_cast<const float*>(smem_p + F32_WBYTES);
  unsigned char* sA0 = smem_p + F32_WBYTES + P32_CBYTES;
  const int li = lane & 31, lh = lane >> 5;
  const float* in = reinterpret_cast<const float*>(a.in);
  const float* resid = reinterpret_cast<const float*>(a.residual);
  float* out = reinterpret_cast<float*>(a.out);
  const int relu_floor = a.relu ? 0 : (int)0x80000000;

  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    for (int i = tid; i < F32_WBYTES / 16; i += 512) *reinterpret_cast<uint4*>(sW + 16 * i) = wp[i];
    float* c = reinterpret_cast<float*>(smem_p + F32_WBYTES);
    if (tid < 32) {
      c[tid] = a.bias[tid];
      c[32 + tid] = a.scale[tid];
      c[64 + tid] = a.shift[tid];
    }
    if (tid < 2 * (F32_LDA / 4))       // the all-zero row of both buffers
      *reinterpret_cast<float4*>(sA0 + (tid / (F32_LDA / 4)) * a_bytes + (rows * F32_LDA + 4 * (tid % (F32_LDA / 4))) * 4) =
          make_float4(0.f, 0.f, 0.f, 0.f);
  }

  int t0, tstep, tend;
  tile_walk(a.ntiles, t0, tstep, tend);
  const int count = t0 < tend ? (tend - t0 + tstep - 1) / tstep : 0;

  // One halo range = rows [tile * BM - W - 1, + rows) clipped to [0, M), read through a buffer descriptor that covers
  // exactly that run: rows outside it (and everything, when there is no such tile) come back as zeros
  float4 areg[NLD];
  __amdgpu_buffer_rsrc_t a_rs;
  int a_shift = 0;
  auto load_a_begin = [&](int tile, bool have) {      // both workgroup-uniform
    const int gbase = tile * BM - W - 1;
    const int g_lo = gbase > 0 ? gbase : 0, g_hi = gbase + rows < a.M ? gbase + rows : a.M;
    a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)g_lo * 32, 0,
                                             have && g_hi > g_lo ? (g_hi - g_lo) * 128 : 0, 0x00020000);
    a_shift = (g_lo - gbase) * 128;
  };
  auto load_a_one = [&](int k) {
    areg[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (tid + 512 * k) * 16 - a_shift, 0, 0));
  };
  auto write_a_one = [&](unsigned char* sA, int k) {
    const int idx = tid + 512 * k;
    if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + ((idx >> 3) * F32_LDA + 4 * (idx & 7)) * 4) = areg[k];
  };
  // byte offset of the lane's operand row per tap (the zero row where padded) for the wave tile that starts at pixel row mw
  auto set_arow = [&](int (&ar)[9], int mw) {
    const unsigned mk = border_mask(mw + li, a.M, H, W, a.mg_hw, a.mg_w);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      ar[tap] = ((((mk >> tap) & 1u) ? wave * 32 + li + (tap / 3) * W + (tap % 3) : rows) * F32_LDA + 4 * lh) * 4;
  };
  auto relu1 = [&](float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), relu_floor)); };

  f32x16 acc, accp;
  float4 res[4];
  int arow[9], arow_n[9];
  int mw_p = 0;
  // channels 8g + 4lh .. + 3 of pixel row m: (acc + bias) * scale + shift (+ residual), ReLU, one float4 store
  auto epilogue_piece = [&](int g, int m) {
    const float* kc = sC + 8 * g + 4 * lh;
    const float4 cb = *reinterpret_cast<const float4*>(kc), cs = *reinterpret_cast<const float4*>(kc + 32),
                 ch = *reinterpret_cast<const float4*>(kc + 64);
    f32x2 lo = __builtin_elementwise_fma((f32x2){accp[4 * g], accp[4 * g + 1]} + (f32x2){cb.x, cb.y}, (f32x2){cs.x, cs.y},
                                         (f32x2){ch.x, ch.y});
    f32x2 hi = __builtin_elementwise_fma((f32x2){accp[4 * g + 2], accp[4 * g + 3]} + (f32x2){cb.z, cb.w}, (f32x2){cs.z, cs.w},
                                         (f32x2){ch.z, ch.w});
    if (RES) {
      lo += (f32x2){res[g].x, res[g].y};
      hi += (f32x2){res[g].z, res[g].w};
    }
    const float o0 = lo.x, o1 = lo.y, o2 = hi.x, o3 = hi.y;
    if (m < a.M) *reinterpret_cast<float4*>(out + (size_t)m * 32 + 8 * g + 4 * lh) = make_float4(relu1(o0), relu1(o1), relu1(o2), relu1(o3));
  };
  auto load_res_one = [&](int g, int m) {
    const size_t off = m < a.M ? (size_t)m * 32 + 8 * g + 4 * lh : 0;
    res[g] = *reinterpret_cast<const float4*>(resid + off);
  };

  if (count > 0) {
    load_a_begin(t0, true);
#pragma unroll
    for (int k = 0; k < NLD; ++k) load_a_one(k);
#pragma unroll
    for (int k = 0; k < NLD; ++k) write_a_one(sA0, k);
    load_a_begin(t0 + tstep, count > 1);
#pragma unroll
    for (int k = 0; k < NLD; ++k) load_a_one(k);
    set_arow(arow, t0 * BM + wave * 32);
  }
  __syncthreads();

  for (int j = 0; j < count; ++j) {
    const int t = t0 + j * tstep;
    const int mw = t * BM + wave * 32;
    const unsigned char* sA = sA0 + (j & 1) * a_bytes;
    unsigned char* sAn = sA0 + ((j + 1) & 1) * a_bytes;
    PSTAMP(j, 0);
    const bool have_next = j + 1 < count, have_prev = j > 0;
    const unsigned char* bl = sW + (lh * 32 + li) * 16;      // + (tap * 4 + c8) * 1024
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 av[2], bv[2];
    av[0] = *reinterpret_cast<const float4*>(sA + arow[0]);
    bv[0] = *reinterpret_cast<const float4*>(bl);
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      if (st + 1 < 36) {
        av[(st + 1) & 1] = *reinterpret_cast<const float4*>(sA + arow[(st + 1) >> 2] + ((st + 1) & 3) * 32);
        bv[(st + 1) & 1] = *reinterpret_cast<const float4*>(bl + (st + 1) * 1024);
      }
      // the piece of side work this step carries
      if (st < NLD) {
        if (have_next) write_a_one(sAn, st);                                   // halo range of tile j+1 -> the other buffer
      } else if (st < 6 + 16) {
        if (have_prev && ((st - 6) & 3) == 0) epilogue_piece((st - 6) >> 2, mw_p + li);   // tile j-1, one channel quad per 4 steps
      } else if (st == 22) {
        load_a_begin(t + 2 * tstep, j + 2 < count);
        load_a_one(0);
      } else if (st < 22 + NLD) {
        load_a_one(st - 22);
      } else if (st < 28 + 4) {
        if (RES) load_res_one(st - 28, mw + li);
      } else if (st == 32) {
        if (have_next) set_arow(arow_n, mw + tstep * BM);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float4 x = av[st & 1], y = bv[st & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, x.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, x.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, x.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, x.w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    PSTAMP(j, 2);
    accp = acc;
    mw_p = mw;
    if (have_next) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) arow[tap] = arow_n[tap];
    }
    PSTAMP(j, 1);
    __syncthreads();
  }
  if (count > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) epilogue_piece(g, mw_p + li);
  }
}

// ------------------------------------------------------------------------------ bf16, fused BasicBlock 32 -> 32 -> 32
// A whole stage-1 BasicBlock (lib/models/backbone_resnet.py:56-72, planes = 32, stride 1: conv1 + BN + ReLU, conv2 + BN +
// residual + ReLU) in one kernel, UMT_PREC_BF16.  Unfused, a block moves 735 KB per crop through HBM (conv1 reads x and
// writes t; conv2 reads t and x and writes y) and both launches are HBM-bound; here the workgroup keeps t on chip:
//   * a tile is eight output rows of one 48x48 image; the workgroup stages the twelve input rows around it (576 pixels,
//     rows outside the image as zeros), computes the ten rows of t that conv2 needs (480 pixels: 1.25x the tile, the
//     price of not exchanging halo rows between workgroups) into LDS as bf16 - rounded exactly as the unfused
//     kernel rounds them for HBM - and then the four output rows, taking the residual from the staged input;
//   * x is read once (1.3-2x with the halo rows, which neighbouring tiles on the same XCD share through L2) and y
//     written once: about 300-440 KB per crop and block;
//   * both weight tensors (2 x 18.4 KB) sit in LDS: with them in registers (2 x 72 per lane) the compiler had none
//     left to read operands ahead of the MFMAs, and the kernel ran at half the speed of the two separate launches;
//     one persistent 512-thread workgroup per CU (122 KB of LDS), the next tile's input rows prefetched into registers;
//   * both convolutions run with the MFMA operands swapped (weights as the row operand), so a lane owns a pixel's
//     channels and neither epilogue needs the fp32 transposing patch of the other kernels.
// k order (tap, 16-channel group) and epilogue arithmetic are those of conv3x3_thin32_bf16 / conv3x3s1_halo_bf16, and
// t rows outside the image are zeros as conv2's padding requires: bit-identical to the two separate launches.
constexpr int FB_W = 48, FB_H = 48, FB_R = 8;              // image size and output rows per tile
constexpr int FB_XPIX = (FB_R + 4) * FB_W;                 // 576 staged input pixels
constexpr int FB_TPIX = (FB_R + 2) * FB_W;                 // 480 intermediate pixels
constexpr int FB_OPIX = FB_R * FB_W;                       // 384 output pixels
constexpr int FB_XBYTES = (FB_XPIX + 1) * T_LDH * 2;       // + the all-zero row
constexpr int FB_TBYTES = (FB_TPIX + 1) * T_LDH * 2;
constexpr int FB_WBYTES = 9 * 2 * 2 * 32 * 16;             // one packed weight tensor
constexpr int FB_CBYTES = 6 * 32 * 4;                     // bias/scale/shift of both layers
constexpr int FB_LDS = FB_XBYTES + FB_TBYTES + 2 * FB_WBYTES + FB_CBYTES;
static_assert(FB_LDS <= 160 * 1024, "block32_fused_bf16: LDS budget");

struct FusedBlockArgs {
  const uint16_t* in;      // rows [n*48*48][32] bf16
  const uint16_t* w1;      // packed [9][2][2][32][8] bf16
  const uint16_t* w2;
  const float *bias1, *scale1, *shift1;
  const float *bias2, *scale2, *shift2;
  uint16_t* out;           // rows [n*48*48][32] bf16
  int ntiles;              // n * 6
};

// The 18 k-steps (tap, 16-channel group) of one 3x3 convolution on one or two 32-pixel tiles, all three operand
// fragments (weights, pixels of tile 0, pixels of tile 1) read from LDS FB_AHEAD steps before the MFMAs that use them and
// pinned there: read in the step that uses them, every MFMA waited a full LDS round trip (matrix pipe 29 % busy).
// (acc + bias) * scale + shift on four consecutive channels, as two packed adds and two packed fused multiply-adds (each
// component rounds exactly as the scalar add + fma the other bf16 kernels use); kc -> this lane's bias quad in LDS, the
// scale and shift quads follow 32 and 64 floats on (read where they are used: held in registers across the MFMAs they cost
// 48 registers and no time)
__device__ __forceinline__ void fb_bn4(const float* kc, float a0, float a1, float a2, float a3, float& v0, float& v1, float& v2,
                                       float& v3) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float4 cb = *reinterpret_cast<const float4*>(kc), cs = *reinterpret_cast<const float4*>(kc + 32),
               ch = *reinterpret_cast<const float4*>(kc + 64);
  const f32x2 lo = __builtin_elementwise_fma((f32x2){a0, a1} + (f32x2){cb.x, cb.y}, (f32x2){cs.x, cs.y}, (f32x2){ch.x, ch.y});
  const f32x2 hi = __builtin_elementwise_fma((f32x2){a2, a3} + (f32x2){cb.z, cb.w}, (f32x2){cs.z, cs.w}, (f32x2){ch.z, ch.w});
  v0 = lo.x; v1 = lo.y; v2 = hi.x; v3 = hi.y;
}

#ifndef FB_AHEAD
#define FB_AHEAD 2
#endif
template <bool TWO>
__device__ __forceinline__ void fb_taps(const unsigned char* wbase, const unsigned char* act, const int (&arow)[2][9],
                                        f32x16 (&acc)[2]) {
  constexpr int NS = 18, D = FB_AHEAD;
  uint4 bw[D + 1], r0[D + 1], r1[D + 1];
  auto rd = [&](int slot, int st) {            // st = tap * 2 + c
    bw[slot] = *reinterpret_cast<const uint4*>(wbase + st * 1024);
    r0[slot] = *reinterpret_cast<const uint4*>(act + arow[0][st >> 1] + (st & 1) * 32);
    if (TWO) r1[slot] = *reinterpret_cast<const uint4*>(act + arow[1][st >> 1] + (st & 1) * 32);
  };
#pragma unroll
  for (int st = 0; st < D; ++st) rd(st, st);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + D < NS) rd((st + D) % (D + 1), st + D);
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[st % (D + 1)]),
                                                     __builtin_bit_cast(bf16x8, r0[st % (D + 1)]), acc[0], 0, 0, 0);
    if (TWO)
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[st % (D + 1)]),
                                                       __builtin_bit_cast(bf16x8, r1[st % (D + 1)]), acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef UMT_FB_STAMP      // diagnostic builds only: phase time stamps of workgroup 3's second tile
__device__ unsigned long long g_fbstamp[16];
#define FBSTAMP(i) do { if (blockIdx.x == 3 && threadIdx.x == 0 && T == first_T + tstep) g_fbstamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FBSTAMP(i) do {} while (0)
#endif
__global__ __launch_bounds__(512, 1) void block32_fused_bf16(FusedBlockArgs a) {
  constexpr int NLD = (FB_XPIX * 4 + 511) / 512;      // 5 uint4 per thread for one input range
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* sX = smem_b;
  unsigned char* sT = smem_b + FB_XBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* sW1 = smem_b + FB_XBYTES + FB_TBYTES;
  unsigned char* sW2 = sW1 + FB_WBYTES;
  float* sK = reinterpret_cast<float*>(sW2 + FB_WBYTES);     // [bias1 | scale1 | shift1 | bias2 | scale2 | shift2] x 32
  const int li = lane & 31, lh = lane >> 5;

  if (tid < 32) {
    sK[tid] = a.bias1[tid]; sK[32 + tid] = a.scale1[tid]; sK[64 + tid] = a.shift1[tid];
    sK[96 + tid] = a.bias2[tid]; sK[128 + tid] = a.scale2[tid]; sK[160 + tid] = a.shift2[tid];
  }
  for (int i = tid; i < FB_WBYTES / 16; i += 512) {
    *reinterpret_cast<uint4*>(sW1 + 16 * i) = reinterpret_cast<const uint4*>(a.w1)[i];
    *reinterpret_cast<uint4*>(sW2 + 16 * i) = reinterpret_cast<const uint4*>(a.w2)[i];
  }
  const int boff = (lh * 32 + li) * 16;        // + (tap * 2 + c) * 1024

  int T, tstep, tend;
  tile_walk(a.ntiles, T, tstep, tend);

  uint4 areg[NLD];
  unsigned aok = 0;
  auto load_x = [&](int tile) {
    const int img = tile / (FB_H / FB_R), y0 = (tile - img * (FB_H / FB_R)) * FB_R;
    aok = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 512 * k;
      const int px = idx >> 2, qq = idx & 3;
      const int xr = px / FB_W;
      const int y = y0 - 2 + xr;
      const bool ok = px < FB_XPIX && y >= 0 && y < FB_H;
      const size_t off = ok ? ((size_t)(img * FB_H + y0 - 2) * FB_W + px) * 32 + 8 * qq : 0;
      areg[k] = *reinterpret_cast<const uint4*>(a.in + off);
      aok |= (ok ? 1u : 0u) << k;
    }
  };
  auto write_x = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 512 * k;
      uint4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_uint4(0, 0, 0, 0);
      if ((idx >> 2) < FB_XPIX) *reinterpret_cast<uint4*>(sX + ((idx >> 2) * T_LDH + 8 * (idx & 3)) * 2) = v;
    }
  };

  if (tid < T_LDH * 2 / 16) {
    *reinterpret_cast<uint4*>(sX + FB_XPIX * T_LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(sT + FB_TPIX * T_LDH * 2 + 16 * tid) = make_uint4(0, 0, 0, 0);
  }
  if (T < tend) {
    load_x(T);
    write_x();
  }
  __syncthreads();

  const int first_T = T;
  (void)first_T;
  for (; T < tend; T += tstep) {
    const int img = T / (FB_H / FB_R), y0 = (T - img * (FB_H / FB_R)) * FB_R;
    const bool has_next = T + tstep < tend;
    FBSTAMP(0);
    if (has_next) load_x(T + tstep);
    __builtin_amdgcn_sched_barrier(0);

    // ---- conv1: 15 tiles of 32 intermediate pixels; wave w takes w and w+8 TOGETHER (one weight read feeds both).
    // The MFMA operands are swapped (weights as the row operand, pixels as the column operand: the same two fragments,
    // the other way round), so the accumulator is t TRANSPOSED: a lane owns one pixel and the 16 channels
    // 8g + 4*lh + e.  Its BN + ReLU result goes to the pixel's LDS row as four 8-byte stores - no transposing patch.
    {
      const float* kc = sK + 4 * lh;     // constants of the lane's channels: read from LDS where they are used
      const int kt[2] = {wave, wave + 8};
      const bool two = kt[1] < FB_TPIX / 32;
      int arow[2][9];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = (i == 0 || two ? kt[i] : kt[0]) * 32 + li;
        const int tr = f / FB_W, col = f - tr * FB_W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const int xx = col + kx - 1;
          const bool ok = xx >= 0 && xx < FB_W;
          arow[i][tap] = ((ok ? (tr + ky) * FB_W + xx : FB_XPIX) * T_LDH + 8 * lh) * 2;
        }
      }
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      auto taps1 = [&](auto twoc) {        // two straight-line bodies: a branch inside the tap loop would pin the schedule
        constexpr bool TWO = decltype(twoc)::value;
        fb_taps<TWO>(sW1 + boff, sX, arow, acc);
      };
      FBSTAMP(1);
      if (two) taps1(std::true_type{});
      else taps1(std::false_type{});
      FBSTAMP(2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 1 && !two) break;
        const int ff = kt[i] * 32 + li;
        const int yt = y0 - 1 + ff / FB_W;              // image row of this intermediate pixel
        const bool pad = yt < 0 || yt >= FB_H;           // conv2's zero padding above and below the image
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v0, v1, v2, v3;
          fb_bn4(kc + 8 * g, acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], v0, v1, v2, v3);
          v0 = thin_relu(v0); v1 = thin_relu(v1); v2 = thin_relu(v2); v3 = thin_relu(v3);
          uint2 o = make_uint2(thin_pack_bf16x2(v0, v1), thin_pack_bf16x2(v2, v3));
          if (pad) o = make_uint2(0, 0);
          *reinterpret_cast<uint2*>(sT + (ff * T_LDH + 8 * g + 4 * lh) * 2) = o;
        }
      }
    }
    FBSTAMP(3);
    __syncthreads();
    FBSTAMP(4);

    // ---- conv2: 12 tiles of 32 output pixels; wave w takes (w+1)%8 and, for four of the waves, that + 8 (wave 7 had
    // one conv1 tile).  Operands swapped as in conv1: a lane owns one output pixel and the channels 8g + 4*lh + e, adds
    // the residual from the pixel's staged input row and stores four 8-byte pieces of the pixel's 64-byte row.
    {
      const float* kc = sK + 96 + 4 * lh;
      const int kt[2] = {(wave + 1) & 7, ((wave + 1) & 7) + 8};
      const bool two = kt[1] < FB_OPIX / 32;
      int arow[2][9];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = (i == 0 || two ? kt[i] : kt[0]) * 32 + li;
        const int orow = f / FB_W, col = f - orow * FB_W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const int xx = col + kx - 1;
          const bool ok = xx >= 0 && xx < FB_W;
          arow[i][tap] = ((ok ? (orow + ky) * FB_W + xx : FB_TPIX) * T_LDH + 8 * lh) * 2;
        }
      }
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      auto taps2 = [&](auto twoc) {
        constexpr bool TWO = decltype(twoc)::value;
        fb_taps<TWO>(sW2 + boff, sT, arow, acc);
      };
      FBSTAMP(5);
      if (two) taps2(std::true_type{});
      else taps2(std::false_type{});
      FBSTAMP(6);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 1 && !two) break;
        const int ff = kt[i] * 32 + li;                        // output pixel within the tile
        const size_t gp = (size_t)(img * FB_H + y0) * FB_W + ff;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 rr = *reinterpret_cast<const uint2*>(sX + ((ff + 2 * FB_W) * T_LDH + 8 * g + 4 * lh) * 2);   // the block input
          float v0, v1, v2, v3;
          fb_bn4(kc + 8 * g, acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], v0, v1, v2, v3);
          v0 += __builtin_bit_cast(float, rr.x << 16); v1 += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v2 += __builtin_bit_cast(float, rr.y << 16); v3 += __builtin_bit_cast(float, rr.y & 0xffff0000u);
          v0 = thin_relu(v0); v1 = thin_relu(v1); v2 = thin_relu(v2); v3 = thin_relu(v3);
          *reinterpret_cast<uint2*>(a.out + gp * 32 + 8 * g + 4 * lh) = make_uint2(thin_pack_bf16x2(v0, v1), thin_pack_bf16x2(v2, v3));
        }
      }
    }
    FBSTAMP(7);
    __syncthreads();          // every wave is done with sX and sT
    FBSTAMP(8);
    if (has_next) write_x();
    __syncthreads();
    FBSTAMP(9);
  }
}

#ifdef UMT_FB_STAMP
}  // namespace umt
extern "C" int umt_fb_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(umt::g_fbstamp), 16 * sizeof(unsigned long long));
}
namespace umt {
#endif
// ---------------------------------------------------------------------------------------- launch
// gfx950 only (umt_model_finalize checks the architecture): an MI355X has 256 CUs; a partition mode with fewer
// only means some persistent workgroups queue behind others.
static int cu_count() { return 256; }

static bool thin64_shape(const ConvLayer& L, int w) {
  return L.taps == 9 && L.stride == 1 && L.cin_s == 64 && L.cout_s == 64 && L.cout_w == 64 && w <= W64_WMAX;
}

static bool stream128_shape(const ConvLayer& L, int w) {
  return !(L.flags & CONV_NO_STREAM128) && L.taps == 9 && L.stride == 1 && L.cin_s == 128 && L.cout_s == 128 && L.cout_w == 128 && w <= S128_WMAX;
}

bool conv_thin_applicable(const ConvLayer& L, int w) {
  if ((L.flags & CONV_NO_THIN) || L.taps != 9 || L.stride != 1) return false;
  return (L.cin_s == 32 && L.cout_s == 32 && L.cout_w == 32 && w <= 48) || thin64_shape(L, w) || stream128_shape(L, w);
}

static int grid_for(int ntiles, int per_cu) {
  int g = cu_count() * per_cu;
  g -= g % 8;
  const int need = (ntiles + 7) / 8 * 8;        // at least one slot per XCD run
  return g < need ? g : need;
}

#ifndef UMT_THIN_TM
#define UMT_THIN_TM 2
#endif

static size_t thin_lds_bytes(int tm, int w) {
  return 2 * ((((size_t)(128 * tm + 2 * w + 2 + 1) * T_LDH * 2 + 15) & ~(size_t)15) + 4 * 32 * T_LDP * sizeof(float));
}
static size_t p64_lds_bytes(int w) { return (size_t)W64_WBYTES + W64_CBYTES + 2 * (size_t)(P64_BM + 2 * w + 2 + 2) * W64_LDA * 2; }
static_assert((size_t)W64_WBYTES + W64_CBYTES + 2 * (size_t)(P64_BM + 2 * W64_WMAX + 2 + 2) * W64_LDA * 2 <= 160 * 1024, "conv3x3_pipe64_bf16: LDS budget");
static size_t w64_lds_bytes(int w) {
  return (size_t)W64_WBYTES + W64_CBYTES +
         2 * ((((size_t)(W64_BM + 2 * w + 2 + 1) * W64_LDA * 2 + 15) & ~(size_t)15) + 4 * 32 * T_LDP * sizeof(float));
}
static_assert((size_t)W64_WBYTES + W64_CBYTES + 2 * ((((size_t)(W64_BM + 2 * W64_WMAX + 2 + 1) * W64_LDA * 2 + 15) & ~(size_t)15) +
                                                       4 * 32 * T_LDP * sizeof(float)) <= 160 * 1024,
              "conv3x3_wlds64_bf16: LDS budget");

static size_t s128_lds_bytes(int w) {
  return (size_t)S128_CBYTES + 2 * ((((size_t)(S128_BM + 2 * w + 2 + 1) * S128_LDA * 2 + 15) & ~(size_t)15) + 4 * 32 * T_LDP * sizeof(float));
}

bool block32_fused_applicable(const ConvLayer& c1, const ConvLayer& c2, int h, int w) {
  const bool off = (c1.flags & (CONV_NO_FUSE | CONV_NO_THIN)) != 0;
  auto ok = [](const ConvLayer& L) {
    return L.taps == 9 && L.stride == 1 && L.cin_s == 32 && L.cout_s == 32 && L.cout_w == 32 && L.wh != nullptr;
  };
  return !off && h == FB_H && w == FB_W && ok(c1) && ok(c2) && c1.relu && c2.relu;
}

int launch_block32_fused_bf16(const ConvLayer& c1, const ConvLayer& c2, const uint16_t* in, int n, uint16_t* out,
                              hipStream_t stream) {
  UMT_REQUIRE(block32_fused_applicable(c1, c2, FB_H, FB_W) && n > 0 && n < (1 << 20), UMT_ERR_INVALID, "block32_fused: bad arguments");
  FusedBlockArgs a;
  a.in = in; a.w1 = c1.wh; a.w2 = c2.wh; a.out = out;
  a.bias1 = c1.bias; a.scale1 = c1.scale; a.shift1 = c1.shift;
  a.bias2 = c2.bias; a.scale2 = c2.scale; a.shift2 = c2.shift;
  a.ntiles = n * (FB_H / FB_R);
  hipLaunchKernelGGL(block32_fused_bf16, dim3(grid_for(a.ntiles, 1)), dim3(512), FB_LDS, stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

static size_t f32_lds_bytes(int w) {
  return (size_t)F32_WBYTES + 2 * ((size_t)(F32_BM + 2 * w + 2 + 1) * F32_LDA * 4 + 4 * 32 * F32_LDA * sizeof(float));
}
static_assert((size_t)F32_WBYTES + 2 * ((size_t)(F32_BM + 2 * 48 + 2 + 1) * F32_LDA * 4 + 4 * 32 * F32_LDA * sizeof(float)) <= 160 * 1024,
              "conv3x3_wlds32_f32: LDS budget");

static size_t p32_lds_bytes(int w) { return (size_t)F32_WBYTES + P32_CBYTES + 2 * (size_t)(P32_BM + 2 * w + 2 + 1) * F32_LDA * 4; }
static_assert((size_t)F32_WBYTES + P32_CBYTES + 2 * (size_t)(P32_BM + 2 * 48 + 2 + 1) * F32_LDA * 4 <= 160 * 1024, "conv3x3_pipe32_f32: LDS budget");

bool conv_thin_f32_applicable(const ConvLayer& L, int w) {
  return !(L.flags & (CONV_NO_THIN_F32 | CONV_NO_THIN)) && L.taps == 9 && L.stride == 1 && L.cin_s == 32 && L.cout_s == 32 && L.cout_w == 32 && w <= 48;
}

int launch_conv_thin_f32(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual, float* out,
                         hipStream_t stream) {
  const int64_t M = (int64_t)n * h * w;
  UMT_REQUIRE(L.w, UMT_ERR_STATE, "conv_thin_f32: layer has no weights");
  UMT_REQUIRE(M > 0 && M < ((int64_t)1 << 31) - 4096, UMT_ERR_INVALID, "conv_thin_f32: %lld pixels", (long long)M);
  ThinArgs a;
  a.in = in; a.w = L.w; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift; a.residual = residual; a.out = out;
  a.H = h; a.W = w; a.M = (int)M; a.relu = L.relu;
  a.ntiles = (int)((M + F32_BM - 1) / F32_BM);
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
  if (!(L.flags & CONV_NO_PIPE32) && M < ((int64_t)1 << 31) - (1 << 20)) {      // (the walk looks one tile step past its last tile)
    a.ntiles = (int)((M + P32_BM - 1) / P32_BM);
    auto kp = residual ? conv3x3_pipe32_f32<true> : conv3x3_pipe32_f32<false>;
    hipLaunchKernelGGL(kp, dim3(grid_for(a.ntiles, 1)), dim3(512), p32_lds_bytes(w), stream, a);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  auto kern = residual ? conv3x3_wlds32_f32<true> : conv3x3_wlds32_f32<false>;
  hipLaunchKernelGGL(kern, dim3(grid_for(a.ntiles, 1)), dim3(512), f32_lds_bytes(w), stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv_thin_prepare_device() {      // largest request: W = 48
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(block32_fused_bf16), FB_LDS));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wlds32_f32<false>), f32_lds_bytes(48)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wlds32_f32<true>), f32_lds_bytes(48)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_pipe32_f32<false>), p32_lds_bytes(48)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_pipe32_f32<true>), p32_lds_bytes(48)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_stream128_bf16<false>), s128_lds_bytes(S128_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_stream128_bf16<true>), s128_lds_bytes(S128_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wlds64_bf16<false>), w64_lds_bytes(W64_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_wlds64_bf16<true>), w64_lds_bytes(W64_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_pipe64_bf16<false>), p64_lds_bytes(W64_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_pipe64_bf16<true>), p64_lds_bytes(W64_WMAX)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_thin32_bf16<UMT_THIN_TM, false>), thin_lds_bytes(UMT_THIN_TM, 48)));
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(conv3x3_thin32_bf16<UMT_THIN_TM, true>), thin_lds_bytes(UMT_THIN_TM, 48)));
  return 0;
}

int launch_conv_thin_bf16(const ConvLayer& L, const uint16_t* in, int n, int h, int w, const uint16_t* residual,
                          uint16_t* out, hipStream_t stream) {
  constexpr int TM = UMT_THIN_TM;
  const int64_t M = (int64_t)n * h * w;
  UMT_REQUIRE(L.wh, UMT_ERR_STATE, "conv_thin: layer has no bf16 weights");
  UMT_REQUIRE(M > 0 && M < ((int64_t)1 << 31) - 4096, UMT_ERR_INVALID, "conv_thin: %lld pixels", (long long)M);
  ThinArgs a;
  a.in = in; a.w = L.wh; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift; a.residual = residual; a.out = out;
  a.H = h; a.W = w; a.M = (int)M; a.relu = L.relu;
  if (stream128_shape(L, w)) {
    a.ntiles = (int)((M + S128_BM - 1) / S128_BM);
    a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
    a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
    auto kern128 = residual ? conv3x3_stream128_bf16<true> : conv3x3_stream128_bf16<false>;
    hipLaunchKernelGGL(kern128, dim3(grid_for(a.ntiles, 1)), dim3(512), s128_lds_bytes(w), stream, a);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (thin64_shape(L, w)) {
    a.ntiles = (int)((M + W64_BM - 1) / W64_BM);
    a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
    a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
    if (!(L.flags & CONV_NO_PIPE64) && M < ((int64_t)1 << 31) - (1 << 20)) {      // (the walk looks one tile step past its last tile)
      a.ntiles = (int)((M + P64_BM - 1) / P64_BM);
      auto kp = residual ? conv3x3_pipe64_bf16<true> : conv3x3_pipe64_bf16<false>;
      hipLaunchKernelGGL(kp, dim3(grid_for(a.ntiles, 1)), dim3(P64_THREADS), p64_lds_bytes(w), stream, a);
      UMT_CHECK_HIP(hipGetLastError());
      return 0;
    }
    auto kern64 = residual ? conv3x3_wlds64_bf16<true> : conv3x3_wlds64_bf16<false>;
    hipLaunchKernelGGL(kern64, dim3(grid_for(a.ntiles, 1)), dim3(512), w64_lds_bytes(w), stream, a);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  a.ntiles = (int)((M + 128 * TM - 1) / (128 * TM));
  a.mg_hw = (unsigned)(((uint64_t)1 << 32) / (uint64_t)(h * w)) + 1u;
  a.mg_w = (unsigned)(((uint64_t)1 << 32) / (uint64_t)w) + 1u;
  const size_t lds = thin_lds_bytes(TM, w);
  const dim3 grid(grid_for(a.ntiles, 1));
  auto kern = residual ? conv3x3_thin32_bf16<TM, true> : conv3x3_thin32_bf16<TM, false>;
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace umt
