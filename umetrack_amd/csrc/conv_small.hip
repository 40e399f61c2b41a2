// Stride-1 3x3 convolution for FEW pixels (the tracker's per-frame call: 4 crops -> 144 pixels at 6x6, 576 at 12x12;
// lib/tracker/tracker.py:141-168 calls the model on one frame's crops).  conv3x3s1_halo_f32 gives such a launch 4-20
// workgroups, and each wave walks its whole K loop on 32x32x2 matrix instructions: 64 cycles per 2 channels whatever
// the tile holds, 25-46 us per layer on an otherwise idle chip.  Here the tile is 16 pixels x 16 output channels per
// wave on v_mfma_f32_16x16x4_f32 (32 cycles per 4 channels): a quarter of the K-loop latency and 4x the workgroups.
//
// Arithmetic: the four k-slots of the 16x16x4 instruction take channels {0, 4, 1, 5} and then {2, 6, 3, 7} of every
// 8-channel group, which is the order in which four 32x32x2 instructions (lane half h = k: channels 8g + e, 8g + 4 + e
// for e = 0..3) add them; chunks of 32 channels outermost, then taps, then groups, as conv3x3s1_halo_f32; the same
// epilogue expression.  Results are bit-identical to that kernel's (tests/test_gpu_edges.py), as head_seq.hip's use of
// the same instruction is to the head's 32x32x2 kernels.
#include "conv_f32_common.h"
#include <type_traits>

namespace umt {

typedef float f32x4s __attribute__((ext_vector_type(4)));

constexpr int SM_BM = 16, SM_BN = 64;        // workgroup tile: 16 pixels x 64 output channels (4 waves x 16)

// TAPS = 9: stride-1 3x3; TAPS = 1: stride-1 1x1 (the backbone projection and the regressor's last layer of a per-frame call: no
// halo, one tap per chunk, bit-identical to conv_igemm_f32)
template <bool RES, int TAPS>
__global__ __launch_bounds__(256, 2) void conv3x3s1_small_f32(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sA[];   // (16 + 2W + 2) rows x LDA + one zero row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * SM_BM, n0w = blockIdx.y * SM_BN + wave * 16;
  // cout_w = 96 (the regressor's 3x3 layers): the last workgroup along cout has two waves without channels; they stage
  // and synchronise with the others, compute on channel 0's weights and store nothing
  const bool has_channels = n0w < a.cout_w;
  const int n0 = has_channels ? n0w : 0;
  const int ks = lane >> 4, lc = lane & 15;                    // k-slot; pixel (B operand) / output channel (A operand) of this lane
  const int W = a.W, H = a.H;
  const int rows = TAPS == 9 ? SM_BM + 2 * W + 2 : SM_BM;
  const int g0 = TAPS == 9 ? m0 - W - 1 : m0;                  // global pixel row of LDS row 0
  const int c8_total = a.cin_s >> 3;
  const int nchunks = (a.cin_s + KC - 1) / KC;
  const int perm = (ks & 1) * 4 + (ks >> 1);                   // channel of this k-slot inside an 8-channel group: 0, 4, 1, 5 (then + 2)

  // the lane's pixel and the LDS byte address of each of its 9 tap rows (the all-zero row where the neighbour does not exist)
  int arow[TAPS];
  if (TAPS == 9) {
    const int m = m0 + lc;
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
        arow[(ky * 3 + kx) % TAPS] = ((ok ? lc + ky * W + kx : rows) * LDA + perm) * 4;
      }
  } else {
    arow[0] = ((m0 + lc < a.M ? lc : rows) * LDA + perm) * 4;
  }
  if (tid < LDA / 4) *reinterpret_cast<float4*>(sA + rows * LDA + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);

  // weights: lane reads packed[tap][c8][h = ks & 1][n0 + lc][0..3] and uses elements (ks >> 1) and (ks >> 1) + 2, three
  // taps ahead of the MFMAs through a ring of four taps (a tap is 8 MFMAs = 0.3 k cycles, an L2 round trip two to three of
  // them; one tap ahead: 13.1 us per launch, three: 12.6).  (Measured and dropped: the whole next chunk in flight as 72
  // four-byte loads per lane - slower; half-width loads as a timing experiment - no change: not bound by the texture path.)
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)((((size_t)(ks & 1) * a.cout_w + n0 + lc) * 4) * sizeof(float));
  const size_t w_c8 = (size_t)2 * a.cout_w * 4 * sizeof(float), w_tap = (size_t)c8_total * w_c8;
  const bool hi = (ks >> 1) != 0;
  float2 bring[4][4];                                          // ring of taps, three ahead of the MFMAs
  auto load_b = [&](float2 (&dst)[4], int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c8 = min(chunk * 4 + c, c8_total - 1);
      const f32x4v v = __builtin_bit_cast(f32x4v, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (int)(tap * w_tap + c8 * w_c8), 0));
      dst[c] = hi ? make_float2(v.y, v.w) : make_float2(v.x, v.z);
    }
  };
#pragma unroll
  for (int i = 0; i < 3; ++i) load_b(bring[i], i % TAPS, min(i / TAPS, nchunks - 1));

  // epilogue constants of the lane's four output channels n0 + 4 ks + {0..3}
  const int co = n0 + 4 * ks;
  const int coc = min(co, a.cout_w - 4);
  const float4 cb = *reinterpret_cast<const float4*>(a.bias + coc), cs = *reinterpret_cast<const float4*>(a.scale + coc),
               ch = *reinterpret_cast<const float4*>(a.shift + coc);
  const int mout = m0 + lc;
  float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (RES && has_channels && mout < a.M && co < a.cout_s) rr = *reinterpret_cast<const float4*>(a.residual + (size_t)mout * a.cout_s + co);

  // A staging as conv3x3s1_halo_f32: slot idx = tid + 256 k (row idx >> 3, channel quad idx & 7), next chunk prefetched
  constexpr int NLD = (SM_BM + 2 * 48 + 2) * 8 / 256 + 1;      // W <= 48
  // two register sets: a chunk's rows are requested two chunks before they are staged (a chunk is 72 MFMAs, about 1.2 us:
  // less than the trip to another XCD's L2 where the previous launch left them)
  float4 areg[2][NLD];
  unsigned aok[2] = {0, 0};
  auto load_a = [&](float4 (&dst)[NLD], unsigned& okm, int chunk) {      // past the last chunk: every slot reads offset 0
    const int c0 = chunk * KC;
    okm = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx >> 3, q = idx & 7;
      const int g = g0 + r;
      const bool ok = r < rows && g >= 0 && g < a.M && (c0 + 4 * q) < a.cin_s;
      const size_t off = ok ? (size_t)g * a.cin_s + c0 + 4 * q : 0;
      dst[k] = *reinterpret_cast<const float4*>(a.in + off);
      okm |= (ok ? 1u : 0u) << k;
    }
  };
  load_a(areg[0], aok[0], 0);
  load_a(areg[1], aok[1], 1);

  f32x4s acc = {0.f, 0.f, 0.f, 0.f};
  const char* sAb = reinterpret_cast<const char*>(sA);
  auto step = [&](float4 (&ar)[NLD], unsigned& okm, int chunk, auto parc) {
    constexpr int PAR = decltype(parc)::value;                 // chunk & 3 (9 taps per chunk: the ring slot of a tap depends on it)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = ar[k];
      if (!((okm >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + (idx >> 3) * LDA + 4 * (idx & 7)) = v;
    }
    __syncthreads();
    // the eight operand values of a tap are read one tap ahead of its MFMAs (the chain is dependent: a read waited for in
    // its own tap leaves the matrix pipe idle for the LDS latency, nine times per chunk)
    float pa[2][8];
    auto read_tap = [&](float (&dst)[8], int tap) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dst[2 * c] = *reinterpret_cast<const float*>(sAb + arow[tap] + c * 32);
        dst[2 * c + 1] = *reinterpret_cast<const float*>(sAb + arow[tap] + c * 32 + 8);
      }
    };
    read_tap(pa[0], 0);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      {   // global tap index T = TAPS chunk + tap; slot T % 4 (chunk & 3 is fixed per call of step: PAR)
        const int T3 = tap + 3;
        load_b(bring[(PAR * TAPS + T3) & 3], T3 % TAPS, min(chunk + T3 / TAPS, nchunks - 1));   // (past the last chunk: a harmless re-read)
      }
      if (tap == TAPS - 1) load_a(ar, okm, chunk + 2);          // behind the chunk's last weight load
      if (tap + 1 < TAPS) read_tap(pa[(tap + 1) & 1], tap + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (chunk * 4 + c >= c8_total) continue;               // zero channel groups past the end (cin_s = 80): skipped, as the halo kernel's one-tile form
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(PAR * TAPS + tap) & 3][c].x, pa[tap & 1][2 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(PAR * TAPS + tap) & 3][c].y, pa[tap & 1][2 * c + 1], acc, 0, 0, 0);
      }
    }
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunks; chunk += 4) {
    step(areg[0], aok[0], chunk, std::integral_constant<int, 0>());
    if (chunk + 1 < nchunks) step(areg[1], aok[1], chunk + 1, std::integral_constant<int, 1>());
    if (chunk + 2 < nchunks) step(areg[0], aok[0], chunk + 2, std::integral_constant<int, 2>());
    if (chunk + 3 < nchunks) step(areg[1], aok[1], chunk + 3, std::integral_constant<int, 3>());
  }

  if (has_channels && mout < a.M && co < a.cout_s) {
    float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    v.x = (v.x + cb.x) * cs.x + ch.x;
    v.y = (v.y + cb.y) * cs.y + ch.y;
    v.z = (v.z + cb.z) * cs.z + ch.z;
    v.w = (v.w + cb.w) * cs.w + ch.w;
    if (RES) { v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
    if (a.relu) {
      v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
      v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
      v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
      v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
    }
    *reinterpret_cast<float4*>(a.out + (size_t)mout * a.cout_s + co) = v;
  }
}

bool conv_small_applicable(const ConvArgs& a, bool force_off) {
  return !force_off && (a.taps == 9 || a.taps == 1) && a.stride == 1 && a.W <= 48 && a.cout_w % 32 == 0 && a.cin_s % 8 == 0 &&
         (a.taps == 9 || !a.residual);
}

void launch_conv_small(const ConvArgs& a, hipStream_t stream) {
  dim3 grid((a.M + SM_BM - 1) / SM_BM, (a.cout_w + SM_BN - 1) / SM_BN);
  const int rows = a.taps == 9 ? SM_BM + 2 * a.W + 2 : SM_BM;
  const size_t lds = (size_t)(rows + 1) * LDA * sizeof(float);
  if (a.taps == 1)
    hipLaunchKernelGGL((conv3x3s1_small_f32<false, 1>), grid, dim3(256), lds, stream, a);
  else if (a.residual)
    hipLaunchKernelGGL((conv3x3s1_small_f32<true, 9>), grid, dim3(256), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3x3s1_small_f32<false, 9>), grid, dim3(256), lds, stream, a);
}

// ------------------------------------------------------------------------------ stride 2 + down-sample, few pixels
// The three fused stride-2 launches of a per-frame call (conv1 3x3 stride 2 + the 1x1 stride-2 down-sample branch) on
// the same 16 x 16 x 4 tiles: 16 output pixels x 64 output channels per workgroup, the input range of those 16 pixels
// (93 rows at W = 12, 177 at W = 48) staged per 32-channel chunk, the centre tap also feeding the down-sample accumulator.  On
// conv3x3s2p_ds_f32 such a launch is 3-36 tiles of 64 pixels, each a chain of 160-640 32x32x2 instructions: 20 us.
// k order per output (chunk, tap, 8-channel group, element; the down-sample product of a chunk at its centre tap) and
// both epilogues are that kernel's: bit-identical (tests/test_gpu_edges.py).
constexpr int SM2_ROWS_MAX = 4 * 48 + 4;

template <bool DS>
__global__ __launch_bounds__(256, 2) void conv3x3s2_small_f32(ConvS2Args g) {
  const ConvArgs& a = g.c1;
  extern __shared__ __attribute__((aligned(16))) float sA[];   // rows x LDA + one zero row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * SM_BM, n0 = blockIdx.y * SM_BN + wave * 16;
  const int ks = lane >> 4, lc = lane & 15;
  const int W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int c8_total = a.cin_s >> 3;
  const int nchunks = (a.cin_s + KC - 1) / KC;
  const int perm = (ks & 1) * 4 + (ks >> 1);

  auto centre = [&](int m) {                                   // mg_w / mg_hw divide by Wo / Ho in the stride-2 launches
    const int Rr = div_magic32(m, Wo, a.mg_w), xo = m - Rr * Wo;
    return 2 * W * Rr + 2 * xo;
  };
  const int m_last = min(m0 + SM_BM, a.M) - 1;
  const int g0 = centre(m0) - W - 1;
  const int rows = centre(m_last) + W + 1 - g0 + 1;            // <= s2_small_rows_cap(W) (host)
  const int zrow = rows;

  int arow[9];
  {
    const int m = m0 + lc;
    const bool live = m < a.M;
    const int mm = live ? m : m0;
    const int Rr = div_magic32(mm, Wo, a.mg_w), xo = mm - Rr * Wo, yo = Rr - div_magic32(Rr, Ho, a.mg_hw) * Ho;
    const int base = 2 * W * Rr + 2 * xo - g0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
        const bool ok = live && yy >= 0 && yy < a.H && xx >= 0 && xx < W;
        arow[ky * 3 + kx] = ((ok ? base + (ky - 1) * W + (kx - 1) : zrow) * LDA + perm) * 4;
      }
  }
  if (tid < LDA / 4) *reinterpret_cast<float4*>(sA + zrow * LDA + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DS ? g.ds_w : a.w), 0, 0x7fffffff, 0x00020000);
  const int wvoff = (int)((((size_t)(ks & 1) * a.cout_w + n0 + lc) * 4) * sizeof(float));
  const size_t w_c8 = (size_t)2 * a.cout_w * 4 * sizeof(float), w_tap = (size_t)c8_total * w_c8;
  const bool hi = (ks >> 1) != 0;
  float2 bcur[4], bnext[4], bds[4];
  auto load_b = [&](float2 (&dst)[4], __amdgpu_buffer_rsrc_t rs, int tap, int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c8 = min(chunk * 4 + c, c8_total - 1);
      const f32x4v v = __builtin_bit_cast(f32x4v, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs, wvoff, (int)(tap * w_tap + c8 * w_c8), 0));
      dst[c] = hi ? make_float2(v.y, v.w) : make_float2(v.x, v.z);
    }
  };
  load_b(bcur, wrs, 0, 0);

  const int co = n0 + 4 * ks;
  const int coc = min(co, a.cout_w - 4);
  const float4 cb = *reinterpret_cast<const float4*>(a.bias + coc), cs = *reinterpret_cast<const float4*>(a.scale + coc),
               ch = *reinterpret_cast<const float4*>(a.shift + coc);
  float4 db = cb, dsc = cs, dh = ch;
  if (DS) {
    db = *reinterpret_cast<const float4*>(g.ds_bias + coc);
    dsc = *reinterpret_cast<const float4*>(g.ds_scale + coc);
    dh = *reinterpret_cast<const float4*>(g.ds_shift + coc);
  }
  const int mout = m0 + lc;

  constexpr int NLD = (SM2_ROWS_MAX * 8 + 255) / 256;          // float4 slots per thread
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

  f32x4s acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
  const char* sAb = reinterpret_cast<const char*>(sA);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = tid + 256 * k;
      float4 v = areg[k];
      if (!((aok >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((idx >> 3) < rows) *reinterpret_cast<float4*>(sA + (idx >> 3) * LDA + 4 * (idx & 7)) = v;
    }
    __syncthreads();
    float pa[2][8];
    auto read_tap = [&](float (&dst)[8], int tap) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dst[2 * c] = *reinterpret_cast<const float*>(sAb + arow[tap] + c * 32);
        dst[2 * c + 1] = *reinterpret_cast<const float*>(sAb + arow[tap] + c * 32 + 8);
      }
    };
    read_tap(pa[0], 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) load_b(bnext, wrs, tap + 1, chunk);
      else if (chunk + 1 < nchunks) load_b(bnext, wrs, 0, chunk + 1);
      if (DS && tap == 2) load_b(bds, drs, 0, chunk);
      if (tap == 0 && chunk + 1 < nchunks) load_a(chunk + 1);
      if (tap < 8) read_tap(pa[(tap + 1) & 1], tap + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (chunk * 4 + c >= c8_total) continue;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[c].x, pa[tap & 1][2 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[c].y, pa[tap & 1][2 * c + 1], acc, 0, 0, 0);
        if (DS && tap == 4) {
          accd = __builtin_amdgcn_mfma_f32_16x16x4f32(bds[c].x, pa[tap & 1][2 * c], accd, 0, 0, 0);
          accd = __builtin_amdgcn_mfma_f32_16x16x4f32(bds[c].y, pa[tap & 1][2 * c + 1], accd, 0, 0, 0);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) bcur[c] = bnext[c];
    }
    __syncthreads();
  }

  if (mout < a.M && co < a.cout_s) {
    float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    v.x = (v.x + cb.x) * cs.x + ch.x;
    v.y = (v.y + cb.y) * cs.y + ch.y;
    v.z = (v.z + cb.z) * cs.z + ch.z;
    v.w = (v.w + cb.w) * cs.w + ch.w;
    if (a.relu) {
      v.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.x), 0));
      v.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.y), 0));
      v.z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.z), 0));
      v.w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v.w), 0));
    }
    *reinterpret_cast<float4*>(a.out + (size_t)mout * a.cout_s + co) = v;
    if (DS) {                                                  // no ReLU on the down-sample branch
      float4 d = make_float4(accd[0], accd[1], accd[2], accd[3]);
      d.x = (d.x + db.x) * dsc.x + dh.x;
      d.y = (d.y + db.y) * dsc.y + dh.y;
      d.z = (d.z + db.z) * dsc.z + dh.z;
      d.w = (d.w + db.w) * dsc.w + dh.w;
      *reinterpret_cast<float4*>(g.ds_out + (size_t)mout * a.cout_s + co) = d;
    }
  }
}

// most staged rows a 16-pixel tile can need: from the first pixel's top-left tap to the last pixel's bottom-right tap
static int s2_small_rows_cap(int w) {
  const int wo = w / 2;
  int worst = 0;
  for (int xo = 0; xo < wo; ++xo) {
    const int d = 2 * w * ((xo + SM_BM - 1) / wo) + 2 * ((xo + SM_BM - 1) % wo - xo);
    worst = d > worst ? d : worst;
  }
  return worst + 2 * w + 3;
}

bool conv_s2_small_applicable(const ConvS2Args& g, bool force_off) {
  const ConvArgs& a = g.c1;
  return !force_off && a.W <= 48 && a.W >= 2 && a.cout_w % SM_BN == 0 && a.cin_s % 8 == 0 &&
         (int64_t)a.M * (a.cout_w / SM_BN) <= 64 * 64 * 4 && s2_small_rows_cap(a.W) <= SM2_ROWS_MAX;
}

void launch_conv_s2_small(const ConvS2Args& g, hipStream_t stream) {
  const ConvArgs& a = g.c1;
  dim3 grid((a.M + SM_BM - 1) / SM_BM, a.cout_w / SM_BN);
  const size_t lds = (size_t)(s2_small_rows_cap(a.W) + 1) * LDA * sizeof(float);      // <= 28.4 KB
  if (g.ds_w)
    hipLaunchKernelGGL(conv3x3s2_small_f32<true>, grid, dim3(256), lds, stream, g);
  else
    hipLaunchKernelGGL(conv3x3s2_small_f32<false>, grid, dim3(256), lds, stream, g);
}

}  // namespace umt
