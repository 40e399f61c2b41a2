// Depth-first stage 1 of the bf16 backbone (lib/models/backbone_resnet.py:14-72, 125-153 with planes = 32, and the
// first block of stage 2: conv1 3x3 stride 2 + the 1x1 stride-2 down-sample branch):
//
//   x0 --c1a--> t1 --c2a(+x0)--> x1 --c1b--> t2 --c2b(+x1)--> x2 --s2--> t (24x24x64), d (24x24x64)
//
// in ONE launch.  As separate launches (block32_fused_bf16 x 2 + conv3x3s2_ds_bf16) these layers moved x1 and x2
// through HBM and spent more time in barriers, epilogues and halo recomputation than in MFMAs.  Here a workgroup
// owns a contiguous run of crops and streams them top to bottom, two pixel rows per step, as a skewed pipeline of
// ROLES, one wave each:
//
//   wave 5      loader   x0 rows 2s+14, 2s+15           HBM -> ring X0 (two steps of loads in flight)
//   wave 0      c1a      t1 rows 2s+11, 2s+12           X0 -> T1
//   wave 1      c2a      x1 rows 2s+8,  2s+9   (+ x0)   T1 -> X1
//   wave 2      c1b      t2 rows 2s+5,  2s+6            X1 -> T2
//   wave 3      c2b      x2 rows 2s+2,  2s+3   (+ x1)   T2 -> X2
//   waves 4, 6  s2       output row s of t (3x3 stride 2), 32 output channels each      X2 -> HBM
//   wave 7      ds       output row s of d (1x1 stride 2), 64 output channels           X2 -> HBM
//
// (row numbers count through all crops of the workgroup: crop c's rows are 48c .. 48c+47; waves w and w + 4 share a
// SIMD).  Every role reads only rows the role before it finished in the PREVIOUS step, so one workgroup barrier per
// step is the only synchronisation; the skew of three rows per layer is what a 3x3 convolution needs to have its lower
// neighbour row.  Each tensor lives in a ring of 6 or 8 rows in LDS (a row = [zero pixel | 48 pixels | zero pixel] x
// 80 B, so the x padding costs no test; 34 rows + a zero row for the y padding + a dump row = 141 KB); nothing but x0
// is read from HBM and nothing but t and d is written (147 + 147 KB per crop instead of 1.0 MB).  A role's weights (18
// packed uint4 per lane = 72 registers) and its BN constants (48) stay in registers for the life of the wave, so a tap
// costs one ds_read_b128 and one MFMA; ring slots and row validity are scalar arithmetic, advanced per step.
//
// What the schedule inside a role is built around (tools/sd_stamps.py, s_memtime stamps per role and step):
//  * a wave issues in order, and another wave's vector instructions do not issue while it streams MFMAs, so side work
//    only hides behind the SAME wave's MFMAs: the epilogues of tiles 0 and 1 are cut into eight pieces of four channels,
//    one behind every second MFMA of tile 2's chain (residual quads read one piece ahead);
//  * one dependent MFMA chain fed from LDS issues every ~64 cycles, two interleaved chains every ~35-40 per MFMA: tiles 0
//    and 1 run interleaved, one weight fragment feeding both;
//  * __syncthreads() drains the vector-memory counter as well: with it the loader's loads in flight made every step wait
//    an HBM round trip (1.41 ms per 4096 crops, against 0.83 ms with the LDS-only barrier below);
//  * per 4096 crops: 1.30 ms as three launches -> 0.83 ms here.
//
// Arithmetic: MFMA operands swapped as in block32_fused_bf16 (weights as the row operand: a lane owns one pixel and 16
// of its channels), k order (tap, 16-channel group) and the epilogue (conv bias, BN, residual, ReLU, one rounding to
// bf16) are those of conv3x3_thin32_bf16 / block32_fused_bf16 / conv3x3s2_ds_bf16: results are BIT-IDENTICAL to the
// three separate launches (tests/test_gpu_bf16.py, UMT_CONV_NODEPTH).
#include "umt_internal.h"

namespace umt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float sd_f32x2 __attribute__((ext_vector_type(2)));

constexpr int SD_W = 48, SD_H = 48;
constexpr int SD_PIXB = 80;                       // bytes per pixel in LDS: 64 B of channels + 16 B (conflict-free b128 reads)
constexpr int SD_ROWB = (SD_W + 2) * SD_PIXB;     // 4000: [zero pixel | 48 pixels | zero pixel] - the x padding is in the row
constexpr int SD_RX0 = 8, SD_RT1 = 6, SD_RX1 = 8, SD_RT2 = 6, SD_RX2 = 6;      // rows per ring (windows 8, 6, 8, 6, 5)
constexpr int SD_X0 = 0, SD_T1 = SD_X0 + SD_RX0 * SD_ROWB, SD_X1 = SD_T1 + SD_RT1 * SD_ROWB, SD_T2 = SD_X1 + SD_RX1 * SD_ROWB,
              SD_X2 = SD_T2 + SD_RT2 * SD_ROWB;
constexpr int SD_ZERO = SD_X2 + SD_RX2 * SD_ROWB; // one all-zero row: the y padding, and what idle lanes read
constexpr int SD_DUMP = SD_ZERO + SD_ROWB;        // where lanes without a real pixel store (no branch around a store)
constexpr int SD_KOFF = SD_DUMP + SD_ROWB;        // constants: four 32-channel layers x (bias | scale | shift), then the two
constexpr int SD_KFLOATS = 4 * 96 + 2 * 192;      // 64-channel stride-2 layers x (bias | scale | shift)
constexpr int SD_LDS = SD_KOFF + SD_KFLOATS * 4;
static_assert(SD_LDS <= 160 * 1024, "stage1_depth_bf16: LDS budget");
constexpr int SD_WAVES = 8, SD_THREADS = SD_WAVES * 64;
constexpr int SD_LEAD = 7;                        // steps before output row 0: the loader starts at row 2 * (-7) + 14 = 0
constexpr int SD_A_LOAD = 14;                     // band offsets of the roles (see the table above)

struct Stage1Args {
  const uint16_t* in;                             // x0 rows [n*48*48][32] bf16
  const uint16_t* w[4];                           // c1a, c2a, c1b, c2b: packed [9][2][2][32][8] bf16
  const float* bias[4];
  const float* scale[4];
  const float* shift[4];
  const uint16_t* w_s2;                           // 3x3 stride 2, 32 -> 64: packed [9][2][2][64][8]
  const float *b_s2, *s_s2, *h_s2;
  const uint16_t* w_ds;                           // 1x1 stride 2, 32 -> 64: packed [1][2][2][64][8]
  const float *b_ds, *s_ds, *h_ds;
  uint16_t* out_t;                                // rows [n*24*24][64] bf16
  uint16_t* out_d;
  int n;
};

// The step barrier.  __syncthreads() would also wait for this wave's GLOBAL loads and stores (its fence drains vmcnt):
// the loader keeps two steps of HBM loads in flight across it and the stride-2 waves their output stores, so only the
// LDS counter is drained here - a ring row is visible to the other waves once its ds_write has completed.
__device__ __forceinline__ void sd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef UMT_SD_STAMP      // diagnostic builds only (tools/sd_stamps.py): cycle stamps of every role of workgroup 3 at one step
__device__ unsigned long long g_sdstamp[8][4];
#define SDSTAMP(i) do { if (blockIdx.x == 3 && s == 100 && (threadIdx.x & 63) == 0) g_sdstamp[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SDSTAMP(i) do {} while (0)
#endif

__device__ __forceinline__ float sd_relu(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }
__device__ __forceinline__ unsigned sd_pack_bf16x2(float x, float y) {
  const __bf16 p = (__bf16)x, r = (__bf16)y;
  return (unsigned)__builtin_bit_cast(unsigned short, p) | ((unsigned)__builtin_bit_cast(unsigned short, r) << 16);
}
// v in [-R, 2R) -> v mod R
__device__ __forceinline__ int sd_wrap(int v, int R) { return v < 0 ? v + R : (v >= R ? v - R : v); }

// BN (+ residual) + ReLU + one rounding of four consecutive channels: (acc + bias) * scale + shift as two packed adds
// and two packed fused multiply-adds, each component rounding exactly as the scalar add + fma of the other bf16 kernels
// (fb_bn4 in conv_thin.hip), then the residual, max(x, 0) on the bit patterns, and the conversion.
typedef __bf16 sd_bf16x2 __attribute__((ext_vector_type(2)));
typedef int sd_i32x2 __attribute__((ext_vector_type(2)));
template <bool RES, bool RELU>
__device__ __forceinline__ uint2 sd_piece(const f32x16& acc, int g, const float4& cb, const float4& cs, const float4& ch, uint2 rr) {
  sd_f32x2 p[2] = {__builtin_elementwise_fma((sd_f32x2){acc[4 * g], acc[4 * g + 1]} + (sd_f32x2){cb.x, cb.y}, (sd_f32x2){cs.x, cs.y},
                                             (sd_f32x2){ch.x, ch.y}),
                   __builtin_elementwise_fma((sd_f32x2){acc[4 * g + 2], acc[4 * g + 3]} + (sd_f32x2){cb.z, cb.w}, (sd_f32x2){cs.z, cs.w},
                                             (sd_f32x2){ch.z, ch.w})};
  const unsigned rw[2] = {rr.x, rr.y};
  unsigned o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (RES) p[i] += (sd_f32x2){__builtin_bit_cast(float, rw[i] << 16), __builtin_bit_cast(float, rw[i] & 0xffff0000u)};
    // ReLU and conversion on the PAIR (as thin_bn_res_relu_pack8 in conv_thin.hip: one v_cvt_pk_bf16_f32 per pair)
    if (RELU) p[i] = __builtin_bit_cast(sd_f32x2, __builtin_elementwise_max(__builtin_bit_cast(sd_i32x2, p[i]), (sd_i32x2){0, 0}));
    o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p[i], sd_bf16x2));
  }
  return make_uint2(o[0], o[1]);
}

// The 18 k-steps (tap, 16-channel group) of one 3x3 convolution on one 32-pixel tile: weight fragments from registers,
// pixel fragments from LDS SD_AHEAD steps ahead of the MFMA that uses them.  ra[ky] = LDS address of the lane's pixel in the
// ring row of tap row ky, at the column of tap kx = 0; kx and the channel group are immediate offsets.  `side(st)` is
// issued behind MFMA st (inside its shadow; everything stays where it is written: scheduling fences on both sides).
// XS: pixels per kx step (1: neighbouring taps are neighbouring pixels, also for the stride-2 layer).  EXTRA: the centre tap also feeds `accd` with the down-sample weights.
#ifndef SD_AHEAD
#define SD_AHEAD 3      // pixel fragments are read this many MFMAs ahead (single chain: ~64 cycles per MFMA)
#endif
#ifndef SD_AHEAD2
#define SD_AHEAD2 3     // ... and this many MFMA pairs ahead in the two-chain form
#endif
template <int XS, bool EXTRA, class Side>
__device__ __forceinline__ void sd_chain(const uint4 (&wr)[18], const uint4 (&wd)[2], const unsigned char* sm, const int (&ra)[3],
                                         f32x16& acc, f32x16& accd, Side side) {
  constexpr int NS = 18, D = SD_AHEAD;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 p[D + 1];
  auto rd = [&](int st) {
    const int tap = st >> 1;
    return *reinterpret_cast<const uint4*>(sm + ra[tap / 3] + (tap % 3) * XS * SD_PIXB + (st & 1) * 32);
  };
#pragma unroll
  for (int st = 0; st < D; ++st) p[st] = rd(st);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + D < NS) p[(st + D) % (D + 1)] = rd(st + D);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[st]), __builtin_bit_cast(bf16x8, p[st % (D + 1)]),
                                                  st == 0 ? zero : acc, 0, 0, 0);
    if (EXTRA && (st >> 1) == 4)
      accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wd[st & 1]), __builtin_bit_cast(bf16x8, p[st % (D + 1)]),
                                                     st == 8 ? zero : accd, 0, 0, 0);
    side(st);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two chains interleaved MFMA by MFMA (two tiles, one weight fragment feeds both).  A single chain is a dependent
// sequence on one accumulator and issues an MFMA every ~64 cycles when its pixel fragments come from LDS (measured,
// tools/sd_stamps.py); two independent accumulators keep the matrix pipe busy.
__device__ __forceinline__ void sd_chain2(const uint4 (&wr)[18], const unsigned char* sm, const int (&ra)[3], const int (&rb)[3],
                                          f32x16& acca, f32x16& accb) {
  constexpr int NS = 18, D = SD_AHEAD2;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 pa[D + 1], pb[D + 1];
  auto off = [](int st) { return ((st >> 1) % 3) * SD_PIXB + (st & 1) * 32; };
#pragma unroll
  for (int st = 0; st < D; ++st) {
    pa[st] = *reinterpret_cast<const uint4*>(sm + ra[(st >> 1) / 3] + off(st));
    pb[st] = *reinterpret_cast<const uint4*>(sm + rb[(st >> 1) / 3] + off(st));
  }
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + D < NS) {
      pa[(st + D) % (D + 1)] = *reinterpret_cast<const uint4*>(sm + ra[((st + D) >> 1) / 3] + off(st + D));
      pb[(st + D) % (D + 1)] = *reinterpret_cast<const uint4*>(sm + rb[((st + D) >> 1) / 3] + off(st + D));
    }
    __builtin_amdgcn_sched_barrier(0);
    acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[st]), __builtin_bit_cast(bf16x8, pa[st % (D + 1)]),
                                                   st == 0 ? zero : acca, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[st]), __builtin_bit_cast(bf16x8, pb[st % (D + 1)]),
                                                   st == 0 ? zero : accb, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One 32 -> 32 stride-1 layer as a pipeline role: per step the band of two pixel rows [g0, g0 + 2), g0 = 2s + A, as three
// 32-pixel tiles.  RES: add the residual from the ring at RES_OFF (the block input) before the ReLU.
// Schedule of a step (a wave issues in order, and vector instructions only hide inside its OWN MFMAs' shadows):
//   chains of tiles 0 and 1 interleaved | chain of tile 2 with the epilogues of tiles 0 and 1 in eight pieces behind its
//   MFMAs | epilogue of tile 2 (the only exposed one) | barrier.
// Everything that depends on the step only (ring slots of the three tap rows, of the output row and of the residual
// row, for each of the band's two rows; which of them exist) is scalar arithmetic; a lane adds its column.
template <bool RES, int IN_OFF, int IN_R, int OUT_OFF, int OUT_R, int RES_OFF, int RES_R, int A>
__device__ __forceinline__ void sd_conv_role(unsigned char* sm, const uint16_t* w, const float* kc, const int total, const int s_end,
                                             const int lane) {
  const int li = lane & 31, lh = lane >> 5;
  uint4 wr[18];
#pragma unroll
  for (int st = 0; st < 18; ++st) wr[st] = reinterpret_cast<const uint4*>(w)[st * 64 + lh * 32 + li];
  const uint4 wnone[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  float4 cb[4], cs[4], ch[4];      // bias, scale, shift of this lane's 16 channels (8g + 4lh + e)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cb[g] = *reinterpret_cast<const float4*>(kc + 8 * g + 4 * lh);
    cs[g] = *reinterpret_cast<const float4*>(kc + 32 + 8 * g + 4 * lh);
    ch[g] = *reinterpret_cast<const float4*>(kc + 64 + 8 * g + 4 * lh);
  }
  // the lane's pixel in tile k: second row of the band?  byte offset of its column (tap kx = 0 is the pixel to its left:
  // row position col, since position 0 is the zero pixel) and of its own 8-byte output piece
  bool row1[3];
  int colb[3], ocolb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int f = 32 * k + li;
    row1[k] = f >= SD_W;
    const int col = f - (row1[k] ? SD_W : 0);
    colb[k] = col * SD_PIXB + 16 * lh;
    ocolb[k] = (col + 1) * SD_PIXB + 8 * lh;
  }
  int g0 = -2 * SD_LEAD + A;                        // -12 <= g0 < 0
  int y0 = g0 + SD_H;                               // g0 mod 48
  // byte offsets inside their rings of the rows g0 - 1 .. g0 + 2 (input), g0, g0 + 1 (output, residual): advanced by two
  // rows per step with a compare and a subtract each, all on the scalar unit
  auto inc = [](int o, int R) { return o + SD_ROWB >= R * SD_ROWB ? o + SD_ROWB - R * SD_ROWB : o + SD_ROWB; };
  int oi[4], oo[2], orr[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) oi[j] = sd_wrap(g0 + j - 1 + 2 * IN_R, IN_R) * SD_ROWB;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    oo[j] = sd_wrap(g0 + j + 2 * OUT_R, OUT_R) * SD_ROWB;
    orr[j] = sd_wrap(g0 + j + 2 * RES_R, RES_R) * SD_ROWB;
  }
  f32x16 unused;

  for (int s = -SD_LEAD; s < s_end; ++s) {
    SDSTAMP(0);
    if (g0 + 1 >= 0 && g0 < total) {                // the band holds at least one real row
      // ---- scalar: which rows exist (inside the workgroup's rows; tap rows inside the pixel's crop)
      const bool live0 = g0 >= 0, live1 = g0 + 1 < total;       // (the band test above covers the other two bounds)
      const int y1 = y0 + 1 == SD_H ? 0 : y0 + 1;
      const int rowS[2][3] = {{live0 && y0 != 0 ? IN_OFF + oi[0] : SD_ZERO, live0 ? IN_OFF + oi[1] : SD_ZERO,
                               live0 && y0 != SD_H - 1 ? IN_OFF + oi[2] : SD_ZERO},
                              {live1 && y1 != 0 ? IN_OFF + oi[1] : SD_ZERO, live1 ? IN_OFF + oi[2] : SD_ZERO,
                               live1 && y1 != SD_H - 1 ? IN_OFF + oi[3] : SD_ZERO}};
      const int outS[2] = {live0 ? OUT_OFF + oo[0] : SD_DUMP, live1 ? OUT_OFF + oo[1] : SD_DUMP};
      const int resS[2] = {RES_OFF + orr[0], RES_OFF + orr[1]};
      // ---- per lane: three tiles (tile 0 lies in the band's first row, tile 2 in its second, tile 1 in both)
      int ra[3][3], oslot[3], rslot[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) ra[k][ky] = (row1[k] ? rowS[1][ky] : rowS[0][ky]) + colb[k];
        oslot[k] = (row1[k] ? outS[1] : outS[0]) + ocolb[k];
        rslot[k] = (row1[k] ? resS[1] : resS[0]) + ocolb[k];
      }
      f32x16 acc[3];
      uint2 rr = make_uint2(0, 0);
      auto store_piece = [&](int k, int g) {
        *reinterpret_cast<uint2*>(sm + oslot[k] + 16 * g) = sd_piece<RES, true>(acc[k], g, cb[g], cs[g], ch[g], rr);
      };
      sd_chain2(wr, sm, ra[0], ra[1], acc[0], acc[1]);
      SDSTAMP(1);
      // chain of tile 2 with the epilogues of tiles 0 and 1 behind its MFMAs: piece j (tile j / 4, channels of quad j % 4)
      // behind MFMA 2j + 1, its residual quad read behind MFMA 2j - 1
      sd_chain<1, false>(wr, wnone, sm, ra[2], acc[2], unused, [&](int st) {
        if ((st & 1) == 0) return;
        const int j = st >> 1;                       // 0..8
        if (j >= 1) store_piece((j - 1) >> 2, (j - 1) & 3);
        if (RES && j < 8) rr = *reinterpret_cast<const uint2*>(sm + rslot[j >> 2] + 16 * (j & 3));
      });
      SDSTAMP(2);
      uint2 rq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) rq[g] = RES ? *reinterpret_cast<const uint2*>(sm + rslot[2] + 16 * g) : make_uint2(0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        rr = rq[g];
        store_piece(2, g);
      }
    }
    SDSTAMP(3);
    sd_barrier();
    g0 += 2;
    y0 = y0 + 2 >= SD_H ? y0 + 2 - SD_H : y0 + 2;
    oi[0] = oi[2];
    oi[1] = oi[3];
    oi[2] = inc(oi[3], IN_R);
    oi[3] = inc(oi[2], IN_R);
    oo[0] = inc(oo[1], OUT_R);
    oo[1] = inc(oo[0], OUT_R);
    orr[0] = inc(orr[1], RES_R);
    orr[1] = inc(orr[0], RES_R);
  }
}

// The stride-2 roles: output row s (24 pixels; lanes 24..31 idle).
//  * sd_s2_role: conv1 (3x3 stride 2, BN, ReLU), output channels 32j .. 32j+31 of t.  Its epilogue goes to HBM, which no
//    barrier guards, so the four pieces of row s - 1 sit behind the MFMAs of row s.
//  * sd_ds_role: the 1x1 stride-2 down-sample branch (BN), all 64 channels of d: four MFMAs on the centre tap.
__device__ __forceinline__ void sd_s2_role(unsigned char* sm, const Stage1Args& a, const int j, const float* kc1, const int c_lo,
                                           const int s_end, const int lane) {
  const int li = lane & 31, lh = lane >> 5;
  uint4 wr[18];
#pragma unroll
  for (int st = 0; st < 18; ++st) wr[st] = reinterpret_cast<const uint4*>(a.w_s2)[(st * 2 + lh) * 64 + j * 32 + li];
  const uint4 wnone[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  float4 cb[4], cs[4], ch[4];      // bias, scale, shift of this lane's 16 channels (32j + 8g + 4lh + e)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cb[g] = *reinterpret_cast<const float4*>(kc1 + j * 32 + 8 * g + 4 * lh);
    cs[g] = *reinterpret_cast<const float4*>(kc1 + 64 + j * 32 + 8 * g + 4 * lh);
    ch[g] = *reinterpret_cast<const float4*>(kc1 + 128 + j * 32 + 8 * g + 4 * lh);
  }
  const bool lane_ok = li < SD_W / 2;
  const int colb = (lane_ok ? 2 * li : 0) * SD_PIXB + 16 * lh;      // tap kx = 0 is input column 2*li - 1 = row position 2*li
  int ro = 0;                                                       // output row inside its crop
  int ox[3] = {(SD_RX2 - 1) * SD_ROWB, 0, SD_ROWB};                 // ring offsets of input rows 2s - 1, 2s, 2s + 1 (s = 0)
  auto inc2 = [](int o) { return o + 2 * SD_ROWB >= SD_RX2 * SD_ROWB ? o + 2 * SD_ROWB - SD_RX2 * SD_ROWB : o + 2 * SD_ROWB; };
  f32x16 acc, pacc, unused;
  uint16_t* ppt = nullptr;                                          // where the previous row's pieces go
  bool have_prev = false;
  for (int s = -SD_LEAD; s < s_end; ++s) {
    SDSTAMP(0);
    if (s >= 0) {
      const int ra[3] = {(ro != 0 ? SD_X2 + ox[0] : SD_ZERO) + colb, SD_X2 + ox[1] + colb, SD_X2 + ox[2] + colb};
      SDSTAMP(1);
      if (have_prev) {
        sd_chain<1, false>(wr, wnone, sm, ra, acc, unused, [&](int st) {
          if ((st & 3) == 1 && st < 16) {                           // behind MFMAs 1, 5, 9, 13
            const uint2 o = sd_piece<false, true>(pacc, st >> 2, cb[st >> 2], cs[st >> 2], ch[st >> 2], make_uint2(0, 0));
            if (lane_ok) *reinterpret_cast<uint2*>(ppt + 8 * (st >> 2)) = o;
          }
        });
      } else {
        sd_chain<1, false>(wr, wnone, sm, ra, acc, unused, [](int) {});
      }
      SDSTAMP(2);
      pacc = acc;
      const size_t m = ((size_t)c_lo * (SD_H / 2) + (size_t)s) * (SD_W / 2) + (lane_ok ? li : 0);      // output pixel row
      ppt = a.out_t + m * 64 + j * 32 + 4 * lh;
      have_prev = true;
      if (++ro == SD_H / 2) ro = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) ox[k] = inc2(ox[k]);
    }
    SDSTAMP(3);
    sd_barrier();
  }
  if (have_prev) {                                                  // the last row's epilogue
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 o = sd_piece<false, true>(pacc, g, cb[g], cs[g], ch[g], make_uint2(0, 0));
      if (lane_ok) *reinterpret_cast<uint2*>(ppt + 8 * g) = o;
    }
  }
}

__device__ __forceinline__ void sd_ds_role(unsigned char* sm, const Stage1Args& a, const float* kcd, const int c_lo, const int s_end,
                                           const int lane) {
  const int li = lane & 31, lh = lane >> 5;
  uint4 wd[2][2];                                                   // [cout tile][16-channel group]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) wd[j][c] = reinterpret_cast<const uint4*>(a.w_ds)[(c * 2 + lh) * 64 + j * 32 + li];
  const bool lane_ok = li < SD_W / 2;
  const int colb = ((lane_ok ? 2 * li : 0) + 1) * SD_PIXB + 16 * lh;      // the centre tap: input column 2*li
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int ox = 0;                                                       // ring offset of input row 2s
  for (int s = -SD_LEAD; s < s_end; ++s) {
    SDSTAMP(0);
    if (s >= 0) {
      const uint4 p0 = *reinterpret_cast<const uint4*>(sm + SD_X2 + ox + colb), p1 = *reinterpret_cast<const uint4*>(sm + SD_X2 + ox + colb + 32);
      f32x16 accd[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        accd[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wd[j][0]), __builtin_bit_cast(bf16x8, p0), zero, 0, 0, 0);
        accd[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wd[j][1]), __builtin_bit_cast(bf16x8, p1), accd[j], 0, 0, 0);
      }
      SDSTAMP(1);
      const size_t m = ((size_t)c_lo * (SD_H / 2) + (size_t)s) * (SD_W / 2) + (lane_ok ? li : 0);
      uint16_t* pd = a.out_d + m * 64 + 4 * lh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* kb = kcd + j * 32 + 8 * g + 4 * lh;
          const uint2 o = sd_piece<false, false>(accd[j], g, *reinterpret_cast<const float4*>(kb), *reinterpret_cast<const float4*>(kb + 64),
                                                 *reinterpret_cast<const float4*>(kb + 128), make_uint2(0, 0));
          if (lane_ok) *reinterpret_cast<uint2*>(pd + j * 32 + 8 * g) = o;
        }
      SDSTAMP(2);
      ox = ox + 2 * SD_ROWB >= SD_RX2 * SD_ROWB ? ox + 2 * SD_ROWB - SD_RX2 * SD_ROWB : ox + 2 * SD_ROWB;
    }
    SDSTAMP(3);
    sd_barrier();
  }
}

// The loader role: x0 rows [2s + 14, 2s + 16) = 96 pixels x 4 quads = 384 uint4, six per lane, requested two steps before
// they are written into the ring (an HBM round trip is longer than a step).  Bands outside the workgroup's rows load
// the first row again (no branch around a load) and are not written.
// (six named registers per band, not an array: the arrays of an earlier form stayed in scratch memory, and a scratch
// access queues behind the HBM loads in the wave's in-order load counter)
#define SD_LD1(r, k) r##k = *reinterpret_cast<const uint4*>(src + (ok_ ? base_ + (size_t)(k) * 1024 : (size_t)0) + lane * 16)
#define SD_BAND_ISSUE(r, step)                                                                                         \
  do {                                                                                                                 \
    const int g0_ = 2 * (step) + SD_A_LOAD;                                                                            \
    const bool ok_ = g0_ >= 0 && g0_ < total; /* total is even: both rows of the band are real or neither is */        \
    const size_t base_ = (size_t)g0_ * SD_W * 64;   /* 16 pixels = 64 uint4 per k */                                   \
    SD_LD1(r, 0); SD_LD1(r, 1); SD_LD1(r, 2); SD_LD1(r, 3); SD_LD1(r, 4); SD_LD1(r, 5);                                \
  } while (0)
#define SD_ST1(r, k)                                                                                                   \
  do {                                                                                                                 \
    const int px = (lane >> 2) + 16 * (k), ro = px >= SD_W ? 1 : 0, col = px - SD_W * ro;                              \
    *reinterpret_cast<uint4*>(sm + SD_X0 + ((g0_ + ro) & (SD_RX0 - 1)) * SD_ROWB + (col + 1) * SD_PIXB + (lane & 3) * 16) = r##k; \
  } while (0)
#define SD_BAND_COMMIT(r, step)                                                                                        \
  do {                                                                                                                 \
    const int g0_ = 2 * (step) + SD_A_LOAD;                                                                            \
    if (g0_ >= 0 && g0_ < total) { SD_ST1(r, 0); SD_ST1(r, 1); SD_ST1(r, 2); SD_ST1(r, 3); SD_ST1(r, 4); SD_ST1(r, 5); } \
  } while (0)
static_assert((SD_RX0 & (SD_RX0 - 1)) == 0, "the loader masks the X0 ring slot");
__device__ __forceinline__ void sd_loader_role(unsigned char* sm, const unsigned char* src, const int total, const int s_end,
                                               const int lane) {
  uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5, rc0, rc1, rc2, rc3, rc4, rc5;
  int s = -SD_LEAD;
  SD_BAND_ISSUE(ra, s);
  SD_BAND_ISSUE(rb, s + 1);
  for (;;) {
    SD_BAND_ISSUE(rc, s + 2); SD_BAND_COMMIT(ra, s); sd_barrier(); if (++s >= s_end) break;
    SD_BAND_ISSUE(ra, s + 2); SD_BAND_COMMIT(rb, s); sd_barrier(); if (++s >= s_end) break;
    SD_BAND_ISSUE(rb, s + 2); SD_BAND_COMMIT(rc, s); sd_barrier(); if (++s >= s_end) break;
  }
}

__global__ __launch_bounds__(SD_THREADS, 1) void stage1_depth_bf16(Stage1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_sd[];
  unsigned char* sm = sm_sd;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c_lo = (int)((int64_t)a.n * blockIdx.x / gridDim.x), c_hi = (int)((int64_t)a.n * (blockIdx.x + 1) / gridDim.x);
  const int crops = c_hi - c_lo;
  const int total = crops * SD_H, s_end = crops * (SD_H / 2);
  float* sK = reinterpret_cast<float*>(sm + SD_KOFF);
  if (tid < 32) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      sK[96 * l + tid] = a.bias[l][tid]; sK[96 * l + 32 + tid] = a.scale[l][tid]; sK[96 * l + 64 + tid] = a.shift[l][tid];
    }
  }
  if (tid < 64) {
    float* k1 = sK + 4 * 96;
    k1[tid] = a.b_s2[tid]; k1[64 + tid] = a.s_s2[tid]; k1[128 + tid] = a.h_s2[tid];
    k1[192 + tid] = a.b_ds[tid]; k1[256 + tid] = a.s_ds[tid]; k1[320 + tid] = a.h_ds[tid];
  }
  // rings and the zero row start as zeros: the two padding pixels of every ring row are never written again
  for (int i = tid; i < SD_KOFF / 16; i += SD_THREADS) *reinterpret_cast<uint4*>(sm + 16 * i) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (crops <= 0) return;

  // Role of each wave.  A workgroup's waves go to the SIMDs in the cyclic order 0, 2, 1, 3, so waves w and w + 4 share a
  // SIMD: the two 3x3 stride-2 waves sit beside the two convolution roles without a residual (the lighter ones), the
  // loader beside c2a and the down-sample wave (four MFMAs a step) beside c2b.
  switch (wave) {
    case 0: sd_conv_role<false, SD_X0, SD_RX0, SD_T1, SD_RT1, SD_X0, SD_RX0, 11>(sm, a.w[0], sK, total, s_end, lane); break;
    case 1: sd_conv_role<true, SD_T1, SD_RT1, SD_X1, SD_RX1, SD_X0, SD_RX0, 8>(sm, a.w[1], sK + 96, total, s_end, lane); break;
    case 2: sd_conv_role<false, SD_X1, SD_RX1, SD_T2, SD_RT2, SD_X1, SD_RX1, 5>(sm, a.w[2], sK + 192, total, s_end, lane); break;
    case 3: sd_conv_role<true, SD_T2, SD_RT2, SD_X2, SD_RX2, SD_X1, SD_RX1, 2>(sm, a.w[3], sK + 288, total, s_end, lane); break;
    case 5:
      sd_loader_role(sm, reinterpret_cast<const unsigned char*>(a.in) + (size_t)c_lo * SD_H * SD_W * 64, total, s_end, lane);
      break;
    case 7: sd_ds_role(sm, a, sK + 384 + 192, c_lo, s_end, lane); break;
    default: sd_s2_role(sm, a, wave == 4 ? 0 : 1, sK + 384, c_lo, s_end, lane); break;
  }
}

bool stage1_depth_applicable(const ConvLayer& c1a, const ConvLayer& c2a, const ConvLayer& c1b, const ConvLayer& c2b,
                             const ConvLayer& s2, const ConvLayer* ds, int h, int w) {
  auto thin = [](const ConvLayer& L) {
    return L.taps == 9 && L.stride == 1 && L.cin_s == 32 && L.cout_s == 32 && L.cout_w == 32 && L.wh != nullptr && L.relu;
  };
  const bool off = (c1a.flags & (CONV_NO_DEPTH | CONV_NO_FUSE | CONV_NO_THIN)) != 0;
  return !off && h == SD_H && w == SD_W && thin(c1a) && thin(c2a) && thin(c1b) && thin(c2b) && ds != nullptr &&
         s2.taps == 9 && s2.stride == 2 && s2.cin_s == 32 && s2.cout_s == 64 && s2.cout_w == 64 && s2.wh && s2.relu &&
         ds->taps == 1 && ds->stride == 2 && ds->cin_s == 32 && ds->cout_s == 64 && ds->cout_w == 64 && ds->wh && !ds->relu;
}

int stage1_depth_prepare_device() {
  UMT_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(stage1_depth_bf16), SD_LDS));
  return 0;
}

int launch_stage1_depth_bf16(const ConvLayer& c1a, const ConvLayer& c2a, const ConvLayer& c1b, const ConvLayer& c2b,
                             const ConvLayer& s2, const ConvLayer& ds, const uint16_t* in, int n, uint16_t* out_t,
                             uint16_t* out_d, hipStream_t stream) {
  UMT_REQUIRE(stage1_depth_applicable(c1a, c2a, c1b, c2b, s2, &ds, SD_H, SD_W) && n > 0 && n < (1 << 20) && in && out_t && out_d,
              UMT_ERR_INVALID, "stage1_depth: bad arguments");
  Stage1Args a;
  const ConvLayer* L[4] = {&c1a, &c2a, &c1b, &c2b};
  for (int i = 0; i < 4; ++i) { a.w[i] = L[i]->wh; a.bias[i] = L[i]->bias; a.scale[i] = L[i]->scale; a.shift[i] = L[i]->shift; }
  a.in = in;
  a.w_s2 = s2.wh; a.b_s2 = s2.bias; a.s_s2 = s2.scale; a.h_s2 = s2.shift;
  a.w_ds = ds.wh; a.b_ds = ds.bias; a.s_ds = ds.scale; a.h_ds = ds.shift;
  a.out_t = out_t; a.out_d = out_d; a.n = n;
  const int wgs = n < 256 ? n : 256;              // one persistent workgroup per CU (gfx950: 256 CUs), each a contiguous run of crops
  hipLaunchKernelGGL(stage1_depth_bf16, dim3(wgs), dim3(SD_THREADS), SD_LDS, stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef UMT_SD_STAMP
}  // namespace umt
extern "C" int umt_sd_stamps(unsigned long long* host32) {
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(umt::g_sdstamp), 32 * sizeof(unsigned long long));
}
namespace umt {
#endif
}  // namespace umt
