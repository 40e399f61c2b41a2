// Stem of the backbone: Conv2d(1,32,3,pad 1,bias) -> BatchNorm2d -> ReLU -> MaxPool2d(2,2)
// (lib/models/model_utils.py:119-124).  K = 9 is too thin for the matrix pipe (tried: same time), so this
// is a VALU kernel bound by its 128-B-per-pixel output stream: each thread owns one
// pooled pixel x 8 channels, reads the 4x4 input patch from an LDS copy of the input rows and
// writes 32 (fp32) / 16 (bf16) contiguous bytes; four neighbouring threads complete a channels-last row.
#include "umt_internal.h"

namespace umt {

__device__ __forceinline__ unsigned stem_pack_bf16x2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// One workgroup = 4 pooled rows of one crop: the 10 input rows they need are staged in LDS with coalesced
// 16-byte loads (the 4x4 patches of neighbouring threads overlap: read straight from global they cost 16
// scattered 4-byte loads per thread and the address path, not the VALU, sets the pace).  A thread owns
// one pooled pixel x 8 channels; 768 such items per workgroup, 3 per thread.
constexpr int STEM_ROWS = 4;                 // pooled rows per workgroup
constexpr int STEM_IN_ROWS = 2 * STEM_ROWS + 2;
constexpr int STEM_LDW = 100;                // floats per staged input row: [2 zeros | 96 pixels | 2 zeros]

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void stem_kernel(StemParams p, const float* __restrict__ images,
                                                   int n, void* __restrict__ out) {
  __shared__ __attribute__((aligned(8))) float sw[9 * 32], sb[32], ss[32], sh[32];
  __shared__ __attribute__((aligned(16))) float sin_[STEM_IN_ROWS * STEM_LDW];
  const int tid = threadIdx.x;
  for (int i = tid; i < 9 * 32; i += 256) sw[i] = p.w[i];
  if (tid < 32) {
    sb[tid] = p.bias[tid];
    ss[tid] = p.scale[tid];
    sh[tid] = p.shift[tid];
  }
  const int img = blockIdx.x / (48 / STEM_ROWS), band = blockIdx.x - img * (48 / STEM_ROWS);
  const int yo0 = band * STEM_ROWS;
  const float* src = images + (size_t)img * 96 * 96;
  // input rows 2*yo0 - 1 .. 2*yo0 + 8 (zero outside the crop), 24 float4 per row
  // torch's convolution, ReLU and MaxPool propagate NaN; the pooled max below starts from 0 and v_max drops a NaN
  // operand.  Non-finite input pixels are therefore detected here, once per staged element, and only a band that
  // holds one takes the per-patch check in the item loop (which writes canonical NaNs for every pooled pixel whose
  // 4x4 patch touches it, the pixels torch would make NaN; an infinite input is treated the same way).
  int bad = 0;
  for (int i = tid; i < STEM_IN_ROWS * 24; i += 256) {
    const int r = i / 24, c4 = i - r * 24;
    const int y = 2 * yo0 - 1 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < 96) v = *reinterpret_cast<const float4*>(src + y * 96 + 4 * c4);
    bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    float* d = sin_ + r * STEM_LDW + 2 + 4 * c4;     // 8-byte aligned
    *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
  }
  if (tid < STEM_IN_ROWS) {
    float* d = sin_ + tid * STEM_LDW;
    d[0] = d[1] = d[98] = d[99] = 0.f;
  }
  const int any_bad = __syncthreads_or(bad);
#pragma unroll 1
  for (int it = 0; it < STEM_ROWS * 48 * 4 / 256; ++it) {
    const int item = tid + 256 * it;
    const int cg = item & 3, pl = item >> 2;           // channel group, pooled pixel inside the band
    const int ry = pl / 48, xo = pl - ry * 48;
    float patch[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      // staged column of input x is x + 2; the patch starts at x = 2*xo - 1 -> column 2*xo + 1
      const float* row = sin_ + (2 * ry + dy) * STEM_LDW + 2 * xo + 1;
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) patch[dy][dx] = row[dx];
    }
    // two channels per instruction (v_pk_fma_f32: the packed form is what reaches the vector fp32 rate); each lane of a
    // pair is the same fmaf chain as before, so the results are unchanged
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float res[8];
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
      const int ch = cg * 8 + c;
      f32x2 w2[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) w2[t] = *reinterpret_cast<const f32x2*>(sw + t * 32 + ch);
      const f32x2 b2 = *reinterpret_cast<const f32x2*>(sb + ch), s2 = *reinterpret_cast<const f32x2*>(ss + ch),
                  h2 = *reinterpret_cast<const f32x2*>(sh + ch);
      f32x2 best = {0.f, 0.f};  // ReLU output is >= 0, so 0 is the identity of the pooled max
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          f32x2 s = {0.f, 0.f};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float pv = patch[py + ky][px + kx];
              const f32x2 p2 = {pv, pv};
              s = __builtin_elementwise_fma(w2[ky * 3 + kx], p2, s);
            }
          const f32x2 v = (s + b2) * s2 + h2;
          best = __builtin_elementwise_max(best, v);
        }
      res[c] = best.x;
      res[c + 1] = best.y;
    }
    if (any_bad) {        // rare: uniform over the workgroup
      bool nf = false;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) nf = nf || !isfinite(patch[dy][dx]);
      if (nf)
#pragma unroll
        for (int c = 0; c < 8; ++c) res[c] = __builtin_nanf("");
    }
    const size_t pix = ((size_t)img * 48 + yo0 + ry) * 48 + xo;
    if (OUT_BF16) {
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + pix * 32 + cg * 8) =
          make_uint4(stem_pack_bf16x2(res[0], res[1]), stem_pack_bf16x2(res[2], res[3]),
                     stem_pack_bf16x2(res[4], res[5]), stem_pack_bf16x2(res[6], res[7]));
    } else {
      float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + pix * 32 + cg * 8);
      dst[0] = make_float4(res[0], res[1], res[2], res[3]);
      dst[1] = make_float4(res[4], res[5], res[6], res[7]);
    }
  }
}

// ---------------------------------------------------------------------------------------- matrix-pipe form
// The same layer on v_mfma_f32_32x32x2_f32: K = 9 taps padded to 10, five instructions per tile of 32 convolution pixels
// x 32 channels.  The instruction is an exact k-ordered fmaf chain starting from 0, so each output is the same chain
// fma(w8, p8, ... fma(w1, p1, fma(w0, p0, 0))) the vector kernel above evaluates (the tenth slot multiplies 0 by 0):
// results are BIT-IDENTICAL to it (tests/test_gpu_edges.py, UMT_STEM_VALU).  Measured per 4096 crops (bf16 output):
// vector kernel 0.42-0.46 ms; this one 0.35-0.38 ms with the matrix pipe busy 49 % of it (rocprofv3 SQ_VALU_MFMA_BUSY):
// the 0.17 ms of MFMA and the ~400 vector instructions per band and wave ADD on this chip rather than overlap, so what is
// left to win is a shorter epilogue, not a better schedule.  How it got there (each step measured, 4096 crops):
//  * a first form with the 2x2 window on four lanes spent as many vector instructions on DPP max as the vector kernel
//    spends on FMAs: 0.64 ms;
//  * operands swapped (weights as the row operand): a lane owns one convolution pixel and the 16 channels 8g + 4lh + e;
//  * a wave works on PAIRS of tiles - the same 32 columns of the two convolution rows 2y and 2y + 1 - so the vertical
//    half of the 2x2 max-pool and the ReLU are one v_max3(a, b, 0) per value with no lane traffic, and the horizontal
//    half one DPP max with the neighbouring lane; BN comes before the max (a scale may be negative), as
//    model_utils.py:119-124 orders the layers: 0.49 ms;
//  * the epilogue of pair i issued behind the MFMAs of pair i + 1 (sched_group_barrier): 0.45 ms;
//  * MFMA results in VGPRs (Makefile: -mllvm -amdgpu-mfma-vgpr-form for this file; 32 v_accvgpr_read per pair gone) and
//    the DPP max as one instruction: 0.37 ms;
//  * the two lanes of a column pair hold the same 16 channels; v_permlane32_swap arranges 8 contiguous channels per lane:
//    one 16-byte store (bf16) per lane, a whole pixel per four lanes.
typedef float stem_f32x16 __attribute__((ext_vector_type(16)));
typedef float stem_f32x2 __attribute__((ext_vector_type(2)));

// max with the neighbouring lane (quad_perm [1,0,3,2]) of eight values in place: one v_max_f32_dpp each.  (The builtin
// form costs a DPP move, a canonicalising max and the max per value.)  Inline assembly is opaque to the compiler's hazard
// recogniser, and a DPP read of a VGPR written by the preceding vector instruction needs two wait states: the s_nop.
__device__ __forceinline__ void stem_pair_max8(float& a, float& b, float& c, float& d, float& e, float& f, float& g, float& h) {
#define STEM_DPP(i) "v_max_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" STEM_DPP(0) STEM_DPP(1) STEM_DPP(2) STEM_DPP(3) STEM_DPP(4) STEM_DPP(5) STEM_DPP(6) STEM_DPP(7)
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
#undef STEM_DPP
}

// A workgroup walks `nb` consecutive bands of one crop with the next band's input rows in flight (one 16-byte load per
// lane, held in registers, written to the other half of a double-buffered LDS tile before the band's last epilogue), so
// no wave waits on HBM between bands; the constants are forced resident before the loop because the compiler's wait-count
// pass otherwise carries their pending loads into the loop as vmcnt waits that drain every band's prefetch and stores.
// Few-crop launches (the tracker's per-frame call) keep one band per workgroup for the parallelism.
//
// SPLIT (the bf16 mode's stem): the same sums on v_mfma_f32_32x32x16_bf16.  A staged pixel is the pair (hi, lo) of bf16
// numbers with hi = bf16(v), lo = bf16(v - hi) in one dword, a weight likewise, and a tile is three instructions
// w_lo . p_hi + w_hi . p_lo + w_hi . p_hi (K = 16 slots: taps 0..7 on the lower half-wave, tap 8 and seven zeros on the
// upper) instead of five of twice the length: 96 cycles of matrix pipe per tile instead of 320.  The dropped w_lo . p_lo
// is 2^-16 of a product, far below the bf16 rounding of the stored output (2^-9): not bit-identical to the fp32 chain,
// equal to it after that rounding for all but about one output in 10^4 (tests/test_gpu_edges.py states the bound).
constexpr int STEM_ZTAIL = 168;      // zero floats behind a staged band: the upper half-wave's seven unused k slots read them

__device__ __forceinline__ void stem_split2(float a, float b, unsigned& pa, unsigned& pb) {
  typedef __bf16 stem_bf16x2 __attribute__((ext_vector_type(2)));
  typedef float stem_f32x2v __attribute__((ext_vector_type(2)));
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((stem_f32x2v){a, b}, stem_bf16x2));
  const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);   // exact
  const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector((stem_f32x2v){ra, rb}, stem_bf16x2));
  pa = (h << 16) | (l & 0xffffu);
  pb = (h & 0xffff0000u) | (l >> 16);
}

template <bool OUT_BF16, bool SPLIT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemParams p, const float* __restrict__ images, int n, int nb,
                                                        void* __restrict__ out) {
  constexpr int SBUF = STEM_IN_ROWS * STEM_LDW + (SPLIT ? STEM_ZTAIL : 0);
  __shared__ __attribute__((aligned(16))) float sbuf[2][SBUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int per_img = (48 / STEM_ROWS) / nb;
  const int img = blockIdx.x / per_img, band0 = (blockIdx.x - img * per_img) * nb;
  const float* src = images + (size_t)img * 96 * 96;
  // staging: lane tid < 240 owns float4 c4 of staged row r of every band (input row 8 band - 1 + r)
  const int sr = tid / 24, sc4 = tid - sr * 24;
  const bool stager = tid < STEM_IN_ROWS * 24;
  auto row_ok = [&](int band) {
    const int y = 2 * STEM_ROWS * band - 1 + sr;
    return stager && y >= 0 && y < 96;
  };
  auto fetch = [&](int band) {                                // unconditional (a branch around a load costs a full vmcnt drain);
    const int y = 2 * STEM_ROWS * band - 1 + sr;              // rows outside the crop are zeroed when staged
    return *reinterpret_cast<const float4*>(src + (row_ok(band) ? y * 96 + 4 * sc4 : 0));
  };
  auto stage = [&](float* dstbuf, float4 v, bool ok) -> int {
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nonfinite = !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    if (SPLIT) {          // (hi, lo) pairs; a non-finite pixel keeps its exponent bits in hi, which is what the check reads
      unsigned q0, q1, q2, q3;
      stem_split2(v.x, v.y, q0, q1);
      stem_split2(v.z, v.w, q2, q3);
      v = make_float4(__builtin_bit_cast(float, q0), __builtin_bit_cast(float, q1), __builtin_bit_cast(float, q2),
                      __builtin_bit_cast(float, q3));
    }
    if (stager) {
      float* d = dstbuf + sr * STEM_LDW + 2 + 4 * sc4;
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    }
    return nonfinite;
  };
  if (tid < 2 * STEM_IN_ROWS) {
    float* d = &sbuf[tid / STEM_IN_ROWS][(tid % STEM_IN_ROWS) * STEM_LDW];   // both buffers: the two zero columns either side of a row
    d[0] = d[1] = d[98] = d[99] = 0.f;
  }
  if (SPLIT) {
    for (int i = tid; i < 2 * STEM_ZTAIL; i += 256) sbuf[i / STEM_ZTAIL][STEM_IN_ROWS * STEM_LDW + i % STEM_ZTAIL] = 0.f;
  }
  int bad = stage(sbuf[0], fetch(band0), row_ok(band0));
  // weights of this lane's channel li for the taps 2j + lh (the tenth slot: 0), BN constants of its 16 channels
  float wq[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) wq[j] = 2 * j + lh < 9 ? p.w[(2 * j + lh) * 32 + li] : 0.f;
  float4 cb[4], cs[4], ch[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cb[g] = *reinterpret_cast<const float4*>(p.bias + 8 * g + 4 * lh);
    cs[g] = *reinterpret_cast<const float4*>(p.scale + 8 * g + 4 * lh);
    ch[g] = *reinterpret_cast<const float4*>(p.shift + 8 * g + 4 * lh);
  }
  if (SPLIT) {          // (x + bias) * scale + shift = x * scale + (bias * scale + shift): 2^-24 apart, under a 2^-9 rounding
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ch[g].x = fmaf(cb[g].x, cs[g].x, ch[g].x); ch[g].y = fmaf(cb[g].y, cs[g].y, ch[g].y);
      ch[g].z = fmaf(cb[g].z, cs[g].z, ch[g].z); ch[g].w = fmaf(cb[g].w, cs[g].w, ch[g].w);
    }
  }
  // float offset of each of the lane's five taps from its tile's origin in the staged rows (tap (ky, kx) of pixel (y, x):
  // staged row y + ky, column x + kx + 1; the tenth slot reads the zero at the start of the tile's first row)
  int toff[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = 2 * j + lh, ky = tap / 3, kx = tap - 3 * ky;
    toff[j] = tap < 9 ? ky * STEM_LDW + li + kx + 1 : -1;
  }
  // SPLIT: k slot s of the lane's half-wave is tap s (lower half) or tap 8 then nothing (upper half); the packed weight
  // pairs of the lane's channel li and the float index of each slot's pixel from the buffer base (tile a, column block 0)
  unsigned wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
  int poff[8];
  if (SPLIT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t0 = lh ? (i == 0 ? 8 : -1) : 2 * i, t1 = lh ? -1 : 2 * i + 1;
      const float w0 = t0 >= 0 ? p.w[(t0 < 0 ? 0 : t0) * 32 + li] : 0.f, w1 = t1 >= 0 ? p.w[(t1 < 0 ? 0 : t1) * 32 + li] : 0.f;
      unsigned q0, q1;
      stem_split2(w0, w1, q0, q1);                    // q = hi << 16 | lo per weight
      wh[i] = (q0 >> 16) | (q1 & 0xffff0000u);
      wl[i] = (q0 & 0xffffu) | (q1 << 16);
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int tap = lh ? (s8 == 0 ? 8 : -1) : s8, ky = tap / 3, kx = tap - 3 * ky;
      poff[s8] = tap >= 0 ? (2 * wave + ky) * STEM_LDW + li + kx + 1 : STEM_IN_ROWS * STEM_LDW;
    }
    asm volatile("" ::"v"(wh[0]), "v"(wh[1]), "v"(wh[2]), "v"(wh[3]), "v"(wl[0]), "v"(wl[1]), "v"(wl[2]), "v"(wl[3]));
  }
  const int px = li & 1;
  // every constant has arrived before the band loop: the compiler's wait-count pass would otherwise carry their pending
  // loads into the loop body as vmcnt waits that also drain each band's prefetch and stores
#pragma unroll
  for (int g = 0; g < 4; ++g)
    asm volatile("" ::"v"(cb[g].x), "v"(cb[g].y), "v"(cb[g].z), "v"(cb[g].w), "v"(cs[g].x), "v"(cs[g].y), "v"(cs[g].z), "v"(cs[g].w),
                 "v"(ch[g].x), "v"(ch[g].y), "v"(ch[g].z), "v"(ch[g].w));
  asm volatile("" ::"v"(wq[0]), "v"(wq[1]), "v"(wq[2]), "v"(wq[3]), "v"(wq[4]));
  int any_bad = __syncthreads_or(bad);
  const float zero_in = sbuf[0][0];

#pragma unroll 1
  for (int ib = 0; ib < nb; ++ib) {
    const int band = band0 + ib, yo0 = band * STEM_ROWS;
    const float* sin_ = sbuf[ib & 1];
    const bool more = ib + 1 < nb;
    const float4 nxt = fetch(more ? band + 1 : band);

    // 12 tile pairs per band (4 pooled rows x 3 column blocks of 16 pooled pixels): wave w takes pooled row w
    auto epilogue = [&](const stem_f32x16& aa, const stem_f32x16& ab, int ry, int tx, bool check) {
      const int xo = tx * 16 + (li >> 1);
      float r4[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const stem_f32x2 sb0 = {cb[g].x, cb[g].y}, sb1 = {cb[g].z, cb[g].w}, ss0 = {cs[g].x, cs[g].y}, ss1 = {cs[g].z, cs[g].w},
                         sh0 = {ch[g].x, ch[g].y}, sh1 = {ch[g].z, ch[g].w};
        stem_f32x2 a0, a1, b0, b1;
        if constexpr (SPLIT) {        // the bias is folded into the shift before the band loop: one instruction per pair of values
          a0 = __builtin_elementwise_fma((stem_f32x2){aa[4 * g], aa[4 * g + 1]}, ss0, sh0);
          a1 = __builtin_elementwise_fma((stem_f32x2){aa[4 * g + 2], aa[4 * g + 3]}, ss1, sh1);
          b0 = __builtin_elementwise_fma((stem_f32x2){ab[4 * g], ab[4 * g + 1]}, ss0, sh0);
          b1 = __builtin_elementwise_fma((stem_f32x2){ab[4 * g + 2], ab[4 * g + 3]}, ss1, sh1);
        } else {                      // the reference's order: (x + bias) * scale + shift
          a0 = __builtin_elementwise_fma((stem_f32x2){aa[4 * g], aa[4 * g + 1]} + sb0, ss0, sh0);
          a1 = __builtin_elementwise_fma((stem_f32x2){aa[4 * g + 2], aa[4 * g + 3]} + sb1, ss1, sh1);
          b0 = __builtin_elementwise_fma((stem_f32x2){ab[4 * g], ab[4 * g + 1]} + sb0, ss0, sh0);
          b1 = __builtin_elementwise_fma((stem_f32x2){ab[4 * g + 2], ab[4 * g + 3]} + sb1, ss1, sh1);
        }
        r4[g][0] = fmaxf(fmaxf(a0.x, b0.x), 0.f);
        r4[g][1] = fmaxf(fmaxf(a0.y, b0.y), 0.f);
        r4[g][2] = fmaxf(fmaxf(a1.x, b1.x), 0.f);
        r4[g][3] = fmaxf(fmaxf(a1.y, b1.y), 0.f);
      }
      stem_pair_max8(r4[0][0], r4[0][1], r4[0][2], r4[0][3], r4[1][0], r4[1][1], r4[1][2], r4[1][3]);
      stem_pair_max8(r4[2][0], r4[2][1], r4[2][2], r4[2][3], r4[3][0], r4[3][1], r4[3][2], r4[3][3]);
      float res[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) res[i][e] = px ? r4[2 + i][e] : r4[i][e];
      if (check) {        // rare: a pooled pixel whose 4x4 patch holds a non-finite input is NaN, as torch makes it
        bool nf = false;
        const float* patch = sin_ + (2 * ry) * STEM_LDW + 2 * xo + 1;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
          for (int dx = 0; dx < 4; ++dx) nf = nf || !isfinite(patch[dy * STEM_LDW + dx]);
        if (nf)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) res[i][e] = __builtin_nanf("");
      }
      // the lane holds channels 8 (2 px + i) + 4 lh + {0..3} for i = 0, 1; v_permlane32_swap trades piece 1 of the lower
      // half-wave for piece 0 of the upper, after which lane (li, lh) holds the 8 channels 8 (2 px + lh) + {0..7}: one
      // contiguous store per lane, a whole 64- or 128-byte pixel per four lanes
      const size_t pix = ((size_t)img * 48 + yo0 + ry) * 48 + xo;
      const int c0 = 8 * (2 * px + lh);
      typedef unsigned int stem_u32x2 __attribute__((ext_vector_type(2)));
      if (OUT_BF16) {
        const stem_u32x2 s0 = __builtin_amdgcn_permlane32_swap(stem_pack_bf16x2(res[0][0], res[0][1]), stem_pack_bf16x2(res[1][0], res[1][1]), false, false);
        const stem_u32x2 s1 = __builtin_amdgcn_permlane32_swap(stem_pack_bf16x2(res[0][2], res[0][3]), stem_pack_bf16x2(res[1][2], res[1][3]), false, false);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + pix * 32 + c0) = make_uint4(s0.x, s1.x, s0.y, s1.y);
      } else {
        stem_u32x2 sw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sw[e] = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, res[0][e]), __builtin_bit_cast(unsigned, res[1][e]), false, false);
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<float*>(out) + pix * 32 + c0);
        dst[0] = make_uint4(sw[0].x, sw[1].x, sw[2].x, sw[3].x);
        dst[1] = make_uint4(sw[0].y, sw[1].y, sw[2].y, sw[3].y);
      }
    };
    // the two interleaved MFMA chains of a pair (column block tx)
    auto chains = [&](int tx, stem_f32x16& aa, stem_f32x16& ab) {
      const stem_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT) {
        typedef __bf16 stem_bf16x8 __attribute__((ext_vector_type(8)));
        typedef unsigned stem_u32x4 __attribute__((ext_vector_type(4)));
        const unsigned* bu = reinterpret_cast<const unsigned*>(sin_) + 32 * tx;
        unsigned da[8], db[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          da[s8] = bu[poff[s8]];
          db[s8] = bu[poff[s8] + STEM_LDW];
        }
        stem_u32x4 ha, la, hb, lb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // slot 2i in the low half of a register, 2i + 1 in the high half
          ha[i] = __builtin_amdgcn_perm(da[2 * i + 1], da[2 * i], 0x07060302u);
          la[i] = __builtin_amdgcn_perm(da[2 * i + 1], da[2 * i], 0x05040100u);
          hb[i] = __builtin_amdgcn_perm(db[2 * i + 1], db[2 * i], 0x07060302u);
          lb[i] = __builtin_amdgcn_perm(db[2 * i + 1], db[2 * i], 0x05040100u);
        }
        const stem_bf16x8 WH = __builtin_bit_cast(stem_bf16x8, (stem_u32x4){wh[0], wh[1], wh[2], wh[3]});
        const stem_bf16x8 WL = __builtin_bit_cast(stem_bf16x8, (stem_u32x4){wl[0], wl[1], wl[2], wl[3]});
        aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WL, __builtin_bit_cast(stem_bf16x8, ha), zero, 0, 0, 0);
        ab = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WL, __builtin_bit_cast(stem_bf16x8, hb), zero, 0, 0, 0);
        aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, __builtin_bit_cast(stem_bf16x8, la), aa, 0, 0, 0);
        ab = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, __builtin_bit_cast(stem_bf16x8, lb), ab, 0, 0, 0);
        aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, __builtin_bit_cast(stem_bf16x8, ha), aa, 0, 0, 0);
        ab = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, __builtin_bit_cast(stem_bf16x8, hb), ab, 0, 0, 0);
      } else {
        const float* base = sin_ + (2 * wave) * STEM_LDW + 32 * tx;   // convolution row 2 (yo0 + wave): staged row 2 wave is its row above
        float va[5], vb[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          va[j] = toff[j] >= 0 ? base[toff[j]] : zero_in;
          vb[j] = toff[j] >= 0 ? base[STEM_LDW + toff[j]] : zero_in;
        }
        aa = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0], va[0], zero, 0, 0, 0);
        ab = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0], vb[0], zero, 0, 0, 0);
#pragma unroll
        for (int j = 1; j < 5; ++j) {
          aa = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j], va[j], aa, 0, 0, 0);
          ab = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j], vb[j], ab, 0, 0, 0);
        }
      }
    };
    // (SPLIT: 0, 12, 20 or 28 vector instructions per MFMA measured the same 281-289 us: the vector pipe sets the pace)
    constexpr int NMF = SPLIT ? 6 : 10, NVA = SPLIT ? 20 : 12;      // the epilogue's vector instructions dealt between the MFMAs
    if (any_bad) {        // uniform; the plain order
#pragma unroll 1
      for (int tx = 0; tx < 3; ++tx) {
        stem_f32x16 aa, ab;
        chains(tx, aa, ab);
        epilogue(aa, ab, wave, tx, true);
      }
      if (more) bad = stage(sbuf[(ib + 1) & 1], nxt, row_ok(band + 1));
    } else {              // the epilogue of a pair is issued behind the MFMAs of the next
      stem_f32x16 a0, b0, a1, b1, a2, b2;
      chains(0, a0, b0);
      chains(1, a1, b1);
      epilogue(a0, b0, wave, 0, false);
#pragma unroll
      for (int k = 0; k < NMF; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NVA, 0);
      }
      chains(2, a2, b2);
      epilogue(a1, b1, wave, 1, false);
#pragma unroll
      for (int k = 0; k < NMF; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NVA, 0);
      }
      // the next band's rows into the other buffer (last read before the previous barrier) ahead of the last epilogue:
      // the wait for the prefetch then leaves that epilogue's stores in flight
      if (more) bad = stage(sbuf[(ib + 1) & 1], nxt, row_ok(band + 1));
      epilogue(a2, b2, wave, 2, false);
    }
    if (more) any_bad = __syncthreads_or(bad);
  }
}

template <bool OUT_BF16>
static int launch_stem_any(const StemParams& p, const float* images, int n, void* out, hipStream_t stream) {
  UMT_REQUIRE(n > 0, UMT_ERR_INVALID, "stem: n = %d", n);
  const int64_t blocks = (int64_t)n * (48 / STEM_ROWS);
  UMT_REQUIRE(blocks < ((int64_t)1 << 31), UMT_ERR_INVALID, "stem: batch too large");
  const int nb = n >= 1024 ? 48 / STEM_ROWS : 1;             // bands per workgroup of the matrix-pipe form
  if (p.valu)
    hipLaunchKernelGGL(stem_kernel<OUT_BF16>, dim3((unsigned)blocks), dim3(256), 0, stream, p, images, n, out);
  else if (OUT_BF16 && p.valu != 2)      // the bf16 mode: split operands on the bf16 matrix instructions
    hipLaunchKernelGGL((stem_mfma_kernel<OUT_BF16, OUT_BF16>), dim3((unsigned)(blocks / nb)), dim3(256), 0, stream, p, images, n, nb, out);
  else
    hipLaunchKernelGGL((stem_mfma_kernel<OUT_BF16, false>), dim3((unsigned)(blocks / nb)), dim3(256), 0, stream, p, images, n, nb, out);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_stem(const StemParams& p, const float* images, int n, float* out, hipStream_t stream) {
  return launch_stem_any<false>(p, images, n, out, stream);
}

// same arithmetic, result rounded once (RNE) to bf16 for the UMT_PREC_BF16 backbone
int launch_stem_bf16(const StemParams& p, const float* images, int n, uint16_t* out, hipStream_t stream) {
  return launch_stem_any<true>(p, images, n, out, stream);
}

}  // namespace umt
