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

template <bool OUT_BF16>
static int launch_stem_any(const StemParams& p, const float* images, int n, void* out, hipStream_t stream) {
  UMT_REQUIRE(n > 0, UMT_ERR_INVALID, "stem: n = %d", n);
  const int64_t blocks = (int64_t)n * (48 / STEM_ROWS);
  UMT_REQUIRE(blocks < ((int64_t)1 << 31), UMT_ERR_INVALID, "stem: batch too large");
  hipLaunchKernelGGL(stem_kernel<OUT_BF16>, dim3((unsigned)blocks), dim3(256), 0, stream, p, images, n, out);
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
