// Stem of the backbone: Conv2d(1,32,3,pad 1,bias) -> BatchNorm2d -> ReLU -> MaxPool2d(2,2)
// (lib/models/model_utils.py:119-124).  K = 9 is too thin for the matrix pipe, so this
// is a VALU kernel bound by its 128-B-per-pixel output stream: each thread owns one
// pooled pixel x 8 channels, reads the 4x4 input patch once and writes 32 contiguous
// bytes; four neighbouring threads complete a 128-B channels-last row.
#include "umt_internal.h"

namespace umt {

__device__ __forceinline__ unsigned stem_pack_bf16x2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void stem_kernel(StemParams p, const float* __restrict__ images,
                                                   int n, void* __restrict__ out) {
  __shared__ float sw[9 * 32], sb[32], ss[32], sh[32];
  for (int i = threadIdx.x; i < 9 * 32; i += 256) sw[i] = p.w[i];
  if (threadIdx.x < 32) {
    sb[threadIdx.x] = p.bias[threadIdx.x];
    ss[threadIdx.x] = p.scale[threadIdx.x];
    sh[threadIdx.x] = p.shift[threadIdx.x];
  }
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n * 48 * 48 * 4;
  if (t >= total) return;
  const int cg = (int)(t & 3);
  const int64_t pix = t >> 2;
  const int img = (int)(pix / (48 * 48));
  const int rem = (int)(pix - (int64_t)img * 48 * 48);
  const int yo = rem / 48, xo = rem - yo * 48;
  const float* src = images + (size_t)img * 96 * 96;
  float patch[4][4];
#pragma unroll
  for (int dy = 0; dy < 4; ++dy) {
    const int y = 2 * yo - 1 + dy;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      const int x = 2 * xo - 1 + dx;
      patch[dy][dx] = (y >= 0 && y < 96 && x >= 0 && x < 96) ? src[y * 96 + x] : 0.f;
    }
  }
  float res[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = cg * 8 + c;
    float best = 0.f;  // ReLU output is >= 0, so 0 is the identity of the pooled max
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) s = fmaf(sw[(ky * 3 + kx) * 32 + ch], patch[py + ky][px + kx], s);
        const float v = (s + sb[ch]) * ss[ch] + sh[ch];
        best = fmaxf(best, v);
      }
    res[c] = best;
  }
  if (OUT_BF16) {
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + (size_t)pix * 32 + cg * 8) =
        make_uint4(stem_pack_bf16x2(res[0], res[1]), stem_pack_bf16x2(res[2], res[3]),
                   stem_pack_bf16x2(res[4], res[5]), stem_pack_bf16x2(res[6], res[7]));
  } else {
    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)pix * 32 + cg * 8);
    dst[0] = make_float4(res[0], res[1], res[2], res[3]);
    dst[1] = make_float4(res[4], res[5], res[6], res[7]);
  }
}

template <bool OUT_BF16>
static int launch_stem_any(const StemParams& p, const float* images, int n, void* out, hipStream_t stream) {
  UMT_REQUIRE(n > 0, UMT_ERR_INVALID, "stem: n = %d", n);
  const int64_t total = (int64_t)n * 48 * 48 * 4;
  const int64_t blocks = (total + 255) / 256;
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
