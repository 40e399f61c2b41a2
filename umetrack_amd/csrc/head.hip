// Small kernels between the convolutions of the head: feature-transform layer (FTL),
// view concatenation, temporal memory gather/scatter, skeleton encoder, global average
// pool, pose decode (Procrustes) and layout converters for the C-ABI stage boundaries.
// All of them move a few KB per hand; none is on the roofline-critical path.
//
// 4x4 algebra is evaluated in double precision from the float32 inputs and rounded once,
// which lands within float32 rounding of the reference's float32 torch.inverse / matmul /
// svd chains (lib/models/feature_extractor.py:61-87, temporal.py:51-78, model_utils.py:17-54,
// umetrack_model.py:77-90).
#include "umt_internal.h"
#include "head_math.h"

namespace umt {

static inline unsigned grid_for(int64_t total, int block) { return (unsigned)((total + block - 1) / block); }

// ---------------------------------------------------------------- layout converters
__global__ void rows_to_nchw_kernel(const float* rows, int cs, int c0, int c, int64_t total, int hw, float* nchw) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int p = (int)(t % hw);
  const int ch = (int)((t / hw) % c);
  const int64_t b = t / ((int64_t)hw * c);
  nchw[t] = rows[(b * hw + p) * cs + c0 + ch];
}

__global__ void nchw_to_rows_kernel(const float* nchw, int c, int64_t total, int hw, float* rows, int cs, int c0) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int ch = (int)(t % c);
  const int p = (int)((t / c) % hw);
  const int64_t b = t / ((int64_t)hw * c);
  rows[(b * hw + p) * cs + c0 + ch] = nchw[(b * c + ch) * hw + p];
}

__global__ void zero_channels_kernel(float* rows, int cs, int c0, int c, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  rows[(t / c) * cs + c0 + (t % c)] = 0.f;
}

int launch_rows_to_nchw(const float* rows, int cs, int c0, int c, int n, int hw, float* nchw, hipStream_t s) {
  const int64_t total = (int64_t)n * c * hw;
  hipLaunchKernelGGL(rows_to_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, rows, cs, c0, c, total, hw, nchw);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_nchw_to_rows(const float* nchw, int c, int n, int hw, float* rows, int cs, int c0, hipStream_t s) {
  const int64_t total = (int64_t)n * c * hw;
  hipLaunchKernelGGL(nchw_to_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, nchw, c, total, hw, rows, cs, c0);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_zero_channels(float* rows, int cs, int c0, int c, int64_t nrows, hipStream_t s) {
  if (c <= 0 || nrows <= 0) return 0;
  const int64_t total = nrows * c;
  hipLaunchKernelGGL(zero_channels_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, rows, cs, c0, c, total);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void gather_cam0_kernel(const float* ext, const int64_t* sr, int n, int bs, float* cam0) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * 16) return;
  const int b = t >> 4;
  int64_t r0 = sr[2 * b];
  r0 = r0 < 0 ? 0 : (r0 >= n ? n - 1 : r0);   // an out-of-range index must not fault the GPU
  cam0[t] = ext[r0 * 16 + (t & 15)];
}

int launch_gather_cam0(const float* extrinsics, const int64_t* sample_range, int n, int bs, float* cam0, hipStream_t s) {
  hipLaunchKernelGGL(gather_cam0_kernel, dim3(grid_for((int64_t)bs * 16, 256)), dim3(256), 0, s, extrinsics, sample_range, n, bs, cam0);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- fusion transforms (A4, A5)
// crop_xf[n][12]: rows of [R|t] taking view features to the canonical space (2-view samples).
// sample_xf[bs][12]: canonical -> cam0 (= S of the sample's first crop).  sample_views[bs]: 1 or 2
// (0 marks a malformed range; its outputs become NaN instead of reading out of bounds).
__global__ void fusion_xforms_kernel(const float* K, const float* E, const int64_t* sr, int n, int bs,
                                     float* crop_xf, float* sample_xf, int32_t* sample_views, int32_t* status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  const int64_t r0 = sr[2 * b], r1 = sr[2 * b + 1];
  const int nv = (int)(r1 - r0);
  const bool okr = r0 >= 0 && r1 <= n && (nv == 1 || nv == 2);
  sample_views[b] = okr ? nv : 0;
  if (!okr) {
    if (status) atomicOr(status, UMT_STATUS_BAD_SAMPLE_RANGE);
    for (int i = 0; i < 12; ++i) sample_xf[b * 12 + i] = __builtin_nanf("");
    return;
  }
  const double s0 = (double)(K[r0 * 9] / 200.0f);   // model_utils.py:188-190, float32 division
  {
    float* o = sample_xf + b * 12;
    for (int i = 0; i < 12; ++i) o[i] = 0.f;
    o[0] = 1.f; o[5] = 1.f; o[10] = (float)s0;
  }
  if (nv == 2) {
    double e0[16], s0inv_e0[16];
    load_mat4(E + r0 * 16, e0);
    for (int i = 0; i < 16; ++i) s0inv_e0[i] = e0[i];
    for (int j = 0; j < 4; ++j) s0inv_e0[8 + j] = e0[8 + j] / s0;   // inverse(S0) @ E0 scales row 2
    for (int v = 0; v < 2; ++v) {
      double ev[16], evi[16], tw[16], tc[16];
      load_mat4(E + (r0 + v) * 16, ev);
      mat4_inv(ev, evi);
      const double sv = (double)(K[(r0 + v) * 9] / 200.0f);
      for (int i = 0; i < 16; ++i) tw[i] = evi[i];
      for (int i = 0; i < 4; ++i) tw[i * 4 + 2] = evi[i * 4 + 2] * sv;     // inverse(E_v) @ S_v scales column 2
      mat4_mul(s0inv_e0, tw, tc);
      float* o = crop_xf + (r0 + v) * 12;
      for (int i = 0; i < 12; ++i) o[i] = (float)tc[i];
    }
  }
}

int launch_fusion_xforms(const float* intrinsics, const float* extrinsics, const int64_t* sample_range,
                         int n, int bs, float* crop_xf, float* sample_xf, int32_t* sample_views, int32_t* status,
                         hipStream_t s) {
  hipLaunchKernelGGL(fusion_xforms_kernel, dim3(grid_for(bs, 64)), dim3(64), 0, s, intrinsics, extrinsics,
                     sample_range, n, bs, crop_xf, sample_xf, sample_views, status);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// FTL on channels-last rows: the reference reshapes (C,6,6) to (3, C/3*36), so channel c,
// c+C/3, c+2C/3 of one pixel form one 3-D point (model_utils.py:89-99).

// cat_rows [bs*36][144]: channels v*72.. of sample b = FTL(to_canon[r0+v]) of crop r0+v; zeros for 1-view samples
__global__ void ftl_concat_kernel(const float* feat, const float* crop_xf, const int64_t* sr,
                                  const int32_t* views, int bs, float* cat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)bs * 36 * 2 * 24) return;
  const int c = (int)(t % 24);
  const int v = (int)((t / 24) % 2);
  const int p = (int)((t / 48) % 36);
  const int b = (int)(t / (48 * 36));
  float* o = cat + ((size_t)b * 36 + p) * 144 + v * 72;
  if (views[b] != 2) {
    const float fill = views[b] == 1 ? 0.f : __builtin_nanf("");
    o[c] = fill; o[c + 24] = fill; o[c + 48] = fill;
    return;
  }
  const int64_t crop = sr[2 * b] + v;
  const float* f = feat + ((size_t)crop * 36 + p) * 72;
  float x, y, z;
  ftl_point(crop_xf + crop * 12, f[c], f[c + 24], f[c + 48], x, y, z);
  o[c] = x; o[c + 24] = y; o[c + 48] = z;
}

int launch_ftl_concat(const float* feat_rows, const float* crop_xf, const int64_t* sample_range,
                      const int32_t* sample_views, int n, int bs, float* cat_rows, hipStream_t s) {
  (void)n;
  hipLaunchKernelGGL(ftl_concat_kernel, dim3(grid_for((int64_t)bs * 36 * 48, 256)), dim3(256), 0, s, feat_rows,
                     crop_xf, sample_range, sample_views, bs, cat_rows);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// out[b] = FTL(S0) of the fused conv output (2 views) or FTL(S) of the crop's own features (1 view)
// zero_rest: also writes zeros to the channels [0, out_c0) and [out_c0 + 72, out_cs) of every output row (at most 24: one
// per thread of a pixel), which saves the caller a memset launch of the whole tensor
__global__ void ftl_select_kernel(const float* conv, const float* feat, const float* sample_xf, const int64_t* sr,
                                  const int32_t* views, int bs, float* out, int out_cs, int out_c0, int zero_rest) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)bs * 36 * 24) return;
  const int c = (int)(t % 24);
  const int p = (int)((t / 24) % 36);
  const int b = (int)(t / (24 * 36));
  const float* f = views[b] == 2 ? conv + ((size_t)b * 36 + p) * 72
                                 : feat + ((size_t)(views[b] == 1 ? sr[2 * b] : 0) * 36 + p) * 72;
  float x, y, z;
  ftl_point(sample_xf + b * 12, f[c], f[c + 24], f[c + 48], x, y, z);
  float* o = out + ((size_t)b * 36 + p) * out_cs + out_c0;
  o[c] = x; o[c + 24] = y; o[c + 48] = z;
  if (zero_rest && c < out_cs - 72) o[(c < out_c0 ? c : c + 72) - out_c0] = 0.f;
}

int launch_ftl_select(const float* conv_rows, const float* feat_rows, const float* sample_xf,
                      const int64_t* sample_range, const int32_t* sample_views, int n, int bs,
                      float* out_rows, int out_cs, int out_c0, hipStream_t s, bool zero_rest) {
  (void)n;
  UMT_REQUIRE(!zero_rest || (out_cs - 72 >= 0 && out_cs - 72 <= 24 && out_c0 >= 0 && out_c0 + 72 <= out_cs), UMT_ERR_INVALID,
              "ftl_select: cannot zero %d other channels", out_cs - 72);
  hipLaunchKernelGGL(ftl_select_kernel, dim3(grid_for((int64_t)bs * 36 * 24, 256)), dim3(256), 0, s, conv_rows,
                     feat_rows, sample_xf, sample_range, sample_views, bs, out_rows, out_cs, out_c0, zero_rest ? 1 : 0);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- temporal memory (A8)
// The temporal step reads ALL previous state before it writes any (temporal.py:66-77 builds every relative
// transform from prev_extrinsics and only then assigns prev_extrinsics[memory_idx] = cur).  Three launches in stream
// order give the same ordering on the device:
//   1. temporal_xforms_kernel   reads prev_E[idx], writes rel_xf[b], claims row idx for sample b
//   2. temporal_gather_kernel   reads mem[idx] (and, in the same launch, the winners write prev_E[idx] = cur_E)
//   3. temporal_split_kernel    (after the three convolutions) winners write mem[idx]
// Duplicate memory_idx values in one batch: every duplicate READS the same previous state; the sample with the
// HIGHEST batch index writes the row (torch's CPU index assignment is last-wins too).  The claim is an atomicMax of
// (epoch << 32 | b) into mem_owner[idx], epoch = one per step, so no claim array has to be cleared.
// row_of[b]: memory row of sample b, or -1 (index out of range: UMT_STATUS_BAD_MEMORY_IDX; sample runs without memory);
// bit 30 set = sample b reads the row but another sample of the batch owns the write.
constexpr int32_t ROW_NOT_OWNER = 1 << 30;

__global__ void temporal_xforms_kernel(const float* prev_ext, unsigned long long* owner, unsigned epoch, int mem_rows,
                                       const float* cam0, const int64_t* midx, const uint8_t* use, int bs, float* rel_xf,
                                       int32_t* row_of, uint8_t* use_eff, int32_t* status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  const int64_t idx = midx[b];
  const bool okr = idx >= 0 && idx < mem_rows;
  row_of[b] = okr ? (int32_t)idx : -1;
  float* o = rel_xf + b * 12;
  for (int i = 0; i < 12; ++i) o[i] = 0.f;
  uint8_t u = 0;
  if (!okr) {
    atomicOr(status, UMT_STATUS_BAD_MEMORY_IDX);
  } else {
    if (use[b]) {
      double pe[16], pinv[16], ce[16], rel[16];
      load_mat4(prev_ext + idx * 16, pe);
      load_mat4(cam0 + b * 16, ce);
      if (mat4_inv(pe, pinv)) {
        mat4_mul(ce, pinv, rel);
        for (int i = 0; i < 12; ++i) o[i] = (float)rel[i];
        u = 1;
      } else {      // torch.inverse raises here (temporal.py:66); the row is treated as use_memory = False and flagged
        atomicOr(status, UMT_STATUS_SINGULAR_PREV_EXTRINSICS);
      }
    }
    atomicMax(owner + idx, ((unsigned long long)epoch << 32) | (unsigned)b);
  }
  use_eff[b] = u;
}

// tin_rows[b*36+p][0..17] = use ? FTL(rel, mem[idx]) : 0   (temporal.py:59-74, 83-85); mem is (rows,18,6,6).
// Thread (c == 0, p < 16) of the winning sample also stores element p of prev_E[idx] = cur_E (temporal.py:77).
__global__ void temporal_gather_kernel(const float* mem, float* prev_ext, const unsigned long long* owner, unsigned epoch,
                                       const float* cam0, const float* rel_xf, int32_t* row_of, const uint8_t* use,
                                       int bs, float* tin, int tin_cs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * 36 * 6) return;
  const int c = t % 6;
  const int p = (t / 6) % 36;
  const int b = t / (6 * 36);
  float x = 0.f, y = 0.f, z = 0.f;
  const int32_t ro = row_of[b];                    // launch 1 wrote plain rows; only this launch adds the bit
  const bool has_row = ro >= 0;
  const int row = ro & ~ROW_NOT_OWNER;
  if (has_row && c == 0) {
    const bool mine = owner[row] == (((unsigned long long)epoch << 32) | (unsigned)b);
    if (mine && p < 16) prev_ext[(size_t)row * 16 + p] = cam0[b * 16 + p];
    if (!mine && p == 35) row_of[b] = row | ROW_NOT_OWNER;      // read by temporal_split_kernel only
  }
  if (has_row && use[b]) {
    const float* m = mem + (size_t)row * 18 * 36;
    ftl_point(rel_xf + b * 12, m[c * 36 + p], m[(c + 6) * 36 + p], m[(c + 12) * 36 + p], x, y, z);
  }
  float* o = tin + ((size_t)b * 36 + p) * tin_cs;
  o[c] = x; o[c + 6] = y; o[c + 12] = z;
}

int launch_temporal_gather(float* mem, float* prev_ext, unsigned long long* owner, unsigned epoch, int mem_rows,
                           const float* cam0_ext, const int64_t* memory_idx, const uint8_t* use_memory, int bs,
                           float* tin_rows, int tin_cs, float* rel_xf, int32_t* row_of, int32_t* status, hipStream_t s) {
  uint8_t* use_eff = reinterpret_cast<uint8_t*>(rel_xf + (size_t)bs * 12);     // carve_head reserves bs bytes behind rel_xf
  hipLaunchKernelGGL(temporal_xforms_kernel, dim3(grid_for(bs, 64)), dim3(64), 0, s, prev_ext, owner, epoch, mem_rows,
                     cam0_ext, memory_idx, use_memory, bs, rel_xf, row_of, use_eff, status);
  hipLaunchKernelGGL(temporal_gather_kernel, dim3(grid_for((int64_t)bs * 36 * 6, 256)), dim3(256), 0, s, mem, prev_ext,
                     owner, epoch, cam0_ext, rel_xf, row_of, use_eff, bs, tin_rows, tin_cs);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// channels 0..17 of the temporal output overwrite the memory row, 18..89 go on (temporal.py:88-89,134)
// zero_pad: also writes zeros to the channels [72, out_cs) of every output row (the skeleton channels, which the skeleton
// kernel fills afterwards, and the padding behind them) instead of a memset launch of the whole tensor by the caller
__global__ void temporal_split_kernel(const float* tout, int tout_cs, float* mem, const int32_t* row_of, int bs,
                                      float* out, int out_cs, int zero_pad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * 36 * 90) return;
  const int c = t % 90;
  const int p = (t / 90) % 36;
  const int b = t / (90 * 36);
  const float v = tout[((size_t)b * 36 + p) * tout_cs + c];
  if (c < 18) {
    const int row = row_of[b];
    if (row >= 0 && !(row & ROW_NOT_OWNER)) mem[(size_t)row * 18 * 36 + c * 36 + p] = v;      // the claiming sample only
  } else {
    out[((size_t)b * 36 + p) * out_cs + (c - 18)] = v;
    if (zero_pad && 72 + (c - 18) < out_cs) out[((size_t)b * 36 + p) * out_cs + 72 + (c - 18)] = 0.f;
  }
}

int launch_temporal_split(const float* tout_rows, int tout_cs, float* mem, const int32_t* row_of, int bs,
                          float* out_rows, int out_cs, hipStream_t s, bool zero_pad) {
  UMT_REQUIRE(!zero_pad || (out_cs >= 72 && out_cs <= 144), UMT_ERR_INVALID, "temporal_split: cannot zero %d channels", out_cs - 72);
  hipLaunchKernelGGL(temporal_split_kernel, dim3(grid_for((int64_t)bs * 36 * 90, 256)), dim3(256), 0, s, tout_rows,
                     tout_cs, mem, row_of, bs, out_rows, out_cs, zero_pad ? 1 : 0);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- skeleton encoder (A9)
// Linear(132,144) on cat(axes, rest) per joint -> view (4,6,6) -> BN(4) -> ReLU (skeleton_encoder.py:36-53);
// a single profile is broadcast to every sample (umetrack_model.py:202-207).
__global__ void skeleton_kernel(SkelParams p, const float* axes, const float* rest, int n_skel, int bs,
                                float* rows, int cs, int c0) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * 144) return;
  const int o = t % 144;
  const int b = t / 144;
  const int sk = n_skel == 1 ? 0 : b;
  const float* a = axes + (size_t)sk * 66;
  const float* r = rest + (size_t)sk * 66;
  const float* w = p.w + (size_t)o * 132;
  float s = 0.f;
  for (int j = 0; j < 22; ++j) {
    s = fmaf(w[j * 6 + 0], a[j * 3 + 0], s);
    s = fmaf(w[j * 6 + 1], a[j * 3 + 1], s);
    s = fmaf(w[j * 6 + 2], a[j * 3 + 2], s);
    s = fmaf(w[j * 6 + 3], r[j * 3 + 0], s);
    s = fmaf(w[j * 6 + 4], r[j * 3 + 1], s);
    s = fmaf(w[j * 6 + 5], r[j * 3 + 2], s);
  }
  const int c = o / 36, px = o % 36;
  const float lin = (s + p.bias[o]) * p.scale[c] + p.shift[c];
  const float v = lin != lin ? lin : fmaxf(lin, 0.f);      // ReLU that keeps a NaN, like torch's
  rows[((size_t)b * 36 + px) * cs + c0 + c] = v;
}

// One profile for the whole batch (umetrack_model.py:202-207; the tracker and the bench): every workgroup evaluates the 144
// outputs once into LDS (the same fmaf chain as above) and writes them, four channels = 16 bytes per thread, to its 256
// (sample, pixel) rows - the kernel above spent 132 multiply-adds and a 4-byte store per output of every sample.
__global__ __launch_bounds__(256) void skeleton_broadcast_kernel(SkelParams p, const float* axes, const float* rest, int bs,
                                                                 float* rows, int cs, int c0) {
  __shared__ float sv[144];
  const int o = threadIdx.x;
  if (o < 144) {
    const float* w = p.w + (size_t)o * 132;
    float s = 0.f;
    for (int j = 0; j < 22; ++j) {
      s = fmaf(w[j * 6 + 0], axes[j * 3 + 0], s);
      s = fmaf(w[j * 6 + 1], axes[j * 3 + 1], s);
      s = fmaf(w[j * 6 + 2], axes[j * 3 + 2], s);
      s = fmaf(w[j * 6 + 3], rest[j * 3 + 0], s);
      s = fmaf(w[j * 6 + 4], rest[j * 3 + 1], s);
      s = fmaf(w[j * 6 + 5], rest[j * 3 + 2], s);
    }
    const int c = o / 36;
    const float lin = (s + p.bias[o]) * p.scale[c] + p.shift[c];
    sv[o] = lin != lin ? lin : fmaxf(lin, 0.f);
  }
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (sample, pixel)
  if (row >= (int64_t)bs * 36) return;
  const int px = (int)(row % 36);
  *reinterpret_cast<float4*>(rows + row * cs + c0) = make_float4(sv[px], sv[36 + px], sv[72 + px], sv[108 + px]);
}

int launch_skeleton(const SkelParams& p, const float* axes, const float* rest, int n_skel, int bs, float* rows,
                    int cs, int c0, hipStream_t s) {
  if (n_skel == 1 && cs % 4 == 0 && c0 % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {
    hipLaunchKernelGGL(skeleton_broadcast_kernel, dim3(grid_for((int64_t)bs * 36, 256)), dim3(256), 0, s, p, axes, rest, bs, rows,
                       cs, c0);
    UMT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(skeleton_kernel, dim3(grid_for((int64_t)bs * 144, 256)), dim3(256), 0, s, p, axes, rest, n_skel,
                     bs, rows, cs, c0);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- global average pool (A10 tail)
__global__ void avgpool_kernel(const float* rows, int cs, int c, int bs, float* raw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * c) return;
  const int ch = t % c;
  const int b = t / c;
  const float* r = rows + (size_t)b * 36 * cs + ch;
  float s = 0.f;
  for (int p = 0; p < 36; ++p) s += r[(size_t)p * cs];
  raw[t] = s / 36.f;
}

int launch_avgpool(const float* rows, int cs, int c, int bs, float* raw, hipStream_t s) {
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for((int64_t)bs * c, 256)), dim3(256), 0, s, rows, cs, c, bs, raw);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- pose decode (A11, A12)
// 3x3 SVD by one-sided Jacobi (Hestenes): rotates column pairs of A = H and of V until the
// columns of A are orthogonal; then A = U*diag(s).  Only R = V*diag(1,1,det(V U^T))*U^T is
// used, which is unique whenever the reference's result is (SURVEY.md section 7 "hard parts").
// Every array index is a compile-time constant (unrolled loops; the smallest singular value's column is placed by
// selects, not by a run-time index), so nothing lives in scratch memory: the kernel is one thread per hand and in the
// tracker's per-frame call its latency is on the critical path.  Same operations in the same order as the indexed form.
__device__ __forceinline__ void kabsch_rotation(const double* H, double* R) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i] = H[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          al += A[i * 3 + p] * A[i * 3 + p];
          be += A[i * 3 + q] * A[i * 3 + q];
          ga += A[i * 3 + p] * A[i * 3 + q];
        }
        if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        off = fmax(off, fabs(ga) / sqrt(al * be));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - s * aq;
          A[i * 3 + q] = s * ap + c * aq;
          const double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - s * vq;
          V[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3], N[9], U[9];
  int kmin = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
  {
    double svmin = sv[0];                       // first of the smallest, as `if (sv[j] < sv[kmin]) kmin = j`
    if (sv[1] < svmin) { svmin = sv[1]; kmin = 1; }
    if (sv[2] < svmin) { svmin = sv[2]; kmin = 2; }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {                 // the columns as they are when j is not the smallest
    const double inv = sv[j] > 0 ? 1.0 / sv[j] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) N[i * 3 + j] = A[i * 3 + j] * inv;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    constexpr int nx[3] = {1, 2, 0}, nn[3] = {2, 0, 1};
    const int ka = nx[j], kb = nn[j];
    // column j if it IS the smallest: divided by its singular value, or for a rank-2 H the cross product of the other two
    // (any sign gives the same R after the det fix)
    const bool full = sv[j] > 1e-12 * fmax(sv[ka], sv[kb]);
    const double x0 = N[1 * 3 + ka] * N[2 * 3 + kb] - N[2 * 3 + ka] * N[1 * 3 + kb];
    const double x1 = N[2 * 3 + ka] * N[0 * 3 + kb] - N[0 * 3 + ka] * N[2 * 3 + kb];
    const double x2 = N[0 * 3 + ka] * N[1 * 3 + kb] - N[1 * 3 + ka] * N[0 * 3 + kb];
    const double s0 = full ? A[0 * 3 + j] / sv[j] : x0, s1 = full ? A[1 * 3 + j] / sv[j] : x1, s2 = full ? A[2 * 3 + j] / sv[j] : x2;
    U[0 * 3 + j] = j == kmin ? s0 : N[0 * 3 + j];
    U[1 * 3 + j] = j == kmin ? s1 : N[1 * 3 + j];
    U[2 * 3 + j] = j == kmin ? s2 : N[2 * 3 + j];
  }
  auto det3 = [](const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  };
  const double d = det3(V) * det3(U);   // det(V U^T), +-1 up to rounding
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += V[i * 3 + k] * (k == kmin ? d : 1.0) * U[j * 3 + k];
      R[i * 3 + j] = s;
    }
}

// A half-wave (32 lanes) per hand.  The Procrustes fit and the wrist transform are one dependent float64 chain and every
// lane of the half-wave walks it (lane 0 stores); what is independent per element - the finiteness test, the joint angles,
// the 21 softplus values (a float64 exp and log1p each) - is spread over the lanes: one thread per hand spent a third of
// its 33 us there, and in the tracker's per-frame call that latency is on the critical path.
constexpr int DECODE_LANES = 32;

__global__ __launch_bounds__(64) void decode_kernel(const float* raw, int nraw, int predict_scale, const float* cam0,
                                                    const int64_t* hand_idx, int bs, float* joint_angles, float* wrist_xfs,
                                                    float* sigmas, float* skel_scales, int32_t* status) {
  const int half = threadIdx.x / DECODE_LANES, l = threadIdx.x % DECODE_LANES;
  const int b = blockIdx.x * (64 / DECODE_LANES) + half;
  const bool have = b < bs;
  const float* r = raw + (size_t)(have ? b : 0) * nraw;
  // torch's ReLU / MaxPool propagate NaN; the integer ReLU and the pooled max of the backbone kernels may drop one
  // (conv_f32_common.h, stem.hip).  A non-finite regressor row is therefore reported, and the whole row of outputs
  // is NaN, so that a corrupted crop or weight cannot come out as a plausible pose.
  bool mine = true;
  for (int i = l; i < nraw; i += DECODE_LANES) mine = mine && isfinite(r[i]);
  const unsigned long long notfin = __ballot(!mine);
  const bool finite = ((notfin >> (DECODE_LANES * half)) & 0xffffffffull) == 0;
  if (!have) return;
  if (!finite) {
    if (status && l == 0) atomicOr(status, UMT_STATUS_NONFINITE_OUTPUT);
    const float q = __builtin_nanf("");
    if (l < 22) joint_angles[b * 22 + l] = q;
    if (l < 16) wrist_xfs[b * 16 + l] = q;
    if (l < 21) sigmas[b * 21 + l] = q;
    if (predict_scale && l == 0) skel_scales[b] = q;
    return;
  }
  // regressor.py:76-85
  if (l < 22) joint_angles[b * 22 + l] = l < 20 ? r[l] : 0.f;
  {
    int o = 41;
    if (predict_scale) {   // regressor.py:107-114
      if (l == 0) skel_scales[b] = expf(r[41]);
      o = 42;
    }
    if (l < 21) {          // regressor.py:117-121: clamp(softplus(x), 1e-5), torch threshold 20
      const float x = r[o + l];
      const float sp = x > 20.f ? x : (float)log1p(exp((double)x));
      sigmas[b * 21 + l] = fmaxf(sp, 1e-5f);
    }
  }
  // regressor.py:19-47, 88-104: seven source points, float32 like the reference's tensor
  const double raw_src[7][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, -1, 0}, {-1, 0, -1}, {0, -1, -1}};
  double src[7][3], dst[7][3], ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double nrm = sqrt(raw_src[i][0] * raw_src[i][0] + raw_src[i][1] * raw_src[i][1] + raw_src[i][2] * raw_src[i][2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      src[i][k] = nrm > 0 ? (double)(float)(raw_src[i][k] / nrm * 0.1) : 0.0;
      dst[i][k] = (double)r[20 + i * 3 + k];
      ms[k] += src[i][k];
      md[k] += dst[i][k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { ms[k] /= 7.0; md[k] /= 7.0; }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // model_utils.py:36-39: from_centered^T @ to_centered
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) H[a * 3 + c] += (src[i][a] - ms[a]) * (dst[i][c] - md[c]);
  double R[9];
  kabsch_rotation(H, R);
  double wc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) wc[i * 4 + j] = R[i * 3 + j];
    wc[i * 4 + 3] = md[i] - (R[i * 3 + 0] * ms[0] + R[i * 3 + 1] * ms[1] + R[i * 3 + 2] * ms[2]);
  }
  // umetrack_model.py:77-90: inverse(cam0 extrinsics) @ wrist, mirror x for right hands
  double e[16], ei[16], ww[16];
  load_mat4(cam0 + b * 16, e);
  mat4_inv(e, ei);
  mat4_mul(ei, wc, ww);
  const bool right = hand_idx[b] == 1;
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) wrist_xfs[b * 16 + i * 4 + j] = (float)((right && j == 0) ? -ww[i * 4 + j] : ww[i * 4 + j]);
  }
}

int launch_decode(const float* raw, int nraw, int predict_scale, const float* cam0_ext, const int64_t* hand_idx,
                  int bs, float* joint_angles, float* wrist_xfs, float* sigmas, float* skel_scales, int32_t* status,
                  hipStream_t s) {
  UMT_REQUIRE(nraw == (predict_scale ? 63 : 62), UMT_ERR_INVALID, "decode: nraw = %d", nraw);
  UMT_REQUIRE(!predict_scale || skel_scales, UMT_ERR_INVALID, "decode: scale head needs a skel_scales buffer");
  hipLaunchKernelGGL(decode_kernel, dim3(grid_for(bs, 64 / DECODE_LANES)), dim3(64), 0, s, raw, nraw, predict_scale, cam0_ext,
                     hand_idx, bs, joint_angles, wrist_xfs, sigmas, skel_scales, status);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace umt
