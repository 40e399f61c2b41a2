// The recurrent part of the model for whole recordings in lock-step, one launch for all T frames:
//   SimpleConvRNN.forward_temporal_features (lib/models/temporal.py:93-139) -> skeleton channels ->
//   PoseRegressor layers (lib/models/regressor.py:163-186, model_utils.py:195-208) -> global average pool,
// for R memory rows (one per (recording, hand)) x T time steps.  Only this part of the network depends on the previous
// frame, and per step it is tiny: stepped with one launch per layer it is bound by launch
// latency (about 19 launches of a few microseconds per step).  Here ONE workgroup owns one memory row for the whole
// recording: its 36 pixel rows, the temporal memory and every intermediate tensor live in LDS, the weights stream from
// L2 (about 1 MB per step), and a step is a chain of small matrix products on v_mfma_f32_16x16x4_f32 (16 x 16 output
// tiles: 36 pixels pad to 48, 76..90 channels to 80..96; the 32 x 32 tiles of conv_igemm.hip would pad 36 -> 64).
//
// Arithmetic is that of the per-layer kernels, bit for bit: the same fmaf chain per output (32-channel chunk, then
// tap, then 8-channel group, inside a group input channels 0,4,1,5,2,6,3,7 - the order in which
// v_mfma_f32_32x32x2_f32 consumes conv_igemm.hip's packed weights), the same epilogue expression, the same FTL and
// average-pool order.  tests/test_gpu_tracker.py holds the two paths equal with torch.equal.
#include "umt_internal.h"
#include "head_math.h"

namespace umt {

constexpr int SQ_THREADS = 1024, SQ_WAVES = SQ_THREADS / 64;
constexpr int SQ_LD = 100;             // floats per LDS activation row: up to 96 channels + 4 (400 B rows)
constexpr int SQ_ZERO = 36;            // index of the all-zero row behind the 36 pixel rows
constexpr int SQ_BUF = (SQ_ZERO + 1) * SQ_LD;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SeqLayer {
  const float* w16;      // packed [kk][4][cout_p][4], see pack_conv_seq in model.hip
  const float* bias;     // [>= cout_p], padded entries 0
  const float* scale;    // padded entries 1
  const float* shift;    // padded entries 0
  int cin_p, cout_p;     // channel counts padded to 16
  int taps, relu;
};

struct SeqArgs {
  const float* fused_t;   // (T, R, 72, 36) fused features, reference layout
  const float* cam0_t;    // (T, R, 16)
  const float* rel_t;     // (T, R, 12) relative transforms cur_E . inverse(prev_E), zeros where unused
  const uint8_t* use_t;   // (T, R) effective use_memory flags
  const float* skel;      // (n_skel, 36, 4) skeleton-encoder channels, or null (scale head)
  float* mem;             // handle state (rows, 18, 36)
  float* prev_ext;        // handle state (rows, 16)
  float* raw_t;           // (T, R, nraw) regressor outputs (one-kernel mode)
  float* rin;             // (T * R * 36, rin_cs) regressor input rows (split mode: the regressor runs batched afterwards)
  int rin_cs;             // 80 (known skeleton: channels 72..75 = skeleton, 76..79 zero) or 72 (scale head)
  SeqLayer temp[3], conv[4], out;
  int T, R, n_skel, nraw;
};

// position of channel c inside an activation row: within each block of 16 channels the four values one lane feeds to
// four consecutive MFMAs are contiguous (one ds_read_b128).  MFMA s of a block uses, in k slot j, channel
// 8 (s >> 1) + 2 (s & 1) + {0, 4, 1, 5}[j]; that channel sits at 4 j + s.
__device__ __forceinline__ int sq_pos(int c) {
  const int c16 = c & 15;
  return (c & ~15) + 4 * (((c16 & 1) << 1) | ((c16 & 7) >> 2)) + 2 * (c16 >> 3) + ((c16 >> 1) & 1);
}

// One convolution (1x1 or 3x3 on the 6x6 map) from LDS rows `in` to LDS rows `out`: 16 x 16 output tiles dealt to the
// 16 waves (pixel tile fastest).  Bit-identical to launch_conv on the same data (see the file header).
// Per k group (16 input channels of one tap) a wave issues one ds_read_b128 (activations, one group ahead), one 16-byte
// buffer load (weights, four groups ahead in a ring of four register sets with static indices, scalar offset) and four
// dependent MFMAs; the address arithmetic is a bit-field extract from the lane's packed table of operand rows plus
// a few scalar instructions (16 waves share the CU's scalar unit: a first version with ~45 scalar instructions per
// group ran at 40 % of the matrix pipe's rate).
// k group kk of a layer with TAPS taps and CINP (padded) input channels: walk order 32-channel chunk, tap, 16-channel
// half of the chunk.  Everything about a group is a compile-time constant of the fully unrolled loop.
template <int TAPS, int CINP>
struct SqWalk {
  static constexpr int NKK = TAPS * (CINP / 16);
  static constexpr int NKK4 = (NKK + 3) & ~3;
  static constexpr int full = (CINP / 32) * TAPS * 2;      // groups in the full 32-channel chunks
  static constexpr int chunk(int kk) { return kk < full ? kk / (2 * TAPS) : CINP / 32; }
  static constexpr int tap(int kk) { return kk >= NKK ? 9 : kk < full ? (kk % (2 * TAPS)) / 2 : kk - full; }
  static constexpr int col(int kk) { return kk >= NKK ? 0 : kk < full ? 32 * chunk(kk) + 16 * (kk & 1) : 32 * (CINP / 32); }
};

template <int TAPS, int CINP>
__device__ __forceinline__ void sq_conv(const SeqLayer& L, const float* __restrict__ in, float* __restrict__ out,
                                        const float* __restrict__ res, int wave, int lane) {
  typedef SqWalk<TAPS, CINP> Wk;
  const int r = lane & 15, j = lane >> 4;
  const int n_nt = L.cout_p >> 4, ntiles = 3 * n_nt;
  const int kbytes = 16 * L.cout_p * 4;                 // bytes between two k groups of the packed weights
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w16), 0, (Wk::NKK4 + 4) * kbytes, 0x00020000);
  for (int tile = wave; tile < ntiles; tile += SQ_WAVES) {
    const int mt = tile % 3, nt = tile / 3;
    const int px = mt * 16 + r;
    const int py = px / 6, pxx = px - py * 6;
    // float index of the lane's operand row per tap (row 36 = the all-zero row); entry 9 serves the padding groups
    int rowoff[10];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      int row = SQ_ZERO;
      if (TAPS == 9) {
        const int yy = py + tap / 3 - 1, xx = pxx + tap % 3 - 1;
        if (px < 36 && (unsigned)yy < 6u && (unsigned)xx < 6u) row = yy * 6 + xx;
      } else if (tap == 0 && px < 36) {
        row = px;
      }
      rowoff[tap] = row * SQ_LD + 4 * j;
    }
    rowoff[9] = SQ_ZERO * SQ_LD + 4 * j;
    const int voff = ((j * L.cout_p + nt * 16 + r) * 4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, i * kbytes, 0));
    float4 av[2];
    av[0] = *reinterpret_cast<const float4*>(in + rowoff[Wk::tap(0)] + Wk::col(0));
#pragma unroll
    for (int kk = 0; kk < Wk::NKK4; ++kk) {
      // the next group's activations (the last trip reads the zero row)
      av[(kk + 1) & 1] = *reinterpret_cast<const float4*>(in + rowoff[Wk::tap(kk + 1)] + Wk::col(kk + 1));
      __builtin_amdgcn_sched_barrier(0);
      const float4 a4 = av[kk & 1];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bq[kk & 3][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bq[kk & 3][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bq[kk & 3][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bq[kk & 3][3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // four groups ahead (all-zero groups past the end)
      bq[kk & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, (kk + 4) * kbytes, 0));
    }
    // accumulator: column (output channel) = lane & 15, rows (pixels) 4 j .. 4 j + 3
    const int col = nt * 16 + r;
    const float cb = L.bias[col], cs = L.scale[col], ch = L.shift[col];
    const int cpos = sq_pos(col);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int po = mt * 16 + 4 * j + k;
      if (po < 36) {
        float v = (acc[k] + cb) * cs + ch;
        if (res) v += res[po * SQ_LD + cpos];
        if (L.relu) v = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
        out[po * SQ_LD + cpos] = v;
      }
    }
  }
}

// REGRESS: the regressor layers and the average pool run here too, step by step (the round-2 form, kept as the check
// and for UMT_SEQ_ONEKERNEL).  Otherwise a step ends after the temporal module: its output rows (+ skeleton channels)
// go to a.rin and the regressor, which feeds nothing back, runs afterwards over all T x R rows with the batched
// kernels - nine tenths of a step's matrix work leaves the serial chain.
template <bool REGRESS>
__global__ __launch_bounds__(SQ_THREADS) void head_seq_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) float buf[3][SQ_BUF];
  __shared__ float smem[18 * 36];
  __shared__ float sskel[36 * 4];
  __shared__ float srel[12];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  for (int i = tid; i < 3 * SQ_BUF; i += SQ_THREADS) (&buf[0][0])[i] = 0.f;
  for (int i = tid; i < 18 * 36; i += SQ_THREADS) smem[i] = a.mem[(size_t)row * 18 * 36 + i];
  if (tid < 36 * 4) sskel[tid] = a.skel ? a.skel[(size_t)(a.n_skel == 1 ? 0 : row) * 36 * 4 + tid] : 0.f;
  __syncthreads();
  float* b0 = buf[0];
  float* b1 = buf[1];
  float* b2 = buf[2];
  // the step's inputs (10 KB of fused features, the relative transform, the use flag) are fetched one step ahead into
  // registers: their HBM latency runs under the previous step's convolutions
  constexpr int NF = (72 * 36 + SQ_THREADS - 1) / SQ_THREADS;
  float fnext[NF], rel_next = 0.f;
  uint8_t use_next;
  auto fetch = [&](int t) {
    const size_t tr = (size_t)t * a.R + row;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const int i = tid + k * SQ_THREADS;
      fnext[k] = i < 72 * 36 ? a.fused_t[tr * 72 * 36 + i] : 0.f;
    }
    use_next = a.use_t[tr];
    if (tid < 12) rel_next = a.rel_t[tr * 12 + tid];
  };
  fetch(0);
  for (int t = 0; t < a.T; ++t) {
    const size_t tr = (size_t)t * a.R + row;
    const bool use = use_next != 0;
    if (tid < 12) srel[tid] = rel_next;
    // tin = [FTL(rel, memory) or 0 (18 channels), fused features (72), zeros (6)]  (temporal.py:59-85)
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const int i = tid + k * SQ_THREADS;
      if (i < 72 * 36) {
        const int c = i / 36, p = i - c * 36;
        b0[p * SQ_LD + sq_pos(18 + c)] = fnext[k];
      }
    }
    __syncthreads();
    if (t + 1 < a.T) fetch(t + 1);
    if (tid < 36 * 6) {
      const int c = tid % 6, p = tid / 6;
      float x = 0.f, y = 0.f, z = 0.f;
      if (use) ftl_point(srel, smem[c * 36 + p], smem[(c + 6) * 36 + p], smem[(c + 12) * 36 + p], x, y, z);
      b0[p * SQ_LD + sq_pos(c)] = x;
      b0[p * SQ_LD + sq_pos(c + 6)] = y;
      b0[p * SQ_LD + sq_pos(c + 12)] = z;
      b0[p * SQ_LD + sq_pos(90 + c)] = 0.f;
    }
    __syncthreads();
    sq_conv<1, 96>(a.temp[0], b0, b1, nullptr, wave, lane);
    __syncthreads();
    sq_conv<1, 96>(a.temp[1], b1, b0, nullptr, wave, lane);
    __syncthreads();
    sq_conv<1, 96>(a.temp[2], b0, b1, nullptr, wave, lane);
    __syncthreads();
    // channels 0..17 overwrite the memory, 18..89 go on with the skeleton channels behind them (temporal.py:88-89,
    // regressor.py:170-176)
    if (REGRESS) {
      for (int i = tid; i < 36 * 96; i += SQ_THREADS) {
        const int p = i / 96, c = i - p * 96;
        if (c < 18) {
          smem[c * 36 + p] = b1[p * SQ_LD + sq_pos(c)];
        } else if (c < 90) {
          b0[p * SQ_LD + sq_pos(c - 18)] = b1[p * SQ_LD + sq_pos(c)];
        } else if (c < 94) {
          b0[p * SQ_LD + sq_pos(72 + c - 90)] = sskel[p * 4 + (c - 90)];
        } else {              // channels 76..79 (two threads per pixel write two zeros each)
          b0[p * SQ_LD + sq_pos(76 + 2 * (c - 94))] = 0.f;
          b0[p * SQ_LD + sq_pos(77 + 2 * (c - 94))] = 0.f;
        }
      }
    } else {
      float* __restrict__ rrow = a.rin + tr * 36 * a.rin_cs;
      for (int i = tid; i < 36 * 98; i += SQ_THREADS) {
        const int p = i / 98, c = i - p * 98;
        if (c < 18) {
          smem[c * 36 + p] = b1[p * SQ_LD + sq_pos(c)];
        } else if (c < 90) {
          rrow[p * a.rin_cs + c - 18] = b1[p * SQ_LD + sq_pos(c)];
        } else if (c < 94) {
          if (a.rin_cs == 80) rrow[p * 80 + 72 + c - 90] = sskel[p * 4 + (c - 90)];
        } else if (a.rin_cs == 80) {
          rrow[p * 80 + 76 + c - 94] = 0.f;
        }
      }
      __syncthreads();
      continue;
    }
    __syncthreads();
    sq_conv<9, 80>(a.conv[0], b0, b1, nullptr, wave, lane);
    __syncthreads();
    sq_conv<9, 80>(a.conv[1], b1, b2, b0, wave, lane);
    __syncthreads();
    sq_conv<9, 80>(a.conv[2], b2, b1, nullptr, wave, lane);
    __syncthreads();
    sq_conv<9, 80>(a.conv[3], b1, b0, b2, wave, lane);
    __syncthreads();
    sq_conv<1, 80>(a.out, b0, b1, nullptr, wave, lane);
    __syncthreads();
    if (tid < a.nraw) {      // global average pool, pixel order as avgpool_kernel
      const int cp = sq_pos(tid);
      float s = 0.f;
      for (int p = 0; p < 36; ++p) s += b1[p * SQ_LD + cp];
      a.raw_t[tr * a.nraw + tid] = s / 36.f;
    }
    __syncthreads();
  }
  for (int i = tid; i < 18 * 36; i += SQ_THREADS) a.mem[(size_t)row * 18 * 36 + i] = smem[i];
  if (tid < 16) a.prev_ext[(size_t)row * 16 + tid] = a.cam0_t[((size_t)(a.T - 1) * a.R + row) * 16 + tid];
}

// Relative transforms of every step at once (temporal.py:65-69): the previous extrinsics of row r at step t are the
// current ones of step t - 1 (every row is in every step's batch and temporal.py:77 stores them whether or not the
// memory is used), and the handle's stored ones at step 0.  A singular previous matrix is reported and the step runs
// without memory, as in temporal_xforms_kernel.
__global__ void seq_xforms_kernel(const float* prev_ext, const float* cam0_t, const uint8_t* use_t, int T, int R,
                                  float* rel_t, uint8_t* use_eff, int32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * R) return;
  const int t = i / R, r = i - t * R;
  float* o = rel_t + (size_t)i * 12;
  for (int k = 0; k < 12; ++k) o[k] = 0.f;
  uint8_t u = 0;
  if (use_t[i]) {
    double pe[16], pinv[16], ce[16], rel[16];
    load_mat4(t == 0 ? prev_ext + (size_t)r * 16 : cam0_t + ((size_t)(t - 1) * R + r) * 16, pe);
    load_mat4(cam0_t + (size_t)i * 16, ce);
    if (mat4_inv(pe, pinv)) {
      mat4_mul(ce, pinv, rel);
      for (int k = 0; k < 12; ++k) o[k] = (float)rel[k];
      u = 1;
    } else {
      atomicOr(status, UMT_STATUS_SINGULAR_PREV_EXTRINSICS);
    }
  }
  use_eff[i] = u;
}

static SeqLayer seq_layer(const ConvLayer& L) {
  SeqLayer s;
  s.w16 = L.w16; s.bias = L.bias; s.scale = L.scale; s.shift = L.shift;
  s.cin_p = round_up(L.cin, 16); s.cout_p = round_up(L.cout, 16);
  s.taps = L.taps; s.relu = L.relu;
  return s;
}

int64_t head_sequence_workspace_bytes(int T, int R) {
  const int64_t n = (int64_t)T * R;
  return round_up64(n * 12 * 4, 256) + round_up64(n, 256) + round_up64(n * 64 * 4, 256) + round_up64((int64_t)R * 36 * 4 * 4, 256);
}

int launch_head_sequence(const ConvLayer* temp, const ConvLayer* convs, const ConvLayer& out, const SkelParams* skel,
                         const float* skel_axes, const float* skel_rest, int n_skel, const float* fused_t,
                         const float* cam0_t, const uint8_t* use_t, int T, int R, int nraw, float* mem, float* prev_ext,
                         int32_t* status, float** raw_t, float* rin, int rin_cs, void* workspace, hipStream_t s) {
  // the kernel is instantiated for the network's shapes: temporal 1x1 with 96 padded input channels, regressor 3x3 and
  // the output 1x1 with 80 (76 or 72 channels, both heads)
  for (int i = 0; i < 3; ++i)
    UMT_REQUIRE(temp[i].w16 && temp[i].taps == 1 && round_up(temp[i].cin, 16) == 96 && temp[i].cout <= 96, UMT_ERR_STATE,
                "head_sequence: temporal layer %d has an unexpected shape", i);
  for (int i = 0; i < 4; ++i)
    UMT_REQUIRE(convs[i].w16 && convs[i].taps == 9 && round_up(convs[i].cin, 16) == 80 && convs[i].cout <= 96, UMT_ERR_STATE,
                "head_sequence: regressor layer %d has an unexpected shape", i);
  UMT_REQUIRE(out.w16 && out.taps == 1 && round_up(out.cin, 16) == 80 && out.cout == nraw && nraw <= 64, UMT_ERR_STATE,
              "head_sequence: output layer has an unexpected shape");
  char* ws = (char*)workspace;
  const int64_t n = (int64_t)T * R;
  float* rel_t = (float*)ws;                 ws += round_up64(n * 12 * 4, 256);
  uint8_t* use_eff = (uint8_t*)ws;           ws += round_up64(n, 256);
  float* raw = (float*)ws;                   ws += round_up64(n * 64 * 4, 256);
  float* skel_rows = (float*)ws;
  hipLaunchKernelGGL(seq_xforms_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, prev_ext, cam0_t, use_t, T, R, rel_t,
                     use_eff, status);
  if (skel) UMT_TRY(launch_skeleton(*skel, skel_axes, skel_rest, n_skel, n_skel == 1 ? 1 : R, skel_rows, 4, 0, s));
  SeqArgs a;
  a.fused_t = fused_t; a.cam0_t = cam0_t; a.rel_t = rel_t; a.use_t = use_eff; a.skel = skel ? skel_rows : nullptr;
  a.mem = mem; a.prev_ext = prev_ext; a.raw_t = raw; a.rin = rin; a.rin_cs = rin_cs;
  for (int i = 0; i < 3; ++i) a.temp[i] = seq_layer(temp[i]);
  for (int i = 0; i < 4; ++i) a.conv[i] = seq_layer(convs[i]);
  a.out = seq_layer(out);
  a.T = T; a.R = R; a.n_skel = n_skel; a.nraw = nraw;
  if (rin) hipLaunchKernelGGL(head_seq_kernel<false>, dim3(R), dim3(SQ_THREADS), 0, s, a);
  else hipLaunchKernelGGL(head_seq_kernel<true>, dim3(R), dim3(SQ_THREADS), 0, s, a);
  UMT_CHECK_HIP(hipGetLastError());
  *raw_t = raw;
  return 0;
}

}  // namespace umt
