// Internal declarations shared by the HIP translation units of libumetrack_hip.so.
// gfx950 only.  Activations between kernels are channels-last ("pixel rows"):
// a tensor (n, C, H, W) of the reference is stored as [n*H*W][Cs] floats with
// Cs = C rounded up to a multiple of 8; channels C..Cs-1 are kept at zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/umetrack_hip.h"

namespace umt {

int set_error(int code, const char* fmt, ...);

#define UMT_CHECK_HIP(expr)                                                               \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return umt::set_error(UMT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                            hipGetErrorString(e__), __FILE__, __LINE__);                  \
  } while (0)

#define UMT_REQUIRE(cond, code, ...)                                                      \
  do {                                                                                    \
    if (!(cond)) return umt::set_error(code, __VA_ARGS__);                                \
  } while (0)

#define UMT_TRY(expr)                                                                     \
  do {                                                                                    \
    int rc__ = (expr);                                                                    \
    if (rc__ != 0) return rc__;                                                           \
  } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// device.  conv_prepare_device() does that for every such kernel, for the current device; umt_model_finalize calls
// it (and tools/conv_bench.hip, which launches without a model).  The call is idempotent and the library keeps no
// record of it: there is no process-wide mutable state behind the launch path.
static inline int raise_dynamic_lds(const void* kernel, size_t bytes) {
  UMT_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}
int conv_prepare_device();      // conv_igemm.hip + conv_thin.hip + conv_x3.hip: for the current device

// Kernel-selection switches (diagnostics: A/B profiling and the bit-identity tests).  umt_model_create reads them
// from the environment ONCE PER MODEL HANDLE and umt_model_finalize copies them into every ConvLayer of that
// handle; the launchers read ConvLayer::flags only.
enum : unsigned {
  CONV_NO_THIN = 1u,        // UMT_CONV_NOTHIN: bf16 stages 1-3 on conv3x3s1_halo_bf16 instead of conv_thin.hip's kernels
  CONV_NO_FUSE = 2u,        // UMT_CONV_NOFUSE: stage-1 BasicBlocks as two launches instead of block32_fused_bf16
  CONV_NO_STREAM128 = 4u,   // UMT_CONV_NOSTREAM128
  CONV_NO_THIN_F32 = 8u,    // UMT_CONV_NOTHIN_F32: fp32 stage 1 on conv3x3s1_halo_f32 instead of conv3x3_wlds32_f32
  CONV_GATHER = 16u,        // UMT_CONV_GATHER: every fp32 convolution on the per-tap gather kernel
  CONV_NO_S2 = 32u,         // UMT_CONV_NOS2: no fused stride-2 + down-sample kernel
  CONV_X3_NO_S2 = 64u,      // UMT_X3_NOS2
  CONV_NO_CHAIN = 256u,     // UMT_CONV_NOCHAIN: the head's 1x1 chains as three conv_igemm_f32 launches instead of conv_chain.hip
  CONV_NO_PIPE64 = 2048u,   // UMT_CONV_NOPIPE64: bf16 64 -> 64 layers on the two-group conv3x3_wlds64_bf16 instead of conv3x3_pipe64_bf16
  CONV_NO_PIPE32 = 1024u,   // UMT_CONV_NOPIPE32: fp32 stage 1 on the two-group conv3x3_wlds32_f32 instead of conv3x3_pipe32_f32
  CONV_NO_S2P = 512u,       // UMT_CONV_NOS2P: fp32 fused stride-2 layers on the one-tile-per-workgroup kernel instead of the persistent one
  CONV_NO_DEPTH = 4096u,    // UMT_CONV_NODEPTH: bf16 stage 1 + the 32 -> 64 stride-2 layer as three launches instead of stage1_depth_bf16
  CONV_NO_SMALL = 8192u,    // UMT_CONV_NOSMALL: few-pixel fp32 launches on conv3x3s1_halo_f32<4,1,1,1> instead of conv_small.hip
  CONV_STEM_VALU = 16384u,  // UMT_STEM_VALU: the stem on the vector pipe (stem_kernel) instead of stem_mfma_kernel
  CONV_NO_WALK = 32768u,    // UMT_CONV_NOWALK: bf16 fused stride-2 layers one tile per workgroup instead of conv3x3s2_walk_bf16
  SEQ_ONE_KERNEL = 65536u,  // UMT_SEQ_ONEKERNEL: umt_head_sequence runs the regressor inside head_seq_kernel, step by step
  STEM_NO_SPLIT = 262144u,  // UMT_STEM_NOSPLIT: the bf16 mode's stem on the fp32 matrix instructions (the fp32 modes' arithmetic)
  HEAD_NO_X3 = 131072u,     // UMT_HEAD_NOX3: the regressor's 3x3 layers on the fp32 kernels in the bf16 / fp32x3 modes too
};
unsigned conv_flags_from_env();
constexpr int64_t HEAD_X3_MIN_PIXELS = 8192;      // below it launch_conv runs its few-pixel kernels
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

#ifdef __HIPCC__
// Tile owned by this workgroup.  Workgroups are dealt to the 8 XCDs round-robin by linear id, so
// neighbouring ids sit behind different L2s; this bijective remap gives each XCD label one
// contiguous run of tiles (cout tile fastest, then pixel tile), so the halo rows two neighbouring
// pixel tiles share and the activation rows two cout tiles share are fetched into one L2 only.
// Placement is a speed matter only; nothing depends on it.
__device__ __forceinline__ void xcd_tile(int& tile_m, int& tile_n) {
#ifdef UMT_X_NOSWIZZLE
  tile_m = blockIdx.x;
  tile_n = blockIdx.y;
#else
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned nwg = gx * gy, lin = blockIdx.y * gx + blockIdx.x;
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = lin & 7u;
  const unsigned w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
  tile_n = (int)(w % gy);
  tile_m = (int)(w / gy);
#endif
}
#endif

// One convolution (1x1 or 3x3, stride 1 or 2, pad = taps==9) with its per-channel epilogue
//   y = (acc + bias[c]) * scale[c] + shift[c]  (+ residual)  (relu)
// which is conv bias followed by eval-mode BatchNorm in the reference's operation order.
struct ConvLayer {
  int taps = 1;        // 1 or 9
  int cin = 0, cout = 0;
  int cin_s = 0;       // channel stride of the input rows  (multiple of 8)
  int cout_s = 0;      // channel stride of the output rows (multiple of 8)
  int cout_w = 0;      // cout padded to the 32-wide MFMA tile
  int stride = 1;
  int relu = 0;
  const float* w = nullptr;      // dev, packed [tap][cin_s/8][2][cout_w][4]
  const float* bias = nullptr;   // dev [cout_w]
  const float* scale = nullptr;  // dev [cout_w]
  const float* shift = nullptr;  // dev [cout_w]
  const uint16_t* wh = nullptr;  // dev, bf16 packed [tap][cin_s/16][2][cout_w][8] (backbone layers only)
  const uint16_t* wx3 = nullptr; // dev, three-way bf16 split [tap][cin_s/16][plane][2][cout_w][8] (backbone 3x3 layers)
  const float* w16 = nullptr;    // dev, packed for v_mfma_f32_16x16x4_f32 (head_seq.hip; temporal and regressor layers only)
  unsigned flags = 0;            // CONV_* kernel-selection switches of the owning model handle
};

// conv_igemm.hip
int launch_conv(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual,
                float* out, hipStream_t stream);

// conv_chain.hip: three consecutive 1x1 layers in one kernel (view fusion, temporal module)
bool conv_chain3_applicable(const ConvLayer& a, const ConvLayer& b, const ConvLayer& c);
int launch_conv_chain3(const ConvLayer& l0, const ConvLayer& l1, const ConvLayer& l2, const float* in, int64_t rows, float* out,
                       hipStream_t stream);

// conv1 (3x3 stride 2) of a stage's first block fused with its 1x1 stride-2 down-sample branch (ds may be null)
bool conv_s2_fused_applicable(const ConvLayer& c1, const ConvLayer* ds, int h, int w);
int launch_conv_s2_fused(const ConvLayer& c1, const ConvLayer* ds, const float* in, int n, int h, int w, float* out,
                         float* ds_out, hipStream_t stream);

// conv_bf16.hip: bf16 rows in, bf16 (or fp32 when out_f32) rows out, fp32 accumulation and epilogue
int launch_conv_bf16(const ConvLayer& L, const uint16_t* in, int n, int h, int w, const uint16_t* residual,
                     void* out, int out_f32, hipStream_t stream);

int launch_conv_s2_fused_bf16(const ConvLayer& c1, const ConvLayer* ds, const uint16_t* in, int n, int h, int w,
                              uint16_t* out, uint16_t* ds_out, hipStream_t stream);

// conv_x3.hip: fp32 rows in and out, operands split into three bf16 each, six bf16 MFMAs per fp32 product
bool conv_x3_applicable(const ConvLayer& L, int w);
int launch_conv_x3(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual, float* out,
                   hipStream_t stream);

bool conv_s2_x3_applicable(const ConvLayer& c1, const ConvLayer* ds, int h, int w);
int launch_conv_s2_fused_x3(const ConvLayer& c1, const ConvLayer* ds, const float* in, int n, int h, int w, float* out,
                            float* ds_out, hipStream_t stream);

// conv_thin.hip: persistent bf16 kernel for the 32->32 stride-1 3x3 layers (weights in registers)
bool conv_thin_applicable(const ConvLayer& L, int w);
int launch_conv_thin_bf16(const ConvLayer& L, const uint16_t* in, int n, int h, int w, const uint16_t* residual,
                          uint16_t* out, hipStream_t stream);

// conv_thin.hip: persistent fp32 kernel for the 32->32 stride-1 3x3 layers (weights in LDS), bit-identical to the halo kernel
bool conv_thin_f32_applicable(const ConvLayer& L, int w);
int launch_conv_thin_f32(const ConvLayer& L, const float* in, int n, int h, int w, const float* residual, float* out,
                         hipStream_t stream);

// conv_thin.hip: a whole 32-channel BasicBlock of stage 1 in one bf16 kernel (intermediate rows stay in LDS)
bool block32_fused_applicable(const ConvLayer& c1, const ConvLayer& c2, int h, int w);
int launch_block32_fused_bf16(const ConvLayer& c1, const ConvLayer& c2, const uint16_t* in, int n, uint16_t* out,
                              hipStream_t stream);

// stage1_depth.hip: both stage-1 BasicBlocks and the first stride-2 layer (+ its down-sample branch) of the bf16
// backbone in one launch, intermediate tensors in LDS rings
bool stage1_depth_applicable(const ConvLayer& c1a, const ConvLayer& c2a, const ConvLayer& c1b, const ConvLayer& c2b,
                             const ConvLayer& s2, const ConvLayer* ds, int h, int w);
int launch_stage1_depth_bf16(const ConvLayer& c1a, const ConvLayer& c2a, const ConvLayer& c1b, const ConvLayer& c2b,
                             const ConvLayer& s2, const ConvLayer& ds, const uint16_t* in, int n, uint16_t* out_t,
                             uint16_t* out_d, hipStream_t stream);
int stage1_depth_prepare_device();

// stem.hip: conv3x3(1->32)+bias+BN+ReLU+maxpool2 ; images (n,96,96) -> rows [n*48*48][32]
struct StemParams {
  const float* w;      // dev [9][32]
  const float* bias;   // dev [32]
  const float* scale;  // dev [32]
  const float* shift;  // dev [32]
  int valu;            // CONV_STEM_VALU: the vector-pipe kernel instead of the matrix-pipe one (bit-identical)
};
int launch_stem(const StemParams& p, const float* images, int n, float* out, hipStream_t stream);
int launch_stem_bf16(const StemParams& p, const float* images, int n, uint16_t* out, hipStream_t stream);

// head.hip
int launch_rows_to_nchw(const float* rows, int cs, int c0, int c, int n, int hw, float* nchw, hipStream_t s);
int launch_nchw_to_rows(const float* nchw, int c, int n, int hw, float* rows, int cs, int c0, hipStream_t s);
int launch_zero_channels(float* rows, int cs, int c0, int c, int64_t nrows, hipStream_t s);
// per-sample transforms for fusion: to_canon (per crop, 12 floats [R|t] rows) and out_xf (per sample)
int launch_fusion_xforms(const float* intrinsics, const float* extrinsics, const int64_t* sample_range,
                         int n, int bs, float* crop_xf, float* sample_xf, int32_t* sample_views,
                         int32_t* status, hipStream_t s);
int launch_ftl_concat(const float* feat_rows, const float* crop_xf, const int64_t* sample_range,
                      const int32_t* sample_views, int n, int bs, float* cat_rows, hipStream_t s);
int launch_ftl_select(const float* conv_rows, const float* feat_rows, const float* sample_xf,
                      const int64_t* sample_range, const int32_t* sample_views, int n, int bs,
                      float* out_rows, int out_cs, int out_c0, hipStream_t s, bool zero_rest = false);
// rel_xf: scratch [bs][12] floats + bs bytes, row_of: scratch [bs] int32 (validated memory row per sample);
// owner: [mem_rows] claim words of the model handle, epoch: this step's claim epoch; status: UMT_STATUS_* word
int launch_temporal_gather(float* mem, float* prev_ext, unsigned long long* owner, unsigned epoch, int mem_rows,
                           const float* cam0_ext, const int64_t* memory_idx, const uint8_t* use_memory, int bs,
                           float* tin_rows, int tin_cs, float* rel_xf, int32_t* row_of, int32_t* status, hipStream_t s);
int launch_temporal_split(const float* tout_rows, int tout_cs, float* mem, const int32_t* row_of, int bs,
                          float* out_rows, int out_cs, hipStream_t s, bool zero_pad = false);
struct SkelParams {
  const float* w;      // dev [144][132]
  const float* bias;   // dev [144]
  const float* scale;  // dev [4]
  const float* shift;  // dev [4]
};
int launch_skeleton(const SkelParams& p, const float* axes, const float* rest, int n_skel, int bs,
                    float* rows, int cs, int c0, hipStream_t s);
int launch_avgpool(const float* rows, int cs, int c, int bs, float* raw, hipStream_t s);
int launch_decode(const float* raw, int nraw, int predict_scale, const float* cam0_ext,
                  const int64_t* hand_idx, int bs, float* joint_angles, float* wrist_xfs,
                  float* sigmas, float* skel_scales, int32_t* status, hipStream_t s);
int launch_gather_cam0(const float* extrinsics, const int64_t* sample_range, int n, int bs,
                       float* cam0, hipStream_t s);

// head_seq.hip: temporal module + regressor layers + average pool for R memory rows x T steps in one launch.
// temp: 3 temporal layers, convs: conv1, conv2 of the two regressor blocks, out: the 1x1 output layer; skel null for the
// scale head.  *raw_t receives the (T, R, nraw) buffer inside `workspace`: filled by the kernel when rin is null (one-kernel
// mode); with rin the kernel stops after the temporal module and writes the regressor's input rows (T * R * 36, rin_cs)
// for the caller to run the batched regressor into *raw_t.
int64_t head_sequence_workspace_bytes(int T, int R);
int launch_head_sequence(const ConvLayer* temp, const ConvLayer* convs, const ConvLayer& out, const SkelParams* skel,
                         const float* skel_axes, const float* skel_rest, int n_skel, const float* fused_t,
                         const float* cam0_t, const uint8_t* use_t, int T, int R, int nraw, float* mem, float* prev_ext,
                         int32_t* status, float** raw_t, float* rin, int rin_cs, void* workspace, hipStream_t s);

}  // namespace umt
