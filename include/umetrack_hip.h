/*
 * umetrack_hip.h - C ABI of libumetrack_hip.so (gfx950 / MI355X only).
 *
 * The reference (facebookresearch/UmeTrack) is pure Python and has no FFI of its
 * own.  Each entry point below therefore cites the Python
 * call site it replaces; the Python host in umetrack_amd/lib/ keeps the
 * reference's names and signatures and forwards to these functions via ctypes.
 *
 * Conventions
 *  - every pointer marked "dev" is device memory on the device the model was
 *    finalised on; the caller owns every input, output and workspace buffer;
 *  - the library allocates device memory only in umt_model_finalize() (weights)
 *    and umt_model_reserve_memory() (temporal state); stage calls never allocate;
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *    synchronises the host;
 *  - return value 0 = ok, non-zero = error; umt_last_error() returns the
 *    thread-local message of the last failing call;
 *  - layouts at this boundary are the reference's (NCHW feature maps, row-major
 *    4x4 transforms, int64 index tensors); the channels-last layout used between
 *    kernels is private to the library.
 */
#ifndef UMETRACK_HIP_H
#define UMETRACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct umt_model umt_model;

enum {
  UMT_OK = 0,
  UMT_ERR_INVALID = 1,     /* bad argument / shape */
  UMT_ERR_STATE = 2,       /* call order (e.g. forward before finalize) */
  UMT_ERR_KEY = 3,         /* strict state_dict mismatch */
  UMT_ERR_HIP = 4,         /* a HIP runtime call failed */
  UMT_ERR_WORKSPACE = 5,   /* workspace too small */
  UMT_ERR_UNSUPPORTED = 6
};

/* UMT_PREC_FP32: fp32 matrix instructions, an exact fmaf chain per output (the reference-parity mode).
 * UMT_PREC_BF16: bf16 backbone activations and weights, fp32 accumulation (reported separately).
 * UMT_PREC_FP32_BF16X3: every tensor stays fp32; inside the stride-1 3x3 backbone layers each fp32 operand is
 *   the exact sum of three bf16 pieces and a product is six bf16 matrix instructions accumulated in fp32
 *   (the dropped cross terms are <= 2^-23 of the product).  fp32-level accuracy, not bit-identical to FP32.
 *   Launches of fewer than 16 384 pixels (a per-frame call's four crops) run the FP32 kernels, whose few-pixel
 *   forms are faster there.
 * In both non-FP32 modes umt_model_forward also runs the known-skeleton regressor's 3x3 layers that way when bs >= 228
 * (8192 head pixels: the batches the wide tiles are for); the stage entry points (umt_regress_decode, umt_head_sequence)
 * take no precision and always use the FP32 kernels. */
enum { UMT_PREC_FP32 = 0, UMT_PREC_BF16 = 1, UMT_PREC_FP32_BF16X3 = 2 };
enum { UMT_HEAD_KNOWN_SKELETON = 0, UMT_HEAD_PREDICT_SCALE = 1 };
enum { UMT_WARP_CV2_FIXED_POINT = 0, UMT_WARP_FLOAT = 1, UMT_WARP_CV2_FIXED_POINT_F64 = 2 };
/* OR-ed into `mode`: _warp_image(depth_check=False), tracker.py:59,71-73 - source points behind the camera keep
 * their projected coordinates instead of mapping to (-1,-1) */
enum { UMT_WARP_NO_DEPTH_CHECK = 0x100 };

const char* umt_last_error(void);
/* bumped on every signature change; bindings compare it with the value they were written for */
#define UMT_ABI_VERSION 4
int umt_abi_version(void);

/* Conditions a kernel can detect but an asynchronous call cannot return: the kernels OR these bits into a status
 * word in the model handle and umt_model_status() reads it (one 4-byte copy + stream synchronisation).  Each bit
 * names what the reference does at that point; the library's behaviour on the device is chosen so that nothing
 * faults and no state is poisoned. */
enum {
  UMT_STATUS_SINGULAR_PREV_EXTRINSICS = 1, /* use_memory on a row whose previous extrinsics are singular (e.g. never
                                              written): torch.inverse raises (temporal.py:66); here the sample runs as
                                              use_memory = False */
  UMT_STATUS_BAD_MEMORY_IDX = 2,           /* memory_idx outside the reserved rows: IndexError in the reference; here the
                                              sample runs without memory and writes none */
  UMT_STATUS_BAD_SAMPLE_RANGE = 4,         /* sample_range row outside [0, n] or not 1 or 2 views: the sample's outputs
                                              are NaN */
  UMT_STATUS_NONFINITE_OUTPUT = 8,         /* a regressor output row was not finite: its decoded outputs are NaN */
  UMT_STATUS_BAD_BONE_INDEX = 16           /* umt_fk_landmarks: a landmark's bone index outside [0, 16] (IndexError in
                                              the reference); that landmark is NaN */
};

/* ---- model handle: replaces load_pretrained_model, lib/models/model_loader.py:53-88 ---- */
int umt_model_create(umt_model** out);
int umt_model_destroy(umt_model* m);
/* the strict key table (== reference state_dict, 252 entries) */
int umt_model_param_count(const umt_model* m, int* count);
int umt_model_param_info(const umt_model* m, int index, const char** key, int* ndim,
                         int64_t shape[4], int* is_int64);
/* host data, float32 (or int64 for num_batches_tracked); copied */
int umt_model_set_param(umt_model* m, const char* key, const void* host_data, int ndim,
                        const int64_t* shape, int is_int64);
/* strict check (every key set exactly once), weight packing, upload to `device` */
int umt_model_finalize(umt_model* m, int device);
/* temporal state: SimpleConvRNN._mem_features/_prev_extrinsics, lib/models/temporal.py:41-42,102-125 */
int umt_model_reserve_memory(umt_model* m, int rows, void* stream);   /* grows, keeps content */
int umt_model_memory_rows(const umt_model* m, int* rows);
int umt_model_clear_memory(umt_model* m);                              /* back to 0 rows */
/* copy state out for inspection: mem (rows,18,6,6), prev_ext (rows,4,4), both dev */
int umt_model_read_memory(const umt_model* m, float* mem_dev, float* prev_ext_dev, void* stream);
/* copies the UMT_STATUS_* word to *status_host (host memory), optionally clears it, and waits for `stream` */
int umt_model_status(umt_model* m, int32_t* status_host, int clear, void* stream);
/* the same word copied to DEVICE memory (status_dev, int32[1]) on `stream`, optionally cleared, without waiting: a caller
 * that brings its outputs to the host in one copy anyway (HandTracker._finish, replacing the per-tensor copies of
 * lib/tracker/tracker.py:156-161) puts the word next to them and pays no extra synchronisation */
int umt_model_status_async(umt_model* m, int32_t* status_dev, int clear, void* stream);

/* ---- workspace sizes (bytes) ---- */
int64_t umt_backbone_workspace_bytes(int n_crops);
int64_t umt_head_workspace_bytes(int n_crops, int n_samples);
int64_t umt_forward_workspace_bytes(int n_crops, int n_samples);

/* ---- A0 (F1): crop cameras for a batch of hands, gen_crop_cameras_from_pose,
 * lib/tracker/perspective_crop.py:54-162 + lib/common/crop.py:15-82 + lib/common/affine.py:34-66,
 * and the per-crop packing of HandTracker._make_inputs, lib/tracker/tracker.py:212-216 ----
 * points        dev float32 (n_hands, n_points, 3): crop points in world mm, n_points in 21..63; the
 *               first 21 are the tracked pose's landmarks (visibility test), the rest the neutral and
 *               open pose (umt_fk_landmarks output for the three poses of perspective_crop.py:89-115)
 * cameras       dev float64 (n_rigs, n_rig_views, 30): fisheye cameras in umt_warp_crops' src_cams format
 * camera_angles dev float64 (n_rigs, n_rig_views): roll in degrees
 * rig_idx       dev int32 (n_hands) or NULL (one rig)
 * hand_idx      dev int64 (n_hands): 1 = right hand, mirrored when mirror_right != 0
 * forced_views  dev int32 (n_hands, max_views) or NULL: use these rig views (-1 = none) instead of the
 *               visibility ranking
 * Ranking: cameras that see >= min_visible of the 21 landmarks inside their image with z > 0; in index
 * order when sort_by_index != 0 (the tracker's setting), else most landmarks first; the first
 * max_views (<= 4) are used.
 * Outputs, all dev: n_views int32 (n_hands); view_idx int32 (n_hands, max_views), -1 = unused slot;
 * dst_cams float64 (n_hands, max_views, 20) in umt_warp_crops' dst_cams format; intrinsics float32
 * (n_hands, max_views, 3, 3); extrinsics float32 (n_hands, max_views, 4, 4) = world->crop-eye with
 * the translation in metres; status int32 (n_hands, max_views): 1 where the reference raises
 * ValueError("Unable to create crop camera") (a point behind the crop camera or focal < 5).       */
int umt_crop_cameras(const float* points, int n_points, const double* cameras, const double* camera_angles,
                     int n_rig_views, const int32_t* rig_idx, const int64_t* hand_idx,
                     const int32_t* forced_views, int n_hands, int max_views, int min_visible,
                     int sort_by_index, int mirror_right, double focal_multiplier, int crop_w, int crop_h,
                     int32_t* n_views, int32_t* view_idx, double* dst_cams, float* intrinsics,
                     float* extrinsics, int32_t* status, void* stream);

/* ---- A1: fisheye -> pinhole crop, _warp_image + /255, lib/tracker/tracker.py:54-80,210-211 ----
 * src_images dev uint8; image i of crop k starts at src_images + src_offset[k] with row pitch
 * src_pitch[k] (bytes) - this is how a 4-view-wide frame is addressed (video_pose_data.py:123-125).
 * src_cams  dev float64 (n,30): w,h,fx,fy,cx,cy,k1,k2,k3,k4,p1,p2,k5,k6, cam_to_world 4x4 row-major
 * dst_cams  dev float64 (n,20): fx,fy,cx,cy, cam_to_world 4x4 row-major (96x96 pinhole crop cameras)
 * out       dev float32 (n,out_h,out_w), values in [0,1]; out_w <= 3072, any out_h (a workgroup takes a band of rows)
 * The kernel copies the source rectangle of a band with aligned 4-byte loads: it may read up to 3 bytes before an image
 * row's first byte and up to 3 bytes past its last byte (never outside the 4-byte words those bytes lie in), so
 * src_images must be an allocation whose size is a multiple of 4 bytes (every hipMalloc'd buffer is).           */
int umt_warp_crops(const uint8_t* src_images, const int64_t* src_offset, const int32_t* src_pitch,
                   const double* src_cams, const double* dst_cams, int n, int out_h, int out_w,
                   int mode, float* out, void* stream);

/* ---- A1': pinhole -> pinhole float resample, _resample_images_batched,
 * lib/batched_dataset/data_transform.py:78-144.  images dev f32 (n,h,w), xfs dev f32 (n,4,4),
 * out dev f32 (n,out_h,out_w) (not divided by 255, as in the reference)                           */
int umt_resample_pinhole(const float* images, const float* xfs, int n, int h, int w, int out_h,
                         int out_w, float* out, void* stream);

/* ---- the first layer group of A3 alone: Conv2d(1,32,3,pad 1) -> BatchNorm2d -> ReLU -> MaxPool2d(2,2),
 * lib/models/model_utils.py:119-124.  images dev f32 (n,96,96) -> out dev (n,48,48,32) channels-last: float for
 * UMT_PREC_FP32 and UMT_PREC_FP32_BF16X3, bf16 (uint16_t) for UMT_PREC_BF16 - the tensor the rest of the backbone
 * reads in that mode.  For the tests that hold the bf16 mode's split-operand stem against the fp32 one.       */
int umt_stem_forward(umt_model* m, const float* images, int n, int precision, void* out, void* stream);

/* ---- A3: _image_backbone, lib/models/umetrack_model.py:127-129, model_utils.py:107-138 ----
 * images dev f32 (n,96,96) -> features dev f32 (n,72,6,6)                                          */
int umt_backbone_forward(umt_model* m, const float* images, int n, int precision, float* features,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* ---- A4-A7: _forward_feature_extractor after the backbone, umetrack_model.py:130-168,
 * feature_extractor.py:61-141.  sample_range dev int64 (bs,2).  out dev f32 (bs,72,6,6)            */
int umt_fuse_views(umt_model* m, const float* features, const float* intrinsics,
                   const float* extrinsics, const int64_t* sample_range, int n, int bs, float* fused,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* ---- A8: SimpleConvRNN.forward_temporal_features, lib/models/temporal.py:93-139 ----
 * cam0_extrinsics dev f32 (bs,4,4) = extrinsics[sample_range[:,0]] (umetrack_model.py:93-97);
 * memory_idx dev int64 (bs), use_memory dev uint8/bool (bs).  The caller must have reserved
 * max(memory_idx)+1 rows.  out dev f32 (bs,72,6,6); updates the handle's state.
 * All previous state is read before any is written (as temporal.py:66-77).  Duplicate memory_idx values in one
 * batch all read the same previous row; the sample with the highest batch index writes it (last wins, as torch's
 * CPU index assignment).  See UMT_STATUS_SINGULAR_PREV_EXTRINSICS / UMT_STATUS_BAD_MEMORY_IDX.                 */
int umt_temporal_step(umt_model* m, const float* fused, const float* cam0_extrinsics,
                      const int64_t* memory_idx, const uint8_t* use_memory, int bs, float* out,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- A9-A12: skeleton encoder + PoseRegressor.regress_poses + _recover_wrist_xfs_in_world,
 * skeleton_encoder.py:43-53, regressor.py:163-186, umetrack_model.py:77-90,198-219,234-240 ----
 * skel_axes/skel_rest dev f32 (n_skel,22,3), n_skel in {1, bs} (ignored for the scale head);
 * outputs dev f32: joint_angles (bs,22), wrist_xfs (bs,4,4) in world, sigmas (bs,21),
 * skel_scales (bs) (scale head only, else may be NULL), raw (bs,62|63) optional (may be NULL)      */
int umt_regress_decode(umt_model* m, int head, const float* temporal, const float* skel_axes,
                       const float* skel_rest, int n_skel, const float* cam0_extrinsics,
                       const int64_t* hand_idx, int bs, float* joint_angles, float* wrist_xfs,
                       float* sigmas, float* skel_scales, float* raw, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ---- whole model in one call: UmeTrackModel.regress_pose_use_skeleton / _pred_skel_scale,
 * lib/models/umetrack_model.py:188-242.  Same buffers as the stage calls; intermediates stay in
 * the library's channels-last layout inside `workspace`.                                            */
int umt_model_forward(umt_model* m, int head, int precision, const float* images,
                      const float* intrinsics, const float* extrinsics, const int64_t* sample_range,
                      const int64_t* memory_idx, const uint8_t* use_memory, const int64_t* hand_idx,
                      int n, int bs, const float* skel_axes, const float* skel_rest, int n_skel,
                      float* joint_angles, float* wrist_xfs, float* sigmas, float* skel_scales,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- whole recordings in lock-step: the T sequential calls of SimpleConvRNN.forward_temporal_features +
 * PoseRegressor.regress_poses that run_eval_known_skeleton.py:68-86 makes frame by frame (temporal.py:93-139,
 * regressor.py:163-186, umetrack_model.py:77-90), for `rows` memory rows at once in ONE launch.  Sample r of step t
 * uses memory row r (memory_idx = arange(rows) in every step); the caller has reserved >= rows rows.  The handle's
 * state is read at the start and written back at the end, so the call composes with umt_temporal_step.
 * Time-major inputs, all dev: fused_t f32 (t, rows, 72, 6, 6) = umt_fuse_views outputs; cam0_t f32 (t, rows, 4, 4);
 * use_memory_t uint8 (t, rows); hand_idx int64 (t, rows); skeleton as umt_regress_decode.
 * Outputs time-major: joint_angles_t (t, rows, 22), wrist_xfs_t (t, rows, 4, 4), sigmas_t (t, rows, 21),
 * skel_scales_t (t, rows) (scale head).  Results are bit-identical to t calls of umt_temporal_step +
 * umt_regress_decode.                                                                                     */
int64_t umt_head_sequence_workspace_bytes(int t, int rows);
int umt_head_sequence(umt_model* m, int head, const float* fused_t, const float* cam0_t, const uint8_t* use_memory_t,
                      const int64_t* hand_idx, const float* skel_axes, const float* skel_rest, int n_skel, int t,
                      int rows, float* joint_angles_t, float* wrist_xfs_t, float* sigmas_t, float* skel_scales_t,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- measurement aid, no counterpart in the reference: when enabled, umt_model_forward records
 * HIP events on the caller's stream before the backbone, after it and at the end of the call;
 * umt_model_timing_read waits for them and returns the summed durations since the last enable.
 * bench.py uses it for the live roofline of the backbone convolutions.                              */
int umt_model_timing_enable(umt_model* m, int on);
int umt_model_timing_read(umt_model* m, double* backbone_ms, double* forward_ms, int* calls);

/* ---- measurement aid, no counterpart in the reference: per-launch timing of the backbone.  When enabled,
 * every backbone pass (umt_model_forward, umt_backbone_forward) records one HIP event on the caller's stream
 * before the stem and one after each launch.  umt_model_layer_timing_read waits for the last event and sums,
 * per position k in the (fixed) launch schedule, over all passes since the last enable: the elapsed time
 * ms[k], the algorithmic FLOP flop[k] (true channel counts) and the number of launches launches[k];
 * names is a caller buffer of max_layers x 48 bytes that receives "<kind> <cin>-><cout> hw<H> s<stride>".
 * The events sit between the launches, so a pass measured this way is a little slower than an unmarked one;
 * bench.py runs it outside its timed region.                                                              */
int umt_model_layer_timing_enable(umt_model* m, int on);
int umt_model_layer_timing_read(umt_model* m, int max_layers, char* names, double* flop, double* ms,
                                int* launches, int* n_layers);

/* ---- A14: skin_landmarks, lib/common/hand_skinning.py:189-209 (+ the right-hand mirror of
 * landmarks_from_hand_pose, lib/tracker/perspective_crop.py:40-51 when mirror_right != 0) ----
 * joint_angles dev f32 (b,22); wrist dev f32 (b,4,4); hand model arrays dev with n_model in {1,b}:
 * axes,rest f32 (n_model,22,3); lm_rest f32 (n_model,21,3); lm_w f32 (n_model,21,3);
 * lm_idx int32 (n_model,21,3); hand_idx dev int64 (b) or NULL; wrist_t_scale multiplies the wrist
 * translation (1000 for the m->mm step of tracker.py:258); out dev f32 (b,21,3);
 * status dev int32[1] or NULL: UMT_STATUS_BAD_BONE_INDEX is OR-ed in when a used bone index (weight != 0) is
 * outside [0, 16] - the reference raises IndexError there; the landmark is written as NaN                      */
int umt_fk_landmarks(const float* joint_angles, const float* wrist, const float* axes,
                     const float* rest, const float* lm_rest, const float* lm_w,
                     const int32_t* lm_idx, int n_model, const int64_t* hand_idx, int mirror_right,
                     float wrist_t_scale, int b, float* out, int32_t* status, void* stream);

/* ---- A15: _compute_metrics + the sums behind aggregate_metrics, load_eval.py:26-89,
 * metric_utils.py:18-84.  gt/pred dev f32 (n_rec,2,T,21,3), valid dev uint8 (n_rec,2,T).
 * partials dev float64[UMT_METRIC_WORDS]: sum_err, n_err, sum_acc, n_acc, sum_gt_acc, n_valid,
 * n_total, then 101 PCK counts (errors <= linspace(0,50,101)[i]).  Written, not accumulated;
 * deterministic (fixed-order two-stage reduction in `workspace`).                                   */
#define UMT_METRIC_WORDS 108
int64_t umt_metric_workspace_bytes(int n_rec, int t);
int umt_metric_partials(const float* gt, const float* pred, const uint8_t* valid, int n_rec, int t,
                        double* partials, void* workspace, int64_t workspace_bytes, void* stream);


/* ---- umt_metric_partials for callers that cannot allocate.  No counterpart in the reference: it is for
 * callers that may not call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation) and have no workspace to pass.  Same inputs, outputs and bit-identical results as
 * umt_metric_partials; the per-workgroup partial sums go to a static device array inside the library
 * (1.77 MB per GPU, held until the process ends) instead of a workspace.  Limit: n_rec * 2 * t <=
 * UMT_METRIC_STATIC_MAX_HAND_FRAMES, larger inputs return UMT_ERR_WORKSPACE.  The call allocates nothing
 * (no hipMalloc / hipFree, no allocating library routine).  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (different streams or threads) race; serialise them.                    */
#define UMT_METRIC_STATIC_MAX_HAND_FRAMES 524288
int umt_metric_partials_static(const float* gt, const float* pred, const uint8_t* valid, int n_rec, int t,
                               double* partials, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UMETRACK_HIP_H */
