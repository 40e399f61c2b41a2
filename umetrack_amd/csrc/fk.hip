// Hand forward kinematics + linear blend skinning to 21 landmarks, one wavefront per hand.
// Replaces skin_landmarks (lib/common/hand_skinning.py:17-209) and the right-hand mirror of
// landmarks_from_hand_pose (lib/tracker/perspective_crop.py:40-51).
//
// Lanes 0..19 build the 20 joint-local transforms (Rodrigues, pytorch3d so3_exp_map formula
// with the squared angle clamped at 1e-4), lanes 0..4 walk one finger chain each
// (W.L0.L1, .L2, .L3), lanes 0..20 blend one landmark each.  The 20 local and 17 skinning
// transforms (3x4 each) live in 1.8 KB of LDS per wave; nothing but the inputs and the 252-B
// result touches HBM, so the kernel is bound by its 932 B (404 B with a shared model) per hand.
#include "umt_internal.h"

namespace umt {

constexpr int FK_WAVES = 4;

__global__ __launch_bounds__(64 * FK_WAVES) void fk_kernel(
    const float* __restrict__ joint_angles, const float* __restrict__ wrist, const float* __restrict__ axes,
    const float* __restrict__ rest, const float* __restrict__ lm_rest, const float* __restrict__ lm_w,
    const int32_t* __restrict__ lm_idx, int n_model, const int64_t* __restrict__ hand_idx, int mirror_right,
    float t_scale, int b_total, float* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ float sL[FK_WAVES][20][12];
  __shared__ float sF[FK_WAVES][17][12];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x * FK_WAVES + wv;
  const bool live = b < b_total;
  const int mb = (n_model == 1 || !live) ? 0 : b;
  float (*L)[12] = sL[wv];
  float (*F)[12] = sF[wv];

  if (live && lane < 20) {   // hand_skinning.py:35-53
    const float th = joint_angles[(size_t)b * 22 + lane];
    const float* ax = axes + ((size_t)mb * 22 + lane) * 3;
    const float* rp = rest + ((size_t)mb * 22 + lane) * 3;
    const float x = ax[0] * th, y = ax[1] * th, z = ax[2] * th;
    const float nrm2 = x * x + y * y + z * z;
    const float ang = sqrtf(fmaxf(nrm2, 1e-4f));
    const float inv = 1.0f / ang;
    const float f1 = inv * sinf(ang);
    const float f2 = inv * inv * (1.0f - cosf(ang));
    // K = hat(v), K^2 = v v^T - |v|^2 I
    float R[9];
    R[0] = f2 * (-(y * y + z * z)) + 1.0f;
    R[1] = f1 * (-z) + f2 * (x * y);
    R[2] = f1 * (y) + f2 * (x * z);
    R[3] = f1 * (z) + f2 * (x * y);
    R[4] = f2 * (-(x * x + z * z)) + 1.0f;
    R[5] = f1 * (-x) + f2 * (y * z);
    R[6] = f1 * (-y) + f2 * (x * z);
    R[7] = f1 * (x) + f2 * (y * z);
    R[8] = f2 * (-(x * x + y * y)) + 1.0f;
    const float px = rp[0], py = rp[1], pz = rp[2];
    for (int r = 0; r < 3; ++r) {
      L[lane][r * 4 + 0] = R[r * 3 + 0];
      L[lane][r * 4 + 1] = R[r * 3 + 1];
      L[lane][r * 4 + 2] = R[r * 3 + 2];
    }
    L[lane][3] = px - (R[0] * px + R[1] * py + R[2] * pz);
    L[lane][7] = py - (R[3] * px + R[4] * py + R[5] * pz);
    L[lane][11] = pz - (R[6] * px + R[7] * py + R[8] * pz);
  }
  __syncthreads();

  if (live && lane < 6) {
    float T[12];
    const float* w = wrist + (size_t)b * 16;
    const bool flip = mirror_right && hand_idx && hand_idx[b] == 1;
    for (int r = 0; r < 3; ++r) {
      T[r * 4 + 0] = flip ? -w[r * 4 + 0] : w[r * 4 + 0];
      T[r * 4 + 1] = w[r * 4 + 1];
      T[r * 4 + 2] = w[r * 4 + 2];
      T[r * 4 + 3] = w[r * 4 + 3] * t_scale;
    }
    if (lane == 5) {   // root and wrist frames, hand_skinning.py:115
      for (int i = 0; i < 12; ++i) { F[0][i] = T[i]; F[1][i] = T[i]; }
    } else {           // hand_skinning.py:17-32, 122-125
      for (int j = 0; j < 4; ++j) {
        const float* M = L[4 * lane + j];
        float N[12];
        for (int r = 0; r < 3; ++r) {
          const float a0 = T[r * 4], a1 = T[r * 4 + 1], a2 = T[r * 4 + 2];
          N[r * 4 + 0] = a0 * M[0] + a1 * M[4] + a2 * M[8];
          N[r * 4 + 1] = a0 * M[1] + a1 * M[5] + a2 * M[9];
          N[r * 4 + 2] = a0 * M[2] + a1 * M[6] + a2 * M[10];
          N[r * 4 + 3] = a0 * M[3] + a1 * M[7] + a2 * M[11] + T[r * 4 + 3];
        }
        for (int i = 0; i < 12; ++i) T[i] = N[i];
        if (j >= 1)
          for (int i = 0; i < 12; ++i) F[2 + 3 * lane + (j - 1)][i] = T[i];
      }
    }
  }
  __syncthreads();

  if (live && lane < 21) {   // hand_skinning.py:56-97, 130-151
    const size_t mo = ((size_t)mb * 21 + lane) * 3;
    const float px = lm_rest[mo], py = lm_rest[mo + 1], pz = lm_rest[mo + 2];
    int idx[3];
    float wt[3];
    // A used bone index (weight != 0) outside [0, 16]: the reference scatters the weight through a flattened index
    // (hand_skinning.py:87-94), so it lands in a neighbouring landmark's row or raises IndexError.  Here the landmark
    // becomes NaN and UMT_STATUS_BAD_BONE_INDEX is reported; the clamp below only keeps the LDS access in range.
    bool bad_index = false;
    for (int k = 0; k < 3; ++k) {
      const int raw_idx = lm_idx[mo + k];
      wt[k] = lm_w[mo + k];
      bad_index = bad_index || (wt[k] != 0.f && (raw_idx < 0 || raw_idx > 16));
      idx[k] = min(max(raw_idx, 0), 16);
    }
    // dense weights are written by assignment: a later non-zero entry for the same bone wins
    if (wt[0] != 0.f && ((wt[1] != 0.f && idx[1] == idx[0]) || (wt[2] != 0.f && idx[2] == idx[0]))) wt[0] = 0.f;
    if (wt[1] != 0.f && wt[2] != 0.f && idx[2] == idx[1]) wt[1] = 0.f;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    for (int k = 0; k < 3; ++k) {
      if (wt[k] == 0.f) continue;
      const float* M = F[idx[k]];
      const float qx = px * wt[k], qy = py * wt[k], qz = pz * wt[k], qw = wt[k];
      ox += M[0] * qx + M[1] * qy + M[2] * qz + M[3] * qw;
      oy += M[4] * qx + M[5] * qy + M[6] * qz + M[7] * qw;
      oz += M[8] * qx + M[9] * qy + M[10] * qz + M[11] * qw;
    }
    if (bad_index) {
      ox = oy = oz = __builtin_nanf("");
      if (status) atomicOr(status, UMT_STATUS_BAD_BONE_INDEX);
    }
    float* o = out + ((size_t)b * 21 + lane) * 3;
    o[0] = ox; o[1] = oy; o[2] = oz;
  }
}

}  // namespace umt

extern "C" int umt_fk_landmarks(const float* joint_angles, const float* wrist, const float* axes, const float* rest,
                                const float* lm_rest, const float* lm_w, const int32_t* lm_idx, int n_model,
                                const int64_t* hand_idx, int mirror_right, float wrist_t_scale, int b, float* out,
                                int32_t* status, void* stream) {
  using namespace umt;
  if (b == 0) return 0;
  UMT_REQUIRE(b > 0, UMT_ERR_INVALID, "fk: b = %d", b);
  UMT_REQUIRE(joint_angles && wrist && axes && rest && lm_rest && lm_w && lm_idx && out, UMT_ERR_INVALID,
              "fk: null pointer");
  UMT_REQUIRE(n_model == 1 || n_model == b, UMT_ERR_INVALID, "fk: n_model must be 1 or b (%d), got %d", b, n_model);
  UMT_REQUIRE(!mirror_right || hand_idx, UMT_ERR_INVALID, "fk: mirror_right needs hand_idx");
  const unsigned blocks = (unsigned)((b + FK_WAVES - 1) / FK_WAVES);
  hipLaunchKernelGGL(fk_kernel, dim3(blocks), dim3(64 * FK_WAVES), 0, (hipStream_t)stream, joint_angles, wrist, axes,
                     rest, lm_rest, lm_w, lm_idx, n_model, hand_idx, mirror_right, wrist_t_scale, b, out, status);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}
