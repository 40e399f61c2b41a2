// Row A0 / F1 of SURVEY.md section 8: crop-camera generation for a batch of hands on the GPU.
//
// Replaces, per hand, gen_crop_cameras_from_pose (lib/tracker/perspective_crop.py:118-162):
//   rank_hand_visibility_in_cameras (perspective_crop.py:54-86): count the tracked pose's 21
//     landmarks that fall inside each fisheye image with z > 0, keep cameras with enough of them;
//   gen_crop_parameters_from_points (lib/common/crop.py:31-82) with make_look_at_matrix /
//     from_two_vectors (lib/common/affine.py:34-66) and gen_intrinsics_from_bounding_pts
//     (crop.py:15-28): look-at rotation towards the bounding-box centre of the crop points, roll by
//     the camera angle, mirror x for right hands, focal length that just contains the points;
// and the per-crop packing of HandTracker._make_inputs (lib/tracker/tracker.py:212-216): K as
// uv_to_window_matrix, E = inv(cam_to_world) with the translation in metres.
//
// One wavefront per hand, lane = crop point (at most 63).  All geometry is float64 like the
// reference's numpy; the only float32 step is the bounding-box centre, which numpy computes on
// the float32 landmarks.  The per-camera 4x4 algebra is evaluated redundantly by every lane (a few
// hundred flops) so no lane ever waits on another except in the three wave reductions.
// The crop points come from the FK kernel (umt_fk_landmarks on the tracked, neutral and open
// pose), so with this kernel no per-frame geometry is left on the host.
#include "umt_internal.h"

namespace umt {

struct Fisheye {
  double w, h, fx, fy, cx, cy, k1, k2, k3, k4, p1, p2, k5, k6, xf[16];
};

__device__ inline void load_fisheye(const double* p, Fisheye& s) {
  s.w = p[0]; s.h = p[1]; s.fx = p[2]; s.fy = p[3]; s.cx = p[4]; s.cy = p[5];
  s.k1 = p[6]; s.k2 = p[7]; s.k3 = p[8]; s.k4 = p[9]; s.p1 = p[10]; s.p2 = p[11]; s.k5 = p[12]; s.k6 = p[13];
  for (int i = 0; i < 16; ++i) s.xf[i] = p[14 + i];
}

// CameraModel.world_to_eye + eye_to_window for Fisheye62 (camera.py:78-85, 120-141, 257-273)
__device__ inline bool landmark_visible(const Fisheye& s, double wx, double wy, double wz) {
  wx -= s.xf[3]; wy -= s.xf[7]; wz -= s.xf[11];
  const double x = wx * s.xf[0] + wy * s.xf[4] + wz * s.xf[8];
  const double y = wx * s.xf[1] + wy * s.xf[5] + wz * s.xf[9];
  const double z = wx * s.xf[2] + wy * s.xf[6] + wz * s.xf[10];
  const double r = sqrt(x * x + y * y);
  const double sc = atan2(r, z) / fmax(r, 2.938735877055719e-39 /* 2^-128 */);
  const double px = x * sc, py = y * sc;
  const double pi2 = 9.869604401089358;
  const double r2 = fmin(fmax(px * px + py * py, -pi2), pi2);
  const double r4 = r2 * r2, r6 = r2 * r4, r8 = r4 * r4, r10 = r4 * r6, r12 = r6 * r6;
  const double radial = 1 + s.k1 * r2 + s.k2 * r4 + s.k3 * r6 + s.k4 * r8 + s.k5 * r10 + s.k6 * r12;
  const double ux = px * radial, uy = py * radial;
  const double x2 = ux * ux, y2 = uy * uy, xy = ux * uy, rr = x2 + y2;
  const double ox = (ux + (2 * s.p2 * xy + s.p1 * (rr + 2 * x2))) * s.fx + s.cx;
  const double oy = (uy + (2 * s.p1 * xy + s.p2 * (rr + 2 * y2))) * s.fy + s.cy;
  return ox >= 0 && ox <= s.w - 1 && oy >= 0 && oy <= s.h - 1 && z > 0;     // perspective_crop.py:68-74
}

// general 4x4 inverse (Gauss-Jordan, partial pivoting) - numpy.linalg.inv on the reference side.
// Every index is a compile-time constant (the loops unroll), so the 4 x 8 tableau lives in registers: with a run-time pivot
// row it sat in scratch memory and the kernel's six inversions per view were a stream of scratch accesses (0.33 ms per
// 8192 hands).  The pivot row comes up by conditional swaps with each lower row in turn: the row that ends in place c is
// the first of largest magnitude, as before, and where the OTHER rows sit does not enter the arithmetic (each is updated
// by its own multiple of the pivot row), so the result is bit-identical to the single-swap form.
__device__ __forceinline__ void inv4(const double* a, double* out) {
  double m[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[i][j] = a[i * 4 + j]; m[i][4 + j] = i == j ? 1.0 : 0.0; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      const bool sw = fabs(m[r][c]) > fabs(m[c][c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double t = m[c][j];
        m[c][j] = sw ? m[r][j] : t;
        m[r][j] = sw ? t : m[r][j];
      }
    }
    const double d = 1.0 / m[c][c];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[c][j] *= d;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (r != c) {
      const double f = m[r][c];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = m[i][4 + j];
}

__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ inline double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ inline float wave_maxf(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ inline float wave_minf(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

constexpr int MAX_RIG_VIEWS = 16;
constexpr int MAX_CROP_VIEWS = 4;

struct CropArgs {
  const float* points;          // [B][P][3]
  const double* cameras;        // [n_rigs][V][30]
  const double* angles;         // [n_rigs][V] degrees
  const int32_t* rig_idx;       // [B] or null
  const int64_t* hand_idx;      // [B]
  const int32_t* forced_views;  // [B][max_views] or null
  int B, P, V, max_views, min_visible, sort_by_index, mirror_right, crop_w, crop_h;
  double focal_multiplier;
  int32_t* n_views;             // [B]
  int32_t* view_idx;            // [B][max_views]
  double* dst_cams;             // [B][max_views][20]
  float* intrinsics;            // [B][max_views][9]
  float* extrinsics;            // [B][max_views][16]
  int32_t* status;              // [B][max_views]
};

__global__ __launch_bounds__(256) void crop_cameras_kernel(CropArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;                                   // whole wave leaves together
  const int rig = a.rig_idx ? a.rig_idx[b] : 0;
  const double* cams = a.cameras + (size_t)rig * a.V * 30;
  const double* angles = a.angles + (size_t)rig * a.V;
  const bool have = lane < a.P;
  float pf[3] = {0.f, 0.f, 0.f};
  if (have) {
    const float* p = a.points + ((size_t)b * a.P + lane) * 3;
    pf[0] = p[0]; pf[1] = p[1]; pf[2] = p[2];
  }
  const double px = pf[0], py = pf[1], pz = pf[2];

  // ---- which cameras ----
  int sel[MAX_CROP_VIEWS];
  int nsel = 0;
  if (a.forced_views) {
    for (int k = 0; k < a.max_views; ++k) {
      const int v = a.forced_views[(size_t)b * a.max_views + k];
      if (v >= 0 && v < a.V) sel[nsel++] = v;
    }
  } else {
    int count[MAX_RIG_VIEWS];
    for (int v = 0; v < a.V; ++v) {
      Fisheye s;
      load_fisheye(cams + (size_t)v * 30, s);
      const bool vis = lane < 21 && landmark_visible(s, px, py, pz);
      count[v] = (int)__popcll(__ballot(vis));
    }
    if (a.sort_by_index) {
      for (int v = 0; v < a.V && nsel < a.max_views; ++v)
        if (count[v] >= a.min_visible) sel[nsel++] = v;
    } else {
      // most landmarks first, ties in index order (list.sort is stable)
      unsigned used = 0;
      while (nsel < a.max_views) {
        int best = -1;
        for (int v = 0; v < a.V; ++v)
          if (!((used >> v) & 1u) && count[v] >= a.min_visible && (best < 0 || count[v] > count[best])) best = v;
        if (best < 0) break;
        used |= 1u << best;
        sel[nsel++] = best;
      }
    }
  }

  // ---- bounding-box centre of the crop points, in float32 like numpy on the float32 landmarks ----
  const float big = 3.0e38f;
  float cen[3];
  for (int i = 0; i < 3; ++i) {
    const float lo = wave_minf(have ? pf[i] : big), hi = wave_maxf(have ? pf[i] : -big);
    cen[i] = (lo + hi) / 2.0f;
  }
  const bool mirror = a.mirror_right && a.hand_idx[b] == 1;

  for (int k = 0; k < a.max_views; ++k) {
    const size_t o = (size_t)b * a.max_views + k;
    if (k >= nsel) {
      if (lane == 0) { a.view_idx[o] = -1; a.status[o] = 0; }
      if (lane < 20) a.dst_cams[o * 20 + lane] = 0.0;
      if (lane < 9) a.intrinsics[o * 9 + lane] = 0.f;
      if (lane < 16) a.extrinsics[o * 16 + lane] = 0.f;
      continue;
    }
    const int v = sel[k];
    double c2w[16], w2e[16];
    for (int i = 0; i < 16; ++i) c2w[i] = cams[(size_t)v * 30 + 14 + i];
    inv4(c2w, w2e);
    // make_look_at_matrix (affine.py:47-66)
    double loc[3];
    for (int i = 0; i < 3; ++i)
      loc[i] = w2e[i * 4] * (double)cen[0] + w2e[i * 4 + 1] * (double)cen[1] + w2e[i * 4 + 2] * (double)cen[2] + w2e[i * 4 + 3];
    const double ln = sqrt(loc[0] * loc[0] + loc[1] * loc[1] + loc[2] * loc[2]);
    double zd[3] = {loc[0] / ln, loc[1] / ln, loc[2] / ln};
    const double bn = fmax(5.43e-20, sqrt(zd[0] * zd[0] + zd[1] * zd[1] + zd[2] * zd[2]));
    const double bx = zd[0] / bn, by = zd[1] / bn, bz = zd[2] / bn;
    // from_two_vectors((0,0,1), b): axis = a x b = (-by, bx, 0)
    const double ax = -by, ay = bx;
    const double s2 = ax * ax + ay * ay;
    const double coef = (1.0 - bz) / fmax(s2, 1e-15);
    // K = skew(axis) with axis_z = 0:  [[0,0,ay],[0,0,-ax],[-ay,ax,0]];  K^2 = [[-ay^2, ax*ay, 0],[ax*ay,-ax^2,0],[0,0,-(ax^2+ay^2)]]
    double delta[9] = {1.0 - ay * ay * coef, ax * ay * coef, ay,
                       ax * ay * coef, 1.0 - ax * ax * coef, -ax,
                       -ay, ax, 1.0 - s2 * coef};
    double e2w[16];
    inv4(w2e, e2w);
    double sn, cs;
    sincos(angles[v] * 0.017453292519943295, &sn, &cs);
    const double rz[9] = {cs, -sn, 0.0, sn, cs, 0.0, 0.0, 0.0, 1.0};
    double tmp[9], rot[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        tmp[i * 3 + j] = e2w[i * 4] * delta[j] + e2w[i * 4 + 1] * delta[3 + j] + e2w[i * 4 + 2] * delta[6 + j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        rot[i * 3 + j] = tmp[i * 3] * rz[j] + tmp[i * 3 + 1] * rz[3 + j] + tmp[i * 3 + 2] * rz[6 + j];
    double nw[16], nw2e[16];
    for (int i = 0; i < 16; ++i) nw[i] = e2w[i];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) nw[i * 4 + j] = rot[i * 3 + j];
    inv4(nw, nw2e);
    if (mirror) for (int j = 0; j < 4; ++j) nw2e[j] = -nw2e[j];     // flip @ new_world_to_eye (crop.py:57-60)
    // gen_intrinsics_from_bounding_pts (crop.py:15-28)
    const double ex = nw2e[0] * px + nw2e[1] * py + nw2e[2] * pz + nw2e[3];
    const double ey = nw2e[4] * px + nw2e[5] * py + nw2e[6] * pz + nw2e[7];
    const double ez = nw2e[8] * px + nw2e[9] * py + nw2e[10] * pz + nw2e[11];
    const double m = wave_max(have ? fmax(fabs(ex / ez), fabs(ey / ez)) : 0.0);
    const double zmin = wave_min(have ? ez : 1.0e300);
    const double ccx = ((double)a.crop_w - 1.0) / 2.0, ccy = ((double)a.crop_h - 1.0) / 2.0;
    double fx = ccx / m, fy = ccy / m;
    const int bad = (zmin < 0.0001 || fx < 5.0 || fy < 5.0 || !(m == m)) ? 1 : 0;   // the reference raises ValueError
    fx *= a.focal_multiplier;
    fy *= a.focal_multiplier;
    double cc2w[16];
    inv4(nw2e, cc2w);
    if (lane == 0) {
      a.view_idx[o] = v;
      a.status[o] = bad;
      double* d = a.dst_cams + o * 20;
      d[0] = fx; d[1] = fy; d[2] = ccx; d[3] = ccy;
      float* K = a.intrinsics + o * 9;
      K[0] = (float)fx; K[1] = 0.f; K[2] = (float)ccx; K[3] = 0.f; K[4] = (float)fy; K[5] = (float)ccy;
      K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
    }
    if (lane < 16) {
      a.dst_cams[o * 20 + 4 + lane] = cc2w[lane];
      // tracker.py:214-216: inv(camera_to_world) with the translation scaled mm -> m
      double e[16];
      inv4(cc2w, e);
      const bool is_t = (lane & 3) == 3 && lane < 12;
      a.extrinsics[o * 16 + lane] = (float)(is_t ? e[lane] * 0.001 : e[lane]);
    }
  }
  if (lane == 0) a.n_views[b] = nsel;
}

}  // namespace umt

using namespace umt;

extern "C" int umt_crop_cameras(const float* points, int n_points, const double* cameras, const double* camera_angles,
                                int n_rig_views, const int32_t* rig_idx, const int64_t* hand_idx,
                                const int32_t* forced_views, int n_hands, int max_views, int min_visible,
                                int sort_by_index, int mirror_right, double focal_multiplier, int crop_w, int crop_h,
                                int32_t* n_views, int32_t* view_idx, double* dst_cams, float* intrinsics,
                                float* extrinsics, int32_t* status, void* stream) {
  UMT_REQUIRE(points && cameras && camera_angles && hand_idx && n_views && view_idx && dst_cams && intrinsics &&
                  extrinsics && status,
              UMT_ERR_INVALID, "crop_cameras: null pointer");
  UMT_REQUIRE(n_hands > 0, UMT_ERR_INVALID, "crop_cameras: n_hands = %d", n_hands);
  UMT_REQUIRE(n_points >= 21 && n_points <= 63, UMT_ERR_INVALID,
              "crop_cameras: %d crop points (the first 21 are the tracked pose's landmarks; at most 63)", n_points);
  UMT_REQUIRE(n_rig_views >= 1 && n_rig_views <= MAX_RIG_VIEWS, UMT_ERR_INVALID, "crop_cameras: %d rig cameras (max %d)",
              n_rig_views, MAX_RIG_VIEWS);
  UMT_REQUIRE(max_views >= 1 && max_views <= MAX_CROP_VIEWS, UMT_ERR_INVALID, "crop_cameras: max_views = %d (1..%d)",
              max_views, MAX_CROP_VIEWS);
  UMT_REQUIRE(crop_w > 1 && crop_h > 1 && focal_multiplier > 0, UMT_ERR_INVALID, "crop_cameras: bad crop size or focal multiplier");
  CropArgs a;
  a.points = points; a.cameras = cameras; a.angles = camera_angles; a.rig_idx = rig_idx; a.hand_idx = hand_idx;
  a.forced_views = forced_views;
  a.B = n_hands; a.P = n_points; a.V = n_rig_views; a.max_views = max_views; a.min_visible = min_visible;
  a.sort_by_index = sort_by_index; a.mirror_right = mirror_right; a.crop_w = crop_w; a.crop_h = crop_h;
  a.focal_multiplier = focal_multiplier;
  a.n_views = n_views; a.view_idx = view_idx; a.dst_cams = dst_cams; a.intrinsics = intrinsics;
  a.extrinsics = extrinsics; a.status = status;
  hipLaunchKernelGGL(crop_cameras_kernel, dim3((n_hands + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}
