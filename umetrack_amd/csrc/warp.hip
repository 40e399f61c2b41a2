// Stage 1 of the hot path: resample source images into 96x96 crops.
//
//  * umt_warp_crops     - fisheye (arctan + 6 radial / 2 tangential) -> pinhole crop, replaces
//                         _warp_image + cv2.remap + /255 (lib/tracker/tracker.py:54-80, 210-211,
//                         camera maths lib/common/camera.py:61-94, 106-141, 257-281).
//  * umt_resample_pinhole - pinhole -> pinhole float bilinear, replaces _resample_images_batched
//                         (lib/batched_dataset/data_transform.py:78-144).
//
// One workgroup per crop.  Pass 1 evaluates the coordinate map and keeps the 20 fixed-point
// coordinates per thread in registers while the workgroup reduces their bounding box.  Pass 2
// copies that source rectangle into LDS with coalesced row reads, pass 3 gathers the four taps
// from LDS and writes whole 384-B output rows.  Rectangles that do not fit fall back to
// gathering from global memory.  Algorithmic traffic: 4 source bytes + 4 output bytes per pixel.
//
// The reference builds the map with float64 numpy and casts it to float32; OpenCV then rounds the
// coordinate to 1/32 px.  In UMT_WARP_CV2_FIXED_POINT mode only that integer matters, so pass 1 runs
// the camera model in float32 first (same formulas on a pre-multiplied rotation) and accepts the
// result where the 1/32-px rounding cannot be affected by the float32 error (bound WARP_FAST_ULPS
// relative roundings, 6x the largest error seen on the test and bench cameras).  The other pixels
// (a few per cent: within the bound of a rounding boundary, near z = 0, not finite) are queued in
// LDS and re-evaluated in float64 by the whole workgroup, densely.  UMT_WARP_CV2_FIXED_POINT_F64
// evaluates every pixel in float64 (the tests hold the two modes bit-identical); UMT_WARP_FLOAT
// always uses float64 maps.
#include "umt_internal.h"

namespace umt {

constexpr int WARP_THREADS = 512;                // 8 waves per crop, 2 crops per CU: 4 waves per SIMD hide the maps' dependent chains
constexpr int WARP_MAX_PX = 20;                 // >= 96*96 / 512, as 5 groups of 4 consecutive pixels
constexpr int WARP_LDS_BYTES = 60 * 1024;       // source rectangle budget (2 workgroups per CU)
constexpr int WARP_QCAP = 3072;                 // float64 re-evaluation queue (aliases the rectangle: 30 KB)
constexpr float WARP_FAST_ULPS = 12.0f;         // accepted float32 error of a source coordinate, in 2^-23 * |coordinate|

struct SrcCam {
  double w, h, fx, fy, cx, cy, k1, k2, k3, k4, p1, p2, k5, k6, xf[16];
};
struct DstCam {
  double fx, fy, cx, cy, xf[16];
};

// float32 camera constants of one crop for the fast map
struct FastCam {
  float m[9], t[3];          // source-eye = m * crop-eye + t
  float ifx, ify, dcx, dcy;  // crop camera
  float fx, fy, cx, cy, k1, k2, k3, k4, k5, k6, p1, p2;
};

__device__ inline FastCam make_fast(const SrcCam& s, const DstCam& d) {
  FastCam f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      f.m[r * 3 + c] = (float)(s.xf[0 + r] * d.xf[0 + c] + s.xf[4 + r] * d.xf[4 + c] + s.xf[8 + r] * d.xf[8 + c]);
    f.t[r] = (float)(s.xf[0 + r] * (d.xf[3] - s.xf[3]) + s.xf[4 + r] * (d.xf[7] - s.xf[7]) + s.xf[8 + r] * (d.xf[11] - s.xf[11]));
  }
  f.ifx = (float)(1.0 / d.fx); f.ify = (float)(1.0 / d.fy); f.dcx = (float)d.cx; f.dcy = (float)d.cy;
  f.fx = (float)s.fx; f.fy = (float)s.fy; f.cx = (float)s.cx; f.cy = (float)s.cy;
  f.k1 = (float)s.k1; f.k2 = (float)s.k2; f.k3 = (float)s.k3; f.k4 = (float)s.k4; f.k5 = (float)s.k5; f.k6 = (float)s.k6;
  f.p1 = (float)s.p1; f.p2 = (float)s.p2;
  return f;
}

// float32 map of one pixel -> 1/32-px integers.  Returns false when the float64 evaluation must decide.
// Uses the hardware reciprocal / reciprocal-square-root / square-root (1 ulp each) and a degree-9
// polynomial in a^2 for atan(a)/a on [0,1] (6e-8 relative in float32); all of it is inside the error
// bound that decides whether the float32 result is accepted.
__device__ inline bool map_point_fast(const FastCam& f, int u, int v, int& sx, int& sy) {
  const float qx = ((float)u - f.dcx) * f.ifx, qy = ((float)v - f.dcy) * f.ify;
  const float inv = __builtin_amdgcn_rsqf(qx * qx + qy * qy + 1.0f);
  const float ex = qx * inv, ey = qy * inv, ez = inv;
  const float x = f.m[0] * ex + f.m[1] * ey + f.m[2] * ez + f.t[0];
  const float y = f.m[3] * ex + f.m[4] * ey + f.m[5] * ez + f.t[1];
  const float z = f.m[6] * ex + f.m[7] * ey + f.m[8] * ez + f.t[2];
  // depth check (tracker.py:71-73): the sign of z must be certain
  const float zmag = fabsf(x) + fabsf(y) + fabsf(z);
  if (!(fabsf(z) > 1e-4f * zmag)) return false;
  if (z < 0.f) { sx = -32; sy = -32; return true; }
  // sc = atan2(r, z) / r with z > 0: a = min/max, theta = a*P(a^2) or pi/2 - a*P(a^2)
  const float r = __builtin_amdgcn_sqrtf(x * x + y * y);
  const float inv_hi = __builtin_amdgcn_rcpf(fmaxf(r, z));
  const float a = fminf(r, z) * inv_hi, s2 = a * a;
  const float P = 1.0f + s2 * (-0.3333328366279602f + s2 * (0.1999826580286026f + s2 * (-0.142619326710701f + s2 * (0.1094203069806099f +
                  s2 * (-0.08377315104007721f + s2 * (0.05755209922790527f + s2 * (-0.030804594978690147f + s2 * (0.01072694268077612f +
                  s2 * -0.0017539297696202993f))))))));
  const float sc = (r <= z ? P : 1.5707963267948966f - a * P) * inv_hi;
  const float px = x * sc, py = y * sc;
  const float r2 = fminf(px * px + py * py, 9.8696044f);
  const float radial = 1.0f + r2 * (f.k1 + r2 * (f.k2 + r2 * (f.k3 + r2 * (f.k4 + r2 * (f.k5 + r2 * f.k6)))));
  // relative sensitivity of the radial factor to r2: g / |radial| with g = r2 * |radial'|.  Where the
  // distortion polynomial is steep (far off axis) the float32 error is amplified: leave it to float64.
  const float dradial = f.k1 + r2 * (2.f * f.k2 + r2 * (3.f * f.k3 + r2 * (4.f * f.k4 + r2 * (5.f * f.k5 + r2 * 6.f * f.k6))));
  const float g = r2 * fabsf(dradial), arad = fabsf(radial);
  if (!(g <= 0.5f * arad)) return false;
  const float ux = px * radial, uy = py * radial;
  const float x2 = ux * ux, y2 = uy * uy, xy = ux * uy, rr = x2 + y2;
  const float tx = 2.0f * f.p2 * xy + f.p1 * (rr + 2.0f * x2), ty = 2.0f * f.p1 * xy + f.p2 * (rr + 2.0f * y2);
  const float dx = ux + tx, dy = uy + ty;
  const float ox = dx * f.fx + f.cx, oy = dy * f.fy + f.cy;
  if (!(fabsf(ox) < 30000.f && fabsf(oy) < 30000.f)) return false;          // also catches NaN
  const float cx32 = ox * 32.0f, cy32 = oy * 32.0f;
  const float rx = rintf(cx32), ry = rintf(cy32);
  // distance to the nearest rounding boundary (k + 0.5) must exceed the error bound, in 1/32 px:
  //   bound = 32 * (ULPS+1) * 2^-23 * (1 + 4 g/|radial|) * magnitude, magnitude = the largest value that
  //   entered the coordinate; compared after multiplying through by |radial|
  const float kb = 32.0f * (WARP_FAST_ULPS + 1.0f) * 1.1920929e-7f * (arad + 4.0f * g);
  const float magx = fmaxf((fabsf(ux) + fabsf(tx)) * fabsf(f.fx) + fabsf(f.cx), 64.f);
  const float magy = fmaxf((fabsf(uy) + fabsf(ty)) * fabsf(f.fy) + fabsf(f.cy), 64.f);
  if (!((0.5f - fabsf(cx32 - rx)) * arad > kb * magx && (0.5f - fabsf(cy32 - ry)) * arad > kb * magy)) return false;
  sx = (int)rx; sy = (int)ry;
  return true;
}

__device__ inline void quantise_map(float mx, float my, int& sx, int& sy) {
  const float cx = fminf(fmaxf(mx * 32.0f, -1048576.f), 1048576.f);   // NaN -> lower bound
  const float cy = fminf(fmaxf(my * 32.0f, -1048576.f), 1048576.f);
  sx = (int)rintf(cx);
  sy = (int)rintf(cy);
}


// window (u,v) of the crop camera -> window coordinates in the fisheye source, as float32
template <bool DEPTH_CHECK = true>
__device__ inline void map_point(const SrcCam& s, const DstCam& d, int u, int v, float& mx, float& my) {
  // camera.py:275-281 + PerspectiveProjection.unproject (camera.py:68-75) + affine.normalized
  double qx = ((double)u - d.cx) / d.fx, qy = ((double)v - d.cy) / d.fy;
  double nrm = fmax(5.43e-20, sqrt(qx * qx + qy * qy + 1.0));
  double ex = qx / nrm, ey = qy / nrm, ez = 1.0 / nrm;
  // eye -> world (camera.py:263-267)
  double wx = ex * d.xf[0] + ey * d.xf[1] + ez * d.xf[2] + d.xf[3];
  double wy = ex * d.xf[4] + ey * d.xf[5] + ez * d.xf[6] + d.xf[7];
  double wz = ex * d.xf[8] + ey * d.xf[9] + ez * d.xf[10] + d.xf[11];
  // world -> source eye: R^T (p - t)  (camera.py:257-261)
  wx -= s.xf[3]; wy -= s.xf[7]; wz -= s.xf[11];
  double x = wx * s.xf[0] + wy * s.xf[4] + wz * s.xf[8];
  double y = wx * s.xf[1] + wy * s.xf[5] + wz * s.xf[9];
  double z = wx * s.xf[2] + wy * s.xf[6] + wz * s.xf[10];
  // ArctanProjection.project (camera.py:78-85)
  double r = sqrt(x * x + y * y);
  double sc = atan2(r, z) / fmax(r, 2.938735877055719e-39 /* 2^-128 */);
  double px = x * sc, py = y * sc;
  // Fisheye62 distortion (camera.py:120-141)
  const double pi2 = 9.869604401089358;
  double r2 = fmin(fmax(px * px + py * py, -pi2), pi2);
  double r4 = r2 * r2, r6 = r2 * r4, r8 = r4 * r4, r10 = r4 * r6, r12 = r6 * r6;
  double radial = 1 + s.k1 * r2 + s.k2 * r4 + s.k3 * r6 + s.k4 * r8 + s.k5 * r10 + s.k6 * r12;
  double ux = px * radial, uy = py * radial;
  double x2 = ux * ux, y2 = uy * uy, xy = ux * uy, rr = x2 + y2;
  double dx = ux + (2 * s.p2 * xy + s.p1 * (rr + 2 * x2));
  double dy = uy + (2 * s.p1 * xy + s.p2 * (rr + 2 * y2));
  double ox = dx * s.fx + s.cx, oy = dy * s.fy + s.cy;
  if (DEPTH_CHECK && z < 0) { ox = -1.0; oy = -1.0; }   // depth check, tracker.py:71-73
  mx = (float)ox;
  my = (float)oy;
}

// pixel handled by slot i of thread tid: groups of 4 consecutive pixels, groups dealt round-robin to the threads
__device__ __forceinline__ int warp_pixel(int tid, int i) { return 4 * (tid + (i >> 2) * WARP_THREADS) + (i & 3); }

template <bool FROM_LDS>
__device__ inline int fetch(const uint8_t* img, int pitch, int W, int H, const uint8_t* tile, int tx0, int ty0,
                            int tpitch, int y, int x) {
  if (x < 0 || y < 0 || x >= W || y >= H) return 0;   // BORDER_CONSTANT 0
  if (FROM_LDS)    // LDS row = the source row from the 4-byte-aligned address at or below its first rectangle byte
    return tile[(y - ty0) * tpitch + (x - tx0) + (int)((reinterpret_cast<uintptr_t>(img) + (size_t)y * pitch + tx0) & 3)];
  return img[(size_t)y * pitch + x];
}

template <bool FROM_LDS>
__device__ inline float sample(int mode, const uint8_t* img, int pitch, int W, int H, const uint8_t* tile, int tx0,
                               int ty0, int tpitch, int sx, int sy, float mx, float my) {
  if (mode == UMT_WARP_CV2_FIXED_POINT) {
    // [external] OpenCV 8-bit INTER_LINEAR: 1/32-px coordinates, 15-bit weights, round to nearest
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const int a = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy, ix);
    const int b = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy, ix + 1);
    const int c = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy + 1, ix);
    const int d = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy + 1, ix + 1);
    const int acc = (a * ((32 - fy) * (32 - fx)) + b * ((32 - fy) * fx) + c * (fy * (32 - fx)) + d * (fy * fx)) * 32;
    const int v = min(max((acc + (1 << 14)) >> 15, 0), 255);
    return (float)v / 255.0f;   // tracker.py:211
  }
  const double dx = (double)mx, dy = (double)my;
  const double flx = floor(dx), fly = floor(dy);
  const double ax = dx - flx, ay = dy - fly;
  const int ix = (int)fmin(fmax(flx, -4.0), 1.0e6), iy = (int)fmin(fmax(fly, -4.0), 1.0e6);
  const double a = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy, ix);
  const double b = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy, ix + 1);
  const double c = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy + 1, ix);
  const double d = fetch<FROM_LDS>(img, pitch, W, H, tile, tx0, ty0, tpitch, iy + 1, ix + 1);
  const float v = (float)(a * (1 - ax) * (1 - ay) + b * ax * (1 - ay) + c * (1 - ax) * ay + d * ax * ay);
  return v / 255.0f;
}

// The float64 camera pair of crop k.  Loaded where it is used (uniform, scalar-cached loads): holding its
// 50 doubles across the float32 pass would cost ~100 registers.
__device__ inline void load_cams(const double* __restrict__ src_cams, const double* __restrict__ dst_cams, int k, SrcCam& s,
                                 DstCam& d) {
  const double* sp = src_cams + (size_t)k * 30;
  s.w = sp[0]; s.h = sp[1]; s.fx = sp[2]; s.fy = sp[3]; s.cx = sp[4]; s.cy = sp[5];
  s.k1 = sp[6]; s.k2 = sp[7]; s.k3 = sp[8]; s.k4 = sp[9]; s.p1 = sp[10]; s.p2 = sp[11]; s.k5 = sp[12]; s.k6 = sp[13];
  for (int i = 0; i < 16; ++i) s.xf[i] = sp[14 + i];
  const double* dp = dst_cams + (size_t)k * 20;
  d.fx = dp[0]; d.fy = dp[1]; d.cx = dp[2]; d.cy = dp[3];
  for (int i = 0; i < 16; ++i) d.xf[i] = dp[4 + i];
}

// MODE: UMT_WARP_CV2_FIXED_POINT (float32 map + float64 re-evaluation queue), UMT_WARP_FLOAT,
// UMT_WARP_CV2_FIXED_POINT_F64 (every pixel in float64)
template <int MODE, bool DEPTH_CHECK = true>
__global__ __launch_bounds__(WARP_THREADS, 4) void warp_kernel(const uint8_t* __restrict__ src_images,
                                                            const int64_t* __restrict__ src_offset,
                                                            const int32_t* __restrict__ src_pitch,
                                                            const double* __restrict__ src_cams,
                                                            const double* __restrict__ dst_cams, int out_h, int out_w,
                                                            float* __restrict__ out) {
  constexpr bool CV2 = MODE != UMT_WARP_FLOAT;
  constexpr int smode = CV2 ? UMT_WARP_CV2_FIXED_POINT : UMT_WARP_FLOAT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ int red[4][WARP_THREADS / 64];
  __shared__ int qcount;
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  const int W = (int)src_cams[(size_t)k * 30], H = (int)src_cams[(size_t)k * 30 + 1];
  const int pitch = src_pitch[k];
  const uint8_t* img = src_images + src_offset[k];
  const int npx = out_h * out_w;

  int sxs[WARP_MAX_PX], sys[WARP_MAX_PX];
  float mxs[CV2 ? 1 : WARP_MAX_PX], mys[CV2 ? 1 : WARP_MAX_PX];

  if (MODE == UMT_WARP_CV2_FIXED_POINT) {
    // float32 map; pixels it cannot decide go to the queue: pixel ids, then their float64 results
    uint16_t* qpix = reinterpret_cast<uint16_t*>(lds);
    int* qres = reinterpret_cast<int*>(lds + WARP_QCAP * 2);
    if (tid == 0) qcount = 0;
    __syncthreads();
    FastCam f;
    {
      SrcCam s;
      DstCam d;
      load_cams(src_cams, dst_cams, k, s, d);
      f = make_fast(s, d);
    }
    unsigned queued = 0;
    int gu = 0, gv = 0;
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) {
      const int p = warp_pixel(tid, i);
      sxs[i] = sys[i] = -32;
      bool undecided = false;
      if (p < npx) {
        if ((i & 3) == 0) { gv = p / out_w; gu = p - gv * out_w; }      // one division per group of 4 pixels
        int u = gu + (i & 3), v = gv;
        if (u >= out_w) { u -= out_w; ++v; }
#ifdef UMT_WARP_X_NOMAP
        sxs[i] = (int)(f.cx * 32.f) + 20 * u + 3 * v; sys[i] = (int)(f.cy * 32.f) + 21 * v - 2 * u;
#else
        undecided = !map_point_fast(f, u, v, sxs[i], sys[i]);
#endif
#ifdef UMT_WARP_X_NOQUEUE
        if (undecided) { undecided = false; sxs[i] = sys[i] = 64; }
#endif
      }
      if (undecided) {
        const int pos = atomicAdd(&qcount, 1);
        if (pos < WARP_QCAP) qpix[pos] = (uint16_t)p;
        sxs[i] = pos;
        queued |= 1u << i;
      }
    }
    __syncthreads();
    const int nq = qcount;
    SrcCam s;
    DstCam d;
    if (nq > 0) load_cams(src_cams, dst_cams, k, s, d);
    if (nq <= WARP_QCAP) {
      for (int q = tid; q < nq; q += WARP_THREADS) {
        const int p = qpix[q];
        const int v = p / out_w, u = p - v * out_w;
        float mx, my;
        map_point(s, d, u, v, mx, my);
        int sx, sy;
        quantise_map(mx, my, sx, sy);
        qres[2 * q] = sx;
        qres[2 * q + 1] = sy;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i)
        if ((queued >> i) & 1u) {
          const int pos = sxs[i];
          sxs[i] = qres[2 * pos];
          sys[i] = qres[2 * pos + 1];
        }
    } else {          // degenerate crop: every thread re-evaluates its own undecided pixels
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i)
        if ((queued >> i) & 1u) {
          const int p = warp_pixel(tid, i);
          const int v = p / out_w, u = p - v * out_w;
          float mx, my;
          map_point(s, d, u, v, mx, my);
          quantise_map(mx, my, sxs[i], sys[i]);
        }
    }
  } else {
    SrcCam s;
    DstCam d;
    load_cams(src_cams, dst_cams, k, s, d);
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) {
      const int p = warp_pixel(tid, i);
      sxs[i] = sys[i] = -32;
      if (!CV2) mxs[i] = mys[i] = -1.f;
      if (p < npx) {
        const int v = p / out_w, u = p - v * out_w;
        float mx, my;
        map_point<DEPTH_CHECK>(s, d, u, v, mx, my);
        if (CV2) {
          quantise_map(mx, my, sxs[i], sys[i]);
        } else {
          mxs[i] = mx; mys[i] = my;
          sxs[i] = (int)fmin(fmax(floor((double)mx), -4.0), 1.0e6);     // integer sample position
          sys[i] = (int)fmin(fmax(floor((double)my), -4.0), 1.0e6);
        }
      }
    }
  }

  int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -(1 << 30), by1 = -(1 << 30);
#pragma unroll
  for (int i = 0; i < WARP_MAX_PX; ++i) {
    const int p = warp_pixel(tid, i);
    if (p < npx) {
      const int ix = CV2 ? sxs[i] >> 5 : sxs[i], iy = CV2 ? sys[i] >> 5 : sys[i];
      if (ix + 1 >= 0 && ix < W && iy + 1 >= 0 && iy < H) {   // footprint touches the image
        bx0 = min(bx0, ix); bx1 = max(bx1, ix + 1);
        by0 = min(by0, iy); by1 = max(by1, iy + 1);
      }
    }
  }
  // workgroup bounding box of the source footprint
  for (int o = 32; o > 0; o >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, o)); by0 = min(by0, __shfl_xor(by0, o));
    bx1 = max(bx1, __shfl_xor(bx1, o)); by1 = max(by1, __shfl_xor(by1, o));
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = bx0; red[1][tid >> 6] = by0; red[2][tid >> 6] = bx1; red[3][tid >> 6] = by1;
  }
  __syncthreads();          // also: every read of the re-evaluation queue is done before the rectangle overwrites it
  bx0 = red[0][0]; by0 = red[1][0]; bx1 = red[2][0]; by1 = red[3][0];
#pragma unroll
  for (int wv = 1; wv < WARP_THREADS / 64; ++wv) {
    bx0 = min(bx0, red[0][wv]); by0 = min(by0, red[1][wv]);
    bx1 = max(bx1, red[2][wv]); by1 = max(by1, red[3][wv]);
  }
  bx0 = max(bx0, 0); by0 = max(by0, 0);
  bx1 = min(bx1, W - 1); by1 = min(by1, H - 1);
  const bool empty = bx1 < bx0 || by1 < by0;
  const int tw = empty ? 0 : bx1 - bx0 + 1, th = empty ? 0 : by1 - by0 + 1;
  const int tpitch = ((tw + 3) & ~3) + 4;          // bytes per LDS row: rectangle width + alignment slack, dword multiple
  const bool use_lds = !empty && (int64_t)tpitch * th <= WARP_LDS_BYTES;   // uniform over the workgroup

  float res[WARP_MAX_PX];
#ifdef UMT_WARP_X_NOGATHER
  if (use_lds || !use_lds) {
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) res[i] = (float)(sxs[i] + sys[i] + bx0 + by1);
  } else
#endif
  if (use_lds) {
    // aligned 4-byte copies: LDS row r holds the dwords that cover source bytes [bx0, bx0 + tw) of row by0 + r
    const int tpd = tpitch >> 2;
    const int total = tpd * th;
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds);
#pragma unroll 4
    for (int idx = tid; idx < total; idx += WARP_THREADS) {
      const int row = idx / tpd, col = idx - row * tpd;
      const uintptr_t a0 = (reinterpret_cast<uintptr_t>(img) + (size_t)(by0 + row) * pitch + bx0) & ~(uintptr_t)3;
      const int mis = (int)((reinterpret_cast<uintptr_t>(img) + (size_t)(by0 + row) * pitch + bx0) & 3);
      uint32_t v = 0;
      if (4 * col < mis + tw) v = *reinterpret_cast<const uint32_t*>(a0 + 4 * (size_t)col);
      lds32[idx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i)
      res[i] = sample<true>(smode, img, pitch, W, H, lds, bx0, by0, tpitch, sxs[i], sys[i], CV2 ? 0.f : mxs[i],
                            CV2 ? 0.f : mys[i]);
  } else {
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i)
      res[i] = warp_pixel(tid, i) < npx
                   ? sample<false>(smode, img, pitch, W, H, lds, 0, 0, 0, sxs[i], sys[i], CV2 ? 0.f : mxs[i], CV2 ? 0.f : mys[i])
                   : 0.f;
  }
  float* obase = out + (size_t)k * npx;
  const bool vec = (npx & 3) == 0;               // then every group of 4 is whole and 16-byte aligned
#pragma unroll
  for (int gq = 0; gq < WARP_MAX_PX / 4; ++gq) {
    const int p = warp_pixel(tid, 4 * gq);
    if (vec) {
      if (p < npx) *reinterpret_cast<float4*>(obase + p) = make_float4(res[4 * gq], res[4 * gq + 1], res[4 * gq + 2], res[4 * gq + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p + j < npx) obase[p + j] = res[4 * gq + j];
    }
  }
}

// data_transform.py:78-144, one thread per output pixel, float64 coordinates like the reference
__global__ void resample_kernel(const float* __restrict__ images, const float* __restrict__ xfs, int n, int h, int w,
                                int out_h, int out_w, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * out_h * out_w;
  if (t >= total) return;
  const int u = (int)(t % out_w);
  const int v = (int)((t / out_w) % out_h);
  const int img = (int)(t / ((int64_t)out_w * out_h));
  const float* m = xfs + (size_t)img * 16;
  const double px = (double)m[0] * u + (double)m[1] * v + (double)m[2] + (double)m[3];
  const double py = (double)m[4] * u + (double)m[5] * v + (double)m[6] + (double)m[7];
  const double pz = (double)m[8] * u + (double)m[9] * v + (double)m[10] + (double)m[11];
  const double x = px / pz, y = py / pz;
  float res = 0.f;
  if (x >= 0 && x < (double)(w - 1) && y >= 0 && y < (double)(h - 1)) {
    const int x0 = (int)x, y0 = (int)y, x1 = x0 + 1, y1 = y0 + 1;
    const float* s = images + (size_t)img * h * w;
    const double f00 = s[y0 * w + x0], f01 = s[y1 * w + x0], f10 = s[y0 * w + x1], f11 = s[y1 * w + x1];
    res = (float)(f00 * (x1 - x) * (y1 - y) + f10 * (x - x0) * (y1 - y) + f01 * (x1 - x) * (y - y0) +
                  f11 * (x - x0) * (y - y0));
  }
  out[t] = res;
}

}  // namespace umt

extern "C" int umt_warp_crops(const uint8_t* src_images, const int64_t* src_offset, const int32_t* src_pitch,
                              const double* src_cams, const double* dst_cams, int n, int out_h, int out_w, int mode,
                              float* out, void* stream) {
  using namespace umt;
  if (n == 0) return 0;
  UMT_REQUIRE(n > 0 && out_h > 0 && out_w >= 4, UMT_ERR_INVALID, "warp: n=%d out=%dx%d (width must be >= 4)", n, out_h, out_w);
  UMT_REQUIRE(out_h * out_w <= WARP_MAX_PX * WARP_THREADS, UMT_ERR_UNSUPPORTED,
              "warp: crops larger than %d pixels are not supported (got %dx%d)", WARP_MAX_PX * WARP_THREADS, out_w, out_h);
  const bool depth_check = (mode & UMT_WARP_NO_DEPTH_CHECK) == 0;
  mode &= ~UMT_WARP_NO_DEPTH_CHECK;
  UMT_REQUIRE(mode == UMT_WARP_CV2_FIXED_POINT || mode == UMT_WARP_FLOAT || mode == UMT_WARP_CV2_FIXED_POINT_F64,
              UMT_ERR_INVALID, "warp: mode %d", mode);
  UMT_REQUIRE(src_images && src_offset && src_pitch && src_cams && dst_cams && out, UMT_ERR_INVALID, "warp: null pointer");
  // depth_check=False (tracker.py:59, never used by the reference's own callers) runs every pixel in float64
  if (!depth_check && mode == UMT_WARP_CV2_FIXED_POINT) mode = UMT_WARP_CV2_FIXED_POINT_F64;
  auto kern = mode == UMT_WARP_CV2_FIXED_POINT ? warp_kernel<UMT_WARP_CV2_FIXED_POINT>
              : mode == UMT_WARP_FLOAT         ? (depth_check ? warp_kernel<UMT_WARP_FLOAT> : warp_kernel<UMT_WARP_FLOAT, false>)
              : depth_check                    ? warp_kernel<UMT_WARP_CV2_FIXED_POINT_F64>
                                               : warp_kernel<UMT_WARP_CV2_FIXED_POINT_F64, false>;
  hipLaunchKernelGGL(kern, dim3(n), dim3(WARP_THREADS), WARP_LDS_BYTES, (hipStream_t)stream, src_images, src_offset,
                     src_pitch, src_cams, dst_cams, out_h, out_w, out);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int umt_resample_pinhole(const float* images, const float* xfs, int n, int h, int w, int out_h, int out_w,
                                    float* out, void* stream) {
  using namespace umt;
  if (n == 0) return 0;
  UMT_REQUIRE(n > 0 && h > 1 && w > 1 && out_h > 0 && out_w > 0, UMT_ERR_INVALID, "resample: bad shape");
  UMT_REQUIRE(images && xfs && out, UMT_ERR_INVALID, "resample: null pointer");
  const int64_t total = (int64_t)n * out_h * out_w;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images,
                     xfs, n, h, w, out_h, out_w, out);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}
