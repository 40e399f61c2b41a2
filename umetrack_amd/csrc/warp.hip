// Stage 1 of the hot path: resample source images into 96x96 crops.
//
//  * umt_warp_crops     - fisheye (arctan + 6 radial / 2 tangential) -> pinhole crop, replaces
//                         _warp_image + cv2.remap + /255 (lib/tracker/tracker.py:54-80, 210-211,
//                         camera maths lib/common/camera.py:61-94, 106-141, 257-281).
//  * umt_resample_pinhole - pinhole -> pinhole float bilinear, replaces _resample_images_batched
//                         (lib/batched_dataset/data_transform.py:78-144).
//
// One workgroup per crop.  Pass 1 evaluates the coordinate map in double precision (the
// reference builds it with float64 numpy and casts to float32) and keeps the 36 fixed-point
// coordinates per thread in registers while the workgroup reduces their bounding box.  Pass 2
// copies that source rectangle into LDS with coalesced row reads, pass 3 gathers the four taps
// from LDS and writes whole 384-B output rows.  Rectangles that do not fit fall back to
// gathering from global memory.  Algorithmic traffic: 4 source bytes + 4 output bytes per pixel.
#include "umt_internal.h"

namespace umt {

constexpr int WARP_THREADS = 256;
constexpr int WARP_MAX_PX = 36;                 // 96*96 / 256
constexpr int WARP_LDS_BYTES = 60 * 1024;       // source rectangle budget (2 workgroups per CU)

struct SrcCam {
  double w, h, fx, fy, cx, cy, k1, k2, k3, k4, p1, p2, k5, k6, xf[16];
};
struct DstCam {
  double fx, fy, cx, cy, xf[16];
};

// window (u,v) of the crop camera -> window coordinates in the fisheye source, as float32
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
  if (z < 0) { ox = -1.0; oy = -1.0; }   // depth check, tracker.py:71-73
  mx = (float)ox;
  my = (float)oy;
}

template <bool FROM_LDS>
__device__ inline int fetch(const uint8_t* img, int pitch, int W, int H, const uint8_t* tile, int tx0, int ty0,
                            int tpitch, int y, int x) {
  if (x < 0 || y < 0 || x >= W || y >= H) return 0;   // BORDER_CONSTANT 0
  if (FROM_LDS) return tile[(y - ty0) * tpitch + (x - tx0)];
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

__global__ __launch_bounds__(WARP_THREADS) void warp_kernel(const uint8_t* __restrict__ src_images,
                                                            const int64_t* __restrict__ src_offset,
                                                            const int32_t* __restrict__ src_pitch,
                                                            const double* __restrict__ src_cams,
                                                            const double* __restrict__ dst_cams, int out_h, int out_w,
                                                            int mode, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ int red[4][WARP_THREADS / 64];
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  SrcCam s;
  DstCam d;
  {
    const double* sp = src_cams + (size_t)k * 30;
    s.w = sp[0]; s.h = sp[1]; s.fx = sp[2]; s.fy = sp[3]; s.cx = sp[4]; s.cy = sp[5];
    s.k1 = sp[6]; s.k2 = sp[7]; s.k3 = sp[8]; s.k4 = sp[9]; s.p1 = sp[10]; s.p2 = sp[11]; s.k5 = sp[12]; s.k6 = sp[13];
    for (int i = 0; i < 16; ++i) s.xf[i] = sp[14 + i];
    const double* dp = dst_cams + (size_t)k * 20;
    d.fx = dp[0]; d.fy = dp[1]; d.cx = dp[2]; d.cy = dp[3];
    for (int i = 0; i < 16; ++i) d.xf[i] = dp[4 + i];
  }
  const int W = (int)s.w, H = (int)s.h;
  const int pitch = src_pitch[k];
  const uint8_t* img = src_images + src_offset[k];
  const int npx = out_h * out_w;

  int sxs[WARP_MAX_PX], sys[WARP_MAX_PX];
  float mxs[WARP_MAX_PX], mys[WARP_MAX_PX];
  int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -(1 << 30), by1 = -(1 << 30);
#pragma unroll
  for (int i = 0; i < WARP_MAX_PX; ++i) {
    const int p = tid + i * WARP_THREADS;
    sxs[i] = sys[i] = -32;
    mxs[i] = mys[i] = -1.f;
    if (p < npx) {
      const int v = p / out_w, u = p - v * out_w;
      float mx, my;
      map_point(s, d, u, v, mx, my);
      mxs[i] = mx; mys[i] = my;
      int ix, iy;
      if (mode == UMT_WARP_CV2_FIXED_POINT) {
        const float cx = fminf(fmaxf(mx * 32.0f, -1048576.f), 1048576.f);   // NaN -> lower bound
        const float cy = fminf(fmaxf(my * 32.0f, -1048576.f), 1048576.f);
        sxs[i] = (int)rintf(cx);
        sys[i] = (int)rintf(cy);
        ix = sxs[i] >> 5; iy = sys[i] >> 5;
      } else {
        ix = (int)fmin(fmax(floor((double)mx), -4.0), 1.0e6);
        iy = (int)fmin(fmax(floor((double)my), -4.0), 1.0e6);
      }
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
  __syncthreads();
  bx0 = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
  by0 = min(min(red[1][0], red[1][1]), min(red[1][2], red[1][3]));
  bx1 = max(max(red[2][0], red[2][1]), max(red[2][2], red[2][3]));
  by1 = max(max(red[3][0], red[3][1]), max(red[3][2], red[3][3]));
  bx0 = max(bx0, 0); by0 = max(by0, 0);
  bx1 = min(bx1, W - 1); by1 = min(by1, H - 1);
  const bool empty = bx1 < bx0 || by1 < by0;
  const int tw = empty ? 0 : bx1 - bx0 + 1, th = empty ? 0 : by1 - by0 + 1;
  const int tpitch = (tw + 3) & ~3;
  const bool use_lds = !empty && (int64_t)tpitch * th <= WARP_LDS_BYTES;   // uniform over the workgroup

  if (use_lds) {
    for (int row = tid >> 6; row < th; row += WARP_THREADS / 64) {
      const uint8_t* srow = img + (size_t)(by0 + row) * pitch + bx0;
      uint8_t* drow = lds + row * tpitch;
      for (int x = tid & 63; x < tw; x += 64) drow[x] = srow[x];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) {
      const int p = tid + i * WARP_THREADS;
      if (p < npx)
        out[(size_t)k * npx + p] = sample<true>(mode, img, pitch, W, H, lds, bx0, by0, tpitch, sxs[i], sys[i], mxs[i], mys[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) {
      const int p = tid + i * WARP_THREADS;
      if (p < npx)
        out[(size_t)k * npx + p] = sample<false>(mode, img, pitch, W, H, lds, 0, 0, 0, sxs[i], sys[i], mxs[i], mys[i]);
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
  UMT_REQUIRE(n > 0 && out_h > 0 && out_w > 0, UMT_ERR_INVALID, "warp: n=%d out=%dx%d", n, out_h, out_w);
  UMT_REQUIRE(out_h * out_w <= WARP_MAX_PX * WARP_THREADS, UMT_ERR_UNSUPPORTED,
              "warp: crops larger than %d pixels are not supported (got %dx%d)", WARP_MAX_PX * WARP_THREADS, out_w, out_h);
  UMT_REQUIRE(mode == UMT_WARP_CV2_FIXED_POINT || mode == UMT_WARP_FLOAT, UMT_ERR_INVALID, "warp: mode %d", mode);
  UMT_REQUIRE(src_images && src_offset && src_pitch && src_cams && dst_cams && out, UMT_ERR_INVALID, "warp: null pointer");
  hipLaunchKernelGGL(warp_kernel, dim3(n), dim3(WARP_THREADS), WARP_LDS_BYTES, (hipStream_t)stream, src_images,
                     src_offset, src_pitch, src_cams, dst_cams, out_h, out_w, mode, out);
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
