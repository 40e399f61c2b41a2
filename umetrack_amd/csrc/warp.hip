// Stage 1 of the hot path: resample source images into 96x96 crops.
//
//  * umt_warp_crops     - fisheye (arctan + 6 radial / 2 tangential) -> pinhole crop, replaces
//                         _warp_image + cv2.remap + /255 (lib/tracker/tracker.py:54-80, 210-211,
//                         camera maths lib/common/camera.py:61-94, 106-141, 257-281).
//  * umt_resample_pinhole - pinhole -> pinhole float bilinear, replaces _resample_images_batched
//                         (lib/batched_dataset/data_transform.py:78-144).
//
// One workgroup per band of 32 rows of a crop.  Pass 1 evaluates the coordinate map and keeps the 12 fixed-point
// coordinates per thread in registers while the workgroup reduces their bounding box.  Pass 2
// copies that source rectangle into LDS with coalesced row reads, pass 3 gathers the four taps
// from LDS and writes whole 384-B output rows.  Rectangles that do not fit fall back to
// gathering from global memory.  Algorithmic traffic: 4 source bytes + 4 output bytes per pixel.
//
// The reference builds the map with float64 numpy and casts it to float32; OpenCV then rounds the
// coordinate to 1/32 px.  In UMT_WARP_CV2_FIXED_POINT mode only that integer matters, so pass 1 evaluates
// the float64 map on a grid of 8-pixel spacing only and interpolates it bicubically in float32 in between,
// accepting a pixel where the 1/32-px rounding cannot be affected by the MEASURED interpolation error of its
// cell plus the float32 arithmetic bound (see "interpolated map" below).  The other pixels (about 5 per cent:
// close to a rounding boundary, next to z = 0, not finite) are queued in LDS and re-evaluated in float64 by
// the whole workgroup, densely.  UMT_WARP_CV2_FIXED_POINT_F64 evaluates every pixel in float64 (the tests
// hold the two modes bit-identical); UMT_WARP_FLOAT always uses float64 maps.
#include "umt_internal.h"

namespace umt {

#ifdef UMT_WARP_STAMP      // diagnostic builds only (tools/warp_bench.py): phase time stamps of one workgroup
__device__ unsigned long long g_wstamp[16];
#define WSTAMP(i) do { if (blockIdx.x == 1000 && threadIdx.x == 0) g_wstamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif

constexpr int WARP_THREADS = 256;                // one workgroup = a band of rows of one crop; 4-6 bands share a CU, so a band's
                                                 // barrier-separated phases (latency chains) overlap with other bands' phases
constexpr int WARP_MAX_PX = 12;                 // pixels per thread, as 3 groups of 4 consecutive pixels
constexpr int WARP_BAND_PX = WARP_MAX_PX * WARP_THREADS;   // 3072 pixels per band: 32 rows of a 96-wide crop
constexpr int WARP_LDS_BYTES = 24 * 1024;       // source rectangle budget of a band
#ifndef WARP_WAVES_PER_SIMD
#define WARP_WAVES_PER_SIMD 6                    // of the interpolated-map kernel: six bands per CU is what its 25.7 KB of LDS allows.
#endif                                           // Measured per 16 384 crops: 4 -> 1.109 ms (123 registers), 5 -> 1.034 (96, 15
                                                 // spilled), 6 -> 1.022 (80, 40 spilled): the phases' latency chains need the company
constexpr int WARP_WAVES_F64 = 4;                // the all-float64 kernels (check modes) keep their registers
constexpr int WARP_QCAP = 1024;                 // float64 re-evaluation queue window (10 KB; the rectangle later reuses the space)

struct SrcCam {
  double w, h, fx, fy, cx, cy, k1, k2, k3, k4, p1, p2, k5, k6, xf[16];
};
struct DstCam {
  double fx, fy, cx, cy, xf[16];
};

__device__ inline void quantise_map(float mx, float my, int& sx, int& sy) {
  const float cx = fminf(fmaxf(mx * 32.0f, -1048576.f), 1048576.f);   // NaN -> lower bound
  const float cy = fminf(fmaxf(my * 32.0f, -1048576.f), 1048576.f);
  sx = (int)rintf(cx);
  sy = (int)rintf(cy);
}


// window (u,v) of the crop camera -> window coordinates in the fisheye source, float64; zc = cosine of the angle between
// the ray and the source camera's axis times the ray length (z of the source-eye point)
__device__ inline void map_point_raw(const SrcCam& s, const DstCam& d, int u, int v, double& ox, double& oy, double& zc,
                                     double& plen2) {
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
  ox = dx * s.fx + s.cx;
  oy = dy * s.fy + s.cy;
  zc = z;
  plen2 = x * x + y * y + z * z;
}

// the reference's map value of one pixel, as float32
template <bool DEPTH_CHECK = true>
__device__ inline void map_point(const SrcCam& s, const DstCam& d, int u, int v, float& mx, float& my) {
  double ox, oy, z, pl;
  map_point_raw(s, d, u, v, ox, oy, z, pl);
  if (DEPTH_CHECK && z < 0) { ox = -1.0; oy = -1.0; }   // depth check, tracker.py:71-73
  mx = (float)ox;
  my = (float)oy;
}

// ---- interpolated map (UMT_WARP_CV2_FIXED_POINT, crops up to 96 wide whose camera shares the source camera's centre) ----
// The map is smooth, so its float64 values on a grid of 8-pixel spacing (15 x 7 nodes around a band of 32 rows) determine it between
// the nodes: bicubic Lagrange interpolation on the 4x4 nodes around a cell.  The interpolation error of every cell is
// MEASURED, not assumed: the float64 map is also evaluated at the midpoint of the central interval of EVERY 1-D stencil
// line the band uses - all 7 node rows (84 points) and all 15 node columns (60 points) - (where the error of the
// 1-D cubic along that edge peaks; the error of the tensor-product interpolant inside the cell is the sum of the two 1-D
// errors to leading order), and a pixel's interpolated coordinate is accepted only when its distance to the nearest 1/32-px
// rounding boundary exceeds twice (1.25 x largest horizontal-edge error + largest vertical-edge error over the stencil's grid
// lines) plus the float32 arithmetic bound.  Nodes that are not regular (behind or within 1.1 degrees of the source camera's
// z = 0 plane, not finite, out of range) are NaN and make every cell they touch undecided; undecided pixels go to the
// float64 queue like those of the per-pixel float32 map.  All 16 nodes regular implies the cell is: with a shared centre the
// set {z > 0.02 |p|} is a circular cone of rays, whose section with the crop's image plane is convex.
constexpr int IG = 8;                       // grid spacing in pixels
constexpr int IBH = 32;                     // rows of a band on the interpolated path
constexpr int INX = 96 / IG + 3;            // 15 node columns: u = IG * (nx - 1)
constexpr int INY = IBH / IG + 3;           // 7 node rows of a band: v = v0 + IG * (ny - 1)
constexpr int ICX = 96 / IG, ICY = IBH / IG;   // 12 x 4 cells
constexpr int I_NODES = INX * INY;          // 105
constexpr int IHR = ICY + 3;                // 7 rows of horizontal-edge midpoints: EVERY node row of the band's grid
constexpr int IVC = ICX + 3;                // 15 columns of vertical-edge midpoints: every node column
constexpr int I_HE = IHR * ICX;             // 84 horizontal-edge midpoints (node row r = 0..6, between node columns c+1 and c+2)
constexpr int I_VE = ICY * IVC;             // 60 vertical-edge midpoints (between node rows cj+1 and cj+2, node column c = 0..14)
constexpr int I_EVALS = I_NODES + I_HE + I_VE;   // 249 float64 evaluations per band of 3072 pixels: one per thread
static_assert(I_EVALS <= WARP_THREADS && IBH * 96 == WARP_BAND_PX, "one evaluation per thread, one band per workgroup");
constexpr int I_OFF_DN = WARP_QCAP * 10;             // double2[249]  (the queue owns [0, 10 * QCAP): 2-byte ids, 8-byte results)
constexpr int I_OFF_NF = I_OFF_DN + 16 * 250;        // float2[105] nodes relative to the band's centre node, 1/32 px
constexpr int I_OFF_ERR = I_OFF_NF + 8 * 106;        // float[144] measured errors
constexpr int I_OFF_CELL = I_OFF_ERR + 4 * 144;      // float[48] per-cell bound
constexpr int I_OFF_WT = I_OFF_CELL + 4 * 48;        // float4[8] cubic weights of the 8 sub-positions
static_assert(I_OFF_WT + 128 <= WARP_LDS_BYTES && I_OFF_WT % 16 == 0 && I_OFF_NF % 8 == 0, "interpolation tables");

__device__ inline void interp_eval_pos(int e, int v0, int& u, int& v) {
  if (e < I_NODES) { const int ny = e / INX, nx = e - ny * INX; u = IG * (nx - 1); v = v0 + IG * (ny - 1); }
  else if (e < I_NODES + I_HE) { const int i = e - I_NODES, r = i / ICX, c = i - r * ICX; u = IG * c + IG / 2; v = v0 + IG * (r - 1); }
  else { const int i = e - I_NODES - I_HE, cj = i / IVC, c = i - cj * IVC; u = IG * (c - 1); v = v0 + IG * cj + IG / 2; }
}

__device__ inline float nan_max(float a, float b) { return (a > b || a != a) ? a : b; }   // NaN if either is

// pixel handled by slot i of thread tid: groups of 4 consecutive pixels, groups dealt round-robin to the threads
__device__ __forceinline__ int warp_pixel(int tid, int i) { return 4 * (tid + (i >> 2) * WARP_THREADS) + (i & 3); }

// LDS tile: row 0 and the last row are zero, every row starts and ends with a zero dword; source byte (x, y) of the
// rectangle lives at (y - ty0) * tpitch + (x - tx0) with (tx0, ty0) the rectangle origin shifted by the border
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
// SQ96: the crop is 96 x 96 (the network's input: three whole bands of 32 rows), so every band is full and the divisions by
// the crop width are by a constant
template <int MODE, bool DEPTH_CHECK = true, bool SQ96 = false>
__global__ __launch_bounds__(WARP_THREADS, MODE == UMT_WARP_CV2_FIXED_POINT ? WARP_WAVES_PER_SIMD : WARP_WAVES_F64) void warp_kernel(const uint8_t* __restrict__ src_images,
                                                            const int64_t* __restrict__ src_offset,
                                                            const int32_t* __restrict__ src_pitch,
                                                            const double* __restrict__ src_cams,
                                                            const double* __restrict__ dst_cams, int out_h_, int out_w_,
                                                            int band_h_, int nbands_, float* __restrict__ out) {
  const int out_h = SQ96 ? 96 : out_h_, out_w = SQ96 ? 96 : out_w_, band_h = SQ96 ? IBH : band_h_, nbands = SQ96 ? 96 / IBH : nbands_;
  constexpr bool CV2 = MODE != UMT_WARP_FLOAT;
  constexpr int smode = CV2 ? UMT_WARP_CV2_FIXED_POINT : UMT_WARP_FLOAT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ int red[4][WARP_THREADS / 64];
  __shared__ int qcount, relmax_bits;
  __shared__ float lut255[256];
  const int k = blockIdx.x / nbands, v0 = (blockIdx.x - k * nbands) * band_h;     // crop, first row of this band
  const int tid = threadIdx.x;
  const int W = (int)src_cams[(size_t)k * 30], H = (int)src_cams[(size_t)k * 30 + 1];
  const int pitch = src_pitch[k];
  const uint8_t* img = src_images + src_offset[k];
  const int npx = min(band_h, out_h - v0) * out_w;         // pixels of this band; pixel p is (p % out_w, v0 + p / out_w)
  WSTAMP(0);
  if (tid < 256) lut255[tid] = (float)tid / 255.0f;      // read after the barriers below

  int sxs[WARP_MAX_PX], sys[WARP_MAX_PX];
  float mxs[CV2 ? 1 : WARP_MAX_PX], mys[CV2 ? 1 : WARP_MAX_PX];

  if (MODE == UMT_WARP_CV2_FIXED_POINT) {
    // interpolated (or per-pixel float32) map; pixels it cannot decide go to the queue: pixel ids, then their float64 results
    uint16_t* qpix = reinterpret_cast<uint16_t*>(lds);
    int* qres = reinterpret_cast<int*>(lds + WARP_QCAP * 2);
    double2* dn = reinterpret_cast<double2*>(lds + I_OFF_DN);
    float2* nf = reinterpret_cast<float2*>(lds + I_OFF_NF);
    float* errs = reinterpret_cast<float*>(lds + I_OFF_ERR);
    float* cell_e = reinterpret_cast<float*>(lds + I_OFF_CELL);
    float4* wtab = reinterpret_cast<float4*>(lds + I_OFF_WT);
    if (tid == 0) { qcount = 0; relmax_bits = 0; }
    bool interp;
    {
      SrcCam s;
      DstCam d;
      load_cams(src_cams, dst_cams, k, s, d);
      const double t0 = d.xf[3] - s.xf[3], t1 = d.xf[7] - s.xf[7], t2 = d.xf[11] - s.xf[11];
      interp = out_w <= 96 && band_h == IBH && (out_w & 3) == 0 && t0 * t0 + t1 * t1 + t2 * t2 <= 1e-12;   // uniform
#ifdef UMT_WARP_X_NOINTERP
      interp = false;
#endif
      if (interp) {
        if (tid < I_EVALS) {
          const int e = tid;
          int u, v;
          interp_eval_pos(e, v0, u, v);
          double ox, oy, z, pl;
          map_point_raw(s, d, u, v, ox, oy, z, pl);
          const bool regular = z > 0 && z * z > 0.0004 * pl && fabs(ox) < 30000.0 && fabs(oy) < 30000.0;   // false for NaN
          const double nan = __longlong_as_double(0x7ff8000000000000ll);
          dn[e] = regular ? make_double2(ox * 32.0, oy * 32.0) : make_double2(nan, nan);
        }
        if (tid < IG) {
          const float t = (float)tid / IG;
          wtab[tid] = make_float4(-t * (t - 1) * (t - 2) / 6, (t + 1) * (t - 1) * (t - 2) / 2, -(t + 1) * t * (t - 2) / 2,
                                  (t + 1) * t * (t - 1) / 6);
        }
      }
    }
    __syncthreads();
    WSTAMP(1);
    unsigned queued = 0;
    if (interp) {
      const double2 ref = dn[(INY / 2) * INX + INX / 2];              // the node at pixel (48, v0 + 16)
      const double rix = floor(ref.x), riy = floor(ref.y);
      if (tid < I_NODES) {
        const double2 nv = dn[tid];
        const float rx = (float)(nv.x - ref.x), ry = (float)(nv.y - ref.y);
        nf[tid] = make_float2(rx, ry);
        const float m = fmaxf(fabsf(rx), fabsf(ry));
        if (rx == rx && ry == ry) atomicMax(&relmax_bits, __float_as_int(m));
      }
      if (tid < I_HE + I_VE) {
        // measured interpolation error at the check points (cubic weights at the midpoint: -1/16, 9/16, 9/16, -1/16)
        double ix, iy;
        if (tid < I_HE) {
          const int r = tid / ICX, c = tid - r * ICX;
          const double2* row = dn + r * INX + c;
          ix = (9.0 * (row[1].x + row[2].x) - (row[0].x + row[3].x)) * 0.0625;
          iy = (9.0 * (row[1].y + row[2].y) - (row[0].y + row[3].y)) * 0.0625;
        } else {
          const int i = tid - I_HE, cj = i / IVC, c = i - cj * IVC;
          const double2* col = dn + cj * INX + c;
          ix = (9.0 * (col[INX].x + col[2 * INX].x) - (col[0].x + col[3 * INX].x)) * 0.0625;
          iy = (9.0 * (col[INX].y + col[2 * INX].y) - (col[0].y + col[3 * INX].y)) * 0.0625;
        }
        const double2 ex = dn[I_NODES + tid];
        const float ea = (float)fabs(ix - ex.x), eb = (float)fabs(iy - ex.y);
        errs[tid] = nan_max(ea, eb) * 1.001f;
      }
      __syncthreads();
      if (tid < ICX * ICY) {
        const int cj = tid / ICX, ci = tid - cj * ICX;
        // f - Iu Iv f = (f - Iv f) + Iv (f - Iu f): the second term is the v-interpolant (Lebesgue constant 1.25) of the
        // u-errors on the stencil's FOUR node rows, so the horizontal-edge error is taken over the four node rows of the
        // stencil (node rows cj .. cj+3 of this band, each measured at its own midpoint), and likewise the vertical-edge
        // error over its four node columns ci .. ci+3: no stencil line is extrapolated from a neighbour's measurement
        float eh = 0.f, ev = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          eh = nan_max(eh, errs[(cj + d) * ICX + ci]);
          ev = nan_max(ev, errs[I_HE + cj * IVC + ci + d]);
        }
        // float32 side of the bound, worst case, with u = 2^-24 and R = relmax (largest node value relative to the centre
        // node): a node is rounded to float32 (u R); the column combination is 4 roundings of partial sums <= 1.25 R (the
        // cubic weights sum to <= 1.25 in absolute value), 3 u relative error of each weight, and the node errors: 10 u R;
        // the row combination adds 4 roundings of sums <= 1.5625 R, the weights again and 1.25 x the column error:
        // 23.5 u R = 1.4e-6 R in all, plus the final addition (u (R + 1))
        // ... plus the reference's float32 cast of the coordinate (up to 2^-24 = 5.96e-8 of its magnitude), bounded for
        // the whole band by the centre node's magnitude plus R + 1; stored as the THRESHOLD a pixel's distance from its
        // rounded value has to stay under, 0.5 - bound (NaN for an irregular cell: every comparison with it fails), and
        // as "nothing passes" when the band's coordinates can leave the range in which the fixed-point conversion is exact
        const float rmax = __int_as_float(relmax_bits);
        const float amax = fmaxf(fabsf((float)ref.x), fabsf((float)ref.y)) + rmax + 1.0f;
        const float bound = 2.0f * (1.25f * eh + ev) + 1.6e-6f * rmax + 0.001f + 6.2e-8f * amax;
        cell_e[tid] = amax < 960000.f ? 0.5f - bound : -1.0f;
      }
      __syncthreads();
      WSTAMP(2);
      const float fxr = (float)(ref.x - rix), fyr = (float)(ref.y - riy);
      const int irx = (int)rix, iry = (int)riy;
#pragma unroll
      for (int gq = 0; gq < WARP_MAX_PX / 4; ++gq) {
        const int p = warp_pixel(tid, 4 * gq);
#pragma unroll
        for (int j = 0; j < 4; ++j) sxs[4 * gq + j] = sys[4 * gq + j] = -32;
        if (p < npx) {
          const int gv = p / out_w, gu = p - gv * out_w;        // band-local row; gu is a multiple of 4: the group lies in one cell
          const int ci = gu >> 3, cj = gv >> 3;
          // (x, y) pairs through packed fused multiply-adds: 16 for the column combination, 4 per pixel; the chain order is
          // fixed here because the error bound in cell_e counts its roundings
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const float4 wy = wtab[gv & 7];
          const f32x2* nb = reinterpret_cast<const f32x2*>(nf) + cj * INX + ci;
          f32x2 cc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            f32x2 t = nb[c] * (f32x2){wy.x, wy.x};
            t = __builtin_elementwise_fma((f32x2){wy.y, wy.y}, nb[INX + c], t);
            t = __builtin_elementwise_fma((f32x2){wy.z, wy.z}, nb[2 * INX + c], t);
            cc[c] = __builtin_elementwise_fma((f32x2){wy.w, wy.w}, nb[3 * INX + c], t);
          }
          const float thr = cell_e[cj * ICX + ci];
          const f32x2 fr = {fxr, fyr};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 4 * gq + j;
            const float4 wx = wtab[(gu & 7) + j];
            f32x2 t = cc[0] * (f32x2){wx.x, wx.x};
            t = __builtin_elementwise_fma((f32x2){wx.y, wx.y}, cc[1], t);
            t = __builtin_elementwise_fma((f32x2){wx.z, wx.z}, cc[2], t);
            t = __builtin_elementwise_fma((f32x2){wx.w, wx.w}, cc[3], t) + fr;
            const float tx = t.x, ty = t.y;
            const float rx = rintf(tx), ry = rintf(ty);
            // accepted when both coordinates stay clear of a rounding boundary by the cell's bound.  The comparison is
            // false for a NaN coordinate: fmaxf would drop a NaN operand, so the two distances are compared one by one
            const bool ok = (int)(fabsf(tx - rx) < thr) & (int)(fabsf(ty - ry) < thr);
            sxs[i] = irx + (int)rx;                 // replaced by the queue position below when not accepted
            sys[i] = iry + (int)ry;
#ifndef UMT_WARP_X_NOQUEUE
            queued |= ok ? 0u : 1u << i;
#endif
          }
        }
      }
    } else {
      // other crop sizes, or a crop camera that does not share the source camera's centre: every pixel in float64
      // (more than the queue holds, so each thread evaluates its own pixels below)
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i) {
        sxs[i] = sys[i] = -32;
        if (warp_pixel(tid, i) < npx) queued |= 1u << i;
      }
    }
    // queue slots: one LDS atomic per wave (a per-pixel atomic on one address serialises the whole workgroup)
    {
      const int cnt = __popc(queued), lane = tid & 63;
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      int base = 0;
      if (lane == 63 && incl > 0) base = atomicAdd(&qcount, incl);
      base = __shfl(base, 63);
      int pos = base + incl - cnt;
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i)
        if ((queued >> i) & 1u) {
          if (pos < WARP_QCAP) qpix[pos] = (uint16_t)warp_pixel(tid, i);
          sxs[i] = pos++;
        }
    }
    __syncthreads();
    WSTAMP(3);
    const int nq = qcount;
    SrcCam s;
    DstCam d;
    if (nq > 0) load_cams(src_cams, dst_cams, k, s, d);
    // the queue in windows of WARP_QCAP entries (one window unless the crop is degenerate): a single float64 code site
    for (int w0 = 0; w0 < nq; w0 += WARP_QCAP) {          // uniform
      if (w0 > 0) {
#pragma unroll
        for (int i = 0; i < WARP_MAX_PX; ++i)
          if (((queued >> i) & 1u) && sxs[i] < w0 + WARP_QCAP) qpix[sxs[i] - w0] = (uint16_t)warp_pixel(tid, i);
        __syncthreads();     // also: the previous window's results have been read
      }
      const int n = min(nq - w0, WARP_QCAP);
      for (int q = tid; q < n; q += WARP_THREADS) {
        const int p = qpix[q];
        const int vl = p / out_w, u = p - vl * out_w;
        float mx, my;
        map_point(s, d, u, v0 + vl, mx, my);
        int sx, sy;
        quantise_map(mx, my, sx, sy);
        qres[2 * q] = sx;
        qres[2 * q + 1] = sy;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i)
        if (((queued >> i) & 1u) && sxs[i] < w0 + WARP_QCAP) {    // positions below w0 were resolved (bit cleared) earlier
          const int pos = sxs[i] - w0;
          sxs[i] = qres[2 * pos];
          sys[i] = qres[2 * pos + 1];
          queued &= ~(1u << i);
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
        const int vl = p / out_w, u = p - vl * out_w;
        float mx, my;
        map_point<DEPTH_CHECK>(s, d, u, v0 + vl, mx, my);
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

  WSTAMP(4);
  int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -(1 << 30), by1 = -(1 << 30);
#pragma unroll
  for (int i = 0; i < WARP_MAX_PX; ++i) {
    const int ix = CV2 ? sxs[i] >> 5 : sxs[i], iy = CV2 ? sys[i] >> 5 : sys[i];
    // The box of ALL the band's footprints, clamped to the image below: a superset of the footprints that touch the image
    // (a pixel whose footprint lies outside only pulls the box to the image border on its side - where, the map being
    // continuous, the band's other pixels reach anyway; degenerate crops get a larger tile or the global-memory gather).
    // Four min / max per pixel instead of an inside test and four selects.
    const bool mine = SQ96 || warp_pixel(tid, i) < npx;
    bx0 = min(bx0, mine ? ix : 1 << 30); bx1 = max(bx1, mine ? ix + 1 : -(1 << 30));
    by0 = min(by0, mine ? iy : 1 << 30); by1 = max(by1, mine ? iy + 1 : -(1 << 30));
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
  WSTAMP(5);
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
  // LDS tile = the rectangle with a zero border: rows 0 and th + 1 are zero, every row is [zero dword][dwords covering
  // source bytes bx0 - mis .. bx0 + tw, bytes outside the rectangle masked to zero][zero dword].  A tap outside the image
  // but next to the rectangle then reads 0 (BORDER_CONSTANT) with no test of its own.
  const bool aligned = (pitch & 3) == 0;           // then every row starts at the same offset within a dword
  const int mis = aligned ? (int)((reinterpret_cast<uintptr_t>(img) + (size_t)by0 * pitch + bx0) & 3) : 0;
  const int tpd = ((mis + tw + 3) >> 2) + 2;       // dwords per LDS row
  const int tpitch = tpd * 4;
  const bool use_lds = !empty && (int64_t)tpitch * (th + 2) <= WARP_LDS_BYTES;   // uniform over the workgroup

  float res[WARP_MAX_PX];
#ifdef UMT_WARP_X_NOGATHER
  if (use_lds || !use_lds) {
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i) res[i] = (float)(sxs[i] + sys[i] + bx0 + by1);
  } else
#endif
  if (use_lds) {
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds);
    // border dwords load a clamped in-rectangle address and store 0; the unaligned form walks the tile's dwords flat, row =
    // idx / tpd through a float reciprocal (exact: idx < 2^14)
    const float inv_tpd = 1.0f / (float)tpd;
    const int total = tpd * (th + 2);
    const uint8_t* g0 = img + (size_t)by0 * pitch + bx0 - mis;
    if (aligned) {
      // a wave per tile row (rows dealt round-robin to the four waves), a lane per dword column: everything about the column
      // (its byte offset, the mask that cuts the partial dwords at the rectangle's ends and the border columns to zero) is
      // computed once, the row's base address is scalar, and a row costs a load, an AND and an LDS store per lane - the
      // flat index form spent ~40 vector instructions per dword on its row / column split, clamps and masks
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
      for (int c0 = 0; c0 < tpd; c0 += 64) {                   // one trip unless the rectangle is wider than 248 pixels
        const int c = c0 + ln;
        const int b0 = 4 * (min(max(c, 1), tpd - 2) - 1);      // byte offset of this dword from the row's aligned start
        const int start = mis - b0, end = mis + tw - b0;       // rectangle bytes are [start, end) of the dword
        uint32_t mask = (c >= 1 && c < tpd - 1) ? 0xffffffffu : 0u;
        if (start > 0) mask &= 0xffffffffu << (8 * start);
        if (end < 4) mask &= (1u << (8 * end)) - 1u;
        const bool col = c < tpd;
        constexpr int NW = WARP_THREADS / 64, RB = 16;          // a wave's rows in batches of RB loads in flight
        for (int r0 = wv; r0 < th + 2; r0 += NW * RB) {
          uint32_t v[RB];
#pragma unroll
          for (int k = 0; k < RB; ++k) {                         // rows past the tile re-read its last row (not stored)
            const uint8_t* grow = g0 + (size_t)(min(max(r0 + NW * k, 1), th) - 1) * pitch;     // scalar
            v[k] = *reinterpret_cast<const uint32_t*>(grow + b0);
          }
#pragma unroll
          for (int k = 0; k < RB; ++k) {
            const int row = r0 + NW * k;
            const uint32_t rmask = (row >= 1 && row <= th) ? mask : 0u;               // the zero rows above and below
            if (col && row < th + 2) lds32[row * tpd + c] = v[k] & rmask;
          }
        }
      }
    } else {
      for (int idx = tid; idx < total; idx += WARP_THREADS) {
        const int row = (int)(((float)idx + 0.5f) * inv_tpd), c = idx - row * tpd;
        uint32_t v = 0;
        if (row >= 1 && row <= th && c >= 1 && c < tpd - 1) {
          const uint8_t* g = g0 + (size_t)(row - 1) * pitch;
          const int b0 = 4 * (c - 1);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (b0 + j < tw) v |= (uint32_t)g[b0 + j] << (8 * j);
        }
        lds32[idx] = v;
      }
    }
    __syncthreads();
    WSTAMP(6);
    if constexpr (CV2) {
      // [external] OpenCV 8-bit INTER_LINEAR: 1/32-px coordinates, 15-bit weights, round to nearest
      const int cst = (1 - by0) * tpitch + 4 + mis - bx0;
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i) {
        const int ix = sxs[i] >> 5, iy = sys[i] >> 5, fx = sxs[i] & 31, fy = sys[i] & 31;
        const bool inside = (int)((unsigned)(ix + 1) <= (unsigned)W) & (int)((unsigned)(iy + 1) <= (unsigned)H) &
                            (int)(warp_pixel(tid, i) < npx);      // footprint touches the image; slots beyond the band read byte 0
        const int idx = inside ? iy * tpitch + ix + cst : 0;       // byte 0: its four taps are border zeros
        const int a = lds[idx], b = lds[idx + 1], c = lds[idx + tpitch], d = lds[idx + tpitch + 1];
        const int acc = (a * ((32 - fy) * (32 - fx)) + b * ((32 - fy) * fx) + c * (fy * (32 - fx)) + d * (fy * fx)) * 32;
        res[i] = lut255[(acc + (1 << 14)) >> 15];                 // tracker.py:211, v / 255 as a correctly rounded float32
      }
    } else {
#pragma unroll
      for (int i = 0; i < WARP_MAX_PX; ++i)
        res[i] = sample<true>(smode, img, pitch, W, H, lds, bx0 - 4 - mis, by0 - 1, tpitch, sxs[i], sys[i], mxs[i], mys[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < WARP_MAX_PX; ++i)
      res[i] = warp_pixel(tid, i) < npx
                   ? sample<false>(smode, img, pitch, W, H, lds, 0, 0, 0, sxs[i], sys[i], CV2 ? 0.f : mxs[i], CV2 ? 0.f : mys[i])
                   : 0.f;
  }
  WSTAMP(7);
  float* obase = out + (size_t)k * out_h * out_w + (size_t)v0 * out_w;
  const bool vec = (((out_h * out_w) | (band_h * out_w) | npx) & 3) == 0;   // then every group of 4 is whole and 16-byte aligned
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
  WSTAMP(8);
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
  UMT_REQUIRE(out_w <= WARP_BAND_PX, UMT_ERR_UNSUPPORTED, "warp: crops wider than %d pixels are not supported (got %d)",
              WARP_BAND_PX, out_w);
  // a workgroup takes a band of rows: 32 for crops up to 96 wide (the interpolated map's band), else what fits its registers
  const int band_h = out_w <= 96 ? IBH : WARP_BAND_PX / out_w;
  const int nbands = (out_h + band_h - 1) / band_h;
  UMT_REQUIRE((int64_t)n * nbands < ((int64_t)1 << 31), UMT_ERR_INVALID, "warp: too many crops (%d x %d bands)", n, nbands);
  const bool depth_check = (mode & UMT_WARP_NO_DEPTH_CHECK) == 0;
  mode &= ~UMT_WARP_NO_DEPTH_CHECK;
  UMT_REQUIRE(mode == UMT_WARP_CV2_FIXED_POINT || mode == UMT_WARP_FLOAT || mode == UMT_WARP_CV2_FIXED_POINT_F64,
              UMT_ERR_INVALID, "warp: mode %d", mode);
  UMT_REQUIRE(src_images && src_offset && src_pitch && src_cams && dst_cams && out, UMT_ERR_INVALID, "warp: null pointer");
  // depth_check=False (tracker.py:59, never used by the reference's own callers) runs every pixel in float64
  if (!depth_check && mode == UMT_WARP_CV2_FIXED_POINT) mode = UMT_WARP_CV2_FIXED_POINT_F64;
  const bool sq96 = out_h == 96 && out_w == 96;
  auto kern = mode == UMT_WARP_CV2_FIXED_POINT ? (sq96 ? warp_kernel<UMT_WARP_CV2_FIXED_POINT, true, true> : warp_kernel<UMT_WARP_CV2_FIXED_POINT>)
              : mode == UMT_WARP_FLOAT         ? (depth_check ? warp_kernel<UMT_WARP_FLOAT> : warp_kernel<UMT_WARP_FLOAT, false>)
              : depth_check                    ? warp_kernel<UMT_WARP_CV2_FIXED_POINT_F64>
                                               : warp_kernel<UMT_WARP_CV2_FIXED_POINT_F64, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n * nbands)), dim3(WARP_THREADS), WARP_LDS_BYTES, (hipStream_t)stream, src_images,
                     src_offset, src_pitch, src_cams, dst_cams, out_h, out_w, band_h, nbands, out);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef UMT_WARP_STAMP
extern "C" int umt_warp_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(umt::g_wstamp), 16 * sizeof(unsigned long long));
}
#endif

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
