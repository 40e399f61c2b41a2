// MPJPE / MPJPA / PCK partial sums on the device (row A15).
// Replaces _compute_metrics and the sums inside aggregate_metrics (load_eval.py:26-89) and the
// threshold counting of PCK_curve (lib/common/metric_utils.py:35-84).  Arithmetic is float64 on
// float32 landmarks, like the reference's float64 numpy arrays.  Two fixed-order stages (per
// workgroup, then one workgroup over the partials) make the result bit-reproducible, so the
// values that RCCL later sums across ranks do not depend on scheduling.
#include "umt_internal.h"

#include <atomic>

namespace umt {

constexpr int MT = 256;
constexpr int MW = UMT_METRIC_WORDS;   // 7 sums + 101 counts

__device__ inline double kp_error(const float* a, const float* b) {
  double s = 0;
  for (int j = 0; j < 21; ++j) {
    const double dx = (double)a[j * 3] - (double)b[j * 3], dy = (double)a[j * 3 + 1] - (double)b[j * 3 + 1],
                 dz = (double)a[j * 3 + 2] - (double)b[j * 3 + 2];
    s += sqrt(dx * dx + dy * dy + dz * dz);
  }
  return s / 21.0;
}

__device__ inline double kp_accel(const float* p0, const float* p1, const float* p2) {
  double s = 0;
  for (int i = 0; i < 21; ++i) {
    double q[3];
    for (int k = 0; k < 3; ++k)
      q[k] = ((double)p0[i * 3 + k] + (double)p2[i * 3 + k]) - 2.0 * (double)p1[i * 3 + k];   // load_eval.py:30
    s += sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  }
  return s / 21.0;
}

__global__ __launch_bounds__(MT) void metric_stage1(const float* __restrict__ gt, const float* __restrict__ pred,
                                                    const uint8_t* __restrict__ valid, int64_t total, int T,
                                                    double* __restrict__ partial) {
  __shared__ double ssum[5][MT];
  __shared__ unsigned hist[102];
  const int tid = threadIdx.x;
  if (tid < 102) hist[tid] = 0;
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * MT + tid;
  double v[5] = {0, 0, 0, 0, 0};   // sum_err, n_err, sum_acc, n_acc, sum_gt_acc
  if (t < total) {
    const int f = (int)(t % T);
    if (valid[t]) {
      const double e = kp_error(gt + t * 63, pred + t * 63);
      v[0] = e; v[1] = 1;
      // errors <= linspace(0,50,101)[i] = 0.5*i  <=>  i >= ceil(2e); 2e is exact in binary
      const double c = ceil(2.0 * e);
      const int bin = c <= 100.0 ? (int)c : 101;
      atomicAdd(&hist[bin < 0 ? 0 : bin], 1u);
      if (f + 2 < T && valid[t + 1] && valid[t + 2]) {   // load_eval.py:35-37
        v[2] = kp_accel(pred + t * 63, pred + (t + 1) * 63, pred + (t + 2) * 63);
        v[3] = 1;
        v[4] = kp_accel(gt + t * 63, gt + (t + 1) * 63, gt + (t + 2) * 63);
      }
    }
  }
  for (int k = 0; k < 5; ++k) ssum[k][tid] = v[k];
  __syncthreads();
  for (int s = MT / 2; s > 0; s >>= 1) {
    if (tid < s)
      for (int k = 0; k < 5; ++k) ssum[k][tid] += ssum[k][tid + s];
    __syncthreads();
  }
  double* o = partial + (size_t)blockIdx.x * MW;
  if (tid < 5) o[tid] = ssum[tid][0];
  if (tid == 5) o[5] = ssum[1][0];                       // n_valid == number of valid hand-frames
  if (tid == 6) {
    const int64_t rem = total - (int64_t)blockIdx.x * MT;
    o[6] = (double)(rem < MT ? rem : MT);                // n_total
  }
  if (tid < 101) o[7 + tid] = (double)hist[tid];         // per-bin counts; made cumulative in stage 2
}

__global__ __launch_bounds__(MT) void metric_stage2(const double* __restrict__ partial, int nblocks,
                                                    double* __restrict__ result) {
  __shared__ double col[MW];
  const int tid = threadIdx.x;
  if (tid < MW) {
    double s = 0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * MW + tid];   // fixed order
    col[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < 7; ++i) result[i] = col[i];
    double run = 0;
    for (int i = 0; i < 101; ++i) { run += col[7 + i]; result[7 + i] = run; }
  }
}

// ---- allocation-free variant (umt_metric_partials_static): the per-workgroup partials of stage 1 live in a
// file-scope device array instead of a caller workspace.  Every GPU the library runs on holds its own copy
// (UMT_METRIC_STATIC_MAX_BLOCKS x 108 doubles = 1.77 MB) until the process ends.  The array's name is only a
// host-side handle; its device address is resolved once per device with hipGetSymbolAddress and kept here.
constexpr int STATIC_MAX_BLOCKS = UMT_METRIC_STATIC_MAX_HAND_FRAMES / MT;
static __device__ __align__(256) double scratch[(size_t)STATIC_MAX_BLOCKS * MW];
constexpr int MAX_DEVICES = 64;
static std::atomic<double*> g_scratch_ptr[MAX_DEVICES];   // zero-initialised; a slot only ever goes null -> address

// device address of `scratch` on the current device (resolved on the first call per device; no allocation)
static int static_scratch(double** out) {
  int dev = -1;
  UMT_CHECK_HIP(hipGetDevice(&dev));
  UMT_REQUIRE(dev >= 0 && dev < MAX_DEVICES, UMT_ERR_UNSUPPORTED, "metrics (static): device index %d >= %d", dev, MAX_DEVICES);
  double* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    UMT_CHECK_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    UMT_REQUIRE(p != nullptr, UMT_ERR_HIP, "metrics (static): hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev].store(p, std::memory_order_release);      // every thread resolves the same address
  }
  *out = p;
  return 0;
}

}  // namespace umt

extern "C" int64_t umt_metric_workspace_bytes(int n_rec, int t) {
  const int64_t total = (int64_t)n_rec * 2 * t;
  const int64_t blocks = (total + umt::MT - 1) / umt::MT;
  return (blocks > 0 ? blocks : 1) * umt::MW * (int64_t)sizeof(double);
}

extern "C" int umt_metric_partials(const float* gt, const float* pred, const uint8_t* valid, int n_rec, int t,
                                   double* partials, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace umt;
  UMT_REQUIRE(n_rec > 0 && t > 0, UMT_ERR_INVALID, "metrics: n_rec=%d t=%d", n_rec, t);
  UMT_REQUIRE(gt && pred && valid && partials && workspace, UMT_ERR_INVALID, "metrics: null pointer");
  UMT_REQUIRE(workspace_bytes >= umt_metric_workspace_bytes(n_rec, t), UMT_ERR_WORKSPACE,
              "metrics: workspace %lld < %lld bytes", (long long)workspace_bytes,
              (long long)umt_metric_workspace_bytes(n_rec, t));
  const int64_t total = (int64_t)n_rec * 2 * t;
  const int blocks = (int)((total + MT - 1) / MT);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(metric_stage1, dim3(blocks), dim3(MT), 0, s, gt, pred, valid, total, t, (double*)workspace);
  hipLaunchKernelGGL(metric_stage2, dim3(1), dim3(MT), 0, s, (const double*)workspace, blocks, partials);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int umt_metric_partials_static(const float* gt, const float* pred, const uint8_t* valid, int n_rec, int t,
                                          double* partials, void* stream) {
  using namespace umt;
  UMT_REQUIRE(n_rec > 0 && t > 0, UMT_ERR_INVALID, "metrics (static): n_rec=%d t=%d", n_rec, t);
  UMT_REQUIRE(gt && pred && valid && partials, UMT_ERR_INVALID, "metrics (static): null pointer");
  const int64_t total = (int64_t)n_rec * 2 * t;
  UMT_REQUIRE(total <= (int64_t)UMT_METRIC_STATIC_MAX_HAND_FRAMES, UMT_ERR_WORKSPACE,
              "metrics (static): %lld hand-frames exceed the static scratch limit of %d; split the call or use "
              "umt_metric_partials with a workspace", (long long)total, UMT_METRIC_STATIC_MAX_HAND_FRAMES);
  const int blocks = (int)((total + MT - 1) / MT);
  double* ws = nullptr;
  UMT_TRY(static_scratch(&ws));
  hipStream_t s = (hipStream_t)stream;
  // stage 1 writes all 108 words of each of its `blocks` rows and stage 2 reads exactly those rows, so what an
  // earlier (larger) call left in the array is never read: nothing has to be cleared
  hipLaunchKernelGGL(metric_stage1, dim3(blocks), dim3(MT), 0, s, gt, pred, valid, total, t, ws);
  hipLaunchKernelGGL(metric_stage2, dim3(1), dim3(MT), 0, s, (const double*)ws, blocks, partials);
  UMT_CHECK_HIP(hipGetLastError());
  return 0;
}
