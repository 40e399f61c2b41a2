// Small device-side linear algebra shared by head.hip and head_seq.hip (4x4 transforms in double precision from
// float32 inputs, rounded once; the feature-transform layer on one 3-D point).
#pragma once
#include <hip/hip_runtime.h>

namespace umt {

__device__ inline void mat4_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      c[i * 4 + j] = s;
    }
}

// Gauss-Jordan with partial pivoting; a singular input yields inf/nan where torch.inverse would raise.
// Returns false when a pivot is exactly zero or not finite (the matrix is singular to working precision).
// Every index is a compile-time constant (the loops unroll) so the tableau stays in registers: the pivot row comes up by
// conditional swaps with each lower row in turn - the row that lands in place is the first of largest magnitude, as with a
// single swap, and where the other rows sit does not enter the arithmetic (bit-identical; see inv4 in cropcam.hip).
__device__ __forceinline__ bool mat4_inv(const double* m, double* inv) {
  bool regular = true;
  double a[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[i][j] = m[i * 4 + j];
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll
  for (int col = 0; col < 4; ++col) {
#pragma unroll
    for (int r = col + 1; r < 4; ++r) {
      const bool sw = fabs(a[r][col]) > fabs(a[col][col]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double t = a[col][j];
        a[col][j] = sw ? a[r][j] : t;
        a[r][j] = sw ? t : a[r][j];
      }
    }
    if (!(fabs(a[col][col]) > 0.0) || !isfinite(a[col][col])) regular = false;
    const double d = 1.0 / a[col][col];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[col][j] *= d;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double f = a[r][col];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[col][j];
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
  return regular;
}

__device__ inline void load_mat4(const float* src, double* dst) {
  for (int i = 0; i < 16; ++i) dst[i] = (double)src[i];
}

// FTL on one point: the reference reshapes (C,6,6) to (3, C/3*36), so channels c, c+C/3, c+2C/3 of one pixel form one
// 3-D point (model_utils.py:89-99); xf = rows of [R|t]
__device__ inline void ftl_point(const float* xf, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = xf[0] * x + xf[1] * y + xf[2] * z + xf[3];
  oy = xf[4] * x + xf[5] * y + xf[6] * z + xf[7];
  oz = xf[8] * x + xf[9] * y + xf[10] * z + xf[11];
}

}  // namespace umt
