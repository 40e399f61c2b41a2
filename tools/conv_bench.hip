// Stand-alone timing harness for the convolution kernels (experiments only, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 [-DUMT_X_...=1] tools/conv_bench.hip -o /tmp/conv_bench && /tmp/conv_bench [crops] [f32|bf16|both]
#include "../umetrack_amd/csrc/conv_igemm.hip"
#include "../umetrack_amd/csrc/conv_bf16.hip"
#include "../umetrack_amd/csrc/conv_thin.hip"
#include "../umetrack_amd/csrc/conv_x3.hip"
#include "../umetrack_amd/csrc/conv_chain.hip"
#include "../umetrack_amd/csrc/conv_small.hip"
#include "../umetrack_amd/csrc/stage1_depth.hip"
#include <vector>
#include <stdio.h>
#include <string.h>
#include <math.h>

namespace umt {
static thread_local char g_err2[512];
int set_error(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err2, sizeof(g_err2), fmt, ap); va_end(ap);
  fprintf(stderr, "error: %s\n", g_err2);
  return code;
}
}  // namespace umt

int main(int argc, char** argv) {
  using namespace umt;
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  const char* which = argc > 2 ? argv[2] : "both";
  struct Shape { int cin, cout, hw, taps, stride, res; };
  const Shape shapes[] = {{32, 32, 48, 9, 1, 0}, {32, 32, 48, 9, 1, 1}, {64, 64, 24, 9, 1, 0}, {64, 64, 24, 9, 1, 1},
                          {128, 128, 12, 9, 1, 0}, {128, 128, 12, 9, 1, 1}, {256, 256, 6, 9, 1, 0}, {256, 256, 6, 9, 1, 1},
                          {32, 64, 48, 9, 2, 0}, {64, 128, 24, 9, 2, 0}, {128, 256, 12, 9, 2, 0}, {32, 64, 48, 1, 2, 0},
                          {128, 256, 12, 1, 2, 0}};
  const int only = argc > 3 ? atoi(argv[3]) : -1;      // index into shapes[], -1 = all
  if (conv_prepare_device() != 0) return 1;            // dynamic-LDS limits (the library does this at umt_model_finalize)


  // fused stride-2 conv1 + down-sample: split-operand kernel vs the fp32 fused kernel
  if (!strcmp(which, "x3s2")) {
    const int cfgs[2][3] = {{64, 128, 24}, {128, 256, 12}};
    auto rne = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto up = [](uint16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
    auto pack = [&](int taps, int cin, int cout, const float* wl, std::vector<float>& wp, std::vector<uint16_t>& wp3) {
      wp.assign((size_t)taps * cin * cout, 0.f); wp3.assign((size_t)taps * cin * cout * 3, 0);
      for (int tap = 0; tap < taps; ++tap)
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co) {
            const float v = wl[((size_t)tap * cin + ci) * cout + co];
            wp[((((size_t)tap * (cin / 8) + ci / 8) * 2 + (ci % 8) / 4) * cout + co) * 4 + ci % 4] = v;
            float r = v;
            for (int pl = 0; pl < 3; ++pl) {
              const uint16_t b = rne(r);
              r -= up(b);
              wp3[(((((size_t)tap * (cin / 16) + ci / 16) * 3 + pl) * 2 + (ci % 16) / 8) * cout + co) * 8 + ci % 8] = b;
            }
          }
    };
    for (auto& cf : cfgs) {
      const int cin = cf[0], cout = cf[1], hw = cf[2], ho = hw / 2;
      ConvLayer C, D;
      C.taps = 9; C.cin = cin; C.cout = cout; C.cin_s = cin; C.cout_s = cout; C.cout_w = cout; C.stride = 2; C.relu = 1;
      D = C; D.taps = 1; D.relu = 0;
      const size_t in_n = (size_t)n * hw * hw * cin, out_n = (size_t)n * ho * ho * cout, w_n = (size_t)9 * cin * cout, wd_n = (size_t)cin * cout;
      float *in, *o1, *o2, *d1, *d2, *w, *wd, *vec; uint16_t *w3, *wd3;
      hipMalloc(&in, in_n * 4); hipMalloc(&o1, out_n * 4); hipMalloc(&o2, out_n * 4); hipMalloc(&d1, out_n * 4); hipMalloc(&d2, out_n * 4);
      hipMalloc(&w, w_n * 4); hipMalloc(&wd, wd_n * 4); hipMalloc(&w3, w_n * 6); hipMalloc(&wd3, wd_n * 6); hipMalloc(&vec, 3 * cout * 4);
      std::vector<float> h(in_n > w_n ? in_n : w_n);
      unsigned s = 999;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffffff) / 16777216.0f - 0.45f; }
      hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice);
      std::vector<float> wl(w_n), wdl(wd_n), wp, wdp; std::vector<uint16_t> wp3, wdp3;
      for (size_t i = 0; i < w_n; ++i) wl[i] = h[i + 5] * 0.1f;
      for (size_t i = 0; i < wd_n; ++i) wdl[i] = h[i + 77777] * 0.2f;
      pack(9, cin, cout, wl.data(), wp, wp3); pack(1, cin, cout, wdl.data(), wdp, wdp3);
      hipMemcpy(w, wp.data(), w_n * 4, hipMemcpyHostToDevice); hipMemcpy(w3, wp3.data(), w_n * 6, hipMemcpyHostToDevice);
      hipMemcpy(wd, wdp.data(), wd_n * 4, hipMemcpyHostToDevice); hipMemcpy(wd3, wdp3.data(), wd_n * 6, hipMemcpyHostToDevice);
      std::vector<float> v3(3 * cout, 0.f);
      for (int i = 0; i < cout; ++i) { v3[i] = 0.01f * (i % 7); v3[cout + i] = 1.f + 0.01f * (i % 5); v3[2 * cout + i] = -0.02f * (i % 3); }
      hipMemcpy(vec, v3.data(), v3.size() * 4, hipMemcpyHostToDevice);
      C.w = w; C.wx3 = w3; D.w = wd; D.wx3 = wd3; C.bias = D.bias = vec; C.scale = D.scale = vec + cout; C.shift = D.shift = vec + 2 * cout;
      if (launch_conv_s2_fused(C, &D, in, n, hw, hw, o1, d1, 0)) return 1;
      hipMemset(o2, 0xff, out_n * 4); hipMemset(d2, 0xff, out_n * 4);
      if (launch_conv_s2_fused_x3(C, &D, in, n, hw, hw, o2, d2, 0)) return 1;
      if (hipDeviceSynchronize() != hipSuccess) { printf("x3s2 kernel failed\n"); return 1; }
      std::vector<float> a(out_n), b(out_n);
      double e1 = 0, m1 = 0, e2 = 0, m2 = 0;
      hipMemcpy(a.data(), o1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, out_n * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out_n; ++i) { e1 = fmax(e1, fabs((double)a[i] - b[i])); m1 = fmax(m1, fabs((double)a[i])); }
      hipMemcpy(a.data(), d1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, out_n * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out_n; ++i) { e2 = fmax(e2, fabs((double)a[i] - b[i])); m2 = fmax(m2, fabs((double)a[i])); }
      hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
      float ms_f, ms_x;
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv_s2_fused(C, &D, in, n, hw, hw, o1, d1, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_f, e0, e1v);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv_s2_fused_x3(C, &D, in, n, hw, hw, o2, d2, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_x, e0, e1v);
      const double fl = 2.0 * cin * cout * 10 * ho * ho * n;
      printf("x3s2 %3d->%3d hw %2d: conv1 max|diff| %.3g (max %.3g), ds max|diff| %.3g (max %.3g); fp32 fused %.1f us (%.1f TF), x3 %.1f us (%.1f TF)\n",
             cin, cout, hw, e1, m1, e2, m2, ms_f / 5 * 1e3, fl / (ms_f / 5) / 1e9, ms_x / 5 * 1e3, fl / (ms_x / 5) / 1e9);
      hipFree(in); hipFree(o1); hipFree(o2); hipFree(d1); hipFree(d2); hipFree(w); hipFree(wd); hipFree(w3); hipFree(wd3); hipFree(vec);
    }
    return 0;
  }
  // fp32 by three-way bf16 splitting (conv_x3.hip) vs the fp32 MFMA kernel, same weights and inputs
  if (!strcmp(which, "x3")) {
    const int cfgs[4][3] = {{32, 32, 48}, {64, 64, 24}, {128, 128, 12}, {256, 256, 6}};
    for (auto& cf : cfgs) {
      for (int withres = 0; withres < 2; ++withres) {
      const int cin = cf[0], cout = cf[1], hw = cf[2];
      ConvLayer L;
      L.taps = 9; L.cin = cin; L.cout = cout; L.cin_s = cin; L.cout_s = cout; L.cout_w = cout; L.stride = 1; L.relu = 1;
      const size_t in_n = (size_t)n * hw * hw * cin, out_n = (size_t)n * hw * hw * cout, w_n = (size_t)9 * cin * cout;
      float *in, *o1, *o2, *res, *w, *vec; uint16_t* w3;
      hipMalloc(&in, in_n * 4); hipMalloc(&o1, out_n * 4); hipMalloc(&o2, out_n * 4); hipMalloc(&res, out_n * 4);
      hipMalloc(&w, w_n * 4); hipMalloc(&w3, w_n * 6); hipMalloc(&vec, 3 * cout * 4);
      std::vector<float> h(in_n > w_n ? in_n : w_n);
      unsigned s = 4242;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffffff) / 16777216.0f - 0.45f; }
      hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice);
      hipMemcpy(res, h.data() + 17, (out_n < h.size() - 17 ? out_n : h.size() - 17) * 4, hipMemcpyHostToDevice);
      // logical weights W[tap][ci][co] (scaled like a trained layer) -> both packings
      std::vector<float> wl(w_n), wp(w_n);
      for (size_t i = 0; i < w_n; ++i) wl[i] = h[i + 5] * 0.1f;
      std::vector<uint16_t> wp3(w_n * 3);
      auto rne = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
      auto up = [](uint16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
      for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co) {
            const float v = wl[((size_t)tap * cin + ci) * cout + co];
            wp[((((size_t)tap * (cin / 8) + ci / 8) * 2 + (ci % 8) / 4) * cout + co) * 4 + ci % 4] = v;
            float r = v;
            for (int pl = 0; pl < 3; ++pl) {
              const uint16_t b = rne(r);
              r -= up(b);
              wp3[(((((size_t)tap * (cin / 16) + ci / 16) * 3 + pl) * 2 + (ci % 16) / 8) * cout + co) * 8 + ci % 8] = b;
            }
          }
      hipMemcpy(w, wp.data(), w_n * 4, hipMemcpyHostToDevice);
      hipMemcpy(w3, wp3.data(), w_n * 6, hipMemcpyHostToDevice);
      std::vector<float> v3(3 * cout, 0.f);
      for (int i = 0; i < cout; ++i) { v3[i] = 0.01f * (i % 7); v3[cout + i] = 1.f + 0.01f * (i % 5); v3[2 * cout + i] = -0.02f * (i % 3); }
      hipMemcpy(vec, v3.data(), v3.size() * 4, hipMemcpyHostToDevice);
      L.w = w; L.wx3 = w3; L.bias = vec; L.scale = vec + cout; L.shift = vec + 2 * cout;
      const float* rp = withres ? res : nullptr;
      launch_conv(L, in, n, hw, hw, rp, o1, 0);
      hipMemset(o2, 0xff, out_n * 4);
      if (launch_conv_x3(L, in, n, hw, hw, rp, o2, 0)) return 1;
      if (hipDeviceSynchronize() != hipSuccess) { printf("x3 kernel failed\n"); return 1; }
      const size_t chk = out_n < ((size_t)1 << 24) ? out_n : ((size_t)1 << 24);
      std::vector<float> a(chk), b(chk);
      hipMemcpy(a.data(), o1, chk * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, chk * 4, hipMemcpyDeviceToHost);
      // tail of the tensor too (partial tiles)
      std::vector<float> at(4096), bt(4096);
      hipMemcpy(at.data(), o1 + out_n - 4096, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(bt.data(), o2 + out_n - 4096, 4096 * 4, hipMemcpyDeviceToHost);
      double e = 0, mx = 0, se = 0, sa = 0;
      for (size_t i = 0; i < chk; ++i) { const double d = (double)a[i] - b[i]; e = fmax(e, fabs(d)); mx = fmax(mx, fabs((double)a[i])); se += d * d; sa += (double)a[i] * a[i]; }
      for (size_t i = 0; i < 4096; ++i) e = fmax(e, fabs((double)at[i] - bt[i]));
      hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
      float ms_f, ms_x;
      for (int i = 0; i < 2; ++i) launch_conv_x3(L, in, n, hw, hw, rp, o2, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv(L, in, n, hw, hw, rp, o1, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_f, e0, e1v);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv_x3(L, in, n, hw, hw, rp, o2, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_x, e0, e1v);
      const double fl = 2.0 * cin * cout * 9 * hw * hw * n;
      printf("x3 %3d->%3d hw %2d res %d: max|diff| %.3g rms rel %.3g (max|x| %.3g); fp32 %.1f us (%.1f TF), x3 %.1f us (%.1f TF)\n",
             cin, cout, hw, withres, e, sqrt(se / sa), mx, ms_f / 5 * 1e3, fl / (ms_f / 5) / 1e9, ms_x / 5 * 1e3, fl / (ms_x / 5) / 1e9);
      hipFree(in); hipFree(o1); hipFree(o2); hipFree(res); hipFree(w); hipFree(w3); hipFree(vec);
      }
    }
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const bool bf = pass == 1;
    if (!bf && !strcmp(which, "bf16")) continue;
    if (bf && !strcmp(which, "f32")) continue;
    const int esz = bf ? 2 : 4;
    int shape_idx = -1;
    for (const Shape& sh : shapes) {
      if (++shape_idx != only && only >= 0) continue;
      ConvLayer L;
      L.taps = sh.taps; L.cin = sh.cin; L.cout = sh.cout; L.cin_s = sh.cin; L.cout_s = sh.cout; L.cout_w = sh.cout;
      L.stride = sh.stride; L.relu = 1;
      if (getenv("NOPIPE64")) L.flags |= CONV_NO_PIPE64;      // the two-group 64 -> 64 bf16 kernel instead of the one-role one
      if (getenv("NOPIPE32")) L.flags |= CONV_NO_PIPE32;
      const int ho = sh.hw / sh.stride;
      const size_t in_n = (size_t)n * sh.hw * sh.hw * sh.cin, out_n = (size_t)n * ho * ho * sh.cout;
      const size_t w_n = (size_t)sh.taps * sh.cin * sh.cout;
      void *in, *out, *res, *w; float* vec;
      hipMalloc(&in, in_n * esz); hipMalloc(&out, out_n * esz); hipMalloc(&res, out_n * esz); hipMalloc(&w, w_n * esz);
      hipMalloc(&vec, 3 * sh.cout * 4);
      {
        const size_t hn = in_n > w_n ? in_n : w_n;
        std::vector<unsigned char> h(hn * esz);
        unsigned s = 12345;
        for (size_t i = 0; i < hn; ++i) {
          s = s * 1664525u + 1013904223u;
          const float v = getenv("ZERO") ? 0.f : ((s >> 8) & 0xffff) / 65536.0f - 0.5f;      // ZERO=1: all-zero operands (clock / throttle diagnostic)
          if (bf) { unsigned u; memcpy(&u, &v, 4); const unsigned short b = (unsigned short)(u >> 16); memcpy(&h[i * 2], &b, 2); }
          else memcpy(&h[i * 4], &v, 4);
        }
        hipMemcpy(in, h.data(), in_n * esz, hipMemcpyHostToDevice);
        hipMemcpy(w, h.data(), w_n * esz, hipMemcpyHostToDevice);
        hipMemcpy(res, h.data(), (out_n < hn ? out_n : hn) * esz, hipMemcpyHostToDevice);
      }
      std::vector<float> v3(3 * sh.cout, 0.f);
      for (int i = 0; i < sh.cout; ++i) v3[sh.cout + i] = 1.f;
      hipMemcpy(vec, v3.data(), v3.size() * 4, hipMemcpyHostToDevice);
      L.w = (const float*)w; L.wh = (const uint16_t*)w; L.bias = vec; L.scale = vec + sh.cout; L.shift = vec + 2 * sh.cout;
      auto run = [&]() {
        if (bf) launch_conv_bf16(L, (const uint16_t*)in, n, sh.hw, sh.hw, sh.res ? (const uint16_t*)res : nullptr, out, 0, 0);
        else launch_conv(L, (const float*)in, n, sh.hw, sh.hw, sh.res ? (const float*)res : nullptr, (float*)out, 0);
      };
      if ((bf && (conv_thin_applicable(L, sh.hw) || (L.flags & CONV_NO_THIN))) ||
          (!bf && sh.cin == 32 && sh.cout == 32 && sh.taps == 9 && sh.stride == 1)) {   // new kernel vs the one-tile-per-workgroup kernel
        ConvLayer L2 = L;
        std::vector<unsigned char> r0(out_n * esz), r1(out_n * esz);
        L.flags |= CONV_NO_THIN; run(); hipDeviceSynchronize(); hipMemcpy(r0.data(), out, out_n * esz, hipMemcpyDeviceToHost);
        hipMemset(out, 0xff, out_n * esz);
        L.flags &= ~CONV_NO_THIN; run(); hipDeviceSynchronize(); hipMemcpy(r1.data(), out, out_n * esz, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < out_n * esz; ++i) bad += r0[i] != r1[i];
        printf("  thin vs halo: %zu differing bytes of %zu\n", bad, out_n * esz);
        if (getenv("THIN_OFF")) L.flags |= CONV_NO_THIN;
      }
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 2; ++i) run();
      hipEventRecord(e0, 0);
      const int iters = 5;
      for (int i = 0; i < iters; ++i) run();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= iters;
      const double fl = 2.0 * sh.cin * sh.cout * sh.taps * ho * ho * n;
      const double bytes = (double)(in_n / (sh.stride * sh.stride) + out_n * (1 + sh.res)) * esz;
      printf("%s cin %3d cout %3d hw %2d taps %d s%d res %d : %8.1f us  %6.1f TF  %5.2f TB/s\n", bf ? "bf16" : "f32 ", sh.cin,
             sh.cout, sh.hw, sh.taps, sh.stride, sh.res, ms * 1e3, fl / ms / 1e9, bytes / ms / 1e9);
#ifdef UMT_STAMP
      {
        unsigned long long h[2][64][4];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
        for (int ph = 0; ph < 22; ++ph) {
          printf("  ph %2d  pre-mfma g0 %6lld g1 %6lld", ph, (long long)(h[0][ph][3] - h[0][ph][0]), (long long)(h[1][ph][3] - h[1][ph][0]));
          printf("  ph %2d  g0: start %8llu work %7llu (epi %7llu)   g1: start %8llu work %7llu (epi %7llu)\n", ph,
                 h[0][ph][0] - h[0][0][0], h[0][ph][1] - h[0][ph][0], h[0][ph][2] > h[0][ph][0] ? h[0][ph][2] - h[0][ph][0] : 0,
                 h[1][ph][0] - h[0][0][0], h[1][ph][1] - h[1][ph][0], h[1][ph][2] > h[1][ph][0] ? h[1][ph][2] - h[1][ph][0] : 0);
        }
      }
#endif
      hipFree(in); hipFree(out); hipFree(res); hipFree(w); hipFree(vec);
    }
  }
  // fused stride-2 conv1 + down-sample vs the two separate launches
  if (strcmp(which, "bf16")) {
    const int cfgs[3][3] = {{32, 64, 48}, {64, 128, 24}, {128, 256, 12}};
    for (auto& cf : cfgs) {
      const int cin = cf[0], cout = cf[1], hw = cf[2], ho = hw / 2;
      ConvLayer C, D;
      C.taps = 9; C.cin = cin; C.cout = cout; C.cin_s = cin; C.cout_s = cout; C.cout_w = cout; C.stride = 2; C.relu = 1;
      D = C; D.taps = 1; D.relu = 0;
      const size_t in_n = (size_t)n * hw * hw * cin, out_n = (size_t)n * ho * ho * cout, w_n = (size_t)9 * cin * cout;
      float *in, *o1, *o2, *d1, *d2, *w, *wd, *vec;
      hipMalloc(&in, in_n * 4); hipMalloc(&o1, out_n * 4); hipMalloc(&o2, out_n * 4); hipMalloc(&d1, out_n * 4); hipMalloc(&d2, out_n * 4);
      hipMalloc(&w, w_n * 4); hipMalloc(&wd, w_n * 4); hipMalloc(&vec, 3 * cout * 4);
      std::vector<float> h(in_n > w_n ? in_n : w_n);
      unsigned s = 777;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
      hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice);
      hipMemcpy(w, h.data(), w_n * 4, hipMemcpyHostToDevice);
      hipMemcpy(wd, h.data() + 1000, (size_t)cin * cout * 4, hipMemcpyHostToDevice);
      std::vector<float> v3(3 * cout, 0.f);
      for (int i = 0; i < cout; ++i) { v3[i] = 0.01f * (i % 7); v3[cout + i] = 1.f + 0.01f * (i % 5); v3[2 * cout + i] = -0.02f * (i % 3); }
      hipMemcpy(vec, v3.data(), v3.size() * 4, hipMemcpyHostToDevice);
      C.w = w; D.w = wd; C.bias = D.bias = vec; C.scale = D.scale = vec + cout; C.shift = D.shift = vec + 2 * cout;
      launch_conv(C, in, n, hw, hw, nullptr, o1, 0);
      launch_conv(D, in, n, hw, hw, nullptr, d1, 0);
      hipMemset(o2, 0xff, out_n * 4); hipMemset(d2, 0xff, out_n * 4);
      if (launch_conv_s2_fused(C, &D, in, n, hw, hw, o2, d2, 0)) return 1;
      hipDeviceSynchronize();
      std::vector<float> a(out_n), b(out_n);
      hipMemcpy(a.data(), o1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, out_n * 4, hipMemcpyDeviceToHost);
      double e1 = 0, mx = 0;
      for (size_t i = 0; i < out_n; ++i) { e1 = fmax(e1, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
      hipMemcpy(a.data(), d1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, out_n * 4, hipMemcpyDeviceToHost);
      size_t bad = 0;
      for (size_t i = 0; i < out_n; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
      hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
      float ms_sep, ms_fused;
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) { launch_conv(C, in, n, hw, hw, nullptr, o1, 0); launch_conv(D, in, n, hw, hw, nullptr, d1, 0); }
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_sep, e0, e1v);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv_s2_fused(C, &D, in, n, hw, hw, o2, d2, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_fused, e0, e1v);
      // the one-tile-per-workgroup fused kernel (CONV_NO_S2P) against the persistent one: must agree bit for bit
      ConvLayer Cold = C;
      Cold.flags |= CONV_NO_S2P;
      hipMemset(o1, 0xff, out_n * 4); hipMemset(d1, 0xff, out_n * 4);
      if (launch_conv_s2_fused(Cold, &D, in, n, hw, hw, o1, d1, 0)) return 1;
      hipDeviceSynchronize();
      size_t bad_p = 0;
      hipMemcpy(a.data(), o1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, out_n * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out_n; ++i) bad_p += memcmp(&a[i], &b[i], 4) != 0;
      hipMemcpy(a.data(), d1, out_n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, out_n * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out_n; ++i) bad_p += memcmp(&a[i], &b[i], 4) != 0;
      float ms_old;
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) launch_conv_s2_fused(Cold, &D, in, n, hw, hw, o1, d1, 0);
      hipEventRecord(e1v, 0); hipEventSynchronize(e1v); hipEventElapsedTime(&ms_old, e0, e1v);
      const double fl = 2.0 * cin * cout * 10 * ho * ho * n;
      printf("s2+ds %3d->%3d hw %2d: conv1 max|diff| %.3g (max|x| %.3g), ds differing words %zu; separate %.1f us, fused %.1f us (%.1f TF); "
             "one-tile kernel %.1f us, words differing from it %zu\n",
             cin, cout, hw, e1, mx, bad, ms_sep / 5 * 1e3, ms_fused / 5 * 1e3, fl / (ms_fused / 5) / 1e9, ms_old / 5 * 1e3, bad_p);
      hipFree(in); hipFree(o1); hipFree(o2); hipFree(d1); hipFree(d2); hipFree(w); hipFree(wd); hipFree(vec);
    }
  }
  return 0;
}
