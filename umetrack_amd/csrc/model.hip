// Model handle, strict parameter table, weight packing and the stage / whole-model entry points.
// Mirrors what lib/models/model_loader.py:53-88 builds and what UmeTrackModel
// (lib/models/umetrack_model.py:100-242) runs, on channels-last rows and the kernels in
// conv_igemm.hip / stem.hip / head.hip.
#include "umt_internal.h"

#include <string.h>
#include <math.h>
#include <stdlib.h>

namespace umt {

static thread_local char g_err[1024] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

struct ParamSpec {
  std::string key;
  int ndim;
  int64_t shape[4];
  bool is_int64;
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

struct BlockLayers {
  ConvLayer conv1, conv2, down;
  bool has_down = false;
};

struct PackJob {   // one conv to pack at finalize
  ConvLayer* dst;
  std::string w, b, bn;   // keys ("" = absent)
};

constexpr float BN_EPS = 1e-5f;
constexpr int N_STAGE_BLOCKS[4] = {2, 3, 5, 2};
constexpr int ACT_FLOATS = 48 * 48 * 32;   // largest activation per crop (stage 1)
constexpr int DS_FLOATS = 24 * 24 * 64;    // largest downsample branch per crop

}  // namespace umt

using namespace umt;

struct umt_model {
  std::vector<ParamSpec> spec;
  std::map<std::string, int> index;
  std::vector<std::vector<float>> values;   // host copies (int64 counters are validated, not kept)
  std::vector<char> seen;
  bool finalized = false;
  int device = -1;
  unsigned conv_flags = 0;   // CONV_* kernel-selection switches, read from the environment at umt_model_create
  int stage1_sub = 0;        // UMT_STAGE1_SUB, likewise
  int32_t* status = nullptr; // dev, UMT_STATUS_* bits OR-ed by the kernels (lives at the end of `slab`)
  float* slab = nullptr;   // all packed weights
  uint16_t* slab_h = nullptr;   // bf16 copies of the backbone conv weights (UMT_PREC_BF16)
  // network
  StemParams stem{};
  std::vector<BlockLayers> blocks;   // 12 backbone blocks
  std::vector<int> block_hw;         // input H (=W) of each block
  ConvLayer proj, fuse1, fuse2, fuse3, temp[3];
  SkelParams skel{};
  BlockLayers reg_blocks[2][2];
  ConvLayer reg_out[2];
  // temporal state (temporal.py:41-42)
  float* mem = nullptr;        // (rows,18,6,6)
  float* prev_ext = nullptr;   // (rows,4,4)
  unsigned long long* mem_owner = nullptr;   // (rows): (epoch << 32 | sample) of the sample that last claimed the row
  unsigned mem_epoch = 0;      // one per temporal step
  int mem_rows = 0;
  int mem_cap = 0;
  // optional stage timing (umt_model_timing_*): a ring of TIMING_RING event triples; a triple that is recycled is
  // folded into the running sums first, so a long run with timing on holds a bounded number of events
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;   // 3 per slot
  size_t ev_head = 0, ev_live = 0;   // oldest live slot, number of live slots
  double t_backbone_ms = 0, t_forward_ms = 0;
  int t_calls = 0;
  // optional per-launch timing of the backbone (umt_model_layer_timing_*): one event before the stem and one
  // after every launch; a mark names the launch that ended at the event of the same index
  struct LayerMark { char name[48]; double flop; };
  bool layer_timing = false;
  std::vector<hipEvent_t> lev_pool;
  std::vector<LayerMark> lev_marks;
  struct LayerAcc { char name[48]; double flop, ms; int launches; };
  std::vector<LayerAcc> lev_acc;     // finished passes, folded per schedule position
};

namespace umt {

static void add_param(umt_model* m, const std::string& key, std::initializer_list<int64_t> shape, bool is_int = false) {
  ParamSpec p;
  p.key = key;
  p.ndim = (int)shape.size();
  int i = 0;
  for (int64_t s : shape) p.shape[i++] = s;
  for (; i < 4; ++i) p.shape[i] = 1;
  p.is_int64 = is_int;
  m->index[key] = (int)m->spec.size();
  m->spec.push_back(p);
}

static void add_bn(umt_model* m, const std::string& p, int c) {
  add_param(m, p + ".weight", {c});
  add_param(m, p + ".bias", {c});
  add_param(m, p + ".running_mean", {c});
  add_param(m, p + ".running_var", {c});
  add_param(m, p + ".num_batches_tracked", {}, true);
}

static void add_conv(umt_model* m, const std::string& p, int cout, int cin, int k, bool bias) {
  add_param(m, p + ".weight", {cout, cin, k, k});
  if (bias) add_param(m, p + ".bias", {cout});
}

static void add_block(umt_model* m, const std::string& p, int cin, int cout, bool down) {
  add_conv(m, p + ".conv1", cout, cin, 3, false);
  add_bn(m, p + ".bn1", cout);
  add_conv(m, p + ".conv2", cout, cout, 3, false);
  add_bn(m, p + ".bn2", cout);
  if (down) {
    add_conv(m, p + ".downsample.0", cout, cin, 1, false);
    add_bn(m, p + ".downsample.1", cout);
  }
}

// the reference's state_dict, in its order
static void build_spec(umt_model* m) {
  const std::string bb = "_feature_extractor._image_backbone";
  add_conv(m, bb + ".0._layers.0.0", 32, 1, 3, true);
  add_bn(m, bb + ".0._layers.0.1", 32);
  const int cin[4] = {32, 32, 64, 128}, cout[4] = {32, 64, 128, 256}, stride[4] = {1, 2, 2, 2};
  for (int s = 0; s < 4; ++s)
    for (int b = 0; b < N_STAGE_BLOCKS[s]; ++b) {
      const bool first = b == 0;
      add_block(m, bb + ".0._layers." + std::to_string(s + 1) + "." + std::to_string(b), first ? cin[s] : cout[s],
                cout[s], first && (stride[s] != 1 || cin[s] != cout[s]));
    }
  add_conv(m, bb + ".1", 72, 256, 1, true);
  const std::string fu = "_feature_extractor._multi_view_fusion";
  add_conv(m, fu + ".0", 108, 144, 1, true);
  add_bn(m, fu + ".1", 108);
  add_conv(m, fu + ".3", 72, 108, 1, true);
  add_bn(m, fu + ".4", 72);
  add_conv(m, fu + ".6", 72, 72, 1, true);
  for (int i = 0; i < 3; ++i) add_conv(m, "_temporal._temporal_module." + std::to_string(2 * i), 90, 90, 1, true);
  add_param(m, "_skeleton_enc._layers.0.weight", {144, 132});
  add_param(m, "_skeleton_enc._layers.0.bias", {144});
  add_bn(m, "_skeleton_enc._layers.2", 4);
  const char* heads[2] = {"_regressor_k", "_regressor_u"};
  const int hc[2] = {76, 72}, ho[2] = {62, 63};
  for (int h = 0; h < 2; ++h) {
    const std::string p = std::string(heads[h]) + "._pose_regression_layers";
    for (int b = 0; b < 2; ++b) add_block(m, p + "." + std::to_string(b), hc[h], hc[h], false);
    add_conv(m, p + ".2", ho[h], hc[h], 1, true);
  }
  m->values.resize(m->spec.size());
  m->seen.assign(m->spec.size(), 0);
}

static const std::vector<float>& val(const umt_model* m, const std::string& key) { return m->values[m->index.at(key)]; }

// Appends one packed conv (weights + bias/scale/shift) to `host`; pointers are stored as offsets first.
struct Packed {
  ConvLayer* layer;
  size_t w_off, b_off, s_off, h_off;
};

static void pack_conv(const umt_model* m, std::vector<float>& host, std::vector<Packed>& packed, ConvLayer* L,
                      const std::string& wkey, const std::string& bkey, const std::string& bn, int stride, int relu) {
  const ParamSpec& ps = m->spec[m->index.at(wkey)];
  const int cout = (int)ps.shape[0], cin = (int)ps.shape[1], k = (int)ps.shape[2];
  L->taps = k * k;
  L->cin = cin;
  L->cout = cout;
  L->cin_s = round_up(cin, 8);
  L->cout_s = round_up(cout, 8);
  L->cout_w = round_up(cout, 32);
  L->stride = stride;
  L->relu = relu;
  L->flags = m->conv_flags;
  const std::vector<float>& w = val(m, wkey);
  const int c8 = L->cin_s / 8;
  Packed p;
  p.layer = L;
  p.w_off = host.size();
  host.resize(host.size() + (size_t)L->taps * c8 * 2 * L->cout_w * 4, 0.f);
  float* dst = host.data() + p.w_off;
  // packed[tap][c8][h][co][q] = W[co][ci = 8*c8 + 4*h + q][ky][kx]   (see conv_igemm.hip)
  for (int tap = 0; tap < L->taps; ++tap)
    for (int g = 0; g < c8; ++g)
      for (int h = 0; h < 2; ++h)
        for (int co = 0; co < cout; ++co)
          for (int q = 0; q < 4; ++q) {
            const int ci = 8 * g + 4 * h + q;
            if (ci < cin)
              dst[((((size_t)tap * c8 + g) * 2 + h) * L->cout_w + co) * 4 + q] = w[((size_t)co * cin + ci) * L->taps + tap];
          }
  p.b_off = host.size();
  host.resize(host.size() + L->cout_w, 0.f);
  p.s_off = host.size();
  host.resize(host.size() + L->cout_w, 1.f);
  p.h_off = host.size();
  host.resize(host.size() + L->cout_w, 0.f);
  if (!bkey.empty()) {
    const std::vector<float>& b = val(m, bkey);
    for (int c = 0; c < cout; ++c) host[p.b_off + c] = b[c];
  }
  if (!bn.empty()) {
    // eval-mode BN as torch evaluates it on CPU: alpha = gamma / sqrt(var + eps), beta = bias - mean * alpha
    const std::vector<float>&g = val(m, bn + ".weight"), &be = val(m, bn + ".bias"), &mu = val(m, bn + ".running_mean"),
                            &var = val(m, bn + ".running_var");
    for (int c = 0; c < cout; ++c) {
      const float alpha = g[c] / sqrtf(var[c] + BN_EPS);
      host[p.s_off + c] = alpha;
      host[p.h_off + c] = be[c] - mu[c] * alpha;
    }
  }
  packed.push_back(p);
}

// round-to-nearest-even fp32 -> bf16 (the weights are finite)
static uint16_t to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// bf16 copy of one conv's weights for conv_bf16.hip:
//   packed[tap][g][h][co][j] = bf16(W[co][ci = 16*g + 8*h + j][ky][kx])
struct PackedH {
  ConvLayer* layer;
  size_t off;
};
static void pack_conv_bf16(const umt_model* m, std::vector<uint16_t>& host, std::vector<PackedH>& packed, ConvLayer* L,
                           const std::string& wkey) {
  const std::vector<float>& w = val(m, wkey);
  const int c16 = L->cin_s / 16;
  PackedH p{L, host.size()};
  host.resize(host.size() + (size_t)L->taps * c16 * 2 * L->cout_w * 8, 0);
  uint16_t* dst = host.data() + p.off;
  for (int tap = 0; tap < L->taps; ++tap)
    for (int g = 0; g < c16; ++g)
      for (int h = 0; h < 2; ++h)
        for (int co = 0; co < L->cout; ++co)
          for (int j = 0; j < 8; ++j) {
            const int ci = 16 * g + 8 * h + j;
            if (ci < L->cin)
              dst[((((size_t)tap * c16 + g) * 2 + h) * L->cout_w + co) * 8 + j] =
                  to_bf16(w[((size_t)co * L->cin + ci) * L->taps + tap]);
          }
  packed.push_back(p);
}

// Three-way bf16 split of one 3x3 conv's fp32 weights for conv_x3.hip (w = w0 + w1 + w2 exactly):
//   packed[tap][g][plane][h][co][j] = plane-th piece of W[co][ci = 16*g + 8*h + j][ky][kx]
static float from_bf16(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static void pack_conv_x3(const umt_model* m, std::vector<uint16_t>& host, std::vector<PackedH>& packed, ConvLayer* L,
                         const std::string& wkey) {
  const std::vector<float>& w = val(m, wkey);
  const int c16 = L->cin_s / 16;
  PackedH p{L, host.size()};
  host.resize(host.size() + (size_t)L->taps * c16 * 3 * 2 * L->cout_w * 8, 0);
  uint16_t* dst = host.data() + p.off;
  for (int tap = 0; tap < L->taps; ++tap)
    for (int ci = 0; ci < L->cin; ++ci)
      for (int co = 0; co < L->cout; ++co) {
        float r = w[((size_t)co * L->cin + ci) * L->taps + tap];
        for (int pl = 0; pl < 3; ++pl) {
          const uint16_t b = to_bf16(r);
          r -= from_bf16(b);       // exact
          dst[(((((size_t)tap * c16 + ci / 16) * 3 + pl) * 2 + (ci % 16) / 8) * L->cout_w + co) * 8 + ci % 8] = b;
        }
      }
  packed.push_back(p);
}

// fp32 weights for head_seq.hip (v_mfma_f32_16x16x4_f32): packed[kk][j][co][s] with kk walking (32-channel chunk,
// tap, 16-channel half) and ci = 32 chunk + 16 half + 8 (s >> 1) + 2 (s & 1) + {0, 4, 1, 5}[j]: k slot j of MFMA s.
// That is the order in which conv_igemm.hip's kernels accumulate, so both paths give the same bits.
struct PackedSeq {
  ConvLayer* layer;
  size_t off;
};
static void pack_conv_seq(const umt_model* m, std::vector<float>& host, std::vector<PackedSeq>& packed, ConvLayer* L,
                          const std::string& wkey) {
  const std::vector<float>& w = val(m, wkey);
  const int cin_p = round_up(L->cin, 16), cout_p = round_up(L->cout, 16);
  static const int T4[4] = {0, 4, 1, 5};
  PackedSeq p{L, host.size()};
  // padded with all-zero groups to a multiple of 4, plus the four groups the kernel's weight ring reads past the end
  const size_t nkk4 = (((size_t)L->taps * (cin_p / 16) + 3) & ~(size_t)3) + 4;
  host.resize(host.size() + nkk4 * 4 * cout_p * 4, 0.f);
  float* dst = host.data() + p.off;
  size_t kk = 0;
  for (int chunk = 0; chunk * 32 < cin_p; ++chunk) {
    const int ng = cin_p - 32 * chunk >= 32 ? 2 : 1;
    for (int tap = 0; tap < L->taps; ++tap)
      for (int g = 0; g < ng; ++g, ++kk)
        for (int j = 0; j < 4; ++j)
          for (int co = 0; co < L->cout; ++co)
            for (int s = 0; s < 4; ++s) {
              const int ci = 32 * chunk + 16 * g + 8 * (s >> 1) + 2 * (s & 1) + T4[j];
              if (ci < L->cin) dst[((kk * 4 + j) * cout_p + co) * 4 + s] = w[((size_t)co * L->cin + ci) * L->taps + tap];
            }
  }
  packed.push_back(p);
}

static void pack_block(const umt_model* m, std::vector<float>& host, std::vector<Packed>& packed, BlockLayers* B,
                       const std::string& p, int stride, bool down) {
  pack_conv(m, host, packed, &B->conv1, p + ".conv1.weight", "", p + ".bn1", stride, 1);
  pack_conv(m, host, packed, &B->conv2, p + ".conv2.weight", "", p + ".bn2", 1, 1);   // ReLU after the residual add
  B->has_down = down;
  if (down) pack_conv(m, host, packed, &B->down, p + ".downsample.0.weight", "", p + ".downsample.1", stride, 0);
}

// ---- workspace carving ----
struct Carver {
  char* base;
  int64_t off = 0, cap;
  Carver(void* p, int64_t c) : base((char*)p), cap(c) {}
  template <typename T>
  T* take(int64_t count) {
    T* r = (T*)(base + off);
    off += round_up64(count * (int64_t)sizeof(T), 256);
    return r;
  }
};

struct BackboneWs {
  float *a, *b, *c, *d, *feat;
};
static int64_t carve_backbone(Carver& cv, int n, BackboneWs* w) {
  w->a = cv.take<float>((int64_t)n * ACT_FLOATS);
  w->b = cv.take<float>((int64_t)n * ACT_FLOATS);
  w->c = cv.take<float>((int64_t)n * ACT_FLOATS);
  w->d = cv.take<float>((int64_t)n * DS_FLOATS);
  w->feat = cv.take<float>((int64_t)n * 36 * 72);
  return cv.off;
}

struct HeadWs {
  float *feat_in;   // used by the stage entry points only
  float *crop_xf, *sample_xf, *cam0, *rel_xf;
  int32_t *views, *row_of;
  float *cat, *f1, *f2, *f3, *tin, *t1, *t2, *rin, *r1, *r2, *r3, *rout, *raw;
};
static int64_t carve_head(Carver& cv, int n, int bs, HeadWs* w) {
  const int64_t px = (int64_t)bs * 36;
  w->feat_in = cv.take<float>((int64_t)n * 36 * 72);
  w->crop_xf = cv.take<float>((int64_t)n * 12);
  w->sample_xf = cv.take<float>((int64_t)bs * 12);
  w->cam0 = cv.take<float>((int64_t)bs * 16);
  w->rel_xf = cv.take<float>((int64_t)bs * 13);      // 12 floats per sample + one effective use_memory byte each
  w->views = cv.take<int32_t>(bs);
  w->row_of = cv.take<int32_t>(bs);
  w->cat = cv.take<float>(px * 144);
  w->f1 = cv.take<float>(px * 112);
  w->f2 = cv.take<float>(px * 72);
  w->f3 = cv.take<float>(px * 72);
  w->tin = cv.take<float>(px * 96);
  w->t1 = cv.take<float>(px * 96);
  w->t2 = cv.take<float>(px * 96);
  w->rin = cv.take<float>(px * 80);
  w->r1 = cv.take<float>(px * 80);
  w->r2 = cv.take<float>(px * 80);
  w->r3 = cv.take<float>(px * 80);
  w->rout = cv.take<float>(px * 64);
  w->raw = cv.take<float>((int64_t)bs * 64);
  return cv.off;
}

// ---- stage timing (measurement aid): a ring of event triples ----
constexpr size_t TIMING_RING = 64;

static int timing_fold_oldest(umt_model* m) {
  hipEvent_t* e = &m->ev_pool[3 * m->ev_head];
  float t1 = 0, t2 = 0;
  UMT_CHECK_HIP(hipEventSynchronize(e[2]));
  UMT_CHECK_HIP(hipEventElapsedTime(&t1, e[0], e[1]));
  UMT_CHECK_HIP(hipEventElapsedTime(&t2, e[0], e[2]));
  m->t_backbone_ms += t1;
  m->t_forward_ms += t2;
  m->t_calls += 1;
  m->ev_head = (m->ev_head + 1) % TIMING_RING;
  m->ev_live -= 1;
  return 0;
}

// the next free triple; when the ring is full the oldest one (TIMING_RING calls back) is folded into the sums first
static int timing_slot(umt_model* m, hipEvent_t** ev) {
  while (m->ev_pool.size() < 3 * TIMING_RING) {
    hipEvent_t e;
    UMT_CHECK_HIP(hipEventCreate(&e));
    m->ev_pool.push_back(e);
  }
  if (m->ev_live == TIMING_RING) UMT_TRY(timing_fold_oldest(m));
  *ev = &m->ev_pool[3 * ((m->ev_head + m->ev_live) % TIMING_RING)];
  m->ev_live += 1;
  return 0;
}

static int check_ready(const umt_model* m) {
  UMT_REQUIRE(m, UMT_ERR_INVALID, "null model handle");
  UMT_REQUIRE(m->finalized, UMT_ERR_STATE, "model is not finalized (call umt_model_finalize first)");
  int dev = -1;
  UMT_CHECK_HIP(hipGetDevice(&dev));
  UMT_REQUIRE(dev == m->device, UMT_ERR_STATE, "current device %d differs from the model's device %d", dev, m->device);
  return 0;
}

// ---- per-launch timing of the backbone (measurement aid) ----
// name == nullptr: the mark that opens a backbone pass.  flop: algorithmic, true channel counts.
// Fold every recorded mark into the per-position sums (waits for the last event) and forget the marks, so the
// event pool is reused from its start: a long run with layer timing on holds at most LAYER_MARK_CAP + one pass.
static int layer_marks_fold(umt_model* m) {
  if (m->lev_marks.empty()) return 0;
  UMT_CHECK_HIP(hipEventSynchronize(m->lev_pool[m->lev_marks.size() - 1]));
  size_t pos = 0;
  for (size_t k = 0; k < m->lev_marks.size(); ++k) {
    const umt_model::LayerMark& mk = m->lev_marks[k];
    if (mk.flop < 0) { pos = 0; continue; }       // opening mark of a pass
    UMT_REQUIRE(k > 0, UMT_ERR_STATE, "layer timing: launch mark without an opening mark");
    float t = 0;
    UMT_CHECK_HIP(hipEventElapsedTime(&t, m->lev_pool[k - 1], m->lev_pool[k]));
    if (m->lev_acc.size() <= pos) {
      umt_model::LayerAcc a{};
      memcpy(a.name, mk.name, 48);
      m->lev_acc.push_back(a);
    }
    m->lev_acc[pos].flop += mk.flop;
    m->lev_acc[pos].ms += t;
    m->lev_acc[pos].launches += 1;
    ++pos;
  }
  m->lev_marks.clear();
  return 0;
}

constexpr size_t LAYER_MARK_CAP = 1024;

static int layer_mark(umt_model* m, hipStream_t s, const char* name, double flop) {
  if (!m->layer_timing) return 0;
  if (!name && m->lev_marks.size() >= LAYER_MARK_CAP) UMT_TRY(layer_marks_fold(m));   // only between passes
  const size_t k = m->lev_marks.size();
  if (m->lev_pool.size() <= k) {
    hipEvent_t e;
    UMT_CHECK_HIP(hipEventCreate(&e));
    m->lev_pool.push_back(e);
  }
  umt_model::LayerMark mk;
  snprintf(mk.name, sizeof(mk.name), "%s", name ? name : "");
  mk.flop = name ? flop : -1.0;
  m->lev_marks.push_back(mk);
  UMT_CHECK_HIP(hipEventRecord(m->lev_pool[k], s));
  return 0;
}

static int conv_mark(umt_model* m, hipStream_t s, const char* kind, const ConvLayer& L, const ConvLayer* ds, int n, int hw) {
  if (!m->layer_timing) return 0;
  const int ho = hw / L.stride;
  double flop = 2.0 * L.taps * L.cin * L.cout * (double)n * ho * ho;
  if (ds) flop += 2.0 * ds->taps * ds->cin * ds->cout * (double)n * ho * ho;
  char name[48];
  snprintf(name, sizeof(name), "%s %d->%d hw%d s%d", kind, L.cin, L.cout, hw, L.stride);
  return layer_mark(m, s, name, flop);
}

// ---- stages on channels-last rows ----
// Blocks [b0, b1) on n crops whose rows start at x; returns the buffer that holds the result.
// x/t/y rotate; the down-sample branch goes to d.
// x3: stride-1 3x3 layers on the bf16 matrix pipe with three-way split operands (conv_x3.hip); every tensor stays fp32.
// A launch with too few pixels to fill the chip with the split kernels' 128-pixel tiles (the tracker's per-frame call: 4 crops)
// stays on the fp32 kernels, which have few-pixel forms (conv_small.hip): 1.05 -> 0.6 ms per frame in the fp32x3 mode.
constexpr int64_t X3_MIN_PIXELS = 16384;
static int conv_f32(const ConvLayer& L, bool x3, const float* in, int n, int hw, const float* res, float* out, hipStream_t s) {
  if (x3 && (int64_t)n * hw * hw >= X3_MIN_PIXELS && conv_x3_applicable(L, hw)) return launch_conv_x3(L, in, n, hw, hw, res, out, s);
  return launch_conv(L, in, n, hw, hw, res, out, s);
}

static int run_blocks(umt_model* m, bool x3, size_t b0, size_t b1, int n, float*& x, float*& t, float*& y, float* d,
                      hipStream_t s) {
  for (size_t i = b0; i < b1; ++i) {
    const BlockLayers& B = m->blocks[i];
    const int hw = m->block_hw[i];
    const float* res = x;
    if (B.has_down && x3 && (int64_t)n * hw * hw / 4 >= X3_MIN_PIXELS && conv_s2_x3_applicable(B.conv1, &B.down, hw, hw)) {
      UMT_TRY(launch_conv_s2_fused_x3(B.conv1, &B.down, x, n, hw, hw, t, d, s));
      UMT_TRY(conv_mark(m, s, "c1+ds", B.conv1, &B.down, n, hw));
      res = d;
    } else if (B.has_down && conv_s2_fused_applicable(B.conv1, &B.down, hw, hw)) {
      UMT_TRY(launch_conv_s2_fused(B.conv1, &B.down, x, n, hw, hw, t, d, s));     // conv1 and the down-sample branch
      UMT_TRY(conv_mark(m, s, "c1+ds", B.conv1, &B.down, n, hw));
      res = d;
    } else {
      UMT_TRY(conv_f32(B.conv1, x3, x, n, hw, nullptr, t, s));
      UMT_TRY(conv_mark(m, s, "c1", B.conv1, nullptr, n, hw));
      if (B.has_down) {
        UMT_TRY(launch_conv(B.down, x, n, hw, hw, nullptr, d, s));
        UMT_TRY(conv_mark(m, s, "ds", B.down, nullptr, n, hw));
        res = d;
      }
    }
    const int hw2 = hw / B.conv1.stride;
    UMT_TRY(conv_f32(B.conv2, x3, t, n, hw2, res, y, s));
    UMT_TRY(conv_mark(m, s, res ? "c2+res" : "c2", B.conv2, nullptr, n, hw2));
    float* old = x;
    x = y;
    y = old;
  }
  return 0;
}

static int backbone_rows(umt_model* m, bool x3, const float* images, int n, const BackboneWs& w, hipStream_t s) {
  float *x = w.a, *t = w.b, *y = w.c;
  const int sub = m->stage1_sub;
  size_t first = 0;
  UMT_TRY(layer_mark(m, s, nullptr, 0));
  if (sub > 0 && sub < n) {
    // Stage 1 (stem + 2 blocks at 48x48x32, 295 KB per crop per tensor) depth-first over groups of
    // `sub` crops, so that a group's three live tensors stay in the 256 MB Infinity Cache between
    // its five launches instead of streaming 1.2 GB per layer through HBM.
    const size_t nb1 = (size_t)N_STAGE_BLOCKS[0];
    float *xe = x, *te = t, *ye = y;
    for (int c0 = 0; c0 < n; c0 += sub) {
      const int nc = n - c0 < sub ? n - c0 : sub;
      const size_t off = (size_t)c0 * ACT_FLOATS;
      float *xs = w.a + off, *ts = w.b + off, *ys = w.c + off;
      UMT_TRY(launch_stem(m->stem, images + (size_t)c0 * 96 * 96, nc, xs, s));
      UMT_TRY(layer_mark(m, s, "stem 1->32 hw96", 2.0 * 9 * 32 * 96.0 * 96.0 * nc));
      UMT_TRY(run_blocks(m, x3, 0, nb1, nc, xs, ts, ys, w.d, s));
      xe = xs - off; te = ts - off; ye = ys - off;    // same rotation for every group
    }
    x = xe; t = te; y = ye;
    first = nb1;
  } else {
    UMT_TRY(launch_stem(m->stem, images, n, x, s));
    UMT_TRY(layer_mark(m, s, "stem 1->32 hw96", 2.0 * 9 * 32 * 96.0 * 96.0 * n));
  }
  UMT_TRY(run_blocks(m, x3, first, m->blocks.size(), n, x, t, y, w.d, s));
  UMT_TRY(launch_conv(m->proj, x, n, 6, 6, nullptr, w.feat, s));
  UMT_TRY(conv_mark(m, s, "proj", m->proj, nullptr, n, 6));
  return 0;
}

// UMT_PREC_BF16: same schedule on bf16 rows (the fp32 workspace buffers hold them, half used); the
// projection writes fp32 feature rows so everything after the backbone is unchanged.
static int run_blocks_bf16(umt_model* m, size_t b0, size_t b1, int n, uint16_t*& x, uint16_t*& t, uint16_t*& y,
                           uint16_t* d, hipStream_t s) {
  for (size_t i = b0; i < b1; ++i) {
    const BlockLayers& B = m->blocks[i];
    const int hw = m->block_hw[i];
    const uint16_t* res = x;
    if (i == 0 && b1 >= 3 && !B.has_down && !m->blocks[1].has_down && m->blocks[2].has_down &&
        stage1_depth_applicable(B.conv1, B.conv2, m->blocks[1].conv1, m->blocks[1].conv2, m->blocks[2].conv1, &m->blocks[2].down, hw, hw)) {
      // both stage-1 blocks, conv1 of stage 2's first block and its down-sample branch depth-first in one launch
      const BlockLayers &B1 = m->blocks[1], &B2 = m->blocks[2];
      UMT_TRY(launch_stage1_depth_bf16(B.conv1, B.conv2, B1.conv1, B1.conv2, B2.conv1, B2.down, x, n, t, d, s));
      if (m->layer_timing) {
        const double px = (double)n * hw * hw;
        UMT_TRY(layer_mark(m, s, "stage1 depth 2xblock32+s2+ds hw48",
                           2.0 * px * (4.0 * 9 * 32 * 32 + (9.0 * 32 * 64 + 32.0 * 64) / 4.0)));
      }
      const int hw2 = hw / 2;
      UMT_TRY(launch_conv_bf16(B2.conv2, t, n, hw2, hw2, d, y, 0, s));
      UMT_TRY(conv_mark(m, s, "c2+res", B2.conv2, nullptr, n, hw2));
      uint16_t* old = x;
      x = y;
      y = old;
      i = 2;
      continue;
    }
    if (!B.has_down && block32_fused_applicable(B.conv1, B.conv2, hw, hw)) {     // conv1 -> conv2 (+x) with t kept on chip
      UMT_TRY(launch_block32_fused_bf16(B.conv1, B.conv2, x, n, y, s));
      if (m->layer_timing) {
        char nm[48];
        snprintf(nm, sizeof(nm), "c1+c2+res %d->%d hw%d s1", B.conv1.cin, B.conv2.cout, hw);
        UMT_TRY(layer_mark(m, s, nm, 2.0 * 2.0 * 9 * B.conv1.cin * B.conv1.cout * (double)n * hw * hw));
      }
      uint16_t* old = x;
      x = y;
      y = old;
      continue;
    }
    if (B.has_down && conv_s2_fused_applicable(B.conv1, &B.down, hw, hw)) {
      UMT_TRY(launch_conv_s2_fused_bf16(B.conv1, &B.down, x, n, hw, hw, t, d, s));
      UMT_TRY(conv_mark(m, s, "c1+ds", B.conv1, &B.down, n, hw));
      res = d;
    } else {
      UMT_TRY(launch_conv_bf16(B.conv1, x, n, hw, hw, nullptr, t, 0, s));
      UMT_TRY(conv_mark(m, s, "c1", B.conv1, nullptr, n, hw));
      if (B.has_down) {
        UMT_TRY(launch_conv_bf16(B.down, x, n, hw, hw, nullptr, d, 0, s));
        UMT_TRY(conv_mark(m, s, "ds", B.down, nullptr, n, hw));
        res = d;
      }
    }
    const int hw2 = hw / B.conv1.stride;
    UMT_TRY(launch_conv_bf16(B.conv2, t, n, hw2, hw2, res, y, 0, s));
    UMT_TRY(conv_mark(m, s, "c2+res", B.conv2, nullptr, n, hw2));
    uint16_t* old = x;
    x = y;
    y = old;
  }
  return 0;
}

static int backbone_rows_bf16(umt_model* m, const float* images, int n, const BackboneWs& w, hipStream_t s) {
  uint16_t *x = (uint16_t*)w.a, *t = (uint16_t*)w.b, *y = (uint16_t*)w.c, *d = (uint16_t*)w.d;
  const int sub = m->stage1_sub;
  size_t first = 0;
  UMT_TRY(layer_mark(m, s, nullptr, 0));
  if (sub > 0 && sub < n) {
    const size_t nb1 = (size_t)N_STAGE_BLOCKS[0];
    uint16_t *xe = x, *te = t, *ye = y;
    for (int c0 = 0; c0 < n; c0 += sub) {
      const int nc = n - c0 < sub ? n - c0 : sub;
      const size_t off = (size_t)c0 * ACT_FLOATS;
      uint16_t *xs = x + off, *ts = t + off, *ys = y + off;
      UMT_TRY(launch_stem_bf16(m->stem, images + (size_t)c0 * 96 * 96, nc, xs, s));
      UMT_TRY(layer_mark(m, s, "stem 1->32 hw96", 2.0 * 9 * 32 * 96.0 * 96.0 * nc));
      UMT_TRY(run_blocks_bf16(m, 0, nb1, nc, xs, ts, ys, d, s));
      xe = xs - off; te = ts - off; ye = ys - off;
    }
    x = xe; t = te; y = ye;
    first = nb1;
  } else {
    UMT_TRY(launch_stem_bf16(m->stem, images, n, x, s));
    UMT_TRY(layer_mark(m, s, "stem 1->32 hw96", 2.0 * 9 * 32 * 96.0 * 96.0 * n));
  }
  UMT_TRY(run_blocks_bf16(m, first, m->blocks.size(), n, x, t, y, d, s));
  UMT_TRY(launch_conv_bf16(m->proj, x, n, 6, 6, nullptr, w.feat, 1, s));
  UMT_TRY(conv_mark(m, s, "proj", m->proj, nullptr, n, 6));
  return 0;
}

static int backbone_any(umt_model* m, int precision, const float* images, int n, const BackboneWs& w, hipStream_t s) {
  if (precision == UMT_PREC_BF16) return backbone_rows_bf16(m, images, n, w, s);
  return backbone_rows(m, precision == UMT_PREC_FP32_BF16X3, images, n, w, s);
}

// feat rows [n*36][72] -> fused features written into tin channels 18..89
static int fuse_rows(umt_model* m, const float* feat, const float* K, const float* E, const int64_t* sr, int n, int bs,
                     const HeadWs& w, hipStream_t s) {
  UMT_TRY(launch_fusion_xforms(K, E, sr, n, bs, w.crop_xf, w.sample_xf, w.views, m->status, s));
  UMT_TRY(launch_ftl_concat(feat, w.crop_xf, sr, w.views, n, bs, w.cat, s));
  if (conv_chain3_applicable(m->fuse1, m->fuse2, m->fuse3)) {
    UMT_TRY(launch_conv_chain3(m->fuse1, m->fuse2, m->fuse3, w.cat, (int64_t)bs * 36, w.f3, s));
  } else {
    UMT_TRY(launch_conv(m->fuse1, w.cat, bs, 6, 6, nullptr, w.f1, s));
    UMT_TRY(launch_conv(m->fuse2, w.f1, bs, 6, 6, nullptr, w.f2, s));
    UMT_TRY(launch_conv(m->fuse3, w.f2, bs, 6, 6, nullptr, w.f3, s));
  }
  // (the memory channels 0..17 and the padding 90..95 are zeroed by the same launch: no memset of the tensor)
  UMT_TRY(launch_ftl_select(w.f3, feat, w.sample_xf, sr, w.views, n, bs, w.tin, 96, 18, s, true));
  return 0;
}

// tin (channels 18..89 set) -> temporal output in out_rows[.., 0..71] with stride out_cs; updates the state
static int temporal_rows(umt_model* m, const float* cam0, const int64_t* midx, const uint8_t* use, int bs,
                         const HeadWs& w, float* out_rows, int out_cs, hipStream_t s, bool zero_pad = false) {
  UMT_REQUIRE(m->mem_rows > 0, UMT_ERR_STATE, "temporal: no memory rows reserved (umt_model_reserve_memory)");
  m->mem_epoch += 1;                      // one claim epoch per step; 2^32 steps before it could wrap
  if (m->mem_epoch == 0) m->mem_epoch = 1;
  UMT_TRY(launch_temporal_gather(m->mem, m->prev_ext, m->mem_owner, m->mem_epoch, m->mem_rows, cam0, midx, use, bs, w.tin, 96,
                                 w.rel_xf, w.row_of, m->status, s));
  if (conv_chain3_applicable(m->temp[0], m->temp[1], m->temp[2])) {
    UMT_TRY(launch_conv_chain3(m->temp[0], m->temp[1], m->temp[2], w.tin, (int64_t)bs * 36, w.t1, s));
  } else {
    UMT_TRY(launch_conv(m->temp[0], w.tin, bs, 6, 6, nullptr, w.t1, s));
    UMT_TRY(launch_conv(m->temp[1], w.t1, bs, 6, 6, nullptr, w.t2, s));
    UMT_TRY(launch_conv(m->temp[2], w.t2, bs, 6, 6, nullptr, w.t1, s));
  }
  UMT_TRY(launch_temporal_split(w.t1, 96, m->mem, w.row_of, bs, out_rows, out_cs, s, zero_pad));
  return 0;
}

// rin (channels 0..71 set; head k: 72..75 filled here) -> decoded outputs
// x3: the four 3x3 layers on the split-operand kernels (conv_x3.hip) when the batch is large enough for the wide tiles -
// the bf16 and fp32x3 modes of umt_model_forward, whose results are not tied bit for bit to the fp32 kernels'; fp32
// tensors and fp32-rounding-level accuracy either way (tests/test_gpu_x3.py)
static int regress_rows(umt_model* m, int head, const float* skel_axes, const float* skel_rest, int n_skel,
                        const float* cam0, const int64_t* hand_idx, int bs, const HeadWs& w, float* joint_angles,
                        float* wrist_xfs, float* sigmas, float* skel_scales, float* raw_out, hipStream_t s,
                        bool x3 = false) {
  const int cs = head == UMT_HEAD_KNOWN_SKELETON ? 80 : 72;
  const int nraw = head == UMT_HEAD_KNOWN_SKELETON ? 62 : 63;
  if (head == UMT_HEAD_KNOWN_SKELETON) {
    UMT_REQUIRE(skel_axes && skel_rest, UMT_ERR_INVALID, "known-skeleton head needs skeleton data");
    UMT_REQUIRE(n_skel == 1 || n_skel == bs, UMT_ERR_INVALID, "n_skel must be 1 or bs (%d), got %d", bs, n_skel);
    UMT_TRY(launch_skeleton(m->skel, skel_axes, skel_rest, n_skel, bs, w.rin, cs, 72, s));
  }
  const BlockLayers* B = m->reg_blocks[head];
  x3 = x3 && !(m->conv_flags & HEAD_NO_X3) && (int64_t)bs * 36 >= HEAD_X3_MIN_PIXELS;
  auto conv = [&](const ConvLayer& L, const float* in, const float* res, float* out) {
    if (x3 && conv_x3_applicable(L, 6)) return launch_conv_x3(L, in, bs, 6, 6, res, out, s);
    return launch_conv(L, in, bs, 6, 6, res, out, s);
  };
  UMT_TRY(conv(B[0].conv1, w.rin, nullptr, w.r1));
  UMT_TRY(conv(B[0].conv2, w.r1, w.rin, w.r2));
  UMT_TRY(conv(B[1].conv1, w.r2, nullptr, w.r1));
  UMT_TRY(conv(B[1].conv2, w.r1, w.r2, w.r3));
  UMT_TRY(launch_conv(m->reg_out[head], w.r3, bs, 6, 6, nullptr, w.rout, s));
  UMT_TRY(launch_avgpool(w.rout, 64, nraw, bs, w.raw, s));
  if (raw_out) UMT_CHECK_HIP(hipMemcpyAsync(raw_out, w.raw, (size_t)bs * nraw * sizeof(float), hipMemcpyDeviceToDevice, s));
  UMT_TRY(launch_decode(w.raw, nraw, head == UMT_HEAD_PREDICT_SCALE, cam0, hand_idx, bs, joint_angles, wrist_xfs, sigmas,
                        skel_scales, m->status, s));
  return 0;
}

}  // namespace umt

// =====================================================================================  C ABI
extern "C" const char* umt_last_error(void) { return g_err; }
extern "C" int umt_abi_version(void) { return UMT_ABI_VERSION; }

extern "C" int umt_model_create(umt_model** out) {
  UMT_REQUIRE(out, UMT_ERR_INVALID, "umt_model_create: null out pointer");
  umt_model* m = new umt_model();
  m->conv_flags = conv_flags_from_env();
  m->stage1_sub = getenv("UMT_STAGE1_SUB") ? atoi(getenv("UMT_STAGE1_SUB")) : 0;
  build_spec(m);
  *out = m;
  return 0;
}

extern "C" int umt_model_destroy(umt_model* m) {
  if (!m) return 0;
  if (m->device >= 0) {      // set as soon as finalize starts to allocate; free whatever exists
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(m->device) == hipSuccess) {
      if (m->slab) (void)hipFree(m->slab);
      if (m->slab_h) (void)hipFree(m->slab_h);
      if (m->mem) (void)hipFree(m->mem);
      if (m->prev_ext) (void)hipFree(m->prev_ext);
      if (m->mem_owner) (void)hipFree(m->mem_owner);
      for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
      for (hipEvent_t e : m->lev_pool) (void)hipEventDestroy(e);
      (void)hipSetDevice(prev);
    }
  }
  delete m;
  return 0;
}

extern "C" int umt_model_param_count(const umt_model* m, int* count) {
  UMT_REQUIRE(m && count, UMT_ERR_INVALID, "null argument");
  *count = (int)m->spec.size();
  return 0;
}

extern "C" int umt_model_param_info(const umt_model* m, int index, const char** key, int* ndim, int64_t shape[4],
                                    int* is_int64) {
  UMT_REQUIRE(m && key && ndim && shape && is_int64, UMT_ERR_INVALID, "null argument");
  UMT_REQUIRE(index >= 0 && index < (int)m->spec.size(), UMT_ERR_INVALID, "parameter index %d out of range", index);
  const ParamSpec& p = m->spec[index];
  *key = p.key.c_str();
  *ndim = p.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  *is_int64 = p.is_int64 ? 1 : 0;
  return 0;
}

extern "C" int umt_model_set_param(umt_model* m, const char* key, const void* host_data, int ndim, const int64_t* shape,
                                   int is_int64) {
  UMT_REQUIRE(m && key && host_data, UMT_ERR_INVALID, "null argument");
  UMT_REQUIRE(!m->finalized, UMT_ERR_STATE, "model already finalized");
  auto it = m->index.find(key);
  UMT_REQUIRE(it != m->index.end(), UMT_ERR_KEY, "Unexpected key in state_dict: \"%s\"", key);
  const ParamSpec& p = m->spec[it->second];
  UMT_REQUIRE(ndim == p.ndim && (is_int64 != 0) == p.is_int64, UMT_ERR_KEY,
              "size mismatch for %s: got %d dims (int64=%d), expected %d dims (int64=%d)", key, ndim, is_int64, p.ndim,
              (int)p.is_int64);
  for (int i = 0; i < ndim; ++i)
    UMT_REQUIRE(shape && shape[i] == p.shape[i], UMT_ERR_KEY, "size mismatch for %s: dim %d is %lld, expected %lld", key,
                i, (long long)(shape ? shape[i] : -1), (long long)p.shape[i]);
  if (!p.is_int64) {
    const float* src = (const float*)host_data;
    m->values[it->second].assign(src, src + p.numel());
  }
  m->seen[it->second] = 1;
  return 0;
}

// packs and uploads with `device` current; on failure the caller (umt_model_finalize) releases what was allocated
static int finalize_on_device(umt_model* m, int device) {
  UMT_TRY(conv_prepare_device());      // raise the dynamic-LDS limit of the kernels that need > 64 KB (idempotent)
  m->device = device;                  // from here on umt_model_destroy / the failure path free the slabs

  std::vector<float> host;
  std::vector<Packed> packed;
  host.reserve(6u << 20);
  // stem: weights as [tap][32]
  const std::string bb = "_feature_extractor._image_backbone";
  size_t stem_off[4];
  {
    const std::vector<float>&w = val(m, bb + ".0._layers.0.0.weight"), &b = val(m, bb + ".0._layers.0.0.bias");
    const std::string bn = bb + ".0._layers.0.1";
    stem_off[0] = host.size();
    host.resize(host.size() + 9 * 32);
    for (int tap = 0; tap < 9; ++tap)
      for (int c = 0; c < 32; ++c) host[stem_off[0] + tap * 32 + c] = w[c * 9 + tap];
    stem_off[1] = host.size();
    host.insert(host.end(), b.begin(), b.end());
    stem_off[2] = host.size();
    host.resize(host.size() + 32);
    stem_off[3] = host.size();
    host.resize(host.size() + 32);
    const std::vector<float>&g = val(m, bn + ".weight"), &be = val(m, bn + ".bias"), &mu = val(m, bn + ".running_mean"),
                            &var = val(m, bn + ".running_var");
    for (int c = 0; c < 32; ++c) {
      const float alpha = g[c] / sqrtf(var[c] + BN_EPS);
      host[stem_off[2] + c] = alpha;
      host[stem_off[3] + c] = be[c] - mu[c] * alpha;
    }
  }
  int nblocks = 0;
  for (int s = 0; s < 4; ++s) nblocks += N_STAGE_BLOCKS[s];
  m->blocks.resize(nblocks);
  m->block_hw.resize(nblocks);
  {
    const int stride[4] = {1, 2, 2, 2};
    int hw = 48, bi = 0;
    for (int s = 0; s < 4; ++s)
      for (int b = 0; b < N_STAGE_BLOCKS[s]; ++b, ++bi) {
        const bool first = b == 0;
        const std::string p = bb + ".0._layers." + std::to_string(s + 1) + "." + std::to_string(b);
        m->block_hw[bi] = hw;
        pack_block(m, host, packed, &m->blocks[bi], p, first ? stride[s] : 1, first && s > 0);
        if (first) hw /= stride[s];
      }
  }
  pack_conv(m, host, packed, &m->proj, bb + ".1.weight", bb + ".1.bias", "", 1, 0);
  std::vector<uint16_t> host_h;
  std::vector<PackedH> packed_h;
  {
    int bi = 0;
    for (int s = 0; s < 4; ++s)
      for (int b = 0; b < N_STAGE_BLOCKS[s]; ++b, ++bi) {
        const std::string p = bb + ".0._layers." + std::to_string(s + 1) + "." + std::to_string(b);
        BlockLayers& B = m->blocks[bi];
        pack_conv_bf16(m, host_h, packed_h, &B.conv1, p + ".conv1.weight");
        pack_conv_bf16(m, host_h, packed_h, &B.conv2, p + ".conv2.weight");
        if (B.has_down) pack_conv_bf16(m, host_h, packed_h, &B.down, p + ".downsample.0.weight");
      }
    pack_conv_bf16(m, host_h, packed_h, &m->proj, bb + ".1.weight");
  }
  std::vector<PackedH> packed_x3;     // same slab, after the bf16 copies
  {
    int bi = 0;
    for (int s = 0; s < 4; ++s)
      for (int b = 0; b < N_STAGE_BLOCKS[s]; ++b, ++bi) {
        const std::string p = bb + ".0._layers." + std::to_string(s + 1) + "." + std::to_string(b);
        BlockLayers& B = m->blocks[bi];
        if (B.conv1.cin_s % 16 == 0) pack_conv_x3(m, host_h, packed_x3, &B.conv1, p + ".conv1.weight");
        if (B.conv2.cin_s % 16 == 0) pack_conv_x3(m, host_h, packed_x3, &B.conv2, p + ".conv2.weight");
        if (B.has_down && B.down.cin_s % 16 == 0) pack_conv_x3(m, host_h, packed_x3, &B.down, p + ".downsample.0.weight");
      }
  }
  const std::string fu = "_feature_extractor._multi_view_fusion";
  pack_conv(m, host, packed, &m->fuse1, fu + ".0.weight", fu + ".0.bias", fu + ".1", 1, 1);
  pack_conv(m, host, packed, &m->fuse2, fu + ".3.weight", fu + ".3.bias", fu + ".4", 1, 1);
  pack_conv(m, host, packed, &m->fuse3, fu + ".6.weight", fu + ".6.bias", "", 1, 0);
  std::vector<PackedSeq> packed_seq;
  for (int i = 0; i < 3; ++i) {
    const std::string p = "_temporal._temporal_module." + std::to_string(2 * i);
    pack_conv(m, host, packed, &m->temp[i], p + ".weight", p + ".bias", "", 1, i < 2 ? 1 : 0);
    pack_conv_seq(m, host, packed_seq, &m->temp[i], p + ".weight");
  }
  const char* heads[2] = {"_regressor_k", "_regressor_u"};
  for (int h = 0; h < 2; ++h) {
    const std::string p = std::string(heads[h]) + "._pose_regression_layers";
    for (int b = 0; b < 2; ++b) {
      const std::string pb = p + "." + std::to_string(b);
      pack_block(m, host, packed, &m->reg_blocks[h][b], pb, 1, false);
      pack_conv_seq(m, host, packed_seq, &m->reg_blocks[h][b].conv1, pb + ".conv1.weight");
      pack_conv_seq(m, host, packed_seq, &m->reg_blocks[h][b].conv2, pb + ".conv2.weight");
    }
    pack_conv(m, host, packed, &m->reg_out[h], p + ".2.weight", p + ".2.bias", "", 1, 0);
    pack_conv_seq(m, host, packed_seq, &m->reg_out[h], p + ".2.weight");
    // split-operand copies for the non-fp32 modes' large batches (regress_rows)
    for (int b = 0; b < 2; ++b) {
      const std::string pb = p + "." + std::to_string(b);
      BlockLayers& B = m->reg_blocks[h][b];
      if (B.conv1.cin_s % 16 == 0) pack_conv_x3(m, host_h, packed_x3, &B.conv1, pb + ".conv1.weight");
      if (B.conv2.cin_s % 16 == 0) pack_conv_x3(m, host_h, packed_x3, &B.conv2, pb + ".conv2.weight");
    }
  }
  // skeleton encoder
  size_t sk_off[4];
  {
    const std::vector<float>&w = val(m, "_skeleton_enc._layers.0.weight"), &b = val(m, "_skeleton_enc._layers.0.bias");
    sk_off[0] = host.size();
    host.insert(host.end(), w.begin(), w.end());
    sk_off[1] = host.size();
    host.insert(host.end(), b.begin(), b.end());
    const std::string bn = "_skeleton_enc._layers.2";
    const std::vector<float>&g = val(m, bn + ".weight"), &be = val(m, bn + ".bias"), &mu = val(m, bn + ".running_mean"),
                            &var = val(m, bn + ".running_var");
    sk_off[2] = host.size();
    host.resize(host.size() + 4);
    sk_off[3] = host.size();
    host.resize(host.size() + 4);
    for (int c = 0; c < 4; ++c) {
      const float alpha = g[c] / sqrtf(var[c] + BN_EPS);
      host[sk_off[2] + c] = alpha;
      host[sk_off[3] + c] = be[c] - mu[c] * alpha;
    }
  }
  const size_t status_off = host.size();      // UMT_STATUS_WORDS int32 status words, zero
  host.resize(host.size() + 4, 0.f);
  // every packed array starts 16-B aligned: all sizes above are multiples of 4 floats
  for (const Packed& p : packed)
    UMT_REQUIRE(p.w_off % 4 == 0 && p.b_off % 4 == 0, UMT_ERR_STATE, "internal: packed offsets not 16-B aligned");
  UMT_CHECK_HIP(hipMalloc((void**)&m->slab, host.size() * sizeof(float)));
  UMT_CHECK_HIP(hipMemcpy(m->slab, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  for (const Packed& p : packed) {
    p.layer->w = m->slab + p.w_off;
    p.layer->bias = m->slab + p.b_off;
    p.layer->scale = m->slab + p.s_off;
    p.layer->shift = m->slab + p.h_off;
  }
  UMT_CHECK_HIP(hipMalloc((void**)&m->slab_h, host_h.size() * sizeof(uint16_t)));
  UMT_CHECK_HIP(hipMemcpy(m->slab_h, host_h.data(), host_h.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  for (const PackedSeq& p : packed_seq) p.layer->w16 = m->slab + p.off;   // sizes are multiples of 4 floats: 16-B aligned
  for (const PackedH& p : packed_h) p.layer->wh = m->slab_h + p.off;   // every size is a multiple of 8 bf16 = 16 B
  for (const PackedH& p : packed_x3) p.layer->wx3 = m->slab_h + p.off;
  m->stem = StemParams{m->slab + stem_off[0], m->slab + stem_off[1], m->slab + stem_off[2], m->slab + stem_off[3],
                       (m->conv_flags & CONV_STEM_VALU) ? 1 : (m->conv_flags & STEM_NO_SPLIT) ? 2 : 0};
  m->skel = SkelParams{m->slab + sk_off[0], m->slab + sk_off[1], m->slab + sk_off[2], m->slab + sk_off[3]};
  m->status = reinterpret_cast<int32_t*>(m->slab + status_off);
  return 0;
}

extern "C" int umt_model_finalize(umt_model* m, int device) {
  UMT_REQUIRE(m, UMT_ERR_INVALID, "null model handle");
  UMT_REQUIRE(!m->finalized, UMT_ERR_STATE, "model already finalized");
  for (size_t i = 0; i < m->spec.size(); ++i)
    UMT_REQUIRE(m->seen[i], UMT_ERR_KEY, "Missing key in state_dict: \"%s\"", m->spec[i].key.c_str());
  int ndev = 0, prev = -1;
  UMT_CHECK_HIP(hipGetDeviceCount(&ndev));
  UMT_REQUIRE(device >= 0 && device < ndev, UMT_ERR_INVALID, "device %d not available (%d devices)", device, ndev);
  hipDeviceProp_t prop;
  UMT_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  UMT_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, UMT_ERR_UNSUPPORTED,
              "libumetrack_hip is built for gfx950 only, device %d is %s", device, prop.gcnArchName);
  UMT_CHECK_HIP(hipGetDevice(&prev));
  UMT_CHECK_HIP(hipSetDevice(device));
  const int rc = finalize_on_device(m, device);
  if (rc != 0) {      // leave the handle as it was before the call (the error text is already set); a retry packs afresh
    if (m->slab) (void)hipFree(m->slab);
    if (m->slab_h) (void)hipFree(m->slab_h);
    m->slab = nullptr;
    m->slab_h = nullptr;
    m->status = nullptr;
    m->device = -1;
    m->blocks.clear();
    m->block_hw.clear();
  } else {
    m->finalized = true;
    std::vector<std::vector<float>>().swap(m->values);
  }
  (void)hipSetDevice(prev);      // the caller's current device is left as it was
  return rc;
}

extern "C" int umt_model_reserve_memory(umt_model* m, int rows, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(rows >= 0, UMT_ERR_INVALID, "reserve_memory: rows = %d", rows);
  if (rows <= m->mem_rows) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (rows > m->mem_cap) {
    const int cap = rows > 2 * m->mem_cap ? rows : 2 * m->mem_cap;
    float *nm = nullptr, *np = nullptr;
    unsigned long long* no = nullptr;
    hipError_t e1 = hipMalloc((void**)&nm, (size_t)cap * 18 * 36 * sizeof(float));
    hipError_t e2 = e1 == hipSuccess ? hipMalloc((void**)&np, (size_t)cap * 16 * sizeof(float)) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipMalloc((void**)&no, (size_t)cap * sizeof(unsigned long long)) : e2;
    if (e3 != hipSuccess) {      // nothing is kept of a failed growth; the old state stays valid
      if (nm) (void)hipFree(nm);
      if (np) (void)hipFree(np);
      return set_error(UMT_ERR_HIP, "reserve_memory: hipMalloc for %d rows failed: %s", cap, hipGetErrorString(e3));
    }
    UMT_CHECK_HIP(hipMemsetAsync(no, 0, (size_t)cap * sizeof(unsigned long long), s));   // epoch 0 is never used
    if (m->mem_rows > 0) {
      UMT_CHECK_HIP(hipMemcpyAsync(nm, m->mem, (size_t)m->mem_rows * 18 * 36 * sizeof(float), hipMemcpyDeviceToDevice, s));
      UMT_CHECK_HIP(hipMemcpyAsync(np, m->prev_ext, (size_t)m->mem_rows * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    UMT_CHECK_HIP(hipStreamSynchronize(s));   // old buffers are freed below; growth is rare (per new hand slot)
    if (m->mem) UMT_CHECK_HIP(hipFree(m->mem));
    if (m->prev_ext) UMT_CHECK_HIP(hipFree(m->prev_ext));
    if (m->mem_owner) UMT_CHECK_HIP(hipFree(m->mem_owner));
    m->mem = nm;
    m->prev_ext = np;
    m->mem_owner = no;
    m->mem_cap = cap;
  }
  // new rows start at zero (temporal.py:106-120)
  UMT_CHECK_HIP(hipMemsetAsync(m->mem + (size_t)m->mem_rows * 18 * 36, 0, (size_t)(rows - m->mem_rows) * 18 * 36 * sizeof(float), s));
  UMT_CHECK_HIP(hipMemsetAsync(m->prev_ext + (size_t)m->mem_rows * 16, 0, (size_t)(rows - m->mem_rows) * 16 * sizeof(float), s));
  m->mem_rows = rows;
  return 0;
}

extern "C" int umt_model_memory_rows(const umt_model* m, int* rows) {
  UMT_REQUIRE(m && rows, UMT_ERR_INVALID, "null argument");
  *rows = m->mem_rows;
  return 0;
}

extern "C" int umt_model_clear_memory(umt_model* m) {
  UMT_REQUIRE(m, UMT_ERR_INVALID, "null model handle");
  m->mem_rows = 0;   // capacity is kept; rows are re-zeroed when reserved again
  return 0;
}

extern "C" int umt_model_read_memory(const umt_model* m, float* mem_dev, float* prev_ext_dev, void* stream) {
  UMT_TRY(check_ready(m));
  if (m->mem_rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (mem_dev)
    UMT_CHECK_HIP(hipMemcpyAsync(mem_dev, m->mem, (size_t)m->mem_rows * 18 * 36 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (prev_ext_dev)
    UMT_CHECK_HIP(hipMemcpyAsync(prev_ext_dev, m->prev_ext, (size_t)m->mem_rows * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int64_t umt_backbone_workspace_bytes(int n_crops) {
  Carver cv(nullptr, 0);
  BackboneWs w;
  return carve_backbone(cv, n_crops > 0 ? n_crops : 1, &w);
}

extern "C" int64_t umt_head_workspace_bytes(int n_crops, int n_samples) {
  Carver cv(nullptr, 0);
  HeadWs w;
  return carve_head(cv, n_crops > 0 ? n_crops : 1, n_samples > 0 ? n_samples : 1, &w);
}

extern "C" int64_t umt_forward_workspace_bytes(int n_crops, int n_samples) {
  return umt_backbone_workspace_bytes(n_crops) + umt_head_workspace_bytes(n_crops, n_samples);
}

static int check_ws(const char* what, void* ws, int64_t have, int64_t need) {
  UMT_REQUIRE(ws, UMT_ERR_WORKSPACE, "%s: null workspace", what);
  UMT_REQUIRE(((uintptr_t)ws & 255) == 0, UMT_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", what);
  UMT_REQUIRE(have >= need, UMT_ERR_WORKSPACE, "%s: workspace has %lld bytes, needs %lld", what, (long long)have,
              (long long)need);
  return 0;
}

extern "C" int umt_stem_forward(umt_model* m, const float* images, int n, int precision, void* out, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(precision >= UMT_PREC_FP32 && precision <= UMT_PREC_FP32_BF16X3, UMT_ERR_INVALID, "stem: precision = %d", precision);
  UMT_REQUIRE(n > 0 && images && out, UMT_ERR_INVALID, "stem: n=%d or null pointer", n);
  if (precision == UMT_PREC_BF16) return launch_stem_bf16(m->stem, images, n, (uint16_t*)out, (hipStream_t)stream);
  return launch_stem(m->stem, images, n, (float*)out, (hipStream_t)stream);
}

extern "C" int umt_backbone_forward(umt_model* m, const float* images, int n, int precision, float* features,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(precision >= UMT_PREC_FP32 && precision <= UMT_PREC_FP32_BF16X3, UMT_ERR_INVALID, "backbone: precision = %d", precision);
  UMT_REQUIRE(n > 0 && images && features, UMT_ERR_INVALID, "backbone: n=%d or null pointer", n);
  UMT_TRY(check_ws("backbone", workspace, workspace_bytes, umt_backbone_workspace_bytes(n)));
  Carver cv(workspace, workspace_bytes);
  BackboneWs w;
  carve_backbone(cv, n, &w);
  hipStream_t s = (hipStream_t)stream;
  UMT_TRY(backbone_any(m, precision, images, n, w, s));
  UMT_TRY(launch_rows_to_nchw(w.feat, 72, 0, 72, n, 36, features, s));
  return 0;
}

extern "C" int umt_fuse_views(umt_model* m, const float* features, const float* intrinsics, const float* extrinsics,
                              const int64_t* sample_range, int n, int bs, float* fused, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(n > 0 && bs > 0 && features && intrinsics && extrinsics && sample_range && fused, UMT_ERR_INVALID,
              "fuse_views: bad argument");
  UMT_TRY(check_ws("fuse_views", workspace, workspace_bytes, umt_head_workspace_bytes(n, bs)));
  Carver cv(workspace, workspace_bytes);
  HeadWs w;
  carve_head(cv, n, bs, &w);
  hipStream_t s = (hipStream_t)stream;
  UMT_TRY(launch_nchw_to_rows(features, 72, n, 36, w.feat_in, 72, 0, s));
  UMT_TRY(fuse_rows(m, w.feat_in, intrinsics, extrinsics, sample_range, n, bs, w, s));
  UMT_TRY(launch_rows_to_nchw(w.tin, 96, 18, 72, bs, 36, fused, s));
  return 0;
}

extern "C" int umt_temporal_step(umt_model* m, const float* fused, const float* cam0_extrinsics,
                                 const int64_t* memory_idx, const uint8_t* use_memory, int bs, float* out,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(bs > 0 && fused && cam0_extrinsics && memory_idx && use_memory && out, UMT_ERR_INVALID,
              "temporal_step: bad argument");
  UMT_TRY(check_ws("temporal_step", workspace, workspace_bytes, umt_head_workspace_bytes(1, bs)));
  Carver cv(workspace, workspace_bytes);
  HeadWs w;
  carve_head(cv, 1, bs, &w);
  hipStream_t s = (hipStream_t)stream;
  UMT_CHECK_HIP(hipMemsetAsync(w.tin, 0, (size_t)bs * 36 * 96 * sizeof(float), s));
  UMT_TRY(launch_nchw_to_rows(fused, 72, bs, 36, w.tin, 96, 18, s));
  UMT_TRY(temporal_rows(m, cam0_extrinsics, memory_idx, use_memory, bs, w, w.f2, 72, s));
  UMT_TRY(launch_rows_to_nchw(w.f2, 72, 0, 72, bs, 36, out, s));
  return 0;
}

extern "C" int umt_regress_decode(umt_model* m, int head, const float* temporal, const float* skel_axes,
                                  const float* skel_rest, int n_skel, const float* cam0_extrinsics,
                                  const int64_t* hand_idx, int bs, float* joint_angles, float* wrist_xfs, float* sigmas,
                                  float* skel_scales, float* raw, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(head == UMT_HEAD_KNOWN_SKELETON || head == UMT_HEAD_PREDICT_SCALE, UMT_ERR_INVALID, "head = %d", head);
  UMT_REQUIRE(bs > 0 && temporal && cam0_extrinsics && hand_idx && joint_angles && wrist_xfs && sigmas, UMT_ERR_INVALID,
              "regress_decode: bad argument");
  UMT_TRY(check_ws("regress_decode", workspace, workspace_bytes, umt_head_workspace_bytes(1, bs)));
  Carver cv(workspace, workspace_bytes);
  HeadWs w;
  carve_head(cv, 1, bs, &w);
  hipStream_t s = (hipStream_t)stream;
  const int cs = head == UMT_HEAD_KNOWN_SKELETON ? 80 : 72;
  UMT_CHECK_HIP(hipMemsetAsync(w.rin, 0, (size_t)bs * 36 * 80 * sizeof(float), s));
  UMT_TRY(launch_nchw_to_rows(temporal, 72, bs, 36, w.rin, cs, 0, s));
  UMT_TRY(regress_rows(m, head, skel_axes, skel_rest, n_skel, cam0_extrinsics, hand_idx, bs, w, joint_angles, wrist_xfs,
                       sigmas, skel_scales, raw, s));
  return 0;
}

extern "C" int umt_model_forward(umt_model* m, int head, int precision, const float* images, const float* intrinsics,
                                 const float* extrinsics, const int64_t* sample_range, const int64_t* memory_idx,
                                 const uint8_t* use_memory, const int64_t* hand_idx, int n, int bs,
                                 const float* skel_axes, const float* skel_rest, int n_skel, float* joint_angles,
                                 float* wrist_xfs, float* sigmas, float* skel_scales, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(precision >= UMT_PREC_FP32 && precision <= UMT_PREC_FP32_BF16X3, UMT_ERR_INVALID, "forward: precision = %d", precision);
  UMT_REQUIRE(head == UMT_HEAD_KNOWN_SKELETON || head == UMT_HEAD_PREDICT_SCALE, UMT_ERR_INVALID, "head = %d", head);
  UMT_REQUIRE(n > 0 && bs > 0 && bs <= n && n <= 2 * bs, UMT_ERR_INVALID, "forward: n=%d crops for bs=%d samples", n, bs);
  UMT_REQUIRE(images && intrinsics && extrinsics && sample_range && memory_idx && use_memory && hand_idx &&
                  joint_angles && wrist_xfs && sigmas,
              UMT_ERR_INVALID, "forward: null pointer");
  UMT_TRY(check_ws("forward", workspace, workspace_bytes, umt_forward_workspace_bytes(n, bs)));
  Carver cv(workspace, workspace_bytes);
  BackboneWs bw;
  HeadWs hw;
  carve_backbone(cv, n, &bw);
  carve_head(cv, n, bs, &hw);
  hipStream_t s = (hipStream_t)stream;
  const int cs = head == UMT_HEAD_KNOWN_SKELETON ? 80 : 72;
  hipEvent_t* ev = nullptr;
  if (m->timing) {
    UMT_TRY(timing_slot(m, &ev));
    UMT_CHECK_HIP(hipEventRecord(ev[0], s));
  }
  UMT_TRY(backbone_any(m, precision, images, n, bw, s));
  if (ev) UMT_CHECK_HIP(hipEventRecord(ev[1], s));
  UMT_TRY(fuse_rows(m, bw.feat, intrinsics, extrinsics, sample_range, n, bs, hw, s));
  UMT_TRY(launch_gather_cam0(extrinsics, sample_range, n, bs, hw.cam0, s));
  UMT_TRY(temporal_rows(m, hw.cam0, memory_idx, use_memory, bs, hw, hw.rin, cs, s, true));   // zeroes rin's channels 72.. too
  UMT_TRY(regress_rows(m, head, skel_axes, skel_rest, n_skel, hw.cam0, hand_idx, bs, hw, joint_angles, wrist_xfs, sigmas,
                       skel_scales, nullptr, s, precision != UMT_PREC_FP32));
  if (ev) UMT_CHECK_HIP(hipEventRecord(ev[2], s));
  return 0;
}

// umt_head_sequence's workspace: head_seq.hip's part, the regressor's input rows for all t x rows hands, and the
// regressor's intermediates for one batch of at most SEQ_REGRESS_BATCH hands
constexpr int64_t SEQ_REGRESS_BATCH = 8192;
struct SeqRegressWs {
  float *rin, *r1, *r2, *r3, *rout;
};
static int64_t carve_seq_regress(Carver& cv, int64_t n, SeqRegressWs* w) {
  const int64_t px = std::min(n, SEQ_REGRESS_BATCH) * 36;
  w->rin = cv.take<float>(n * 36 * 80);
  w->r1 = cv.take<float>(px * 80);
  w->r2 = cv.take<float>(px * 80);
  w->r3 = cv.take<float>(px * 80);
  w->rout = cv.take<float>(px * 64);
  return cv.off;
}

extern "C" int64_t umt_head_sequence_workspace_bytes(int t, int rows) {
  t = t > 0 ? t : 1;
  rows = rows > 0 ? rows : 1;
  Carver cv(nullptr, 0);
  SeqRegressWs w;
  return round_up64(head_sequence_workspace_bytes(t, rows), 256) + carve_seq_regress(cv, (int64_t)t * rows, &w);
}

extern "C" int umt_head_sequence(umt_model* m, int head, const float* fused_t, const float* cam0_t,
                                 const uint8_t* use_memory_t, const int64_t* hand_idx, const float* skel_axes,
                                 const float* skel_rest, int n_skel, int t, int rows, float* joint_angles_t,
                                 float* wrist_xfs_t, float* sigmas_t, float* skel_scales_t, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(head == UMT_HEAD_KNOWN_SKELETON || head == UMT_HEAD_PREDICT_SCALE, UMT_ERR_INVALID, "head = %d", head);
  UMT_REQUIRE(t > 0 && rows > 0 && (int64_t)t * rows < ((int64_t)1 << 24), UMT_ERR_INVALID, "head_sequence: t=%d rows=%d", t, rows);
  UMT_REQUIRE(fused_t && cam0_t && use_memory_t && hand_idx && joint_angles_t && wrist_xfs_t && sigmas_t, UMT_ERR_INVALID,
              "head_sequence: null pointer");
  UMT_REQUIRE(rows <= m->mem_rows, UMT_ERR_STATE, "head_sequence: %d rows but %d memory rows reserved (umt_model_reserve_memory)",
              rows, m->mem_rows);
  const bool known = head == UMT_HEAD_KNOWN_SKELETON;
  if (known) {
    UMT_REQUIRE(skel_axes && skel_rest, UMT_ERR_INVALID, "known-skeleton head needs skeleton data");
    UMT_REQUIRE(n_skel == 1 || n_skel == rows, UMT_ERR_INVALID, "n_skel must be 1 or rows (%d), got %d", rows, n_skel);
  } else {
    UMT_REQUIRE(skel_scales_t, UMT_ERR_INVALID, "scale head needs a skel_scales buffer");
  }
  UMT_TRY(check_ws("head_sequence", workspace, workspace_bytes, umt_head_sequence_workspace_bytes(t, rows)));
  hipStream_t s = (hipStream_t)stream;
  const BlockLayers* B = m->reg_blocks[head];
  const ConvLayer convs[4] = {B[0].conv1, B[0].conv2, B[1].conv1, B[1].conv2};
  const int nraw = known ? 62 : 63;
  float* raw = nullptr;
  const int64_t n = (int64_t)t * rows;
  if (m->conv_flags & SEQ_ONE_KERNEL) {
    UMT_TRY(launch_head_sequence(m->temp, convs, m->reg_out[head], known ? &m->skel : nullptr, skel_axes, skel_rest, n_skel,
                                 fused_t, cam0_t, use_memory_t, t, rows, nraw, m->mem, m->prev_ext, m->status, &raw, nullptr, 0,
                                 workspace, s));
  } else {
    // the serial part (temporal module, one workgroup per memory row) leaves the regressor's input rows behind; the
    // regressor, which feeds nothing back, then runs over all t x rows hands with the batched kernels
    const int64_t seq_bytes = round_up64(head_sequence_workspace_bytes(t, rows), 256);
    Carver cv((char*)workspace + seq_bytes, workspace_bytes - seq_bytes);
    SeqRegressWs w;
    carve_seq_regress(cv, n, &w);
    const int cs = known ? 80 : 72;
    UMT_TRY(launch_head_sequence(m->temp, convs, m->reg_out[head], known ? &m->skel : nullptr, skel_axes, skel_rest, n_skel,
                                 fused_t, cam0_t, use_memory_t, t, rows, nraw, m->mem, m->prev_ext, m->status, &raw, w.rin, cs,
                                 workspace, s));
    for (int64_t b0 = 0; b0 < n; b0 += SEQ_REGRESS_BATCH) {
      const int bs = (int)std::min(SEQ_REGRESS_BATCH, n - b0);
      const float* rin = w.rin + b0 * 36 * cs;
      UMT_TRY(launch_conv(convs[0], rin, bs, 6, 6, nullptr, w.r1, s));
      UMT_TRY(launch_conv(convs[1], w.r1, bs, 6, 6, rin, w.r2, s));
      UMT_TRY(launch_conv(convs[2], w.r2, bs, 6, 6, nullptr, w.r1, s));
      UMT_TRY(launch_conv(convs[3], w.r1, bs, 6, 6, w.r2, w.r3, s));
      UMT_TRY(launch_conv(m->reg_out[head], w.r3, bs, 6, 6, nullptr, w.rout, s));
      UMT_TRY(launch_avgpool(w.rout, 64, nraw, bs, raw + b0 * nraw, s));
    }
  }
  // Procrustes decode + wrist recovery for all t x rows outputs at once (hand_idx is time-major like the outputs)
  UMT_TRY(launch_decode(raw, nraw, !known, cam0_t, hand_idx, t * rows, joint_angles_t, wrist_xfs_t, sigmas_t, skel_scales_t,
                        m->status, s));
  return 0;
}

extern "C" int umt_model_timing_enable(umt_model* m, int on) {
  UMT_REQUIRE(m, UMT_ERR_INVALID, "null model handle");
  m->timing = on != 0;
  m->ev_head = m->ev_live = 0;      // events still in flight from an earlier enable are simply recorded over later
  m->t_backbone_ms = m->t_forward_ms = 0;
  m->t_calls = 0;
  return 0;
}

extern "C" int umt_model_timing_read(umt_model* m, double* backbone_ms, double* forward_ms, int* calls) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(backbone_ms && forward_ms && calls, UMT_ERR_INVALID, "null argument");
  while (m->ev_live > 0) UMT_TRY(timing_fold_oldest(m));
  *backbone_ms = m->t_backbone_ms;
  *forward_ms = m->t_forward_ms;
  *calls = m->t_calls;
  return 0;
}

extern "C" int umt_model_layer_timing_enable(umt_model* m, int on) {
  UMT_REQUIRE(m, UMT_ERR_INVALID, "null model handle");
  m->layer_timing = on != 0;
  m->lev_marks.clear();
  m->lev_acc.clear();
  return 0;
}

extern "C" int umt_model_status(umt_model* m, int32_t* status_host, int clear, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(status_host, UMT_ERR_INVALID, "model_status: null output");
  hipStream_t s = (hipStream_t)stream;
  UMT_CHECK_HIP(hipMemcpyAsync(status_host, m->status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (clear) UMT_CHECK_HIP(hipMemsetAsync(m->status, 0, sizeof(int32_t), s));
  UMT_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int umt_model_status_async(umt_model* m, int32_t* status_dev, int clear, void* stream) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(status_dev, UMT_ERR_INVALID, "model_status_async: null output");
  hipStream_t s = (hipStream_t)stream;
  UMT_CHECK_HIP(hipMemcpyAsync(status_dev, m->status, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (clear) UMT_CHECK_HIP(hipMemsetAsync(m->status, 0, sizeof(int32_t), s));
  return 0;
}

extern "C" int umt_model_layer_timing_read(umt_model* m, int max_layers, char* names, double* flop, double* ms,
                                           int* launches, int* n_layers) {
  UMT_TRY(check_ready(m));
  UMT_REQUIRE(max_layers > 0 && names && flop && ms && launches && n_layers, UMT_ERR_INVALID, "layer_timing_read: bad argument");
  // launch k of a backbone pass is aggregated with launch k of every other pass (the schedule is fixed per precision)
  UMT_TRY(layer_marks_fold(m));
  for (int i = 0; i < max_layers; ++i) { flop[i] = 0; ms[i] = 0; launches[i] = 0; names[(size_t)i * 48] = 0; }
  int count = 0;
  for (size_t pos = 0; pos < m->lev_acc.size() && (int)pos < max_layers; ++pos) {
    const umt_model::LayerAcc& a = m->lev_acc[pos];
    memcpy(names + pos * 48, a.name, 48);
    flop[pos] = a.flop;
    ms[pos] = a.ms;
    launches[pos] = a.launches;
    count = (int)pos + 1;
  }
  *n_layers = count;
  return 0;
}
