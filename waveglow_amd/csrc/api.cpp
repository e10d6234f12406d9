// Host side of libwaveglow_amd: C ABI (include/waveglow_amd.h), weight packing, launch sequencing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/waveglow_amd.h"
#include "wg_common.h"

using namespace wg;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return fail(WG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// fp64 Gauss-Jordan inverse with partial pivoting; also log|det| and sign.
bool invert(const std::vector<double>& A, int n, std::vector<double>& inv, double& logabsdet, int& sign) {
  std::vector<double> M(A);
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  logabsdet = 0.0;
  sign = 1;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = r;
    if (M[(size_t)piv * n + c] == 0.0) return false;
    if (piv != c) {
      for (int k = 0; k < n; ++k) {
        std::swap(M[(size_t)piv * n + k], M[(size_t)c * n + k]);
        std::swap(inv[(size_t)piv * n + k], inv[(size_t)c * n + k]);
      }
      sign = -sign;
    }
    const double d = M[(size_t)c * n + c];
    logabsdet += std::log(std::fabs(d));
    if (d < 0) sign = -sign;
    for (int k = 0; k < n; ++k) {
      M[(size_t)c * n + k] /= d;
      inv[(size_t)c * n + k] /= d;
    }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) {
        M[(size_t)r * n + k] -= f * M[(size_t)c * n + k];
        inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k];
      }
    }
  }
  return true;
}

// tanh(a) = 1 - 2/(1 + 2^(a*kTanhScale)), sigmoid(b) = 1/(1 + 2^(b*kSigmScale)): folded into GEMM-1 rows
constexpr float kTanhScale = 2.8853900817779268f;    // 2*log2(e)
constexpr float kSigmScale = -1.4426950408889634f;   // -log2(e)

struct LayerOffsets {
  size_t wA1, bias1, wA2, bias2, wEs;
  size_t wA1f = 0;   // layer 0 only: in_layers[0] o start folded onto the a0 plane, one gathered K-step ([tap][8] along K)
  size_t wA1x = 0, wA1fx = 0;   // the same two as 16x16x32 fragments (wn_frag16: the 128-column tile's K loop)
};
struct FlowOffsets {
  std::vector<LayerOffsets> layers;
  size_t wstart, bstart, out_init, winv, wfwd;
  size_t wStA = 0;   // start weights as the A fragment of the first layer's residual step (wn_res_a0)
  int c, h;
  double logdet;   // log|det W_k|, NaN when det < 0 (torch.logdet semantics, model.py:63)
};

}  // namespace

struct wg_handle {
  wg_config cfg;
  int device;
  int NS;                 // n_mel * n_group
  std::vector<int> c_k;   // remaining channels per flow (model.py:160-176)
  std::vector<std::string> expected;
  std::map<std::string, HostTensor> tensors;
  bool finalized = false;
  char* d_blob = nullptr;
  size_t blob_bytes = 0;
  char* d_cond = nullptr;   // derived: folded cond_layer o upsample A fragments [flow][layer][phase]...
  char* d_cond16 = nullptr; // ... as 16x16x32 fragments (wn_frag16), same sizes
  size_t cond_layer_bytes = 0, cond_flow_bytes = 0;
  std::vector<FlowOffsets> flows;
  int n_cu = 256;         // multiProcessorCount, read in wg_finalize
  int force_bn = 0;       // WG_FORCE_BN=64|128 (tests): pin the WN tile width instead of choosing by workload size
  bool fold_start = true; // WG_NO_START_FOLD=1 (tests / A-B runs): first WN layer reads x_0 like every other layer
  unsigned long long* dbg_stamps = nullptr;   // diagnostic builds only
  // profiling
  bool prof = false;
  unsigned prof_mask = ~0u;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_class;
  size_t ev_used = 0;
  double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // library-owned streams of the training direction (independent chains of one call run side by side, joined back into
  // the caller's stream before the call returns) and a ring of ordering events for them
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> sync_ev;
  size_t sync_next = 0;
  // events of wg_train_backward's long-lived marks (recorded on one stream, waited on one or two flows later): a pool of
  // their own, one event per role, so that no number of ring events consumed in between can re-record one under a wait
  hipEvent_t mark_ev[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // pinned staging buffers for small host -> device table uploads (wg_train_prepare / wg_train_param_grads): a rotating set,
  // each guarded by an event recorded behind its copy, so a buffer is never rewritten while a copy out of it is pending
  static constexpr int kPins = 4;
  void* pin[kPins] = {nullptr, nullptr, nullptr, nullptr};
  size_t pin_bytes[kPins] = {0, 0, 0, 0};
  hipEvent_t pin_ev[kPins] = {nullptr, nullptr, nullptr, nullptr};
  int pin_next = 0;
};

namespace {

std::vector<int> flow_channels(const wg_config& c) {
  std::vector<int> out;
  int rem = c.n_group;
  for (int k = 0; k < c.n_flows; ++k) {
    if (k % c.n_early_every == 0 && k > 0) rem -= c.n_early_size;
    out.push_back(rem);
  }
  return out;
}

bool is_early(const wg_config& c, int k) { return k % c.n_early_every == 0 && k > 0; }

RowGeom make_geom(const wg_config& c, int B, int L, int T) {
  RowGeom g;
  g.frames = nullptr;
  g.B = B;
  g.L = L;
  g.F = (L + kPhases - 1) / kPhases;
  // guard frames: a dilated tap reaches (phase + dilation) >> 5 frames past an utterance's ends; dilation <= 2^(n_layers-1)
  // (model.py:97): 4 frames up to 8 layers, 8 / 16 for 9 / 10 layers
  const int max_dil = 1 << (c.n_layers - 1);
  g.Gf = (kPhases - 1 + max_dil) / kPhases;
  if (g.Gf < 4) g.Gf = 4;
  g.Fp = g.Gf + g.F + g.Gf;
  g.Rp = (B * g.Fp + 127) / 128 * 128;
  g.R = kPhases * g.Rp + 2 * kRowPad;
  g.T = T;
  return g;
}

struct Workspace {
  _Float16 *melT, *X0, *X1, *A0;
  float *Z, *OUT;
  size_t bytes;
  size_t x_bytes, mel_bytes, a0_bytes;
};

Workspace carve(const wg_handle* h, const RowGeom& g, char* base) {
  Workspace w;
  size_t off = 0;
  const int C = h->cfg.n_channels;
  w.mel_bytes = align_up((size_t)(3 + g.B * (g.T + 6)) * h->cfg.n_mel_channels * 2);
  w.x_bytes = align_up((size_t)C * g.R * 2);
  w.melT = (_Float16*)(base + off); off += w.mel_bytes;
  w.X0 = (_Float16*)(base + off); off += w.x_bytes;
  w.X1 = (_Float16*)(base + off); off += w.x_bytes;
  w.a0_bytes = align_up((size_t)g.R * 128);
  w.A0 = (_Float16*)(base + off); off += w.a0_bytes;
  w.Z = (float*)(base + off); off += align_up((size_t)g.B * g.L * 8 * 4);
  w.OUT = (float*)(base + off); off += align_up((size_t)g.B * g.L * 8 * 4);
  w.bytes = off;
  return w;
}

const HostTensor* find(const wg_handle* h, const std::string& name) {
  auto it = h->tensors.find(name);
  return it == h->tensors.end() ? nullptr : &it->second;
}

// WG_DEBUG_SYNC=1: synchronise after every launch and name it on stderr (fault localisation only)
static bool dbg_sync() { static int v = -1; if (v < 0) { const char* e = getenv("WG_DEBUG_SYNC"); v = e && *e == '1'; } return v == 1; }
#define WG_DBG(stream, what) do { if (dbg_sync()) { hipError_t _e = hipStreamSynchronize(stream); fprintf(stderr, "[wg] %s -> %s\n", what, hipGetErrorString(_e)); fflush(stderr); } } while (0)

struct Prof {
  wg_handle* h;
  hipStream_t s;
  int cls;
  Prof(wg_handle* h_, hipStream_t s_, int cls_) : h(h_), s(s_), cls(cls_) {
    if (h->prof) rec();
  }
  ~Prof() {
    if (h->prof) rec();
  }
  void rec() {
    if (h->ev_used == h->ev.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      h->ev.push_back(e);
      h->ev_class.push_back(0);
    }
    h->ev_class[h->ev_used] = cls;
    hipEventRecord(h->ev[h->ev_used++], s);
  }
};

}  // namespace

namespace wg { extern unsigned long long* g_wgrad_stamps; }   // train.hip

int wg_set_error(int code, const char* msg) { return fail(code, "%s", msg); }

// accessors for the other translation units of the library (train_api.cpp)
const wg_config* wg_internal_config(const wg_handle* h) { return h ? &h->cfg : nullptr; }
const int* wg_internal_flow_channels(const wg_handle* h) { return h ? h->c_k.data() : nullptr; }
int wg_internal_device(const wg_handle* h) { return h ? h->device : -1; }
int wg_internal_n_cu(const wg_handle* h) { return h ? h->n_cu : 256; }   // of the HANDLE's device (wg_create)
wg::RowGeom wg_internal_geom(const wg_handle* h, int B, int L, int T) { return make_geom(h->cfg, B, L, T); }
// one profiling event of class cls on stream s (a no-op unless wg_profile_enable is on); events come in begin/end pairs
void wg_internal_prof_event(wg_handle* h, void* s, int cls) {
  if (!h || !h->prof || !((h->prof_mask >> cls) & 1u)) return;
  if (h->ev_used == h->ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    h->ev.push_back(e);
    h->ev_class.push_back(0);
  }
  h->ev_class[h->ev_used] = cls;
  hipEventRecord(h->ev[h->ev_used++], (hipStream_t)s);
}

// Library-owned stream i (0: second chain, normal priority; 1, 2: work with slack, lowest priority) of the handle's device.
hipStream_t wg_internal_aux_stream(wg_handle* h, int i) {
  if (!h || i < 0 || i > 2) return nullptr;
  if (!h->aux[i]) {
    int lo = 0, hi = 0, prev = -1;
    hipStream_t st = nullptr;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device) (void)hipSetDevice(h->device);
    if (i >= 1 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo) != hipSuccess) st = nullptr;
    }
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    h->aux[i] = st;
  }
  return h->aux[i];
}
// Copies `bytes` of host memory to `dst` (device) on stream `s` through a pinned staging buffer of the handle: the caller's
// source may be freed or rewritten as soon as this returns, and the copy itself is asynchronous.
hipError_t wg_internal_upload(wg_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (!h) return hipErrorInvalidValue;
  const int i = h->pin_next;
  h->pin_next = (i + 1) % wg_handle::kPins;
  hipError_t e;
  if (h->pin_ev[i] && (e = hipEventSynchronize(h->pin_ev[i])) != hipSuccess) return e;     // its last copy has run
  if (h->pin_bytes[i] < bytes) {
    if (h->pin[i]) (void)hipHostFree(h->pin[i]);
    h->pin[i] = nullptr;
    h->pin_bytes[i] = 0;
    const size_t cap = bytes < 65536 ? 65536 : bytes;
    if ((e = hipHostMalloc(&h->pin[i], cap, hipHostMallocDefault)) != hipSuccess) return e;
    h->pin_bytes[i] = cap;
  }
  if (!h->pin_ev[i] && (e = hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming)) != hipSuccess) return e;
  memcpy(h->pin[i], src, bytes);
  if ((e = hipMemcpyAsync(dst, h->pin[i], bytes, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
  return hipEventRecord(h->pin_ev[i], s);
}
// Event `slot` of the mark pool (no timing), created on first use.
hipEvent_t wg_internal_mark_event(wg_handle* h, int slot) {
  if (!h || slot < 0 || slot >= 16) return nullptr;
  if (!h->mark_ev[slot] && hipEventCreateWithFlags(&h->mark_ev[slot], hipEventDisableTiming) != hipSuccess) h->mark_ev[slot] = nullptr;
  return h->mark_ev[slot];
}
// Next event of the ordering ring (no timing).  A wait captures the record that precedes it in host order, so an event
// may be recorded again while earlier waits on it are still pending on the device.  Ring events are for record-then-wait
// pairs issued back to back (order_after); anything waited on LATER uses wg_internal_mark_event.
hipEvent_t wg_internal_sync_event(wg_handle* h) {
  constexpr size_t kRing = 1024;
  if (!h) return nullptr;
  if (h->sync_ev.size() < kRing) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    h->sync_ev.push_back(e);
    return e;
  }
  hipEvent_t e = h->sync_ev[h->sync_next];
  h->sync_next = (h->sync_next + 1) % kRing;
  return e;
}

extern "C" {

const char* wg_version(void) { return "waveglow_amd 0.1 (gfx950)"; }
const char* wg_last_error(void) { return g_err.c_str(); }

int wg_create(const wg_config* cfg, int device_id, wg_handle** out) {
  if (!cfg || !out) return fail(WG_ERR_INVALID, "null argument");
  const wg_config& c = *cfg;
  if (c.n_group != 8) return fail(WG_ERR_INVALID, "n_group=%d unsupported (only 8)", c.n_group);
  if (c.kernel_size != 3) return fail(WG_ERR_INVALID, "kernel_size=%d unsupported (only 3)", c.kernel_size);
  if (c.n_channels != 64 && c.n_channels != 128 && c.n_channels != 256 && c.n_channels != 512)
    return fail(WG_ERR_INVALID, "n_channels=%d unsupported (64, 128, 256, 512)", c.n_channels);
  if (c.n_layers < 1 || c.n_layers > 10) return fail(WG_ERR_INVALID, "n_layers=%d unsupported (1..10)", c.n_layers);
  if (c.upsample_kernel != 1024 || c.upsample_stride != 256)
    return fail(WG_ERR_INVALID, "upsample geometry %d/%d unsupported (1024/256)", c.upsample_kernel, c.upsample_stride);
  if (c.n_mel_channels < 16 || c.n_mel_channels > 80 || c.n_mel_channels % 16 != 0)
    return fail(WG_ERR_INVALID, "n_mel_channels=%d unsupported (multiple of 16, <= 80)", c.n_mel_channels);
  if (c.n_flows < 1 || c.n_early_every < 1 || c.n_early_size < 0 || c.n_early_size % 2 != 0)
    return fail(WG_ERR_INVALID, "bad flow configuration");
  std::vector<int> ck = flow_channels(c);
  if (ck.back() < 2 || ck.back() % 2 != 0) return fail(WG_ERR_INVALID, "flow configuration leaves %d channels", ck.back());
  wg_handle* h = new wg_handle();
  h->cfg = c;
  h->device = device_id;
  {
    int ncu = 0;   // of THIS device, whatever the caller's current device is
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0) h->n_cu = ncu;
  }
  h->NS = c.n_mel_channels * c.n_group;
  h->c_k = ck;
  if (const char* e = getenv("WG_FORCE_BN")) h->force_bn = atoi(e);
  if (const char* e = getenv("WG_NO_START_FOLD")) h->fold_start = !(*e == '1');
  h->expected.push_back("upsample.weight");
  h->expected.push_back("upsample.bias");
  for (int k = 0; k < c.n_flows; ++k) {
    const std::string ks = std::to_string(k);
    h->expected.push_back("convinv." + ks + ".conv.weight");
    const std::string p = "WN." + ks + ".";
    for (const char* n : {"start", "cond_layer", "end"}) {
      h->expected.push_back(p + n + ".weight");
      h->expected.push_back(p + n + ".bias");
    }
    for (int i = 0; i < c.n_layers; ++i) {
      const std::string is = std::to_string(i);
      h->expected.push_back(p + "in_layers." + is + ".weight");
      h->expected.push_back(p + "in_layers." + is + ".bias");
      h->expected.push_back(p + "res_skip_layers." + is + ".weight");
      h->expected.push_back(p + "res_skip_layers." + is + ".bias");
    }
  }
  *out = h;
  return WG_OK;
}

int wg_destroy(wg_handle* h) {
  if (!h) return WG_OK;
  if (h->d_blob) hipFree(h->d_blob);
  if (h->d_cond) hipFree(h->d_cond);
  if (h->d_cond16) hipFree(h->d_cond16);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  for (hipEvent_t e : h->sync_ev) hipEventDestroy(e);
  for (hipEvent_t e : h->mark_ev)
    if (e) hipEventDestroy(e);
  for (int i = 0; i < wg_handle::kPins; ++i) {
    if (h->pin_ev[i]) hipEventDestroy(h->pin_ev[i]);
    if (h->pin[i]) (void)hipHostFree(h->pin[i]);
  }
  for (hipStream_t st : h->aux)
    if (st) hipStreamDestroy(st);
  delete h;
  return WG_OK;
}

int wg_num_expected_tensors(const wg_handle* h) { return h ? (int)h->expected.size() : 0; }
const char* wg_expected_tensor_name(const wg_handle* h, int32_t i) {
  if (!h || i < 0 || i >= (int)h->expected.size()) return nullptr;
  return h->expected[i].c_str();
}

int wg_set_tensor(wg_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!h || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(WG_ERR_INVALID, "bad argument to wg_set_tensor");
  bool known = false;
  for (const auto& e : h->expected) known |= (e == name);
  if (!known) return fail(WG_ERR_INVALID, "unexpected tensor name '%s'", name);
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return fail(WG_ERR_INVALID, "bad shape for '%s'", name);
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.data.assign(data, data + n);
  h->tensors[name] = std::move(t);
  h->finalized = false;
  return WG_OK;
}

static int check_shape(const wg_handle* h, const std::string& name, std::initializer_list<int64_t> want,
                       const HostTensor** out) {
  const HostTensor* t = find(h, name);
  if (!t) return fail(WG_ERR_STATE, "missing tensor '%s'", name.c_str());
  std::vector<int64_t> w(want);
  if (t->shape != w) {
    std::string got, exp;
    for (auto v : t->shape) got += std::to_string(v) + ",";
    for (auto v : w) exp += std::to_string(v) + ",";
    return fail(WG_ERR_INVALID, "tensor '%s' has shape [%s] expected [%s]", name.c_str(), got.c_str(), exp.c_str());
  }
  *out = t;
  return WG_OK;
}

int wg_finalize(wg_handle* h) {
  if (!h) return fail(WG_ERR_INVALID, "null handle");
  const wg_config& c = h->cfg;
  const int C = c.n_channels, M = c.n_mel_channels, NS = h->NS, NL = c.n_layers;
  const int NW = wn_waves(C), MB = C / (32 * NW), MT = 2 * MB, CC = C / 64, K2 = C / 16, nKx = 3 * CC;
  std::vector<char> blob;
  auto reserve = [&](size_t bytes) {
    size_t off = align_up(blob.size());
    blob.resize(off + bytes, 0);
    return off;
  };
  int rc;
  // ---- upsample weights: folded into the conditioning GEMM on the device below (cond_fold_kernel)
  const HostTensor *upw, *upb;
  if ((rc = check_shape(h, "upsample.weight", {M, M, c.upsample_kernel}, &upw))) return rc;
  if ((rc = check_shape(h, "upsample.bias", {M}, &upb))) return rc;
  h->flows.assign(c.n_flows, FlowOffsets());
  for (int k = 0; k < c.n_flows; ++k) {
    FlowOffsets& fo = h->flows[k];
    const int ck = h->c_k[k], hk = ck / 2;
    fo.c = ck;
    fo.h = hk;
    const std::string ks = std::to_string(k), p = "WN." + ks + ".";
    const HostTensor *wci, *wst, *bst, *wcond, *bcond, *wend, *bend;
    if ((rc = check_shape(h, "convinv." + ks + ".conv.weight", {ck, ck, 1}, &wci))) return rc;
    if ((rc = check_shape(h, p + "start.weight", {C, hk, 1}, &wst))) return rc;
    if ((rc = check_shape(h, p + "start.bias", {C}, &bst))) return rc;
    if ((rc = check_shape(h, p + "cond_layer.weight", {2 * C * NL, NS, 1}, &wcond))) return rc;
    if ((rc = check_shape(h, p + "cond_layer.bias", {2 * C * NL}, &bcond))) return rc;
    if ((rc = check_shape(h, p + "end.weight", {2 * hk, C, 1}, &wend))) return rc;
    if ((rc = check_shape(h, p + "end.bias", {2 * hk}, &bend))) return rc;
    // 1x1 invertible conv: forward matrix, fp64 inverse, log|det|
    {
      std::vector<double> A((size_t)ck * ck), inv;
      for (int i = 0; i < ck * ck; ++i) A[i] = wci->data[i];
      double lad;
      int sign;
      if (!invert(A, ck, inv, lad, sign)) return fail(WG_ERR_INVALID, "convinv.%d weight is singular", k);
      fo.logdet = sign > 0 ? lad : std::nan("");
      fo.winv = reserve((size_t)ck * ck * 4);
      fo.wfwd = reserve((size_t)ck * ck * 4);
      float* wi = (float*)(blob.data() + fo.winv);
      float* wf = (float*)(blob.data() + fo.wfwd);
      for (int i = 0; i < ck * ck; ++i) {
        wi[i] = (float)inv[i];
        wf[i] = wci->data[i];
      }
    }
    // start conv, position-major rows
    fo.wstart = reserve((size_t)C * hk * 4);
    fo.bstart = reserve((size_t)C * 4);
    {
      float* ws = (float*)(blob.data() + fo.wstart);
      float* bs = (float*)(blob.data() + fo.bstart);
      for (int P = 0; P < C; ++P) {
        const int ch = pos_to_chan(P);
        for (int j = 0; j < hk; ++j) ws[P * hk + j] = wst->data[(size_t)ch * hk + j];
        bs[P] = bst->data[ch];
      }
    }
    if (wn_res_a0(C)) {
      // A fragment of the 32x32x16 MFMA, [wave][lane = (row r, half)][8]: row r = channel 32 w + r; lanes 0-31 hold the fp16
      // hi parts of (W_start[ch][0..h-1], b_start[ch] at k = 4), lanes 32-63 the lo parts -- against the a0 plane row, which
      // carries (a0 | 1 | 0 0 0) at positions 0-7 and again at 8-15 (flow_kernel)
      fo.wStA = reserve((size_t)(C / 32) * 64 * 8 * 2);
      _Float16* d = (_Float16*)(blob.data() + fo.wStA);
      for (int w = 0; w < C / 32; ++w)
        for (int lane = 0; lane < 64; ++lane) {
          const int ch = 32 * w + (lane & 31);
          for (int j = 0; j < 8; ++j) {
            const float v = j < hk ? wst->data[(size_t)ch * hk + j] : j == 4 ? bst->data[ch] : 0.0f;
            const _Float16 hi = (_Float16)v;
            d[((size_t)w * 64 + lane) * 8 + j] = lane < 32 ? hi : (_Float16)(v - (float)hi);
          }
        }
    }
    std::vector<double> out_bias(8, 0.0);
    for (int r = 0; r < 2 * hk; ++r) out_bias[r] = bend->data[r];
    fo.layers.assign(NL, LayerOffsets());
    for (int i = 0; i < NL; ++i) {
      LayerOffsets& lo = fo.layers[i];
      const std::string is = std::to_string(i);
      const bool has_res = i < NL - 1;
      const int RS = has_res ? 2 * C : C;
      const HostTensor *win, *bin, *wrs, *brs;
      if ((rc = check_shape(h, p + "in_layers." + is + ".weight", {2 * C, C, 3}, &win))) return rc;
      if ((rc = check_shape(h, p + "in_layers." + is + ".bias", {2 * C}, &bin))) return rc;
      if ((rc = check_shape(h, p + "res_skip_layers." + is + ".weight", {RS, C, 1}, &wrs))) return rc;
      if ((rc = check_shape(h, p + "res_skip_layers." + is + ".bias", {RS}, &brs))) return rc;
      // GEMM1 tap A fragments [2*3C/64 half K-steps][NW][MT][2 k16][64 lanes][8]; wave w owns 32-channel blocks
      // w*MB .. w*MB+MB-1: M-tiles mt < MB are their tanh rows, mt >= MB their sigmoid rows (+C)
      lo.wA1 = reserve((size_t)nKx * 2 * NW * MT * 2 * 64 * 8 * 2);
      {
        _Float16* dst = (_Float16*)(blob.data() + lo.wA1);
        for (int u = 0; u < 2 * nKx; ++u)
          for (int w = 0; w < NW; ++w)
            for (int mt = 0; mt < MT; ++mt)
              for (int k2 = 0; k2 < 2; ++k2)
                for (int lane = 0; lane < 64; ++lane) {
                  const int ksx = u >> 1, k16 = (u & 1) * 2 + k2;
                  const int r = lane & 31, hh = lane >> 5;
                  const bool tanh_row = mt < MB;
                  const int m = (tanh_row ? 0 : C) + 32 * (w * MB + (tanh_row ? mt : mt - MB)) + r;
                  const float rs = tanh_row ? kTanhScale : kSigmScale;   // gate pre-scale (kernels.hip gate_act)
                  _Float16* d = dst + (((((size_t)u * NW + w) * MT + mt) * 2 + k2) * 64 + lane) * 8;
                  for (int j = 0; j < 8; ++j) {
                    const int kk = k16 * 16 + 8 * hh + j;
                    const int tap = ksx / CC, cc = ksx % CC;
                    const int ch = pos_to_chan(cc * 64 + kk);
                    d[j] = (_Float16)(win->data[((size_t)m * C + ch) * 3 + tap] * rs);
                  }
                }
      }
      const bool f16 = wn_frag16(C, wn_block_n(C));
      // 16x16x32 fragments [half K-step u][wave][tile m][64 lanes = (row i = lane & 15, K group lane >> 4)][8]: tile m of
      // wave w = rows 32 w + 16 (m & 1) + i of the tanh (m < 2) / sigmoid half; K = 32 (u & 1) + 8 (lane >> 4) + j of the step
      auto pack16 = [&](size_t off, int n_half, auto&& value) {
        _Float16* dst = (_Float16*)(blob.data() + off);
        for (int u = 0; u < n_half; ++u)
          for (int w = 0; w < NW; ++w)
            for (int m4 = 0; m4 < 4; ++m4)
              for (int lane = 0; lane < 64; ++lane) {
                const int m = (m4 < 2 ? 0 : C) + 32 * w + 16 * (m4 & 1) + (lane & 15);
                const float rs = m4 < 2 ? kTanhScale : kSigmScale;
                _Float16* d = dst + ((((size_t)u * NW + w) * 4 + m4) * 64 + lane) * 8;
                for (int j = 0; j < 8; ++j) d[j] = (_Float16)(value(m, u >> 1, 32 * (u & 1) + 8 * (lane >> 4) + j) * rs);
              }
      };
      if (f16) {
        lo.wA1x = reserve((size_t)nKx * 2 * NW * 4 * 64 * 8 * 2);
        pack16(lo.wA1x, 2 * nKx, [&](int m, int ksx, int kk) {
          const int tap = ksx / CC, cc = ksx % CC;
          return win->data[((size_t)m * C + pos_to_chan(cc * 64 + kk)) * 3 + tap];
        });
      }
      if (i == 0) {
        // in_layers[0] o start (model.py:117, :123): column kk of tap `tap` on the a0 plane row (a0 | 1 | 0...):
        //   kk < h: sum_ch W_in[m][ch][tap] W_start[ch][kk];  kk == 4: sum_ch W_in[m][ch][tap] b_start[ch]
        std::vector<double> fold((size_t)2 * C * 3 * 8, 0.0);
        for (int m = 0; m < 2 * C; ++m)
          for (int tap = 0; tap < 3; ++tap) {
            double* fr = &fold[((size_t)m * 3 + tap) * 8];
            for (int ch = 0; ch < C; ++ch) {
              const double wv = win->data[((size_t)m * C + ch) * 3 + tap];
              for (int j = 0; j < hk; ++j) fr[j] += wv * wst->data[(size_t)ch * hk + j];
              fr[4] += wv * bst->data[ch];
            }
          }
        // ONE K-step for the three taps (wn_layer_kernel A0G): K index kk = 8 tap + j against a B tile whose 16-byte chunk
        // `tap` is the first chunk (a0 | 1 | 0 0 0) of that tap's a0-plane row
        lo.wA1f = reserve((size_t)2 * NW * MT * 2 * 64 * 8 * 2);
        _Float16* dst = (_Float16*)(blob.data() + lo.wA1f);
        for (int u = 0; u < 2; ++u)
          for (int w = 0; w < NW; ++w)
            for (int mt = 0; mt < MT; ++mt)
              for (int k2 = 0; k2 < 2; ++k2)
                for (int lane = 0; lane < 64; ++lane) {
                  const int k16 = u * 2 + k2;
                  const int r = lane & 31, hh = lane >> 5;
                  const bool tanh_row = mt < MB;
                  const int m = (tanh_row ? 0 : C) + 32 * (w * MB + (tanh_row ? mt : mt - MB)) + r;
                  const float rs = tanh_row ? kTanhScale : kSigmScale;
                  _Float16* d = dst + (((((size_t)u * NW + w) * MT + mt) * 2 + k2) * 64 + lane) * 8;
                  for (int j = 0; j < 8; ++j) {
                    const int kk = k16 * 16 + 8 * hh + j;          // 8 tap + value index of the gathered tile row
                    d[j] = (_Float16)(kk < 24 ? (float)(fold[((size_t)m * 3 + (kk >> 3)) * 8 + (kk & 7)] * rs) : 0.0f);
                  }
                }
        if (f16) {
          lo.wA1fx = reserve((size_t)2 * NW * 4 * 64 * 8 * 2);
          pack16(lo.wA1fx, 2, [&](int m, int, int kk) { return kk < 24 ? (float)fold[((size_t)m * 3 + (kk >> 3)) * 8 + (kk & 7)] : 0.0f; });
        }
      }
      lo.bias1 = reserve((size_t)2 * C * 4);
      {
        float* b1 = (float*)(blob.data() + lo.bias1);
        for (int m = 0; m < 2 * C; ++m) {
          // constant part of the folded conditioning: W_cond . (upsample bias broadcast over the 8 group phases)
          double cb = 0.0;
          const float* wr = &wcond->data[((size_t)(2 * C * i + m)) * NS];
          for (int sch = 0; sch < NS; ++sch) cb += (double)wr[sch] * upb->data[sch >> 3];
          b1[m] = (float)((bin->data[m] + bcond->data[2 * C * i + m] + cb) * (m < C ? kTanhScale : kSigmScale));
        }
      }
      // GEMM2 (res) A fragments [NW][MB][K2][64][8], bias2
      lo.wA2 = reserve((size_t)NW * MB * K2 * 64 * 8 * 2);
      lo.bias2 = reserve((size_t)C * 4);
      if (has_res) {
        _Float16* dst = (_Float16*)(blob.data() + lo.wA2);
        for (int w = 0; w < NW; ++w)
          for (int mb = 0; mb < MB; ++mb)
            for (int k16 = 0; k16 < K2; ++k16)
              for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, hh = lane >> 5;
                const int m = 32 * (w * MB + mb) + r;
                _Float16* d = dst + ((((size_t)w * MB + mb) * K2 + k16) * 64 + lane) * 8;
                for (int j = 0; j < 8; ++j) {
                  const int ch = pos_to_chan(k16 * 16 + 8 * hh + j);
                  d[j] = (_Float16)wrs->data[(size_t)m * C + ch];
                }
              }
        float* b2 = (float*)(blob.data() + lo.bias2);
        for (int m = 0; m < C; ++m) b2[m] = brs->data[m];
      }
      // folded end x skip: Wes = W_end (2h x C) * W_skip_i (C x C), fp64; hi/lo fp16 split
      const int skip_row0 = has_res ? C : 0;
      std::vector<double> Wes((size_t)8 * C, 0.0);
      for (int r = 0; r < 2 * hk; ++r) {
        for (int m = 0; m < C; ++m) {
          const double we = wend->data[(size_t)r * C + m];
          const float* wrow = &wrs->data[(size_t)(skip_row0 + m) * C];
          for (int cch = 0; cch < C; ++cch) Wes[(size_t)r * C + cch] += we * wrow[cch];
          out_bias[r] += we * brs->data[skip_row0 + m];
        }
      }
      lo.wEs = reserve((size_t)(C / 32) * 64 * 8 * 2);
      {
        _Float16* dst = (_Float16*)(blob.data() + lo.wEs);
        for (int s = 0; s < C / 32; ++s)
          for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, l4 = lane >> 4;
            for (int j = 0; j < 8; ++j) {
              const int ch = pos_to_chan(32 * s + 8 * l4 + j);
              const float v = (float)Wes[(size_t)(row & 7) * C + ch];
              const _Float16 hi = (_Float16)v;
              const _Float16 lo16 = (_Float16)(v - (float)hi);
              dst[((size_t)s * 64 + lane) * 8 + j] = row < 8 ? hi : lo16;
            }
          }
      }
    }
    fo.out_init = reserve(8 * 4);
    {
      float* oi = (float*)(blob.data() + fo.out_init);
      for (int r = 0; r < 8; ++r) oi[r] = (float)out_bias[r];
    }
  }
  // the library works on h->device without changing the caller's current device for good
  struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  } dev_guard;
  HIP_TRY(hipGetDevice(&dev_guard.prev));
  HIP_TRY(hipSetDevice(h->device));
  {
    int ncu = 0;
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));
    if (ncu > 0) h->n_cu = ncu;
  }
  if (h->d_blob) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(h->d_blob));
    h->d_blob = nullptr;
  }
  HIP_TRY(hipMalloc((void**)&h->d_blob, blob.size()));
  HIP_TRY(hipMemcpy(h->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
  h->blob_bytes = blob.size();
  // ---- derived weights: cond_layer o upsample, one matrix per (flow, layer, phase), built on the device
  {
    if (h->d_cond) {
      HIP_TRY(hipFree(h->d_cond));
      h->d_cond = nullptr;
    }
    const int n_half = 2 * (M / 16);
    h->cond_layer_bytes = (size_t)kPhases * n_half * NW * MT * 2 * 64 * 8 * 2;
    h->cond_flow_bytes = h->cond_layer_bytes * NL;
    HIP_TRY(hipMalloc((void**)&h->d_cond, h->cond_flow_bytes * c.n_flows));
    if (h->d_cond16) {
      HIP_TRY(hipFree(h->d_cond16));
      h->d_cond16 = nullptr;
    }
    if (wn_frag16(C, wn_block_n(C))) HIP_TRY(hipMalloc((void**)&h->d_cond16, h->cond_flow_bytes * c.n_flows));
    float *d_wc = nullptr, *d_up = nullptr;
    const size_t wc_bytes = (size_t)2 * C * NL * NS * 4, up_bytes = (size_t)M * M * c.upsample_kernel * 4;
    HIP_TRY(hipMalloc((void**)&d_wc, wc_bytes));
    HIP_TRY(hipMalloc((void**)&d_up, up_bytes));
    HIP_TRY(hipMemcpy(d_up, upw->data.data(), up_bytes, hipMemcpyHostToDevice));
    for (int k = 0; k < c.n_flows; ++k) {
      const HostTensor* wcond = find(h, "WN." + std::to_string(k) + ".cond_layer.weight");
      HIP_TRY(hipMemcpy(d_wc, wcond->data.data(), wc_bytes, hipMemcpyHostToDevice));
      HIP_TRY(launch_cond_fold(d_wc, d_up, (_Float16*)(h->d_cond + h->cond_flow_bytes * k), C, NW, M, NL,
                               c.upsample_kernel, kTanhScale, kSigmScale, 0, nullptr));
      if (h->d_cond16)
        HIP_TRY(launch_cond_fold(d_wc, d_up, (_Float16*)(h->d_cond16 + h->cond_flow_bytes * k), C, NW, M, NL,
                                 c.upsample_kernel, kTanhScale, kSigmScale, 1, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      if (dbg_sync()) { fprintf(stderr, "[wg] cond_fold flow %d ok\n", k); fflush(stderr); }
    }
    HIP_TRY(hipFree(d_wc));
    HIP_TRY(hipFree(d_up));
  }
  h->finalized = true;
  return WG_OK;
}

size_t wg_infer_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames) {
  if (!h || B < 1 || n_frames < 1) return 0;
  const int L = n_frames * h->cfg.upsample_stride / h->cfg.n_group;
  RowGeom g = make_geom(h->cfg, B, L, n_frames);
  return carve(h, g, nullptr).bytes;
}

size_t wg_forward_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames, int32_t audio_len) {
  if (!h || B < 1 || n_frames < 1 || audio_len < h->cfg.n_group || audio_len % h->cfg.n_group) return 0;
  RowGeom g = make_geom(h->cfg, B, audio_len / h->cfg.n_group, n_frames);
  return carve(h, g, nullptr).bytes;
}

static int run_wn(wg_handle* h, int k, const RowGeom& g, Workspace& w, _Float16*& cur, _Float16*& oth, hipStream_t s) {
  const wg_config& c = h->cfg;
  const int C = c.n_channels;
  int BN = wn_block_n(C);
  // Small workloads: 64-column tiles double the workgroup count when 128-column tiles would leave CUs idle -- when that
  // buys whole rounds.  A 64-column tile takes ~0.6 of a 128-column tile's time (measured at 256 channels: 22.7 vs 38.5 us
  // for a one-round launch), so compare rounds x tile time: 128 tiles on 256 CUs -> 64-column tiles fill the chip in one
  // round (configs[0]); 224 tiles -> two rounds of 64-column tiles lose to one round of 128-column ones (+18 % at 1 x 80x864).
  if (BN == 128) {
    const int64_t t128 = (int64_t)kPhases * (g.Rp / 128), ncu = h->n_cu > 0 ? h->n_cu : 1;
    const int64_t r128 = (t128 + ncu - 1) / ncu, r64 = (2 * t128 + ncu - 1) / ncu;
    if (t128 < 4 * ncu && 6 * r64 < 10 * r128) BN = 64;
  }
  if (BN == 128 && h->force_bn == 64) BN = 64;
  if (h->force_bn == 128 && wn_block_n(C) == 128) BN = 128;
  const FlowOffsets& fo = h->flows[k];
  for (int i = 0; i < c.n_layers; ++i) {
    const LayerOffsets& lo = fo.layers[i];
    WnLayerArgs a;
    a.x_in = cur;
    a.x_out = oth;
    const bool fold0 = (i == 0) && h->fold_start;       // in_layers[0] o start on the a0 plane (one gathered K-step)
    a.x_tap = fold0 ? w.A0 : cur;
    a.x_chunks_per_tap = fold0 ? 1 : C / 64;
    a.melT = w.melT;
    a.frag16 = wn_frag16(C, BN) ? 1 : 0;
    a.wStA = fold0 && wn_res_a0(C) ? (const _Float16*)(h->d_blob + fo.wStA) : nullptr;
    a.a0_fold = fold0 ? 1 : 0;
    a.wA1c = (const _Float16*)((a.frag16 ? h->d_cond16 : h->d_cond) + h->cond_flow_bytes * k + h->cond_layer_bytes * i);
    a.wA1 = (const _Float16*)(h->d_blob + (a.frag16 ? (fold0 ? lo.wA1fx : lo.wA1x) : (fold0 ? lo.wA1f : lo.wA1)));
    a.bias1 = (const float*)(h->d_blob + lo.bias1);
    a.wA2 = (const _Float16*)(h->d_blob + lo.wA2);
    a.bias2 = (const float*)(h->d_blob + lo.bias2);
    a.wEs = (const _Float16*)(h->d_blob + lo.wEs);
    a.out = w.OUT;
    a.g = g;
    a.dil = 1 << i;
    a.n_cond_steps = c.n_mel_channels / 16;
    a.M = c.n_mel_channels;
    a.has_res = i < c.n_layers - 1;
    a.row0 = 0;
    a.tiles_per_phase = g.Rp / BN;
    a.sp = nullptr;
    a.save_t = a.save_s = a.save_a = nullptr;
    a.in0 = a.in1 = nullptr;
    a.out0 = nullptr;
    a.n_tiles = kPhases * a.tiles_per_phase;
    a.stamps = h->dbg_stamps;
    a.n_cu = h->n_cu;
    {
      Prof p(h, s, 2);
      HIP_TRY(launch_wn_layer(a, C, BN, s));
    }
    WG_DBG(s, "wn_layer");
    if (a.has_res) std::swap(cur, oth);
  }
  return WG_OK;
}

int wg_infer(wg_handle* h, const void* mel, const void* z_init, const void* const* z_early, int32_t n_z_early,
             float sigma, void* audio, int32_t B, int32_t n_frames, int32_t io_dtype, void* workspace,
             size_t workspace_bytes, void* stream) {
  return wg_infer_ragged(h, mel, nullptr, z_init, z_early, n_z_early, sigma, audio, B, n_frames, io_dtype, workspace,
                         workspace_bytes, stream);
}

int wg_infer_ragged(wg_handle* h, const void* mel, const int32_t* frames, const void* z_init, const void* const* z_early,
                    int32_t n_z_early, float sigma, void* audio, int32_t B, int32_t n_frames, int32_t io_dtype,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return fail(WG_ERR_INVALID, "null handle");
  if (!h->finalized) return fail(WG_ERR_STATE, "wg_finalize has not been called");
  if (!mel || !z_init || !audio || !workspace) return fail(WG_ERR_INVALID, "null buffer");
  if (B < 1 || n_frames < 1) return fail(WG_ERR_INVALID, "bad B/n_frames");
  if (io_dtype != WG_F32 && io_dtype != WG_F16) return fail(WG_ERR_INVALID, "bad io_dtype");
  const wg_config& c = h->cfg;
  int n_early = 0;
  for (int k = 0; k < c.n_flows; ++k) n_early += is_early(c, k);
  if (n_z_early != n_early || (n_early && !z_early)) return fail(WG_ERR_INVALID, "expected %d early-noise tensors", n_early);
  const int L = n_frames * c.upsample_stride / c.n_group;
  if ((int64_t)B * L * 8 >= (1ll << 31)) return fail(WG_ERR_INVALID, "batch too large for 32-bit row indexing");
  RowGeom g = make_geom(c, B, L, n_frames);
  g.frames = (const int*)frames;
  Workspace w = carve(h, g, (char*)workspace);
  if (w.bytes > workspace_bytes) return fail(WG_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.bytes);
  if ((size_t)g.R * 128 >= (1ull << 32)) return fail(WG_ERR_INVALID, "plane too large for 32-bit offsets");
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_channels;
  {
    Prof p(h, s, 3);
    // guard rows / rows >= L of both x planes must read as zero padding (model.py:98-102)
    HIP_TRY(launch_zero_fill(w.X0, 2 * w.x_bytes + w.a0_bytes, s));
  }
  {
    MelPackArgs u;
    u.mel = mel;
    u.frames = g.frames;
    u.melT = w.melT;
    u.B = B;
    u.M = c.n_mel_channels;
    u.T = n_frames;
    u.io_f16 = io_dtype == WG_F16;
    Prof p(h, s, 0);
    HIP_TRY(launch_mel_pack(u, s));
  }
  WG_DBG(s, "mel_pack");
  _Float16 *cur = w.X0, *oth = w.X1;
  int ze_idx = 0;
  auto base_flow_args = [&]() {
    FlowArgs f;
    memset(&f, 0, sizeof f);
    f.direction = 0;
    f.sigma = sigma;
    f.Z = w.Z;
    f.out = w.OUT;
    f.Z_w = w.Z;
    f.out_w = w.OUT;
    f.g = g;
    f.C = C;
    f.io_f16 = io_dtype == WG_F16;
    return f;
  };
  {
    const int k = c.n_flows - 1;
    FlowArgs f = base_flow_args();
    f.first = 1;
    f.z_extra = z_init;
    f.c_next = h->c_k[k];
    f.h_next = f.c_next / 2;
    f.wstart = (const float*)(h->d_blob + h->flows[k].wstart);
    f.bstart = (const float*)(h->d_blob + h->flows[k].bstart);
    f.out_init = (const float*)(h->d_blob + h->flows[k].out_init);
    f.x = cur;
    f.a0p = w.A0;
    f.skip_x = h->fold_start && wn_res_a0(C);
    Prof p(h, s, 1);
    HIP_TRY(launch_flow(f, s));
  }
  WG_DBG(s, "flow");
  for (int k = c.n_flows - 1; k >= 0; --k) {
    int rc = run_wn(h, k, g, w, cur, oth, s);
    if (rc) return rc;
    FlowArgs f = base_flow_args();
    f.c_in = h->c_k[k];
    f.h_in = f.c_in / 2;
    f.winv = (const float*)(h->d_blob + h->flows[k].winv);
    if (is_early(c, k)) {
      f.z_extra = z_early[ze_idx++];
      f.n_extra = c.n_early_size;
      if (!f.z_extra) return fail(WG_ERR_INVALID, "null early-noise tensor");
    }
    f.c_next = f.c_in + f.n_extra;
    f.last = (k == 0);
    if (f.last) {
      if (f.c_next != c.n_group) return fail(WG_ERR_STATE, "flow bookkeeping error");
      f.audio_out = audio;
    } else {
      f.h_next = h->c_k[k - 1] / 2;
      if (h->c_k[k - 1] != f.c_next) return fail(WG_ERR_STATE, "flow bookkeeping error");
      f.wstart = (const float*)(h->d_blob + h->flows[k - 1].wstart);
      f.bstart = (const float*)(h->d_blob + h->flows[k - 1].bstart);
      f.out_init = (const float*)(h->d_blob + h->flows[k - 1].out_init);
      f.x = cur;
      f.a0p = w.A0;
      f.skip_x = h->fold_start && wn_res_a0(C);
    }
    Prof p(h, s, 1);
    HIP_TRY(launch_flow(f, s));
  }
  WG_DBG(s, "flow");
  return WG_OK;
}

int wg_forward(wg_handle* h, const void* mel, const void* audio, float* z, float* const* log_s, float* log_det_W,
               int32_t B, int32_t n_frames, int32_t audio_len, int32_t io_dtype, void* workspace,
               size_t workspace_bytes, void* stream) {
  if (!h) return fail(WG_ERR_INVALID, "null handle");
  if (!h->finalized) return fail(WG_ERR_STATE, "wg_finalize has not been called");
  if (!mel || !audio || !z || !log_s || !log_det_W || !workspace) return fail(WG_ERR_INVALID, "null buffer");
  if (io_dtype != WG_F32 && io_dtype != WG_F16) return fail(WG_ERR_INVALID, "bad io_dtype");
  const wg_config& c = h->cfg;
  if (B < 1 || n_frames < 1 || audio_len < c.n_group || audio_len % c.n_group)
    return fail(WG_ERR_INVALID, "audio_len must be a positive multiple of n_group");
  // assert spect.size(2) >= audio.size(1)   (model.py:187)
  if ((int64_t)(n_frames - 1) * c.upsample_stride + c.upsample_kernel < audio_len)
    return fail(WG_ERR_INVALID, "upsampled mel (%d frames) shorter than audio (%d)", n_frames, audio_len);
  const int L = audio_len / c.n_group;
  if ((int64_t)B * L * 8 >= (1ll << 31)) return fail(WG_ERR_INVALID, "batch too large for 32-bit row indexing");
  RowGeom g = make_geom(c, B, L, n_frames);
  Workspace w = carve(h, g, (char*)workspace);
  if (w.bytes > workspace_bytes) return fail(WG_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.bytes);
  if ((size_t)g.R * 128 >= (1ull << 32)) return fail(WG_ERR_INVALID, "plane too large for 32-bit offsets");
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_channels;
  for (int k = 0; k < c.n_flows; ++k) log_det_W[k] = (float)((double)B * L * h->flows[k].logdet);   // model.py:63
  HIP_TRY(launch_zero_fill(w.X0, 2 * w.x_bytes + w.a0_bytes, s));
  {
    MelPackArgs u;
    u.mel = mel;
    u.frames = nullptr;
    u.melT = w.melT;
    u.B = B;
    u.M = c.n_mel_channels;
    u.T = n_frames;
    u.io_f16 = io_dtype == WG_F16;
    Prof p(h, s, 0);
    HIP_TRY(launch_mel_pack(u, s));
  }
  WG_DBG(s, "mel_pack");
  _Float16 *cur = w.X0, *oth = w.X1;
  int z_ch = 0;
  for (int k = 0; k <= c.n_flows; ++k) {
    FlowArgs f;
    memset(&f, 0, sizeof f);
    f.direction = 1;
    f.Z = w.Z;
    f.out = w.OUT;
    f.Z_w = w.Z;
    f.out_w = w.OUT;
    f.g = g;
    f.C = C;
    f.io_f16 = io_dtype == WG_F16;
    f.z_out = z;
    f.z_out_ch0 = z_ch;
    f.first = (k == 0);
    f.last = (k == c.n_flows);
    if (f.first) {
      f.audio_in = audio;
      f.c_in = c.n_group;
    } else {
      f.c_in = h->c_k[k - 1];
      f.h_in = f.c_in / 2;
      f.log_s_out = log_s[k - 1];
      if (!f.log_s_out) return fail(WG_ERR_INVALID, "null log_s[%d]", k - 1);
    }
    if (!f.last) {
      f.n_peel = is_early(c, k) ? c.n_early_size : 0;
      f.c_next = h->c_k[k];
      f.h_next = f.c_next / 2;
      if (f.c_in - f.n_peel != f.c_next) return fail(WG_ERR_STATE, "flow bookkeeping error");
      f.winv = (const float*)(h->d_blob + h->flows[k].wfwd);
      f.wstart = (const float*)(h->d_blob + h->flows[k].wstart);
      f.bstart = (const float*)(h->d_blob + h->flows[k].bstart);
      f.out_init = (const float*)(h->d_blob + h->flows[k].out_init);
      f.x = cur;
      f.a0p = w.A0;
      f.skip_x = h->fold_start && wn_res_a0(C);
      z_ch += f.n_peel;
    }
    {
      Prof p(h, s, 1);
      HIP_TRY(launch_flow(f, s));
    }
    if (!f.last) {
      int rc = run_wn(h, k, g, w, cur, oth, s);
      if (rc) return rc;
    }
  }
  return WG_OK;
}

static int loss_impl(const float* z, int64_t z_elems, const float* const* log_s, const int64_t* log_s_elems, int32_t n_flows,
                     const float* log_det_host, const float* log_det_dev, float sigma, float* loss_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!z || !log_s || !log_s_elems || (!log_det_host && !log_det_dev) || !loss_out || !workspace) return fail(WG_ERR_INVALID, "null buffer");
  if (z_elems < 1 || n_flows < 1 || !(sigma > 0.f)) return fail(WG_ERR_INVALID, "bad sizes/sigma");
  if (workspace_bytes < 16) return fail(WG_ERR_WORKSPACE, "workspace %zu < required 16", workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  double* acc = (double*)workspace;
  HIP_TRY(hipMemsetAsync(acc, 0, 16, s));
  HIP_TRY(launch_reduce_sum(z, (size_t)z_elems, 1, acc, s));                              // sum z*z      train.py:43
  double log_det_total = 0.0;
  for (int k = 0; k < n_flows; ++k) {
    if (!log_s[k] || log_s_elems[k] < 1) return fail(WG_ERR_INVALID, "bad log_s[%d]", k);
    HIP_TRY(launch_reduce_sum(log_s[k], (size_t)log_s_elems[k], 0, acc + 1, s));          // sum log_s    train.py:36-40
    if (log_det_host) log_det_total += (double)log_det_host[k];                           // train.py:37,41
  }
  HIP_TRY(launch_loss_final(acc, log_det_total, log_det_dev, log_det_dev ? n_flows : 0, sigma, (double)z_elems, loss_out, s));   // train.py:43-44
  return WG_OK;
}

int wg_loss(const float* z, int64_t z_elems, const float* const* log_s, const int64_t* log_s_elems, int32_t n_flows,
            const float* log_det_W, float sigma, float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  return loss_impl(z, z_elems, log_s, log_s_elems, n_flows, log_det_W, nullptr, sigma, loss_out, workspace, workspace_bytes, stream);
}

int wg_loss_dev(const float* z, int64_t z_elems, const float* const* log_s, const int64_t* log_s_elems, int32_t n_flows,
                const float* log_det_W_dev, float sigma, float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  return loss_impl(z, z_elems, log_s, log_s_elems, n_flows, nullptr, log_det_W_dev, sigma, loss_out, workspace, workspace_bytes, stream);
}

double wg_macs_per_group_step(const wg_handle* h) {
  if (!h) return 0.0;
  const wg_config& c = h->cfg;
  const double C = c.n_channels, NS = h->NS, M = c.n_mel_channels;
  double macs = 8.0 * M * M * 4.0;   // upsample: 8 samples x M x M x 4 taps
  for (int k = 0; k < c.n_flows; ++k) {
    const double ck = h->c_k[k], hk = ck / 2;
    macs += hk * C + NS * 2 * C * c.n_layers + c.n_layers * (C * 2 * C * c.kernel_size) +
            (c.n_layers - 1) * (C * 2 * C) + C * C + C * 2 * hk + ck * ck;
  }
  return macs;
}

int wg_debug_set_stamp_buffer(wg_handle* h, void* device_buffer) {
  if (!h) return fail(WG_ERR_INVALID, "null handle");
  h->dbg_stamps = (unsigned long long*)device_buffer;
  wg::g_wgrad_stamps = (unsigned long long*)device_buffer;
  return WG_OK;
}

int wg_profile_enable(wg_handle* h, int32_t on) {
  if (!h) return fail(WG_ERR_INVALID, "null handle");
  h->prof = on != 0;
  h->prof_mask = on == 1 ? ~0u : (unsigned)on;      // on > 1: bit c set = class c is timed (fewer events per step)
  h->ev_used = 0;
  for (int i = 0; i < 8; ++i) {
    h->prof_ms[i] = 0;
    h->prof_n[i] = 0;
  }
  return WG_OK;
}

int wg_profile_read(wg_handle* h, double* ms, int64_t* n, int32_t n_classes) {
  if (!h || !ms || !n || n_classes < 4) return fail(WG_ERR_INVALID, "bad argument");
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(h->ev[i + 1]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
    h->prof_ms[h->ev_class[i]] += t;
    h->prof_n[h->ev_class[i]] += 1;
  }
  h->ev_used = 0;
  for (int i = 0; i < (n_classes < 8 ? n_classes : 8); ++i) {
    ms[i] = h->prof_ms[i];
    n[i] = h->prof_n[i];
  }
  return WG_OK;
}

}  // extern "C"
