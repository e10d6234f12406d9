// C ABI of the training direction (include/waveglow_amd.h: wg_train_*): launch sequencing of train.hip.
// Reference: WaveGlow.forward under autograd (src/waveglow/model.py:178-221) and loss.backward() (train.py:190-199).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/waveglow_amd.h"
#include "wg_train.h"

using namespace wg;

int wg_set_error(int code, const char* msg);                              // api.cpp
const wg_config* wg_internal_config(const wg_handle* h);                  // api.cpp
const int* wg_internal_flow_channels(const wg_handle* h);                 // api.cpp
wg::RowGeom wg_internal_geom(const wg_handle* h, int B, int L, int T);    // api.cpp
void wg_internal_prof_event(wg_handle* h, void* stream, int cls);         // api.cpp
hipStream_t wg_internal_aux_stream(wg_handle* h, int i);                  // api.cpp
hipEvent_t wg_internal_sync_event(wg_handle* h);                          // api.cpp
hipEvent_t wg_internal_mark_event(wg_handle* h, int slot);                // api.cpp
hipError_t wg_internal_upload(wg_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s);   // api.cpp
int wg_internal_n_cu(const wg_handle* h);                                 // api.cpp

namespace {

#define TR_TRY(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
      return wg_set_error(WG_ERR_HIP, m.c_str());                                            \
    }                                                                                        \
    if (dbg_sync()) {                                                                        \
      hipError_t _s = hipDeviceSynchronize();                                                \
      fprintf(stderr, "[wg-train] %s -> %s\n", #expr, hipGetErrorString(_s));                \
      fflush(stderr);                                                                        \
      if (_s != hipSuccess) return wg_set_error(WG_ERR_HIP, hipGetErrorString(_s));          \
    }                                                                                        \
  } while (0)

// per-class device timing of the training launches (wg_profile_enable / wg_profile_read, classes 4..7)
#define TR_PROF(st, cls, stmt)                      \
  do {                                             \
    wg_internal_prof_event(h, (void*)(st), cls);   \
    stmt;                                          \
    wg_internal_prof_event(h, (void*)(st), cls);   \
  } while (0)

bool dbg_sync() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("WG_DEBUG_SYNC"); v = e && *e == '1'; }
  return v == 1;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

bool is_early(const wg_config& c, int k) { return k % c.n_early_every == 0 && k > 0; }

struct TrainWs {
  // fp16 planes (elements)
  _Float16 *X, *T, *S, *A;      // [FL] x C/64 chunks each (plane_c elements per fl)
  _Float16 *GP;                 // [FL] x 2C/64 chunks, contiguous: the K operand of the cond_layer dgrad
  _Float16 *GXL;                // [n_layers] x C/64 chunks: d x_i of the flow in flight (one buffer per layer: see chains)
  _Float16 *GO[2], *SP, *MELP, *GSP;   // d out plane (per flow parity), spectrogram planes, mel planes, d spect planes
  float *Zpost, *OUT;           // [n_flows][B*L*8]
  float *GZ;                    // [B*L*8]
  float *slab[2], *slab2[2];    // blocked wgrad slabs (dW1 | dW2), two sets: a layer's slabs are reduced beside the NEXT layer's launch
  float *part[2], *part2[2];    // column-sum partials of the two weight-gradient jobs (two sets)
  float *ext[2], *extb[2];      // d (W_end W_skip) partials [n_slabs][16][C] and their column sums [n_slabs][16]
  float *slab_up, *part3;       // d upsample slabs (one per phase) / partials of the row kernels of a flow and of d upsample
  size_t plane_c;               // elements of one C-channel plane set
  size_t rows8;                 // B*L*8
  size_t zero_bytes;            // prefix that `fresh` clears (all planes)
  size_t bytes;
};

TrainWs carve(const wg_config& c, const RowGeom& g, const int* n_slabs, char* base) {
  TrainWs w;
  const int C = c.n_channels, FL = c.n_flows * c.n_layers, M8 = c.n_mel_channels * 8;
  const size_t chunk = (size_t)g.R * 64;           // elements of one 64-channel plane
  w.plane_c = (size_t)(C / 64) * chunk;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base + off; off += align_up(bytes); return p; };
  w.X = (_Float16*)take((size_t)FL * w.plane_c * 2);
  w.T = (_Float16*)take((size_t)FL * w.plane_c * 2);
  w.S = (_Float16*)take((size_t)FL * w.plane_c * 2);
  w.A = (_Float16*)take((size_t)FL * w.plane_c * 2);
  w.GP = (_Float16*)take((size_t)FL * 2 * w.plane_c * 2);
  w.GXL = (_Float16*)take((size_t)c.n_layers * w.plane_c * 2);
  w.GO[0] = (_Float16*)take(chunk * 2);
  w.GO[1] = (_Float16*)take(chunk * 2);
  w.SP = (_Float16*)take((size_t)(M8 / 64) * chunk * 2);
  w.MELP = (_Float16*)take(2 * chunk * 2);
  w.GSP = (_Float16*)take((size_t)(M8 / 64) * chunk * 2);
  w.zero_bytes = off;
  w.rows8 = (size_t)g.B * g.L * 8;
  w.Zpost = (float*)take((size_t)c.n_flows * w.rows8 * 4);
  w.OUT = (float*)take((size_t)c.n_flows * w.rows8 * 4);
  w.GZ = (float*)take(w.rows8 * 4);
  const int cc = C / 64, mc = M8 / 64;
  const size_t t1 = (size_t)wgrad_tiles(2 * cc, 3 * cc + mc), t2 = (size_t)wgrad_tiles(cc, cc);
  for (int q = 0; q < 2; ++q) {
    w.slab[q] = (float*)take((size_t)n_slabs[0] * t1 * kWgradTileFloats * 4);
    w.slab2[q] = (float*)take((size_t)n_slabs[1] * t2 * kWgradTileFloats * 4);
    w.part[q] = (float*)take((size_t)n_slabs[0] * 2 * C * 4);
    w.part2[q] = (float*)take((size_t)n_slabs[1] * C * 4);
    w.ext[q] = (float*)take((size_t)n_slabs[1] * 16 * C * 4);
    w.extb[q] = (float*)take((size_t)n_slabs[1] * 16 * 4);
  }
  w.slab_up = (float*)take((size_t)kPhases * wgrad_tiles(mc, 8) * kWgradTileFloats * 4);
  w.part3 = (float*)take(max_sz(max_sz((size_t)flow_bwd_workgroups(g) * 64, (size_t)start_wgrad_workgroups(g) * 5 * C),
                                (size_t)kPhases * M8) * 4);
  w.bytes = off;
  return w;
}

struct Ctx {
  const wg_config* c;
  const int* ck;
  RowGeom g;
  TrainWs w;
  int C, FL, M8, K1, nl;
  int n_cu;
  int halves;      // 2: the batch runs as two independent half-batch chains (see setup)
  int n_slabs[2];  // row ranges of the two jobs of a weight-gradient launch (see setup)
  bool serial;     // WG_TRAIN_SERIAL=1: everything on the caller's stream (profiling of single kernels, A/B runs)
};

// sets rc_ and returns from the enclosing function on a HIP error of an ordering call
#define TR_ORDER(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return wg_set_error(WG_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
  } while (0)

// `to` continues after everything enqueued on `from` so far
hipError_t order_after(wg_handle* h, hipStream_t from, hipStream_t to) {
  if (from == to) return hipSuccess;
  hipEvent_t e = wg_internal_sync_event(h);
  if (!e) return hipErrorOutOfMemory;
  hipError_t r = hipEventRecord(e, from);
  return r != hipSuccess ? r : hipStreamWaitEvent(to, e, 0);
}

int setup(wg_handle* h, int32_t B, int32_t n_frames, int32_t audio_len, void* workspace, size_t workspace_bytes, Ctx& x) {
  if (!h) return wg_set_error(WG_ERR_INVALID, "null handle");
  x.c = wg_internal_config(h);
  x.ck = wg_internal_flow_channels(h);
  const wg_config& c = *x.c;
  if (B < 1 || n_frames < 1 || audio_len < c.n_group || audio_len % c.n_group)
    return wg_set_error(WG_ERR_INVALID, "audio_len must be a positive multiple of n_group");
  if ((int64_t)(n_frames - 1) * c.upsample_stride + c.upsample_kernel < audio_len)    // model.py:187
    return wg_set_error(WG_ERR_INVALID, "upsampled mel shorter than audio");
  const int L = audio_len / c.n_group;
  x.g = wg_internal_geom(h, B, L, n_frames);
  x.n_cu = wg_internal_n_cu(h);     // of the handle's device: decides the chain geometry (workspace layout) below
  // Chains.  A WN-layer launch runs ceil(tiles / CUs) rounds and the next layer waits for its last, partly filled
  // round (config 4: 576 tiles of 128 columns on 256 CUs = 2.25 rounds, a quarter of the chip-time idle).  The two
  // halves of the batch never exchange data inside a WN, so they run as two chains of half-size launches on two
  // streams and fill each other's idle CUs.  That needs the first B/2 utterances to end on a tile boundary: the rows
  // per utterance Fp are padded up (71 -> 72 at config 4: 16 x 72 = 9 tiles, the same 2304 rows per phase as before).
  // Rows past an utterance's last frame are guard rows like the others: never valid, always written as zeros -- the
  // only rows of the other half a chain's dilated taps can reach.
  x.halves = 1;
  {
    const char* e = getenv("WG_TRAIN_HALVES");
    const int want = e ? atoi(e) : 0;               // 1: never, 2: always (tests), otherwise by shape
    if (B % 2 == 0 && want != 1) {
      int fp = x.g.Fp;
      while ((B / 2 * fp) % 128) ++fp;
      const int rp = B * fp;
      const long long tiles = (long long)kPhases * (x.g.Rp / 128);
      const long long idle = (tiles + x.n_cu - 1) / x.n_cu * x.n_cu - tiles;
      const bool helps = tiles > x.n_cu && idle * 10 >= tiles && rp <= x.g.Rp + x.g.Rp / 32;
      if (want == 2 || helps) {
        x.g.Fp = fp;
        x.g.Rp = rp;
        x.g.R = kPhases * rp + 2 * kRowPad;
        x.halves = 2;
      }
    }
    const char* se = getenv("WG_TRAIN_SERIAL");
    x.serial = se && *se == '1';
  }
  // Slabs of the weight-gradient launches (train.hip: wgrad_kernel): a layer's workgroups -- tiles of d W1 x its slabs + tiles
  // of d W2 x its slabs -- should be ONE round of one workgroup per CU, all equally long.  d W1 gets CUs / (all tiles) slabs;
  // d W2, whose steps cost ~17 % more (the extra d out plane, its column sums), gets the CUs that are left: more, shorter
  // slabs (config 4: 11 tiles x 21 slabs + 1 tile x 25 slabs = 256 workgroups of 110 / 92 steps).
  {
    const int cc = c.n_channels / 64, mc = c.n_mel_channels * 8 / 64;
    const int t1 = wgrad_tiles(2 * cc, 3 * cc + mc), t2 = wgrad_tiles(cc, cc);
    const long long total_steps = (long long)kPhases * (x.g.Rp / 32);
    long long s1 = x.n_cu / (t1 + t2), s2;
    if (s1 < 1) s1 = 1;
    s2 = ((long long)x.n_cu - s1 * t1) / t2;
    if (s2 < s1) s2 = s1;
    if (const char* e = getenv("WG_TRAIN_SLABS")) {      // tests: pin the number of slabs, "<d W1>[,<d W2>]"
      int a_ = 0, b_ = 0;
      const int n = sscanf(e, "%d,%d", &a_, &b_);
      if (n >= 1 && a_ >= 1) { s1 = a_; s2 = (n == 2 && b_ >= 1) ? b_ : a_; }
    }
    if (s1 > total_steps) s1 = total_steps;
    if (s2 > total_steps) s2 = total_steps;
    x.n_slabs[0] = (int)s1;
    x.n_slabs[1] = (int)s2;
  }
  x.w = carve(c, x.g, x.n_slabs, (char*)workspace);
  if (workspace && x.w.bytes > workspace_bytes) return wg_set_error(WG_ERR_WORKSPACE, "training workspace too small");
  if ((size_t)x.g.R * 128 >= (1ull << 32)) return wg_set_error(WG_ERR_INVALID, "plane too large for 32-bit offsets");
  x.C = c.n_channels;
  x.nl = c.n_layers;
  x.FL = c.n_flows * c.n_layers;
  x.M8 = c.n_mel_channels * 8;
  x.K1 = 3 * x.C + x.M8;
  return WG_OK;
}

// every pointer of the weight / gradient blocks that the call sequence dereferences
int check_weights(const wg_train_weights* w, int n_flows) {
  if (!w->a1 || !w->a1c || !w->b1 || !w->a2 || !w->b2 || !w->es || !w->wat || !w->wbt || !w->wct || !w->wup || !w->bup ||
      !w->wstart || !w->bstart || !w->out_init || !w->w1x1)
    return wg_set_error(WG_ERR_INVALID, "wg_train_weights has a null member");
  for (int k = 0; k < n_flows; ++k)
    if (!w->wstart[k] || !w->bstart[k] || !w->out_init[k] || !w->w1x1[k])
      return wg_set_error(WG_ERR_INVALID, "wg_train_weights has a null per-flow pointer");
  return WG_OK;
}
int check_grads(const wg_train_grads* g, int n_flows) {
  if (!g->dw1 || !g->db1 || !g->dw2 || !g->db2 || !g->dwes || !g->dwup || !g->dbup || !g->dstart || !g->dout_init ||
      !g->dw1x1)
    return wg_set_error(WG_ERR_INVALID, "wg_train_grads has a null member");
  for (int k = 0; k < n_flows; ++k)
    if (!g->dstart[k] || !g->dout_init[k] || !g->dw1x1[k])
      return wg_set_error(WG_ERR_INVALID, "wg_train_grads has a null per-flow pointer");
  return WG_OK;
}

// One WN-layer-kernel GEMM over part `part` of `parts` equal row ranges of every phase (parts = 1: every column).
// (Measured and dropped: splitting a layer whose tile count is not a multiple of the CU count into a whole-rounds
// launch of 128-column tiles and a tail launch of 64-column tiles on the same stream.  A 64-column tile streams the same
// A fragments and takes ~80 % of a 128-column tile's time, so 2 + 0.8 rounds plus a second launch is no faster than 3.)
template <class Launch>
hipError_t launch_part(WnLayerArgs a, const RowGeom& g, int bn, int part, int parts, Launch launch) {
  const int rows = g.Rp / parts;
  a.row0 = part * rows;
  a.tiles_per_phase = rows / bn;
  a.n_tiles = kPhases * a.tiles_per_phase;
  return launch(a, bn);
}

SlabSeg make_seg(const float* slabs, int n_slabs, size_t stride, size_t n, float scale, float* out, int row_len, int perm) {
  SlabSeg g;
  g.slabs = slabs; g.out = out; g.stride = stride; g.n = n; g.n_slabs = n_slabs; g.scale = scale;
  g.row_len = row_len; g.perm = perm;
  g.blocked = 0; g.m_chunks = 0; g.k_chunks = 0; g.n_groups = 1; g.out_group_stride = 0;
  return g;
}

PRun run_of(const _Float16* base, int n_chunks, int dt) {
  PRun r;
  r.base = base;
  r.n_chunks = n_chunks;
  r.dt = dt;
  return r;
}

}  // namespace

extern "C" {

int32_t wg_wn_waves(int32_t n_channels) { return wn_waves(n_channels); }

int wg_train_pack(wg_handle* h, const wg_train_plain* in, const wg_train_weights* out, void* stream) {
  if (!h || !in || !out) return wg_set_error(WG_ERR_INVALID, "null argument");
  if (!in->w1 || !in->w2 || !in->wes || !in->wup || !out->a1 || !out->a1c || !out->a2 || !out->es || !out->wat ||
      !out->wbt || !out->wct || !out->wup)
    return wg_set_error(WG_ERR_INVALID, "wg_train_pack: null tensor");
  const wg_config& c = *wg_internal_config(h);
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_channels, M8 = c.n_mel_channels * 8, FL = c.n_flows * c.n_layers, NW = wn_waves(C);
  if (NW <= 0 || M8 % 64) return wg_set_error(WG_ERR_INVALID, "wg_train_pack: unsupported channel counts");
  const size_t K1 = 3 * (size_t)C + M8;
  PackArgs a;
  memset(&a, 0, sizeof a);
  a.C = C; a.M8 = M8; a.FL = FL; a.NW = NW;
  a.w1 = in->w1; a.w2 = in->w2; a.wes = in->wes; a.wup = in->wup;
  auto run = [&](int kind, const void* dst, const void* dst2, size_t elements) -> hipError_t {
    a.kind = kind;
    a.dst = (_Float16*)const_cast<void*>(dst);
    a.dst2 = (_Float16*)const_cast<void*>(dst2);
    a.n_pieces = elements / 8;
    return launch_pack(a, s);
  };
  TR_TRY(run(PACK_A1, out->a1, out->a1c, (size_t)FL * 2 * C * K1));
  TR_TRY(run(PACK_A2, out->a2, nullptr, (size_t)FL * C * C));
  TR_TRY(run(PACK_ES, out->es, nullptr, (size_t)FL * 16 * C));
  TR_TRY(run(PACK_WAT, out->wat, nullptr, (size_t)FL * C * (C + 64)));
  TR_TRY(run(PACK_WBT, out->wbt, nullptr, (size_t)FL * C * 6 * C));
  TR_TRY(run(PACK_WCT, out->wct, nullptr, (size_t)M8 * FL * 2 * C));
  TR_TRY(run(PACK_WUP, out->wup, nullptr, (size_t)32 * M8 * 512));
  return WG_OK;
}

size_t wg_train_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames, int32_t audio_len) {
  Ctx x;
  if (setup(const_cast<wg_handle*>(h), B, n_frames, audio_len, nullptr, 0, x) != WG_OK) return 0;
  return x.w.bytes;
}

int wg_train_forward(wg_handle* h, const wg_train_weights* wt, const void* mel, const void* audio, float* z,
                     float* const* log_s, int32_t B, int32_t n_frames, int32_t audio_len, int32_t fresh,
                     void* workspace, size_t workspace_bytes, void* stream) {
  if (!wt || !mel || !audio || !z || !log_s || !workspace) return wg_set_error(WG_ERR_INVALID, "null argument");
  Ctx x;
  int rc = setup(h, B, n_frames, audio_len, workspace, workspace_bytes, x);
  if (rc) return rc;
  if ((rc = check_weights(wt, x.c->n_flows))) return rc;
  const wg_config& c = *x.c;
  const RowGeom& g = x.g;
  TrainWs& w = x.w;
  hipStream_t s = (hipStream_t)stream;
  const int C = x.C, nl = x.nl, M8 = x.M8, K1 = x.K1;
  const int cc = C / 64, mc = M8 / 64;
  // per-layer sizes (fp16 elements) of the forward fragment tensors (include/waveglow_amd.h: wg_train_weights)
  const int NW = wn_waves(C), MBw = C / (32 * NW), MTw = 2 * MBw;
  const size_t a1_n = (size_t)2 * (3 * cc) * NW * MTw * 2 * 64 * 8, a1c_n = (size_t)2 * mc * NW * MTw * 2 * 64 * 8;
  const size_t a2_n = (size_t)NW * MBw * (C / 16) * 64 * 8, es_n = (size_t)(C / 32) * 64 * 8;
  const int n_cu = x.n_cu;
  // second chain (the caller's stream is the first): see setup
  hipStream_t sB = s;
  if (x.halves == 2 && !x.serial) {
    sB = wg_internal_aux_stream(h, 0);
    if (!sB) return wg_set_error(WG_ERR_HIP, "cannot create the second chain's stream");
  }
  // tile width of the fused layer kernel: as the inference path chooses it (api.cpp: run_wn)
  int BNw = wn_block_n(C);
  if (BNw == 128 && (int64_t)kPhases * (g.Rp / 128) < (int64_t)n_cu) BNw = 64;
  if (const char* e = getenv("WG_FORCE_BN")) {
    const int f = atoi(e);
    if (f == 64 || (f == 128 && wn_block_n(C) == 128)) BNw = f;
  }
  (void)K1;

  if (fresh) TR_TRY(hipMemsetAsync(workspace, 0, w.zero_bytes, s));
  TR_TRY(launch_mel_plane(mel, 0, c.n_mel_channels, g, w.MELP, s));
  {
    // upsample (ConvTranspose1d 1024/256, model.py:145-150, :186-189) + squeeze (:191-193): one matrix per phase
    PGemmArgs a;
    memset(&a, 0, sizeof a);
    a.n_runs = 4;
    for (int j = 0; j < 4; ++j) a.run[j] = run_of(w.MELP, 2, -32 * j);
    a.A = (const _Float16*)wt->wup;
    a.a_phase_stride = (long long)M8 * 512;
    a.ktot = 512;
    a.n_blk = M8 / 32;
    a.M = M8;
    a.bias = wt->bup;
    a.g = g;
    a.o0 = w.SP;
    TR_TRY(launch_plane_gemm(a, s));
  }
  int z_ch = 0;
  for (int k = 0; k <= c.n_flows; ++k) {
    FlowArgs f;
    memset(&f, 0, sizeof f);
    f.direction = 1;
    f.g = g;
    f.C = C;
    f.io_f16 = 0;
    f.z_out = z;
    f.z_out_ch0 = z_ch;
    f.first = (k == 0);
    f.last = (k == c.n_flows);
    if (f.first) {
      f.audio_in = audio;
      f.c_in = c.n_group;
    } else {
      f.Z = w.Zpost + (size_t)(k - 1) * w.rows8;
      f.out = w.OUT + (size_t)(k - 1) * w.rows8;
      f.c_in = x.ck[k - 1];
      f.h_in = f.c_in / 2;
      f.log_s_out = log_s[k - 1];
      if (!f.log_s_out) return wg_set_error(WG_ERR_INVALID, "null log_s entry");
    }
    if (!f.last) {
      f.Z_w = w.Zpost + (size_t)k * w.rows8;
      f.out_w = w.OUT + (size_t)k * w.rows8;
      f.n_peel = is_early(c, k) ? c.n_early_size : 0;
      f.c_next = x.ck[k];
      f.h_next = f.c_next / 2;
      if (f.c_in - f.n_peel != f.c_next) return wg_set_error(WG_ERR_STATE, "flow bookkeeping error");
      f.winv = wt->w1x1[k];
      f.wstart = wt->wstart[k];
      f.bstart = wt->bstart[k];
      f.out_init = wt->out_init[k];
      f.x = w.X + (size_t)(k * nl) * w.plane_c;
      z_ch += f.n_peel;
    }
    // (measured and dropped: each chain running its own half of the flow step, so that the chains never meet -- they
    //  drift apart and the forward pass took 0.2-1.0 ms longer than with this join per flow)
    TR_TRY(launch_flow(f, s));
    if (f.last) break;
    TR_ORDER(order_after(h, s, sB));
    for (int i = 0; i < nl; ++i) {
      const int fl = k * nl + i, d = 1 << i;
      const _Float16* Xi = w.X + (size_t)fl * w.plane_c;
      _Float16* Ai = w.A + (size_t)fl * w.plane_c;
      {
        // ONE fused launch per layer (kernels.hip: wn_layer_kernel<..., TR = true>): in_layers[i] + cond_layer slice as one
        // K-extended GEMM, gate in registers (tanh / sigmoid / acts saved as planes for the backward pass), res rows +
        // residual add -> x_{i+1}, skip rows folded through WN.end -> OUT   (model.py:123-137)
        WnLayerArgs a;
        memset(&a, 0, sizeof a);
        a.x_in = Xi;
        a.x_tap = Xi;
        a.x_chunks_per_tap = cc;
        a.x_out = (i < nl - 1) ? w.X + (size_t)(fl + 1) * w.plane_c : nullptr;
        a.wA1 = (const _Float16*)wt->a1 + (size_t)fl * a1_n;
        a.wA1c = (const _Float16*)wt->a1c + (size_t)fl * a1c_n;
        a.bias1 = wt->b1 + (size_t)fl * 2 * C;
        a.wA2 = (const _Float16*)wt->a2 + (size_t)fl * a2_n;
        a.bias2 = wt->b2 + (size_t)fl * C;
        a.wEs = (const _Float16*)wt->es + (size_t)fl * es_n;
        a.out = w.OUT + (size_t)k * w.rows8;
        a.g = g;
        a.dil = d;
        a.n_cond_steps = mc;
        a.M = c.n_mel_channels;
        a.has_res = i < nl - 1;
        a.n_cu = n_cu;
        a.sp = w.SP;
        a.save_t = w.T + (size_t)fl * w.plane_c;
        a.save_s = w.S + (size_t)fl * w.plane_c;
        a.save_a = Ai;
        for (int half = 0; half < x.halves; ++half) {
          hipStream_t sh = half ? sB : s;
          TR_PROF(sh, 4, TR_TRY(launch_part(a, g, BNw, half, x.halves, [&](const WnLayerArgs& q, int bn) { return launch_wn_layer_train(q, C, bn, sh); })));
        }
      }
    }
    TR_ORDER(order_after(h, sB, s));
  }
  return WG_OK;
}

int wg_train_backward(wg_handle* h, const wg_train_weights* wt, const wg_train_grads* gr, const float* g_z,
                      const float* const* g_log_s, float scale, const void* audio, int32_t B, int32_t n_frames,
                      int32_t audio_len, void* workspace, size_t workspace_bytes, void* stream) {
  const wg_config* c = wg_internal_config(h);
  if (!c) return wg_set_error(WG_ERR_INVALID, "null handle");
  return wg_train_backward_flows(h, wt, gr, g_z, g_log_s, scale, audio, B, n_frames, audio_len, workspace, workspace_bytes,
                                 c->n_flows - 1, 0, stream);
}

int wg_train_backward_flows(wg_handle* h, const wg_train_weights* wt, const wg_train_grads* gr, const float* g_z,
                            const float* const* g_log_s, float scale, const void* audio, int32_t B, int32_t n_frames,
                            int32_t audio_len, void* workspace, size_t workspace_bytes, int32_t flow_hi, int32_t flow_lo,
                            void* stream) {
  if (!wt || !gr || !audio || !workspace) return wg_set_error(WG_ERR_INVALID, "null argument");
  if (!(scale > 0.f)) return wg_set_error(WG_ERR_INVALID, "scale must be positive");
  Ctx x;
  int rc = setup(h, B, n_frames, audio_len, workspace, workspace_bytes, x);
  if (rc) return rc;
  if ((rc = check_weights(wt, x.c->n_flows)) || (rc = check_grads(gr, x.c->n_flows))) return rc;
  const wg_config& c = *x.c;
  const RowGeom& g = x.g;
  TrainWs& w = x.w;
  hipStream_t s = (hipStream_t)stream;
  const int C = x.C, nl = x.nl, M8 = x.M8, K1 = x.K1, FL = x.FL;
  const int cc = C / 64, mc = M8 / 64;
  const float inv = 1.0f / scale;
  const _Float16* wat = (const _Float16*)wt->wat;
  const _Float16* wbt = (const _Float16*)wt->wbt;
  // fragment tensors of the two dgrad GEMMs (include/waveglow_amd.h: wat, wbt): elements per 64-deep K-step and per layer
  const int NW = wn_waves(C), MBw = C / (32 * NW);
  const size_t kstep_n = (size_t)2 * NW * MBw * 2 * 64 * 8;
  const size_t wat_n = (size_t)(cc + 1) * kstep_n, wbt_n = (size_t)(6 * cc) * kstep_n;
  const int n_cu = x.n_cu;
  // Streams of one backward call.  The caller's stream `s` carries chain 0 (d acts / gate derivative and d x of the
  // first half of the batch) and the row kernels of every flow; sB carries chain 1 (the second half, see setup); sW
  // (lowest priority) carries the weight-gradient launches and their slab reductions, which nothing downstream in
  // the same call waits for: they fill the CUs the chains leave idle.  Everything is joined back into `s` before the
  // call returns, so the caller sees the usual stream semantics.  WG_TRAIN_SERIAL=1: all three are `s`.
  // (Two chains pay in the forward pass, -23 % per layer at config 4; in the backward pass the weight-gradient stream
  //  already fills the idle CUs and a second chain measured +0.7 ms per step: off unless WG_TRAIN_BWD_HALVES=2, tests.)
  int bh = 1;
  if (const char* e = getenv("WG_TRAIN_BWD_HALVES")) bh = (atoi(e) == 2) ? x.halves : 1;
  hipStream_t sB = s, sW = s, sR = s;
  if (!x.serial) {
    if (bh == 2 && !(sB = wg_internal_aux_stream(h, 0))) return wg_set_error(WG_ERR_HIP, "cannot create the second chain's stream");
    if (!(sW = wg_internal_aux_stream(h, 1))) return wg_set_error(WG_ERR_HIP, "cannot create the weight-gradient stream");
    if (!(sR = wg_internal_aux_stream(h, 2))) return wg_set_error(WG_ERR_HIP, "cannot create the slab-reduction stream");
  }
  const int* const n_slabs = x.n_slabs;
  bool fuse = true;                       // WG_TRAIN_NO_FUSE=1 (tests, A/B): d acts + gate derivative as launches of their own
  if (const char* e = getenv("WG_TRAIN_NO_FUSE")) fuse = !(*e == '1');
  int BNw = wn_block_n(C);
  if (BNw == 128 && (int64_t)kPhases * (g.Rp / 128) < (int64_t)n_cu) BNw = 64;
  if (const char* e = getenv("WG_FORCE_BN")) {
    const int f = atoi(e);
    if (f == 64 || (f == 128 && wn_block_n(C) == 128)) BNw = f;
  }
  if (const char* e = getenv("WG_TRAIN_BWD_BN")) {      // A/B runs: tile width of the backward launches alone
    const int f = atoi(e);
    if (f == 64 || (f == 128 && wn_block_n(C) == 128)) BNw = f;
  }
  // where entry fl of a per-layer gradient tensor lives: dense, or in interleaved per-layer records (wg_train_grads)
  if ((gr->layer_stride == 0) != (gr->flow_stride == 0) || gr->layer_stride < 0 || gr->flow_stride < 0)
    return wg_set_error(WG_ERR_INVALID, "wg_train_grads: layer_stride and flow_stride must both be 0 or both positive");
  auto gofs = [&](int fl, size_t dense) -> size_t {
    return gr->layer_stride ? (size_t)(fl / nl) * (size_t)gr->flow_stride + (size_t)(fl % nl) * (size_t)gr->layer_stride
                            : (size_t)fl * dense;
  };

  if (flow_lo < 0 || flow_hi >= c.n_flows || flow_lo > flow_hi) return wg_set_error(WG_ERR_INVALID, "bad flow range");
  // channel offsets of the peeled outputs in z (model.py:201-203, :220): early outputs of the flows <= k
  auto early_channels_upto = [&](int k) {
    int n = 0;
    for (int q = 0; q <= k; ++q)
      if (is_early(c, q)) n += c.n_early_size;
    return n;
  };
  int z_final_ch0 = early_channels_upto(flow_hi);

  // Buffers the streams share and what orders their reuse:
  //   GXL[i] (d x_i)   written by the chains' d x launch of layer i, read by their layer i-1 launches (same stream) and by
  //                    sW's second job of layer i-1: the next flow's layer-i launch waits for that job (w_done[i-1]);
  //   GO[k & 1]        written by flow k's pre kernel on s, read by the chains and by all of flow k's jobs on sW: the pre
  //                    kernel of flow k-2 waits for the last of them (w_flow[k & 1]);
  //   GP, X, T, S, A   one set per layer of the whole model: no reuse inside a call.
  TR_ORDER(order_after(h, s, sW));
  // A layer's slabs are reduced by a launch of its own on a third stream (sR, lowest priority too): it is HBM-bound and
  // small in registers and LDS, so its workgroups run beside the NEXT layer's weight-gradient workgroups (MFMA / L2-bound,
  // one per CU).  Two slab sets: launch n + 2 on sW waits for the reduction of launch n (r_done), the reduction of launch
  // n for launch n itself (l_done).
  int n_layer = 0;
  // (marks are waited on one or two flows / launches after their record: events of their own, wg_internal_mark_event,
  //  slots 0-9 = w_done[layer], 10-11 = w_flow[parity], 12-13 = l_done[set], 14-15 = r_done[set]; a null entry = not recorded
  //  in this call, nothing to wait for)
  hipEvent_t w_done[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t w_flow[2] = {nullptr, nullptr}, l_done[2] = {nullptr, nullptr}, r_done[2] = {nullptr, nullptr};
  if (nl > 10) return wg_set_error(WG_ERR_INVALID, "more than 10 layers");
  auto mark = [&](hipStream_t st, hipEvent_t& e, int slot) -> hipError_t {
    e = nullptr;
    if (x.serial) return hipSuccess;
    e = wg_internal_mark_event(h, slot);
    return e ? hipEventRecord(e, st) : hipErrorOutOfMemory;
  };
  auto wait_for = [&](hipStream_t st, hipEvent_t e) -> hipError_t { return e ? hipStreamWaitEvent(st, e, 0) : hipSuccess; };

  for (int k = flow_hi; k >= flow_lo; --k) {
    const int ck = x.ck[k], hk = ck / 2;
    FlowBwdArgs fb;
    memset(&fb, 0, sizeof fb);
    fb.g = g;
    fb.C = C;
    fb.c = ck;
    fb.h = hk;
    fb.scale = scale;
    fb.Zpost = w.Zpost + (size_t)k * w.rows8;
    fb.OUT = w.OUT + (size_t)k * w.rows8;
    fb.g_z = g_z;
    fb.g_log_s = g_log_s ? g_log_s[k] : nullptr;
    fb.from_z = (k == c.n_flows - 1);
    fb.z_ch0 = early_channels_upto(c.n_flows - 1);
    fb.GZ = w.GZ;
    _Float16* const GOk = w.GO[k & 1];
    fb.GO = GOk;
    TR_ORDER(wait_for(s, w_flow[k & 1]));
    TR_TRY(launch_flow_bwd_pre(fb, s));
    TR_ORDER(order_after(h, s, sB));

    _Float16* gx = nullptr;                     // gx = d x_{i+1} (null: zero, the last layer has no res output)
    for (int i = nl - 1; i >= 0; --i) {
      const int fl = k * nl + i, d = 1 << i;
      const _Float16* Xi = w.X + (size_t)fl * w.plane_c;
      const _Float16* Ai = w.A + (size_t)fl * w.plane_c;
      _Float16* GPi = w.GP + (size_t)fl * 2 * w.plane_c;
      // Fused (round 3): the d x launch of layer i + 1 has already produced d pre of this layer behind its own result
      // (wn_layer_kernel MODE 4: the d x tile goes through LDS into the next GEMM instead of out to the planes and back in
      // through a launch of its own); only a flow's last layer, which has no d x above it, runs MODE 3 alone.
      if (!fuse || i == nl - 1) {
        // d acts = W_res^T d x_{i+1} + (W_end W_skip_i)^T d out ; gate derivative -> d pre   (wn_layer_kernel MODE 3)
        WnLayerArgs a;
        memset(&a, 0, sizeof a);
        const _Float16* Am = wat + (size_t)fl * wat_n;
        if (gx) {
          a.x_tap = gx;
          a.x_chunks_per_tap = cc;
          a.sp = GOk;
          a.n_cond_steps = 1;
          a.wA1 = Am;
          a.wA1c = Am + (size_t)cc * kstep_n;
        } else {
          a.x_tap = GOk;                     // last layer of a flow: no d x_{i+1}, the d out plane alone
          a.x_chunks_per_tap = 1;
          a.n_cond_steps = 0;
          a.wA1 = Am + (size_t)cc * kstep_n;
          a.wA1c = a.wA1;
        }
        a.dil = 0;
        a.g = g;
        a.M = c.n_mel_channels;
        a.n_cu = n_cu;
        a.in0 = w.T + (size_t)fl * w.plane_c;
        a.in1 = w.S + (size_t)fl * w.plane_c;
        a.out0 = GPi;
        for (int half = 0; half < bh; ++half) {
          hipStream_t sh = half ? sB : s;
          TR_PROF(sh, 5, TR_TRY(launch_part(a, g, BNw, half, bh, [&](const WnLayerArgs& q, int bn) { return launch_wn_plain(q, C, 3, bn, sh); })));
        }
      }
      // the weight-gradient stream continues once both chains have written their half of d pre (and of d x_{i+1} before it)
      TR_ORDER(order_after(h, s, sW));
      TR_ORDER(order_after(h, sB, sW));
      {
        // d W1 = d pre x [x taps | spect]^T, d b1;
        // d W2 = d x_{i+1} x acts^T, d b2  and  d (W_end W_skip_i) = d out x acts^T  share the X operand (acts): one
        // job with the d out plane as the `extra` 16 rows (the last layer has no d x: the d out plane stands in as the
        // job's G as well, and that part of the result is not used).
        // Both jobs in ONE launch over the same slab partition (train.hip: wgrad_kernel).
        WgradJob jb[2];
        const int set = n_layer & 1;
        ++n_layer;
        memset(jb, 0, sizeof jb);
        jb[0].G = GPi;
        jb[0].m_chunks = 2 * cc;
        jb[0].n_runs = 4;
        jb[0].run[0] = run_of(Xi, cc, -d);
        jb[0].run[1] = run_of(Xi, cc, 0);
        jb[0].run[2] = run_of(Xi, cc, d);
        jb[0].run[3] = run_of(w.SP, mc, 0);
        jb[0].k_chunks = 3 * cc + mc;
        jb[0].slabs = w.slab[set];
        jb[0].bias_out = w.part[set];
        jb[1].G = gx ? gx : GOk;
        jb[1].m_chunks = gx ? cc : 1;
        jb[1].G_extra = GOk;
        jb[1].n_runs = 1;
        jb[1].run[0] = run_of(Ai, cc, 0);
        jb[1].k_chunks = cc;
        jb[1].slabs = w.slab2[set];
        jb[1].bias_out = w.part2[set];
        jb[1].extra_out = w.ext[set];
        jb[1].extra_bias_out = w.extb[set];
        TR_ORDER(wait_for(sW, r_done[set]));          // the reduction of the launch before last has read this slab set
        TR_PROF(sW, 6, TR_TRY(launch_wgrad(jb, 2, g, n_slabs, sW)));
        TR_ORDER(mark(sW, l_done[set], 12 + set));
        // ---- reduction of everything this launch left behind, in NATURAL channel order (SlabSeg: perm bit 0 = rows are
        // channels, bit 1 = columns are)
        SlabSeg seg[kMaxSlabSegs];
        int n_seg = 0;
        const size_t n1 = (size_t)2 * C * K1;
        auto add_flat = [&](int job, const float* slabs, size_t stride, size_t n, float* out, int row_len, int perm) {
          seg[n_seg++] = make_seg(slabs, n_slabs[job], stride, n, inv, out, row_len, perm);
        };
        auto add_blocked = [&](int job, const float* slabs, int m_ch, int k_ch, float* out) {
          SlabSeg q = make_seg(slabs, n_slabs[job], (size_t)wgrad_tiles(m_ch, k_ch) * kWgradTileFloats,
                               (size_t)wgrad_tiles(m_ch, k_ch) * kWgradTileFloats, inv, out, k_ch * 64, 3);
          q.blocked = 1; q.m_chunks = m_ch; q.k_chunks = k_ch; q.n_groups = 1;
          seg[n_seg++] = q;
        };
        add_blocked(0, w.slab[set], 2 * cc, 3 * cc + mc, gr->dw1 + gofs(fl, n1));
        add_flat(0, w.part[set], (size_t)2 * C, (size_t)2 * C, gr->db1 + gofs(fl, (size_t)2 * C), 2 * C, 2);
        if (gx) {
          add_blocked(1, w.slab2[set], cc, cc, gr->dw2 + gofs(fl, (size_t)C * C));
          add_flat(1, w.part2[set], (size_t)C, (size_t)C, gr->db2 + gofs(fl, (size_t)C), C, 2);
        }
        add_flat(1, w.ext[set], (size_t)16 * C, (size_t)8 * C, gr->dwes + gofs(fl, (size_t)8 * C), C, 2);
        // d out_init = sum over columns of (d b | d log_s), once per flow
        if (i == 0) add_flat(1, w.extb[set], 16, 8, gr->dout_init[k], 0, 0);
        TR_ORDER(wait_for(sR, l_done[set]));
        TR_TRY(launch_slab_reduce_multi(seg, n_seg, sR));
        TR_ORDER(mark(sR, r_done[set], 14 + set));
        TR_ORDER(mark(sW, w_done[i], i));
      }
      {
        // d x_i = d x_{i+1} + sum_tap W_in[tap]^T d pre(t - (tap-1) d)   (wn_layer_kernel MODE 2: taps at +d, 0, -d)
        WnLayerArgs a;
        memset(&a, 0, sizeof a);
        a.x_tap = GPi;
        a.x_chunks_per_tap = 2 * cc;
        a.n_cond_steps = 0;
        a.wA1 = wbt + (size_t)fl * wbt_n;
        a.wA1c = a.wA1;
        a.dil = -d;
        a.g = g;
        a.M = c.n_mel_channels;
        a.n_cu = n_cu;
        a.in0 = gx;
        _Float16* const gxi = w.GXL + (size_t)i * w.plane_c;
        a.out0 = gxi;
        const int kind = (fuse && i > 0) ? 4 : 2;
        if (kind == 4) {      // ... and d acts + gate derivative of layer i - 1 on the tile (see above)
          a.wat_prev = wat + (size_t)(fl - 1) * wat_n;
          a.gout = GOk;
          a.t_prev = w.T + (size_t)(fl - 1) * w.plane_c;
          a.s_prev = w.S + (size_t)(fl - 1) * w.plane_c;
          a.dpre_prev = w.GP + (size_t)(fl - 1) * 2 * w.plane_c;
        }
        for (int half = 0; half < bh; ++half) {
          hipStream_t sh = half ? sB : s;
          if (i > 0) TR_ORDER(wait_for(sh, w_done[i - 1]));     // the previous flow's reader of GXL[i] (not yet re-marked: layer i-1 of this flow comes later)
          TR_PROF(sh, 5, TR_TRY(launch_part(a, g, BNw, half, bh, [&](const WnLayerArgs& q, int bn) { return launch_wn_plain(q, C, kind, bn, sh); })));
        }
        gx = gxi;
      }
    }
    TR_ORDER(mark(sW, w_flow[k & 1], 10 + (k & 1)));
    TR_ORDER(order_after(h, sB, s));
    {
      StartWgradArgs a;
      a.g = g;
      a.C = C;
      a.h = hk;
      a.GX = gx;
      a.Zpost = fb.Zpost;
      a.partial = w.part3;
      TR_TRY(launch_start_wgrad(a, s));
      const SlabSeg sg = make_seg(w.part3, start_wgrad_workgroups(g), (size_t)5 * C, (size_t)5 * C, inv, gr->dstart[k], C, 2);
      TR_TRY(launch_slab_reduce_multi(&sg, 1, s));
    }
    fb.GX = gx;
    fb.wstart = wt->wstart[k];
    fb.w1x1 = wt->w1x1[k];
    if (k > 0) {
      fb.Zprev = w.Zpost + (size_t)(k - 1) * w.rows8;
      fb.OUTprev = w.OUT + (size_t)(k - 1) * w.rows8;
      fb.h_prev = x.ck[k - 1] / 2;
      fb.n_peel = is_early(c, k) ? c.n_early_size : 0;
      if (fb.n_peel) z_final_ch0 -= fb.n_peel;
      fb.z_peel_ch0 = z_final_ch0;
    } else {
      fb.audio = (const float*)audio;
    }
    fb.dw_partial = w.part3;
    TR_TRY(launch_flow_bwd_post(fb, s));
    const SlabSeg sg = make_seg(w.part3, flow_bwd_workgroups(g), 64, 64, inv, gr->dw1x1[k], 0, 0);
    TR_TRY(launch_slab_reduce_multi(&sg, 1, s));
  }
  if (flow_lo > 0) {                      // the upsample gradient needs the d pre planes of every flow
    TR_ORDER(order_after(h, sW, s));      // every gradient of the call is final on the caller's stream
    TR_ORDER(order_after(h, sR, s));
    return WG_OK;
  }
  {
    // d spect = sum over every layer of cond_layer^T d pre: ONE GEMM with K = FL*2C over the kept d pre planes
    PGemmArgs a;
    memset(&a, 0, sizeof a);
    a.n_runs = 1;
    a.run[0] = run_of(w.GP, FL * 2 * cc, 0);
    a.A = (const _Float16*)wt->wct;
    a.ktot = FL * 2 * C;
    a.n_blk = M8 / 32;
    a.M = M8;
    a.g = g;
    a.o0 = w.GSP;
    TR_TRY(launch_plane_gemm(a, s));
  }
  {
    WgradJob a;   // d upsample: per phase, d spect x mel frames q..q-3 (no sum over phases: one slab = one phase = one result)
    memset(&a, 0, sizeof a);
    a.G = w.GSP;
    a.m_chunks = mc;
    a.n_runs = 4;
    for (int j = 0; j < 4; ++j) a.run[j] = run_of(w.MELP, 2, -32 * j);
    a.k_chunks = 8;
    a.slabs = w.slab_up;
    a.bias_out = w.part3;
    const int up_slabs = kPhases;
    TR_TRY(launch_wgrad(&a, 1, g, &up_slabs, s));
    SlabSeg sg[2];
    const size_t tile_n = (size_t)wgrad_tiles(mc, 8) * kWgradTileFloats;
    sg[0] = make_seg(w.slab_up, kPhases, tile_n, tile_n, inv, gr->dwup, 512, 1);      // rows to natural order, columns are mel taps
    sg[0].blocked = 1; sg[0].m_chunks = mc; sg[0].k_chunks = 8; sg[0].n_groups = kPhases; sg[0].out_group_stride = (size_t)M8 * 512;
    sg[1] = make_seg(w.part3, kPhases, M8, M8, inv, gr->dbup, M8, 2);
    TR_TRY(launch_slab_reduce_multi(sg, 2, s));
  }
  TR_ORDER(order_after(h, sW, s));        // every gradient of the call is final on the caller's stream
  TR_ORDER(order_after(h, sR, s));
  return WG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Training plumbing on the device (train_prep.hip): parameters in their own tensors -> wg_train_weights, packed gradients ->
// one gradient per parameter.
// ---------------------------------------------------------------------------------------------
namespace {

struct ParamDesc { int sec, idx; long long numel; std::string name; };

// canonical parameter list: section-major (wg_train.h: PrepSec), g sections only for weight-normed modules
std::vector<ParamDesc> param_list(const wg_config& c, const int* ck, bool wn) {
  std::vector<ParamDesc> out;
  const int C = c.n_channels, nl = c.n_layers, nf = c.n_flows, M = c.n_mel_channels, M8 = 8 * M, FL = nf * nl;
  auto wname = [&](const std::string& mod, int which) {     // 0: v / dense weight, 1: g
    if (!wn) return mod + ".weight";
    return mod + (which ? ".parametrizations.weight.original0" : ".parametrizations.weight.original1");
  };
  for (int sec = 0; sec <= SEC_UP_B; ++sec) {
    const bool is_g = sec == SEC_IN_G || sec == SEC_RS_G || sec == SEC_CO_G || sec == SEC_ST_G;
    if (is_g && !wn) continue;
    const int n = prep_sec_len(FL, nf, sec);
    for (int q = 0; q < n; ++q) {
      ParamDesc d;
      d.sec = sec;
      d.idx = q;
      const int k = sec <= SEC_RS_B ? q / nl : q, i = sec <= SEC_RS_B ? q % nl : 0, h = ck[k < nf ? k : 0] / 2;
      const std::string wnp = "WN." + std::to_string(k) + ".";
      const std::string in = wnp + "in_layers." + std::to_string(i), rs = wnp + "res_skip_layers." + std::to_string(i);
      const long long rs_rows = i < nl - 1 ? 2 * C : C;
      switch (sec) {
        case SEC_IN_V: d.name = wname(in, 0); d.numel = (long long)2 * C * C * 3; break;
        case SEC_IN_G: d.name = wname(in, 1); d.numel = 2 * C; break;
        case SEC_IN_B: d.name = in + ".bias"; d.numel = 2 * C; break;
        case SEC_RS_V: d.name = wname(rs, 0); d.numel = rs_rows * C; break;
        case SEC_RS_G: d.name = wname(rs, 1); d.numel = rs_rows; break;
        case SEC_RS_B: d.name = rs + ".bias"; d.numel = rs_rows; break;
        case SEC_CO_V: d.name = wname(wnp + "cond_layer", 0); d.numel = (long long)2 * C * nl * M8; break;
        case SEC_CO_G: d.name = wname(wnp + "cond_layer", 1); d.numel = (long long)2 * C * nl; break;
        case SEC_CO_B: d.name = wnp + "cond_layer.bias"; d.numel = (long long)2 * C * nl; break;
        case SEC_ST_V: d.name = wname(wnp + "start", 0); d.numel = (long long)C * h; break;
        case SEC_ST_G: d.name = wname(wnp + "start", 1); d.numel = C; break;
        case SEC_ST_B: d.name = wnp + "start.bias"; d.numel = C; break;
        case SEC_EN_W: d.name = wnp + "end.weight"; d.numel = (long long)2 * h * C; break;
        case SEC_EN_B: d.name = wnp + "end.bias"; d.numel = 2 * h; break;
        case SEC_CV_W: d.name = "convinv." + std::to_string(k) + ".conv.weight"; d.numel = (long long)ck[k] * ck[k]; break;
        case SEC_UP_W: d.name = "upsample.weight"; d.numel = (long long)M * M * c.upsample_kernel; break;
        default: d.name = "upsample.bias"; d.numel = M; break;
      }
      out.push_back(d);
    }
  }
  return out;
}

struct PrepLayout {
  size_t tab, goff, s_in, s_co, s_rs, s_st, wend8, bsum, wes, bytes;
  int n_slots, n_scale[4];
};
PrepLayout prep_layout(const wg_config& c) {
  PrepLayout L;
  const int C = c.n_channels, nl = c.n_layers, nf = c.n_flows, FL = nf * nl;
  L.n_slots = prep_slot_n(FL, nf, SEC_COUNT, 0);
  L.n_scale[0] = FL * 2 * C; L.n_scale[1] = nf * 2 * C * nl; L.n_scale[2] = FL * 2 * C; L.n_scale[3] = nf * C;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
  L.tab = take((size_t)L.n_slots * sizeof(void*));
  L.goff = take((size_t)L.n_slots * sizeof(long long));
  L.s_in = take((size_t)2 * L.n_scale[0] * 4);
  L.s_co = take((size_t)2 * L.n_scale[1] * 4);
  L.s_rs = take((size_t)2 * L.n_scale[2] * 4);
  L.s_st = take((size_t)2 * L.n_scale[3] * 4);
  L.wend8 = take((size_t)nf * 8 * C * 4);
  L.bsum = take((size_t)nf * C * 4);
  L.wes = take((size_t)FL * 8 * C * 4);
  L.bytes = off;
  return L;
}

// fills the argument block and uploads the pointer table (parameters + the per-flow buffers of `wt` / `gr`)
int prep_args(wg_handle* h, const void* const* params, int weight_normed, const wg_train_weights* wt, const wg_train_grads* gr,
              void* aux, size_t aux_bytes, float* flat, hipStream_t s, PrepArgs& a) {
  if (!h || !params || !aux) return wg_set_error(WG_ERR_INVALID, "null argument");
  const wg_config& c = *wg_internal_config(h);
  const int* ck = wg_internal_flow_channels(h);
  if (c.n_flows > kPrepMaxFlows) return wg_set_error(WG_ERR_INVALID, "more than 32 flows are not supported by the training direction");
  if (c.upsample_kernel != 1024) return wg_set_error(WG_ERR_INVALID, "upsample kernel must be 1024");
  const PrepLayout L = prep_layout(c);
  if (aux_bytes < L.bytes) return wg_set_error(WG_ERR_WORKSPACE, "wg_train_prepare: aux buffer too small");
  const int nf = c.n_flows, FL = nf * c.n_layers;
  std::vector<void*> tab((size_t)L.n_slots, nullptr);
  std::vector<long long> goff((size_t)L.n_slots, 0);
  const std::vector<ParamDesc> pl = param_list(c, ck, weight_normed != 0);
  long long off = 0;
  for (size_t i = 0; i < pl.size(); ++i) {
    if (!params[i]) return wg_set_error(WG_ERR_INVALID, ("null parameter pointer: " + pl[i].name).c_str());
    const int slot = prep_slot_n(FL, nf, pl[i].sec, pl[i].idx);
    tab[slot] = const_cast<void*>(params[i]);
    goff[slot] = off;
    off += pl[i].numel;
  }
  for (int k = 0; k < nf; ++k) {
    if (wt) {
      tab[prep_slot_n(FL, nf, SEC_O_WSTART, k)] = const_cast<float*>(wt->wstart[k]);
      tab[prep_slot_n(FL, nf, SEC_O_BSTART, k)] = const_cast<float*>(wt->bstart[k]);
      tab[prep_slot_n(FL, nf, SEC_O_OINIT, k)] = const_cast<float*>(wt->out_init[k]);
      tab[prep_slot_n(FL, nf, SEC_O_W1X1, k)] = const_cast<float*>(wt->w1x1[k]);
    }
    if (gr) {
      tab[prep_slot_n(FL, nf, SEC_G_DSTART, k)] = gr->dstart[k];
      tab[prep_slot_n(FL, nf, SEC_G_DOINIT, k)] = gr->dout_init[k];
      tab[prep_slot_n(FL, nf, SEC_G_DW1X1, k)] = gr->dw1x1[k];
    }
  }
  char* base = (char*)aux;
  // (through the handle's pinned staging buffers: `tab` / `goff` die with this call, the copies run later on the stream)
  TR_TRY(wg_internal_upload(h, base + L.tab, tab.data(), tab.size() * sizeof(void*), s));
  TR_TRY(wg_internal_upload(h, base + L.goff, goff.data(), goff.size() * sizeof(long long), s));
  memset(&a, 0, sizeof a);
  a.tab = (void* const*)(base + L.tab);
  a.goff = (const long long*)(base + L.goff);
  a.flat = flat;
  a.s_in = (float*)(base + L.s_in); a.s_co = (float*)(base + L.s_co); a.s_rs = (float*)(base + L.s_rs); a.s_st = (float*)(base + L.s_st);
  a.wend8 = (float*)(base + L.wend8); a.bsum = (float*)(base + L.bsum); a.wes = (float*)(base + L.wes);
  a.C = c.n_channels; a.M8 = c.n_mel_channels * 8; a.nl = c.n_layers; a.nf = nf; a.FL = FL;
  for (int k = 0; k < nf; ++k) { a.ck[k] = ck[k]; a.hk[k] = ck[k] / 2; }
  for (int q = 0; q < 4; ++q) a.n_scale[q] = L.n_scale[q];
  if (wt) { a.b1 = const_cast<float*>(wt->b1); a.b2 = const_cast<float*>(wt->b2); a.bup = const_cast<float*>(wt->bup); }
  if (gr) {
    a.dw1 = gr->dw1; a.db1 = gr->db1; a.dw2 = gr->dw2; a.db2 = gr->db2; a.dwes = gr->dwes; a.dwup = gr->dwup; a.dbup = gr->dbup;
    a.layer_stride = gr->layer_stride; a.flow_stride = gr->flow_stride;
  }
  return WG_OK;
}

}  // namespace

extern "C" {

int32_t wg_train_param_count(const wg_handle* h, int32_t weight_normed) {
  if (!h) return 0;
  return (int32_t)param_list(*wg_internal_config(h), wg_internal_flow_channels(h), weight_normed != 0).size();
}

const char* wg_train_param_name(const wg_handle* h, int32_t weight_normed, int32_t i) {
  static thread_local std::string name;
  if (!h) return "";
  const std::vector<ParamDesc> pl = param_list(*wg_internal_config(h), wg_internal_flow_channels(h), weight_normed != 0);
  if (i < 0 || (size_t)i >= pl.size()) return "";
  name = pl[i].name;
  return name.c_str();
}

int64_t wg_train_param_numel(const wg_handle* h, int32_t weight_normed, int32_t i) {
  if (!h) return 0;
  const std::vector<ParamDesc> pl = param_list(*wg_internal_config(h), wg_internal_flow_channels(h), weight_normed != 0);
  return (i < 0 || (size_t)i >= pl.size()) ? 0 : pl[i].numel;
}

size_t wg_train_prepare_bytes(const wg_handle* h) { return h ? prep_layout(*wg_internal_config(h)).bytes : 0; }

int wg_train_prepare(wg_handle* h, const void* const* params, int32_t weight_normed, const wg_train_weights* out, void* aux,
                     size_t aux_bytes, void* stream) {
  if (!out) return wg_set_error(WG_ERR_INVALID, "null argument");
  if (!out->a1 || !out->a1c || !out->b1 || !out->a2 || !out->b2 || !out->es || !out->wat || !out->wbt || !out->wct || !out->wup ||
      !out->bup || !out->wstart || !out->bstart || !out->out_init || !out->w1x1)
    return wg_set_error(WG_ERR_INVALID, "wg_train_prepare: wg_train_weights has a null member");
  hipStream_t s = (hipStream_t)stream;
  PrepArgs pa;
  int rc = prep_args(h, params, weight_normed, out, nullptr, aux, aux_bytes, nullptr, s, pa);
  if (rc) return rc;
  TR_TRY(launch_prepare(pa, s));
  const wg_config& c = *wg_internal_config(h);
  const int C = c.n_channels, M8 = c.n_mel_channels * 8, FL = c.n_flows * c.n_layers, NW = wn_waves(C);
  if (NW <= 0 || M8 % 64) return wg_set_error(WG_ERR_INVALID, "wg_train_prepare: unsupported channel counts");
  const size_t K1 = 3 * (size_t)C + M8;
  PackArgs a;
  memset(&a, 0, sizeof a);
  a.C = C; a.M8 = M8; a.FL = FL; a.NW = NW;
  a.native = 1;
  a.prep = pa;
  a.wes = pa.wes;
  auto run = [&](int kind, const void* dst, const void* dst2, size_t elements) -> hipError_t {
    a.kind = kind;
    a.dst = (_Float16*)const_cast<void*>(dst);
    a.dst2 = (_Float16*)const_cast<void*>(dst2);
    a.n_pieces = elements / 8;
    return launch_pack(a, s);
  };
  TR_TRY(run(PACK_A1, out->a1, out->a1c, (size_t)FL * 2 * C * K1));
  TR_TRY(run(PACK_A2, out->a2, nullptr, (size_t)FL * C * C));
  TR_TRY(run(PACK_ES, out->es, nullptr, (size_t)FL * 16 * C));
  TR_TRY(run(PACK_WAT, out->wat, nullptr, (size_t)FL * C * (C + 64)));
  TR_TRY(run(PACK_WBT, out->wbt, nullptr, (size_t)FL * C * 6 * C));
  TR_TRY(run(PACK_WCT, out->wct, nullptr, (size_t)M8 * FL * 2 * C));
  TR_TRY(run(PACK_WUP, out->wup, nullptr, (size_t)32 * M8 * 512));
  return WG_OK;
}

int wg_train_param_grads(wg_handle* h, const void* const* params, int32_t weight_normed, const wg_train_grads* grads, void* aux,
                         size_t aux_bytes, float* flat, void* stream) {
  if (!grads || !flat) return wg_set_error(WG_ERR_INVALID, "null argument");
  int rc = check_grads(grads, wg_internal_config(h) ? wg_internal_config(h)->n_flows : 0);
  if (rc) return rc;
  PrepArgs pa;
  rc = prep_args(h, params, weight_normed, nullptr, grads, aux, aux_bytes, flat, (hipStream_t)stream, pa);
  if (rc) return rc;
  TR_TRY(launch_param_grads(pa, (hipStream_t)stream));
  return WG_OK;
}

}  // extern "C"
