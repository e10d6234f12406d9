// Training direction (WaveGlow.forward + backward, src/waveglow/model.py:178-221 under autograd; train.py:190-199):
// kernel argument blocks shared by train.hip and train_api.cpp.  gfx950 only.
//
// Everything here works on the same fp16 planes as the inference path (wg_common.h: RowGeom, phase-major rows,
// position-major channels inside every 32-block).  Unlike inference, weights change every optimiser step, so no
// weight is pre-packed on the host: the caller hands over every matrix as an fp16 device buffer in "(pos,pos)"
// order (rows and K columns permuted by chan_to_pos inside 32-blocks) laid out in MFMA-fragment order (PGemmArgs::A;
// one torch gather per matrix type and step), and the
// cond_layer o upsample fold is NOT used (it would have to be rebuilt every step) -- the upsampled, squeezed
// spectrogram exists as planes and cond_layer is a K-segment of GEMM 1.
#pragma once
#include "wg_common.h"

namespace wg {

// One run of K-chunks of a plane-GEMM's B operand: n_chunks consecutive 64-channel planes starting at `base`
// (chunk stride = R*64 elements), read at time offset dt (group-timesteps; phase-major rows make that a phase
// change plus a frame shift, see RowGeom).  dt = -d / 0 / +d for the dilated taps, -32j for mel tap j.
struct PRun {
  const _Float16* base;
  int n_chunks;
  int dt;
};
constexpr int kMaxRuns = 4;

// D[M x columns] = A[M x K] . B[K x columns] (+ bias) for every column tile of every phase, stored as fp16 planes
// (invalid columns as zeros).
struct PGemmArgs {
  PRun run[kMaxRuns];
  int n_runs;
  const _Float16* A;        // fp16 matrix in MFMA-fragment order [K/64 steps][n_blk][4 sub-steps][64 lanes][8]:
                            //   lane (r = lane & 31, h = lane >> 5), element j of sub-step s of block b, step t
                            //   = Mat[32 b + chan_to_pos(r)][64 t + 32 h + 8 s + j]   with Mat in (pos,pos) order
                            // (a K offset of c chunks = A + c * n_blk * 2048)
  long long a_phase_stride; // elements added to A per phase (upsample: one matrix per phase), else 0
  int ktot;                 // sum of n_chunks * 64
  int n_blk;                // 32-row blocks of the whole matrix
  int M;                    // number of matrix rows
  const float* bias;        // fp32, pos order, or null
  RowGeom g;
  _Float16* o0;             // output planes [M/64][R][64]
};

// Weight gradient dW[m][k'] = sum over rows of G[row][m] * X[row (+ shift)][k'] (train.hip: wgrad_kernel).
// The rows (32 phases x Rp) are cut into n_slabs contiguous ranges of 32-row steps (a range may cross phase boundaries);
// one workgroup = one output tile x one slab; it leaves its fp32 partial tile in `slabs` in BLOCKED order (the order of
// its accumulator registers: every store instruction writes one contiguous KiB) and slab_reduce sums the slabs and
// un-blocks.  Tiles: 256 x 256 (4 x 4 chunks of 64) over the whole multiples of 4 K'-chunks, then 512 x 128 (8 x 2)
// over the remaining 1-3 K'-chunks (wgrad_tile below: the same decode on the host, in the kernel and in the reduction).
struct WgradJob {
  const _Float16* G;        // planes [m_chunks][R][64]
  int m_chunks;
  const _Float16* G_extra;  // optional: one more 64-channel plane whose first 16 channels give 16 extra output rows
                            // (the d out plane behind d x: d (W_end W_skip) shares the acts operand with d W_res)
  PRun run[kMaxRuns];       // the X operand (same runs as the forward GEMM's B operand)
  int n_runs;
  int k_chunks;             // sum n_chunks
  float* slabs;             // [n_slabs][wgrad_tiles][kWgradTileFloats] blocked partial tiles
  float* bias_out;          // optional [n_slabs][m_chunks*64]: sum over the slab's rows of G (bias gradients)
  float* extra_out;         // with G_extra: [n_slabs][16][k_chunks*64]
  float* extra_bias_out;    // with G_extra, optional: [n_slabs][16]
};
constexpr int kWgradTileFloats = 8 * 32 * 64 * 4;   // 8 waves x 32 accumulator quads x 64 lanes x 4 = 256 KB
struct WgradTile {
  int shape;                // 0: 4 G chunks x 4 X chunks (waves 2 x 4); 1: 8 x 2 (waves 4 x 2); a wave owns 2 x 1 chunks
  int mc0, kc0;             // first G / X chunk
  int bias_chunks;          // bit c: this tile's workgroups also sum G chunk mc0 + c over the rows (every chunk is summed by
                            // exactly one tile of its row of tiles: spread over the row, so that no tile is much slower)
  int extra_duty;           // ... and multiply G_extra with their X chunks (done once per X chunk)
};
__host__ __device__ inline int wgrad_tiles(int m_chunks, int k_chunks) {
  const int gyA = k_chunks / 4, rem = k_chunks - 4 * gyA;
  return ((m_chunks + 3) / 4) * gyA + ((m_chunks + 7) / 8) * ((rem + 1) / 2);
}
__host__ __device__ inline WgradTile wgrad_tile(int m_chunks, int k_chunks, int t) {
  const int gxA = (m_chunks + 3) / 4, gyA = k_chunks / 4, nA = gxA * gyA, gxB = (m_chunks + 7) / 8;
  WgradTile q;
  if (t < nA) {
    const int by = t / gxA, bx = t - by * gxA, spread = gyA < 4 ? gyA : 4;
    q.shape = 0; q.mc0 = 4 * bx; q.kc0 = 4 * by; q.extra_duty = bx == 0;
    q.bias_chunks = 0;
    for (int c = 0; c < 4; ++c)
      if (by < spread && c % spread == by) q.bias_chunks |= 1 << c;
  } else {
    const int u = t - nA, by = u / gxB, bx = u - by * gxB;
    q.shape = 1; q.mc0 = 8 * bx; q.kc0 = 4 * gyA + 2 * by; q.extra_duty = bx == 0;
    q.bias_chunks = (gyA == 0 && by == 0) ? 0xff : 0;
  }
  return q;
}

// Row kernels of the flow backward (coupling, 1x1, start): model.py:200-218 differentiated.
struct FlowBwdArgs {
  RowGeom g;
  int C, c, h;              // WN channels, flow channels c_k, h_k = c_k/2
  float scale;              // loss scale applied to the incoming gradients
  // saved forward state
  const float* Zpost;       // [B*L][8]  W.z of this flow (a0 | a1)
  const float* OUT;         // [B*L][8]  (b | log_s) of this flow
  // incoming gradients
  const float* g_z;         // [B][8][L] gradient of the returned z (model.py:220-221), or null
  const float* g_log_s;     // [B][h][L] gradient of the returned log_s of this flow, or null
  int from_z;               // pre: 1 -> the flow output's gradient comes from g_z channels [z_ch0, z_ch0+c)
  int z_ch0;
  float* GZ;                // [B*L][8] fp32: pre: in = d(flow output), out = (d a0 direct | d a1)
                            //                post: in = that, out = d(previous flow's output)
  _Float16* GO;             // pre: fp16 plane (1 chunk): ch [0,h) = d b, [h,2h) = d log_s, rest 0
  // post
  const _Float16* GX;       // d x_0 planes [C/64][R][64]
  const float* wstart;      // [C][h] pos rows
  const float* w1x1;        // [c][c] row-major W of this flow (model.py:64)
  const float* Zprev;       // Zpost of flow k-1 (or null when k == 0)
  const float* OUTprev;     // OUT of flow k-1
  int h_prev;               // h of flow k-1
  int n_peel;               // channels peeled before this flow (model.py:201-203)
  int z_peel_ch0;           // their channel offset in z
  const float* audio;       // k == 0: the input audio [B][8L] fp32
  float* dw_partial;        // [n_workgroups][64]: per-workgroup partial of d W[r][cc]
};

struct StartWgradArgs {
  RowGeom g;
  int C, h;
  const _Float16* GX;       // d x_0 planes
  const float* Zpost;       // a0 = Zpost[:, :h]
  float* partial;           // [n_workgroups][5][C]: j < 4: d Wstart[:, j], j = 4: d bstart
};

hipError_t launch_plane_gemm(const PGemmArgs& a, hipStream_t s);

// ---- training plumbing on the device (train_prep.hip): the module's own parameter tensors in, per-parameter gradients out.
// One device table of pointers, section-major; sections 0..16 are the PARAMETERS in the library's canonical order
// (wg_train_param_name), the rest are per-flow buffers of wg_train_weights / wg_train_grads.
enum PrepSec : int {
  SEC_IN_V = 0, SEC_IN_G, SEC_IN_B,      // in_layers[i]: weight (v or dense) [2C][C][3], g [2C] or null, bias [2C]     x FL
  SEC_RS_V, SEC_RS_G, SEC_RS_B,          // res_skip_layers[i]: [2C or C][C][1], g, bias                                x FL
  SEC_CO_V, SEC_CO_G, SEC_CO_B,          // cond_layer: [2C*nl][M8][1], g, bias                                         x n_flows
  SEC_ST_V, SEC_ST_G, SEC_ST_B,          // start: [C][h_k][1], g, bias                                                 x n_flows
  SEC_EN_W, SEC_EN_B, SEC_CV_W,          // end.weight [2h_k][C][1], end.bias [2h_k], convinv.conv.weight [c_k][c_k][1]  x n_flows
  SEC_UP_W, SEC_UP_B,                    // upsample.weight [M][M][1024], upsample.bias [M]                              x 1
  SEC_O_WSTART, SEC_O_BSTART, SEC_O_OINIT, SEC_O_W1X1,      // outputs of prepare (wg_train_weights per-flow pointers)   x n_flows
  SEC_G_DSTART, SEC_G_DOINIT, SEC_G_DW1X1,                  // packed gradients (wg_train_grads per-flow pointers)       x n_flows
  SEC_COUNT
};
constexpr int kPrepMaxFlows = 32;
struct PrepArgs {
  void* const* tab;          // device pointer table (prep_slot)
  const long long* goff;     // per parameter slot: offset (floats) of its gradient in `flat`
  float* flat;               // per-parameter gradients, canonical order
  float *s_in, *s_co, *s_rs, *s_st;     // weight-norm row scales g / ||v||, followed by 1 / ||v|| (n_scale entries each)
  float *wend8, *bsum, *wes;            // W_end zero-padded to 8 rows [nf][8][C], sum_i b_skip_i [nf][C], W_end.W_skip_i [FL][8][C]
  float *b1, *b2, *bup;                 // wg_train_weights: b1 (pre-scaled), b2, bup
  const float *dw1, *db1, *dw2, *db2, *dwes, *dwup, *dbup;    // wg_train_grads
  long long layer_stride, flow_stride;
  int C, M8, nl, nf, FL;
  int hk[kPrepMaxFlows], ck[kPrepMaxFlows];
  int n_scale[4];            // rows of the four weight-normed module classes (in_layers, cond_layer, res_skip_layers, start)
};
__host__ __device__ inline int prep_sec_len(int FL, int nf, int sec) {
  return sec <= SEC_RS_B ? FL : (sec == SEC_UP_W || sec == SEC_UP_B) ? 1 : nf;
}
__host__ __device__ inline int prep_slot_n(int FL, int nf, int sec, int idx) {
  int base = 0;
  for (int q = 0; q < sec; ++q) base += prep_sec_len(FL, nf, q);
  return base + idx;
}
__host__ __device__ inline int prep_slot(const PrepArgs& a, int sec, int idx) { return prep_slot_n(a.FL, a.nf, sec, idx); }
hipError_t launch_prepare(const PrepArgs& a, hipStream_t s);
hipError_t launch_param_grads(const PrepArgs& a, hipStream_t s);

// weight packing (train.hip: pack_kernel), one launch per output tensor
enum PackKind : int { PACK_A1 = 0, PACK_A2, PACK_ES, PACK_WAT, PACK_WBT, PACK_WCT, PACK_WUP };
struct PackArgs {
  int kind;
  int C, M8, FL, NW;
  const float *w1, *w2, *wes, *wup;   // natural-order sources (include/waveglow_amd.h: wg_train_plain)
  int native;                         // 1: w1 / w2 / wup are read from the module's own tensors instead (prep.tab, weight-norm
  PrepArgs prep;                      //    scales applied on the fly): wg_train_prepare; wes is prep.wes either way
  _Float16 *dst, *dst2;               // PACK_A1: a1 (tap K-steps) and a1c (conditioning K-steps)
  size_t n_pieces;                    // 16-byte output pieces
};
hipError_t launch_pack(const PackArgs& a, hipStream_t s);
// one launch for up to two jobs (a layer's d W1 and its d W2 / end x skip), each with its own slab count: sum_j tiles_j x
// n_slabs[j] workgroups fill the chip in ONE round when that sum is the CU count, and a job whose steps cost more (the extra
// plane) gets more, shorter slabs
hipError_t launch_wgrad(const WgradJob* jobs, int n_jobs, const RowGeom& g, const int* n_slabs, hipStream_t s);
// out[i] = scale * sum_{s < n_slabs} slabs[s * stride + i],  i < n  (fixed order: bitwise reproducible), for up to
// kMaxSlabSegs independent (slabs, out) pairs in ONE launch: a layer's weight-gradient
// launches leave six small-to-large slab sets behind, and six launches of a few microseconds each cost more in launch
// gaps than in traffic
constexpr int kMaxSlabSegs = 8;
struct SlabSeg {
  const float* slabs;
  float* out;
  size_t stride, n;      // n % 4 == 0, stride % 4 == 0
  int n_slabs;
  float scale;
  // The slabs hold [rows][row_len] in the kernels' channel-POSITION order; the sums are written in NATURAL channel order
  // (wg_common.h: pos_to_chan, a permutation inside 32-blocks that keeps aligned runs of four together, so a float4 of
  // positions is a float4 of channels): perm bit 0 = the rows are channels, bit 1 = the columns are.  row_len = 0: flat.
  int row_len, perm;
  // blocked = 1: the slabs are wgrad_kernel's blocked tiles (WgradJob::slabs): n = tiles * kWgradTileFloats, (m_chunks,
  // k_chunks) decode a float4's (row, 4 columns), row_len = k_chunks * 64; the result is n_groups matrices (at
  // out + group * out_group_stride), each the sum of n_slabs / n_groups consecutive slabs (groups > 1: d upsample, one per phase)
  int blocked, m_chunks, k_chunks, n_groups;
  size_t out_group_stride;
};
hipError_t launch_slab_reduce_multi(const SlabSeg* segs, int n_segs, hipStream_t s);
hipError_t launch_mel_plane(const void* mel, int io_f16, int M, const RowGeom& g, _Float16* melp, hipStream_t s);
hipError_t launch_flow_bwd_pre(const FlowBwdArgs& a, hipStream_t s);
hipError_t launch_flow_bwd_post(const FlowBwdArgs& a, hipStream_t s);
int flow_bwd_workgroups(const RowGeom& g);
hipError_t launch_start_wgrad(const StartWgradArgs& a, hipStream_t s);
int start_wgrad_workgroups(const RowGeom& g);

}  // namespace wg
