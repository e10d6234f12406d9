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

// dW[m][k'] = sum over the rows of one slab of G[row][m] * X[row(+shift)][k']  -> out[slab][m][k'] * out_scale
// slab = phase * row_split + part: a phase's Rp rows are cut into row_split parts (small GEMMs: more workgroups);
// or slab = phase / phases_per_slab (large GEMMs: fewer, longer workgroups -- half the slab bytes to write and reduce)
struct WgradArgs {
  const _Float16* G;        // planes [m_chunks][R][64]  (the last chunk comes from G_last when that is set)
  const _Float16* G_last;   // optional: one more 64-channel plane appended to G (e.g. the d out plane behind d x)
  int m_chunks;             // chunks in all, G_last included
  PRun run[kMaxRuns];       // the X operand (same runs as the forward GEMM's B operand)
  int n_runs;
  int k_chunks;             // sum n_chunks
  RowGeom g;
  int row_split;            // 1, 2 or 4
  int phases_per_slab;      // 1, or (row_split == 1 only) 2 / 4: one workgroup runs that many consecutive phases into one slab
  float* out;               // [32*row_split/phases_per_slab slabs][m_chunks*64][k_chunks*64] fp32
  float out_scale;
  float* bias_out;          // optional [slabs][m_chunks*64]: sum over the slab's rows of G (bias gradients), unscaled
  int natural_rows;         // 1: `out` is a final result (no reduction follows): its rows m go to natural channel order
};

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

// weight packing (train.hip: pack_kernel), one launch per output tensor
enum PackKind : int { PACK_A1 = 0, PACK_A2, PACK_ES, PACK_WAT, PACK_WBT, PACK_WCT, PACK_WUP };
struct PackArgs {
  int kind;
  int C, M8, FL, NW;
  const float *w1, *w2, *wes, *wup;   // natural-order sources (include/waveglow_amd.h: wg_train_plain)
  _Float16 *dst, *dst2;               // PACK_A1: a1 (tap K-steps) and a1c (conditioning K-steps)
  size_t n_pieces;                    // 16-byte output pieces
};
hipError_t launch_pack(const PackArgs& a, hipStream_t s);
// one launch for job `a` and, optionally, a second job `b` whose workgroups fill the slots a's last round leaves idle and a
// slab reduction `red` (of slabs an earlier launch wrote) whose memory-bound workgroups run beside them
struct SlabSeg;
hipError_t launch_wgrad(const WgradArgs& a, const WgradArgs* b, const SlabSeg* red, int n_red, hipStream_t s);
// out[i] = scale * sum_{s < n_slabs} slabs[s * stride + i],  i < n  (fixed order: bitwise reproducible), for up to
// kMaxSlabSegs independent (slabs, out) pairs in ONE launch: a layer's weight-gradient
// launches leave six small-to-large slab sets behind, and six launches of a few microseconds each cost more in launch
// gaps than in traffic
constexpr int kMaxSlabSegs = 6;
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
};
hipError_t launch_slab_reduce_multi(const SlabSeg* segs, int n_segs, hipStream_t s);
hipError_t launch_mel_plane(const void* mel, int io_f16, int M, const RowGeom& g, _Float16* melp, hipStream_t s);
hipError_t launch_flow_bwd_pre(const FlowBwdArgs& a, hipStream_t s);
hipError_t launch_flow_bwd_post(const FlowBwdArgs& a, hipStream_t s);
int flow_bwd_workgroups(const RowGeom& g);
hipError_t launch_start_wgrad(const StartWgradArgs& a, hipStream_t s);
int start_wgrad_workgroups(const RowGeom& g);

}  // namespace wg
