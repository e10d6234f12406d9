// Shared host/device declarations for libwaveglow_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wg {

// ---------------------------------------------------------------------------------------------
// Channel-position permutation.
//
// Activations x (and the gated `acts` tile in LDS) are stored "position-major": within every block
// of 32 channels, storage position p = 16*h + 4*g + i holds channel 8*g + 4*h + i (g<4, h<2, i<4).
// That is exactly the order in which one lane of a 32x32 MFMA accumulator holds its 16 rows
// (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so a lane's 16 results are 32 contiguous bytes of
// fp16 -- written with two 16-byte stores and read back as MFMA B-operand fragments with
// ds_read_b128 / global_load_lds, no transpose anywhere.  The GEMM contracts over channels, so the
// weights are packed with the same permutation on the host and the order cancels.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int pos_to_chan(int P) {
  const int blk = P >> 5, p = P & 31;
  const int h = p >> 4, g = (p >> 2) & 3, i = p & 3;
  return (blk << 5) + 8 * g + 4 * h + i;
}
__host__ __device__ inline int chan_to_pos(int c) {
  const int blk = c >> 5, o = c & 31;
  const int g = o >> 3, h = (o >> 2) & 1, i = o & 3;
  return (blk << 5) + 16 * h + 4 * g + i;
}

constexpr int kMaxGroup = 8;     // n_group (flow state channels)
constexpr int kBK = 64;          // GEMM K-step (fp16 elements) = one 128-byte row of a chunk plane

// Row geometry of the activation planes x: [C/64 chunks][rows][64 channels], rows in PHASE-MAJOR time order.
// A group-timestep t = 32*q + p (q = mel frame, p = phase inside the frame, 32 = upsample_stride / n_group) of
// utterance b lives in row  p*Rp + b*Fp + Gf + q:  every phase owns a block of Rp rows; inside it every
// utterance owns Fp = Gf + F + Gf rows (F = ceil(L/32) frames, Gf zero guard frames each side).  With this order
//   * a dilated tap t +- d of a run of consecutive frames of ONE phase is again a run of consecutive rows (of
//     phase (p +- d) & 31, shifted by (p +- d) >> 5 frames): every GEMM B tile stays BN contiguous 128-byte rows;
//   * all columns of a tile share the phase, so the cond_layer o upsample fold (one weight matrix per phase,
//     api.cpp) applies to the whole tile.
// Guard rows are never written and read as the convolution's zero padding (model.py:98-102).
struct RowGeom {
  int B;        // utterances
  int L;        // valid group-timesteps per utterance
  int F;        // frames per utterance = ceil(L / 32)
  int Gf;       // guard frames each side (>= max dilation / 32)
  int Fp;       // Gf + F + Gf
  int Rp;       // rows per phase block = B * Fp rounded up to 128
  int R;        // 32 * Rp + 2 * kRowPad  (plane rows incl. slack before row 0 and after the last row)
  int T;        // mel frames of the input (melT rows per utterance = 3 + T + 3)
  const int* frames;   // optional (device, [B]): mel frames of every utterance of a ragged batch, each <= T; columns
                       // t >= 32 * frames[b] of utterance b are padding: never written, so they read as the zero padding
                       // of the convolutions exactly as in a batch-of-one call.  null: every utterance has L columns.
};
#ifdef WG_ROWPAD                 // A/B builds only
constexpr int kRowPad = WG_ROWPAD;
#else
constexpr int kRowPad = 16;      // zero slack rows in front of / behind every plane chunk (taps of the first / last tile reach up to
                                 // Gf <= 16 rows past it: the guard rows of a tile read as far outside as its valid rows do)
#endif
constexpr int kPhases = 32;

// GEMM 1 of the inference layer on 16x16x32 MFMAs (wn_layer_kernel M16) for this tile shape: its A fragments are packed
// differently (api.cpp wg_finalize packs both orders when a model can run either tile width)
#ifdef WG_NO_M16                 // A/B builds only
constexpr bool wn_frag16(int, int) { return false; }
#else
constexpr bool wn_frag16(int C, int BN) { return C == 256 && BN == 128; }
#endif

// First layer of a WN in the inference kernels: the residual input x_0 = W_start a0 + b_start (model.py:117) is rebuilt from
// the a0 plane by one MFMA step of the epilogue (weights hi + lo fp16, WnLayerArgs::wStA), so flow_kernel does not write
// the x_0 planes at all (2C bytes per group-timestep and flow).
constexpr bool wn_res_a0(int) { return true; }

struct WnLayerArgs {
  const _Float16* x_in;     // [C/64][R][64] position-major
  const _Float16* x_tap;    // B operand of the three dilated taps of GEMM 1: x_in, or for the first layer of a WN the
                            // a0 plane [1][R][64] = (a0_0..a0_3, 1, 0...) with in_layers[0] o start folded into wA1
  int x_chunks_per_tap;     // C/64, or 1 with the a0 plane
  _Float16* x_out;          // same layout (ping-pong), unused when !has_res
  const _Float16* melT;     // [3 + B*(3+T+3)][M] fp16 frame-major mel (3 zero rows of slack, then per utterance
                            // 3 zero rows, T frames, 3 zero rows)
  const _Float16* wA1;      // packed GEMM1 A fragments of the 3 taps  [2*3C/64 half-steps][NW][MT][2][64][8]
  const _Float16* wA1c;     // ... of the folded cond_layer o upsample, per phase  [32][2*M/16 half-steps][NW][MT][2][64][8]
  const float* bias1;       // [2C]  b_in + b_cond slice
  const _Float16* wA2;      // packed GEMM2 A fragments  [NW][MB][C/16][64][8]
  const float* bias2;       // [C]   b_res
  const _Float16* wEs;      // packed folded end x skip  [C/32][64][8]  (rows 0-7 hi, 8-15 lo)
  float* out;               // [B*L][8] fp32, accumulated across layers
  RowGeom g;
  int dil;                  // dilation of this layer
  int n_cond_steps;         // K-steps of the folded conditioning = 4*M/64 = M/16
  int M;                    // n_mel_channels
  int has_res;              // 0 for the last layer of a WN (model.py:106-110)
  int row0;                 // first row (inside every phase block) of this launch's tiles: 0, or the start of the narrow-tile
                            // tail when a layer is split into a wide-tile and a narrow-tile launch (train_api.cpp)
  int tiles_per_phase;      // tiles of this launch per phase: (Rp - row0) / BN, or fewer
  int n_tiles;              // 32 * tiles_per_phase
  int n_cu;                 // compute units of the device (persistent grid size)
  int frag16 = 0;           // wA1 / wA1c are 16x16x32 fragments (must equal wn_frag16(C, BN) of the inference launch)
  int a0_fold = 0;          // first layer of a WN with the start fold: x_tap is the a0 plane, wA1 is in_layers[0] o start packed
                            // [tap][8] along ONE K-step (wn_layer_kernel A0G); x_chunks_per_tap == 1
  const _Float16* wStA = nullptr;   // first layer with the start fold (x_chunks_per_tap == 1), wn_res_a0(C): [C/32][64][8] A fragments
                            // (row r = channel 32 blk + r; k = 0..3 W_start, k = 4 b_start; lanes 0-31 hi, 32-63 lo parts)
  unsigned long long* stamps;   // diagnostic build only (-DWG_STAMPS): [n_tiles][8] s_memtime per phase
  // ---- training forward only (wn_layer_kernel<..., TR = true>, train_api.cpp); null / unused for inference
  const _Float16* sp;       // upsampled, squeezed spectrogram planes [M8/64 chunks][R][64] (position-major): the B operand
                            // of the conditioning K-steps (the cond_layer o upsample fold is NOT used: weights change every
                            // step); wA1c then holds cond_layer's slice itself, no phase dimension, n_cond_steps = M8/64
  _Float16* save_t;         // saved activations for the backward pass, planes [C/64][R][64] like x:
  _Float16* save_s;         //   tanh, sigmoid and their product (acts) of this layer
  _Float16* save_a;
  // ---- backward dgrad GEMMs only (MODE 2 / 3, launch_wn_plain): epilogue inputs / output planes
  const _Float16* in0;      // MODE 2: d x_{i+1} planes added to the result (or null); MODE 3: saved tanh planes
  const _Float16* in1;      // MODE 3: saved sigmoid planes
  _Float16* out0;           // MODE 2: d x_i planes [C/64]; MODE 3: d pre planes [2C/64] (tanh half, then sigmoid half)
  // ---- MODE 4 = MODE 2 of layer i fused with MODE 3 of layer i-1 (the d x_i tile never leaves the workgroup between them):
  const _Float16* wat_prev; // plain-row-block fragments [W_res^T | (W_end W_skip)^T] of layer i-1 (wg_train_weights::wat)
  const _Float16* gout;     // the flow's d out plane [1][R][64]
  const _Float16* t_prev;   // saved tanh / sigmoid planes of layer i-1
  const _Float16* s_prev;
  _Float16* dpre_prev;      // d pre planes [2C/64] of layer i-1
};

struct MelPackArgs {
  const int* frames;        // optional per-utterance frame counts (RowGeom::frames): frames beyond them are zero
  const void* mel;          // [B][M][T] io dtype
  _Float16* melT;           // see WnLayerArgs::melT
  int B, M, T, io_f16;
};

struct FlowArgs {
  // direction: 0 = infer (inverse flow, model.py:246-271), 1 = forward (model.py:200-218)
  int direction;
  // ---- inverse-flow step (skipped when `first`): coupling of flow k, W^-1 mix, optional early concat
  int first;                // infer: z = sigma * z_init ; forward: z = squeeze(audio)
  int c_in;                 // channels of Z on entry
  int h_in;                 // c_in / 2
  const float* winv;        // [c_in][c_in] row-major (infer: W^-1 of flow k; forward: W of flow k_next)
  const void* z_extra;      // infer: z_early[k] [B][n_early][L] io dtype or null; first: z_init [B][c][L]
  int n_extra;              // channels of z_extra (infer), 0 if none
  float sigma;
  // ---- forward-direction extras
  float* log_s_out;         // forward: [B][h_in][L] fp32 for the flow just finished (or null)
  float* z_out;             // forward: [B][8][L] fp32 final output
  int z_out_ch0;            // forward: channel offset in z_out where peeled / final channels go
  int n_peel;               // forward: channels peeled off before the next flow (model.py:201-203)
  const void* audio_in;     // forward+first: [B][audio_len] io dtype
  // ---- next WN start (skipped when `last`)
  int last;                 // infer: write audio ; forward: write remaining z
  int c_next;               // channels of Z on exit
  int h_next;               // c_next / 2
  const float* wstart;      // [C][h_next] position-major rows (row P <-> channel pos_to_chan(P))
  const float* bstart;      // [C] position-major
  const float* out_init;    // [8] folded end bias for the next WN
  // ---- buffers
  float* Z;                 // [B*L][8] fp32 flow state (channels-last)
  float* out;               // [B*L][8] fp32  b | log_s of the flow just computed, re-initialised for next
  float* Z_w;               // where the new state / re-initialised out are written: = Z / out for inference (in place);
  float* out_w;             // the training forward keeps every flow's state and gives each flow its own buffers
  _Float16* x;              // [C/64][R][64] start output
  _Float16* a0p;            // [1][R][64] a0 plane: channels 0..3 = a0 (fp16), channel 4 = 1, 8..12 the same again, rest 0 (or null)
  int skip_x;               // do not write the x planes (the next WN's first layer rebuilds x_0 from the a0 plane, wn_res_a0)
  void* audio_out;          // infer+last: [B][8L] io dtype
  RowGeom g;
  int C;
  int io_f16;
};

// launch wrappers (kernels.hip)
hipError_t launch_mel_pack(const MelPackArgs& a, hipStream_t s);
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t s);   // 16-byte aligned pointer and size
// derived weights: A fragments of (W_cond slice of layer l) x (upsample taps of phase p), see api.cpp
hipError_t launch_cond_fold(const float* w_cond, const float* w_up, _Float16* out, int C, int NW, int M, int n_layers,
                            int up_kernel, float tanh_scale, float sigm_scale, int frag16, hipStream_t s);
hipError_t launch_flow(const FlowArgs& a, hipStream_t s);
hipError_t launch_wn_layer(const WnLayerArgs& a, int C, int bn, hipStream_t s);   // bn = 128 (default) or 64
hipError_t launch_wn_layer_train(const WnLayerArgs& a, int C, int bn, hipStream_t s);   // training forward (a.sp, a.save_*)
hipError_t launch_wn_plain(const WnLayerArgs& a, int C, int kind, int bn, hipStream_t s);   // backward dgrad GEMMs (kind 2 / 3)
int wn_block_n(int C);   // default BN for channel count C
int wn_waves(int C);     // waves per workgroup for channel count C

hipError_t launch_reduce_sum(const float* x, size_t n, int square, double* acc, hipStream_t s);
hipError_t launch_loss_final(const double* acc, double log_det_total, const float* log_det_dev, int n_dev, float sigma, double denom,
                             float* out, hipStream_t s);

}  // namespace wg
