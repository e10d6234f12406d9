/*
 * waveglow_amd.h -- C ABI of the MI355X-native WaveGlow hot path (libwaveglow_amd.so).
 *
 * The reference (stefantaubert/waveglow) is pure Python and has no FFI of its own; the
 * interface this library replaces is the set of Python methods on its model object
 * (paths relative to /root/reference/):
 *
 *   wg_create / wg_set_tensor / wg_finalize  <-  WaveGlow.__init__ (src/waveglow/model.py:141-176),
 *       load_model -> load_state_dict (src/waveglow/train.py:48-55) and
 *       WaveGlow.remove_weightnorm (model.py:276-297): tensors are handed over by their
 *       state_dict names in weight-norm-removed form.
 *   wg_infer    <-  WaveGlow.infer(spect, sigma)          (model.py:223-274)
 *   wg_forward  <-  WaveGlow.forward((spect, audio))       (model.py:178-221)
 *
 * Conventions: plain pointers and sizes only.  `mel`, noise, outputs and `workspace` are DEVICE
 * pointers owned by the caller; weights given to wg_set_tensor are HOST fp32 pointers and are
 * copied.  Every call returns 0 on success or a negative wg_status; wg_last_error() gives the
 * message of the last failure on the calling thread.  wg_infer / wg_forward only enqueue work on
 * `stream` (a hipStream_t passed as void*); they never synchronise, allocate or free.
 * One handle per device; a handle may be used from one stream at a time.
 */
#ifndef WAVEGLOW_AMD_H
#define WAVEGLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wg_handle wg_handle;

typedef enum wg_status {
  WG_OK = 0,
  WG_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  WG_ERR_STATE = -2,        /* call order (e.g. infer before finalize, missing tensor) */
  WG_ERR_HIP = -3,          /* HIP runtime error (message has the hipError string) */
  WG_ERR_WORKSPACE = -4     /* workspace too small */
} wg_status;

typedef enum wg_dtype { WG_F32 = 0, WG_F16 = 1 } wg_dtype;

/* Model-shaping fields of ModelHParams (src/waveglow/hparams.py:19-31) plus the fixed
 * upsample geometry of model.py:145-150. */
typedef struct wg_config {
  int32_t n_mel_channels;   /* 80 */
  int32_t n_flows;          /* 12 */
  int32_t n_group;          /* 8 (only 8 is supported) */
  int32_t n_early_every;    /* 4 */
  int32_t n_early_size;     /* 2 */
  int32_t n_layers;         /* 8 (dilation 2^i, i < n_layers; <= 8) */
  int32_t n_channels;       /* 64, 128, 256 or 512 */
  int32_t kernel_size;      /* 3 (only 3 is supported) */
  int32_t upsample_kernel;  /* 1024 */
  int32_t upsample_stride;  /* 256 */
} wg_config;

const char* wg_version(void);
const char* wg_last_error(void);

/* WaveGlow.__init__ (model.py:141-176): validates the configuration, selects device `device_id`. */
int wg_create(const wg_config* cfg, int device_id, wg_handle** out);
int wg_destroy(wg_handle* h);

/* One tensor of the weight-norm-removed state_dict (model.py:276-297), fp32, host memory, C order.
 * Names and shapes are the reference's: "upsample.weight" [M,M,K], "upsample.bias" [M],
 * "convinv.k.conv.weight" [c_k,c_k,1], "WN.k.start.{weight [C,h_k,1],bias [C]}",
 * "WN.k.cond_layer.{weight [2C*n_layers, M*8, 1], bias}", "WN.k.in_layers.i.{weight [2C,C,3], bias}",
 * "WN.k.res_skip_layers.i.{weight [2C or C, C, 1], bias}", "WN.k.end.{weight [2h_k,C,1], bias}". */
int wg_set_tensor(wg_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);

/* Number of tensors wg_finalize expects, and the i-th expected name (for host-side checks). */
int wg_num_expected_tensors(const wg_handle* h);
const char* wg_expected_tensor_name(const wg_handle* h, int32_t i);

/* Packs all tensors into the kernels' MFMA fragment layouts (fp16 operands, fp32 biases), inverts the
 * 1x1 matrices in fp64 (model.py:51-60: W.float().inverse()), and uploads them.  May be called again
 * after further wg_set_tensor calls (derived state is rebuilt; no stale W^-1, unlike model.py:52-58). */
int wg_finalize(wg_handle* h);

/* Bytes of device workspace wg_infer / wg_forward need for batch B and n_frames mel frames
 * (forward: audio_len samples per utterance, a multiple of n_group). 0 on invalid arguments. */
size_t wg_infer_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames);
size_t wg_forward_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames, int32_t audio_len);

/* WaveGlow.infer (model.py:223-274) with the noise injected (the caller draws it, keeping the
 * reference's RNG order: z_init [B, n_rem, L] first, then one [B, n_early_size, L] per early-output
 * flow in DESCENDING flow index; L = n_frames*upsample_stride/n_group).
 *   mel      [B, n_mel, n_frames]           io_dtype
 *   z_init   [B, n_rem, L]                  io_dtype
 *   z_early  n_z_early device pointers, each [B, n_early_size, L], io_dtype
 *   audio    [B, n_frames*upsample_stride]  io_dtype (output)
 * Arithmetic: fp16 MFMA operands, fp32 accumulation, fp32 flow state. */
int wg_infer(wg_handle* h, const void* mel, const void* z_init, const void* const* z_early,
             int32_t n_z_early, float sigma, void* audio, int32_t B, int32_t n_frames,
             int32_t io_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* wg_infer for a batch of utterances of DIFFERENT lengths (serving; the reference's commented-out --batch-size,
 * src/waveglow_cli/inference_v2.py:64): `frames` is a device array of B mel-frame counts, each <= n_frames; mel / noise /
 * audio are padded to n_frames as in wg_infer (contents behind an utterance's own length are ignored, its audio tail is
 * zero).  Every utterance gets exactly the result of a batch-of-one call on its own frames: the padding columns are never
 * written, so they are the zero padding the convolutions see at the end of the sequence (model.py:98-102).
 * frames == NULL is wg_infer. */
int wg_infer_ragged(wg_handle* h, const void* mel, const int32_t* frames, const void* z_init,
                    const void* const* z_early, int32_t n_z_early, float sigma, void* audio, int32_t B,
                    int32_t n_frames, int32_t io_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* WaveGlow.forward (model.py:178-221), inference of the normalising direction (no autograd).
 *   mel      [B, n_mel, n_frames]     io_dtype
 *   audio    [B, audio_len]           io_dtype; audio_len % n_group == 0 and
 *                                     audio_len <= (n_frames-1)*stride + kernel  (model.py:187)
 *   z        [B, n_group, L]          fp32 out, L = audio_len / n_group
 *   log_s    n_flows device pointers, log_s[k] is [B, h_k, L] fp32 out
 *   log_det_W  HOST pointer to n_flows floats: B*L*logdet(W_k) (model.py:63), written before return */
int wg_forward(wg_handle* h, const void* mel, const void* audio, float* z, float* const* log_s,
               float* log_det_W, int32_t B, int32_t n_frames, int32_t audio_len, int32_t io_dtype,
               void* workspace, size_t workspace_bytes, void* stream);

/* WaveGlowLoss.forward (src/waveglow/train.py:31-45) on the outputs of wg_forward:
 *   loss = (sum z^2 / (2 sigma^2) - sum_k sum log_s[k] - sum_k log_det_W[k]) / z_elems      (z_elems = B*n_group*L)
 * z, log_s[k]: device fp32 with z_elems / log_s_elems[k] elements; log_det_W: n_flows HOST floats;
 * loss_out: DEVICE float; workspace: >= 16 bytes of device memory.  Enqueue-only; sums accumulate in fp64. */
int wg_loss(const float* z, int64_t z_elems, const float* const* log_s, const int64_t* log_s_elems, int32_t n_flows,
            const float* log_det_W, float sigma, float* loss_out, void* workspace, size_t workspace_bytes,
            void* stream);
/* The same with log_det_W as n_flows DEVICE floats: nothing of the loss passes through the host, so a training step
 * needs no stream synchronisation between forward and backward (the reference reads the 12 scalars as tensors too,
 * train.py:37-41). */
int wg_loss_dev(const float* z, int64_t z_elems, const float* const* log_s, const int64_t* log_s_elems, int32_t n_flows,
                const float* log_det_W_dev, float sigma, float* loss_out, void* workspace, size_t workspace_bytes,
                void* stream);

/* Algorithmic MACs per group-timestep (8 samples) of one infer pass, as SURVEY.md section 8(d) counts
 * them (for roofline reporting). */
double wg_macs_per_group_step(const wg_handle* h);

/* Per-kernel device timing: when enabled, wg_infer brackets its kernels with hipEvents on `stream`;
 * wg_profile_read synchronises those events and returns accumulated milliseconds per kernel class.
 * classes: 0 = mel_pack, 1 = flow/start, 2 = wn_layer, 3 = memset; with n_classes = 8 also the training direction
 * (wg_train_forward / wg_train_backward): 4 = fused layer forward, 5 = dgrad GEMMs, 6 = wgrad.  `on` = 1 times every class;
 * a value > 1 is a bit mask of the classes to time (bit c = class c). */
int wg_profile_enable(wg_handle* h, int32_t on);
int wg_profile_read(wg_handle* h, double* ms_per_class, int64_t* launches_per_class, int32_t n_classes);

/* ---- Denoiser (src/waveglow/denoiser.py:14-57 on the conv-STFT of src/waveglow/stft.py:98-198), fp32 -----------------
 * wg_stft_create: fwd_basis / inv_basis are the reference's windowed bases [2*513][1024] (STFT.__init__,
 *   stft.py:108-132: real rows then imaginary rows), win_sq the squared zero-centred window [1024]; HOST pointers.
 * wg_stft_denoise: audio [B][n_samples] fp32 device (n_samples % 256 == 0) ->
 *   audio_out [B][n_samples] = istft(max(|X| - strength*bias_mag, 0) * exp(i*arg X))   (Denoiser.forward), and/or
 *   mag0_out [B][513] = |X| of frame 0 (what Denoiser.__init__ keeps as bias_spec).  bias_mag null => no subtraction;
 *   audio_out null => transform only.  Enqueue-only. */
typedef struct wg_stft wg_stft;
int wg_stft_create(const float* fwd_basis, const float* inv_basis, const float* win_sq, int32_t filter_length,
                   int32_t hop_length, int32_t device_id, wg_stft** out);
int wg_stft_destroy(wg_stft* h);
size_t wg_stft_workspace_bytes(const wg_stft* h, int32_t B, int32_t n_samples);
int wg_stft_denoise(wg_stft* h, const float* audio, const float* bias_mag, float strength, float* audio_out,
                    float* mag0_out, int32_t B, int32_t n_samples, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Mel front-end, TacotronSTFT.mel_spectrogram (src/waveglow/taco_stft.py:84-104): STFT magnitudes (reflect padding,
 * stft.py:141-152) x mel filterbank -> log(clamp(., 1e-5)).  audio [B][n_samples] fp32 device (any n_samples > 512),
 * mel_basis [n_mel][513] fp32 DEVICE, mel_out [B][n_mel][n_samples/256 + 1] fp32 device.  Enqueue-only. */
size_t wg_stft_mel_workspace_bytes(const wg_stft* h, int32_t B, int32_t n_samples);
int wg_stft_mel(wg_stft* h, const float* mel_basis, int32_t n_mel, const float* audio, float* mel_out, int32_t B,
                int32_t n_samples, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Training direction: WaveGlow.forward under autograd and loss.backward() ---------------------------------------
 * (src/waveglow/model.py:178-221, train.py:190-199).  Weights change every optimiser step, so they are NOT taken from
 * the handle: the caller passes device buffers.  Every fp16 matrix below is given as [rows][K] in "(pos,pos)" order
 * -- rows and K columns permuted inside 32-blocks so that position 16h+4g+i holds channel 8g+4h+i -- and then laid out
 * in MFMA-fragment order [K/64][rows/32][4][64 lanes][8]: lane (r = lane&31, h = lane>>5), element j of sub-step s,
 * block b, K-step t = Mat[32b + pos(r)][64t + 32h + 8s + j]  (waveglow_amd/train.py: differentiable torch ops build
 * the (pos,pos) matrices, so weight norm and the folds below are differentiated by the caller's autograd; one gather
 * per matrix type produces the fragment order).  Gradients come back as plain [rows][K] (pos,pos) fp32.
 * C = n_channels, M8 = 8*n_mel_channels, fl = flow*n_layers + layer, K1 = 3C + M8, h_k / c_k per flow.
 * Needs only wg_create (no wg_set_tensor / wg_finalize). */
typedef struct wg_train_weights {
  /* Forward (one fused launch per WN layer, the inference kernel's structure).  MFMA A-fragment order of that kernel:
   * NW = wg_wn_waves(C) waves own MB = C/(32 NW) channel blocks each; M tile mt < MB = tanh rows of block w*MB+mt,
   * mt >= MB = sigmoid rows (+C) of block w*MB+mt-MB; lane (r = lane&31, hh = lane>>5) element j of k16 step
   * 2*(u&1)+k2 of half K-step u holds  W[row 32*blk + r][64*(u>>1) + 32*(u&1) + 16*k2 + 8*hh + j]  -- rows in NATURAL
   * channel order, K in position order (tap-major: tap0 | tap1 | tap2, then the cond_layer slice).  The tanh rows and
   * bias entries are pre-scaled by 2*log2(e), the sigmoid rows by -log2(e) (the gate works on exp2). */
  const void* a1;      /* fp16 [FL][2*3C/64][NW][2MB][2][64][8]   in_layers (model.py:98-102) */
  const void* a1c;     /* fp16 [FL][2*M8/64][NW][2MB][2][64][8]   cond_layer slice of the layer (model.py:121-128) */
  const float* b1;     /* [FL][2C]  (in_layers.bias + cond_layer.bias slice), natural order, pre-scaled */
  const void* a2;      /* fp16 [FL][NW][MB][C/16][64][8]   res rows of res_skip_layers (model.py:131-134): lane (r, hh)
                          element j of k16 step k = W_res[32*blk + r][16k + 8hh + j]; last layer of a flow unused */
  const float* b2;     /* [FL][C] natural order */
  const void* es;      /* fp16 [FL][C/32][64][8]  end x skip fold W_end.W_skip_i [8][C]: lane (row = lane&15, l4 = lane>>4)
                          element j of step s = hi (row < 8) / lo (row >= 8) fp16 half of  Wes[row&7][32s + 8 l4 + j] */
  /* Backward dgrad GEMMs on the same kernel (plain row blocks: [FL][2*K/64][NW][MB][2][64][8], lane (r, hh) element j of
   * k16 step 2*(u&1)+k2 of half K-step u = Mat[32*blk + r][64*(u>>1) + 32*(u&1) + 16*k2 + 8*hh + j], rows natural, K in
   * the position order of the planes it multiplies): */
  const void* wat;     /* fp16, Mat [C][C+64] = [ W_res^T | (W_end.W_skip_i)^T hi halves, padded to 64 ]  (K: d x positions,
                          then the d out plane's 64 channels) */
  const void* wbt;     /* fp16, Mat [C][6C]   = W_in[:, :, tap]^T for tap 0, 1, 2  (K per tap: the 2C d-pre positions) */
  /* ... and on the plane GEMM: [rows][K] in "(pos,pos)" order, fragment order [K/64][rows/32][4][64][8]: lane (r, h)
   * element j of sub-step s of block b, K-step t = Mat[32b + pos(r)][64t + 32h + 8s + j] */
  const void* wct;     /* fp16 [M8][FL*2C]    cond_layer^T of every layer */
  const void* wup;     /* fp16 [32][M8][512]  upsample per phase p: row (o,g) , K = [tap j][128]: W_up[i][o][8p+g+256j] */
  const float* bup;    /* [M8]                upsample.bias[o] repeated over g */
  const float* const* wstart;    /* n_flows pointers: [C][h_k] fp32 */
  const float* const* bstart;    /* [C] */
  const float* const* out_init;  /* [8]   W_end.(sum_i b_skip_i) + b_end, zero padded */
  const float* const* w1x1;      /* [c_k][c_k] fp32 row-major (model.py:64) */
} wg_train_weights;

/* Gradients: fp32 device buffers in NATURAL channel order, w.r.t. the matrices of wg_train_plain (dw1, dw2, dwes, dwup)
 * and the natural-order bias vectors; dwes is w.r.t. the effective end x skip matrix [8][C]. */
typedef struct wg_train_grads {
  float* dw1;          /* [FL][2C][K1] */
  float* db1;          /* [FL][2C] */
  float* dw2;          /* [FL][C][C]   (last layer of each flow: untouched) */
  float* db2;          /* [FL][C] */
  float* dwes;         /* [FL][8][C] */
  float* dwup;         /* [32][M8][512] */
  float* dbup;         /* [M8] */
  float* const* dstart;      /* n_flows pointers: [5][C]: rows j < 4 = d Wstart[:, j] (zero for j >= h_k), row 4 = d bstart */
  float* const* dout_init;   /* [8] */
  float* const* dw1x1;       /* [8][8], top-left [c_k][c_k] used: the W.z term only; the logdet term (model.py:63) is
                                the caller's */
  /* Record layout of dw1 / db1 / dw2 / db2 / dwes (elements of float).  Both 0: every tensor is dense, entry fl at
   * fl * (its own size).  Otherwise entry fl = flow*n_layers + layer of each of the five lies at
   * base + flow*flow_stride + layer*layer_stride: a data-parallel caller interleaves the five tensors of a layer in one
   * record and the records of a flow (plus its dstart / dout_init / dw1x1) in one contiguous region, so that a flow's
   * gradients travel as ONE all-reduce message (waveglow_amd/train.py: GradBuffers). */
  int64_t layer_stride;
  int64_t flow_stride;
} wg_train_grads;

size_t wg_train_workspace_bytes(const wg_handle* h, int32_t B, int32_t n_frames, int32_t audio_len);

/* The stacked weight matrices in NATURAL channel order, fp32, as the caller's autograd packing produces them every step
 * (waveglow_amd/train.py: pack_weights; reference modules: WN.in_layers / cond_layer / res_skip_layers / end,
 * model.py:85-113, upsample model.py:145-150). */
typedef struct wg_train_plain {
  const float* w1;     /* [FL][2C][K1]   in_layers[i].weight as K = tap*C + c_in (tap-major), then the layer's cond_layer rows [M8] */
  const float* w2;     /* [FL][C][C]     res rows of res_skip_layers[i] (unused for the last layer of a flow) */
  const float* wes;    /* [FL][8][C]     W_end . W_skip_i, zero padded to 8 rows */
  const float* wup;    /* [32][M8][512]  upsample per phase: row (o,g), K = [tap j][128]: W_up[i][o][8p+g+256j] */
} wg_train_plain;

/* Fills the fp16 fragment tensors a1, a1c, a2, es, wat, wbt, wct, wup of *out (device buffers of the sizes documented on
 * wg_train_weights; its other members are not touched) from the natural-order matrices: the permutations, transposes,
 * gate pre-scales and fragment orders above in one pass per tensor.  Not differentiable -- the gradients come back in
 * natural order (wg_train_grads).  Enqueue-only. */
int wg_train_pack(wg_handle* h, const wg_train_plain* in, const wg_train_weights* out, void* stream);

/* ---- Training plumbing on the device (round 3): the module's OWN parameter tensors in, one gradient per parameter out.
 * Replaces wg_train_plain / wg_train_pack plus the caller-side autograd ops around them (weight norm, stacking, the
 * W_end x W_skip fold and their backward): reference modules WN.start / in_layers / cond_layer / res_skip_layers
 * (torch weight_norm: w = g v / ||v||, model.py:85-113), WN.end (model.py:90-92), Invertible1x1Conv.conv
 * (model.py:29-43), WaveGlow.upsample (model.py:145-150).
 * The parameters come in the library's canonical order: wg_train_param_count / _name (the state_dict key of the
 * reference's module tree: "...parametrizations.weight.original0|1" for weight-normed modules, "...weight" otherwise,
 * weight_normed selects which) / _numel.  All tensors fp32, contiguous, in their native layouts. */
int32_t wg_train_param_count(const wg_handle* h, int32_t weight_normed);
const char* wg_train_param_name(const wg_handle* h, int32_t weight_normed, int32_t i);   /* valid until the next call on this thread */
int64_t wg_train_param_numel(const wg_handle* h, int32_t weight_normed, int32_t i);
/* Scratch the two calls below share (row norms, W_end x W_skip, pointer tables): must stay untouched between a
 * wg_train_prepare and the wg_train_param_grads of the same step. */
size_t wg_train_prepare_bytes(const wg_handle* h);
/* Fills EVERY member of *out (fragment tensors and the small fp32 vectors; all buffers caller-allocated with the
 * sizes documented on wg_train_weights) from params[i] = device pointer of canonical parameter i.  Enqueue-only. */
int wg_train_prepare(wg_handle* h, const void* const* params, int32_t weight_normed, const wg_train_weights* out, void* aux,
                     size_t aux_bytes, void* stream);
/* From the packed gradients that wg_train_backward left in *grads to one gradient per parameter: flat[offset_i ..
 * offset_i + numel_i) with offset_i = sum of the numel of the canonical parameters before i.  The 1x1 weights get the
 * W.z term only (the logdet term, model.py:63, is the caller's).  Enqueue-only. */
int wg_train_param_grads(wg_handle* h, const void* const* params, int32_t weight_normed, const wg_train_grads* grads, void* aux,
                         size_t aux_bytes, float* flat, void* stream);

/* Waves per workgroup of the WN-layer kernel for n_channels (the NW of the fragment orders above); 0 = unsupported. */
int32_t wg_wn_waves(int32_t n_channels);

/* Forward with saved activations.  mel [B][n_mel][n_frames] fp32, audio [B][audio_len] fp32 (audio_len % 8 == 0),
 * z [B][8][L] fp32 out, log_s[k] [B][h_k][L] fp32 out.  `fresh` != 0: the workspace has not been used with this
 * geometry before (it is cleared: guard rows must read as zero).  The workspace must stay untouched until
 * wg_train_backward has run.  Enqueue-only.  At large batch the call (and wg_train_backward) also enqueues on two
 * streams the handle owns: they are forked from `stream` inside the call and joined back into it before it returns,
 * so the caller sees ordinary stream order.  Environment: WG_TRAIN_SERIAL=1 keeps every launch on `stream`
 * (timing single kernels); WG_TRAIN_HALVES=1|2 never / always runs the forward as two half-batch chains. */
int wg_train_forward(wg_handle* h, const wg_train_weights* w, const void* mel, const void* audio, float* z,
                     float* const* log_s, int32_t B, int32_t n_frames, int32_t audio_len, int32_t fresh,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the last wg_train_forward on this workspace.  g_z [B][8][L] (or null), g_log_s[k] [B][h_k][L]
 * (null entries = zero) are the gradients of the returned tensors; `scale` multiplies them on entry (fp16 gradient
 * planes) and is divided out of every result.  Enqueue-only. */
int wg_train_backward(wg_handle* h, const wg_train_weights* w, const wg_train_grads* grads, const float* g_z,
                      const float* const* g_log_s, float scale, const void* audio, int32_t B, int32_t n_frames,
                      int32_t audio_len, void* workspace, size_t workspace_bytes, void* stream);

/* The same backward pass cut at flow boundaries: processes flows flow_hi, flow_hi-1, ..., flow_lo (0 <= flow_lo <=
 * flow_hi < n_flows); calls must come in descending, contiguous order starting at n_flows-1, and the call with
 * flow_lo == 0 also produces the upsample gradients.  After a call returns, the gradient slices of its flows
 * (dw1[fl], db1[fl], dw2[fl], db2[fl], dwes[fl] for fl in [flow_lo*n_layers, (flow_hi+1)*n_layers), dstart / dout_init /
 * dw1x1 of those flows) are final on `stream` -- a data-parallel caller can start their all-reduce while the earlier
 * flows are still being computed (waveglow_amd/train.py). */
int wg_train_backward_flows(wg_handle* h, const wg_train_weights* w, const wg_train_grads* grads, const float* g_z,
                            const float* const* g_log_s, float scale, const void* audio, int32_t B, int32_t n_frames,
                            int32_t audio_len, void* workspace, size_t workspace_bytes, int32_t flow_hi,
                            int32_t flow_lo, void* stream);

/* Diagnostic builds only (-DWG_STAMPS): device buffer of n_tiles*8 uint64 that the WN-layer kernel fills with
 * s_memtime stamps at its phase boundaries (last launch wins).  A no-op pointer in the shipped library. */
int wg_debug_set_stamp_buffer(wg_handle* h, void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* WAVEGLOW_AMD_H */
