// C ABI of the STFT denoiser (include/waveglow_amd.h: wg_stft_*).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/waveglow_amd.h"

namespace wg {
constexpr int kFL = 1024, kHop = 256, kCut = 513, kRows = 1056;
struct StftArgs {
  const float* audio; const float* fwdA; const float* bias; float strength; float* rec; float* mag0; int N, F, Fs;
  float* mag;
};
struct MelArgs {
  const float* mag; const float* basis; float* mel; int n_mel, F;
};
hipError_t launch_mel(const MelArgs& a, int B, hipStream_t s);
struct IstftArgs {
  const float* rec; const float* invA; const float* win_sq; float* out; int N, F, Fs;
};
hipError_t launch_stft(const StftArgs& a, int B, hipStream_t s);
hipError_t launch_istft(const IstftArgs& a, int B, hipStream_t s);
}  // namespace wg
using namespace wg;

int wg_set_error(int code, const char* msg);   // api.cpp

struct wg_stft {
  int device;
  float *d_fwdA = nullptr, *d_invA = nullptr, *d_win = nullptr;
};

#define HIP_TRY2(expr)                                                      \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) return wg_set_error(WG_ERR_HIP, hipGetErrorString(_e)); \
  } while (0)

extern "C" {

int wg_stft_create(const float* fwd_basis, const float* inv_basis, const float* win_sq, int32_t filter_length,
                   int32_t hop_length, int32_t device_id, wg_stft** out) {
  if (!fwd_basis || !inv_basis || !win_sq || !out) return wg_set_error(WG_ERR_INVALID, "null argument");
  if (filter_length != kFL || hop_length != kHop)
    return wg_set_error(WG_ERR_INVALID, "only filter_length 1024 / hop_length 256 are supported");
  // basis row c of the library = interleaved (re_k, im_k): c = 2k -> reference row k, c = 2k+1 -> row 513 + k
  auto ref_row = [](int c) { return (c & 1) ? kCut + (c >> 1) : (c >> 1); };
  std::vector<float> fa((size_t)(kRows / 32) * (kFL / 2) * 64, 0.f), ia((size_t)8 * 4 * (kRows / 2) * 64, 0.f);
  for (int mt = 0; mt < kRows / 32; ++mt)
    for (int ks = 0; ks < kFL / 2; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int c = mt * 32 + (lane & 31), k = 2 * ks + (lane >> 5);
        if (c < 2 * kCut) fa[((size_t)mt * (kFL / 2) + ks) * 64 + lane] = fwd_basis[(size_t)ref_row(c) * kFL + k];
      }
  for (int w = 0; w < 8; ++w)
    for (int j = 0; j < 4; ++j)
      for (int ks = 0; ks < kRows / 2; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int r = w * 32 + (lane & 31), c = 2 * ks + (lane >> 5);
          if (c < 2 * kCut)
            ia[(((size_t)w * 4 + j) * (kRows / 2) + ks) * 64 + lane] = inv_basis[(size_t)ref_row(c) * kFL + r + kHop * j];
        }
  wg_stft* h = new wg_stft();
  h->device = device_id;
  struct DeviceGuard {   // restore the caller's current device on every exit path
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  } dev_guard;
  HIP_TRY2(hipGetDevice(&dev_guard.prev));
  HIP_TRY2(hipSetDevice(device_id));
  HIP_TRY2(hipMalloc((void**)&h->d_fwdA, fa.size() * 4));
  HIP_TRY2(hipMalloc((void**)&h->d_invA, ia.size() * 4));
  HIP_TRY2(hipMalloc((void**)&h->d_win, kFL * 4));
  HIP_TRY2(hipMemcpy(h->d_fwdA, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY2(hipMemcpy(h->d_invA, ia.data(), ia.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY2(hipMemcpy(h->d_win, win_sq, kFL * 4, hipMemcpyHostToDevice));
  *out = h;
  return WG_OK;
}

int wg_stft_destroy(wg_stft* h) {
  if (!h) return WG_OK;
  if (h->d_fwdA) (void)hipFree(h->d_fwdA);
  if (h->d_invA) (void)hipFree(h->d_invA);
  if (h->d_win) (void)hipFree(h->d_win);
  delete h;
  return WG_OK;
}

static int frames_padded(int F) { return 3 + (F + 3 + 31) / 32 * 32 + 32; }

size_t wg_stft_workspace_bytes(const wg_stft* h, int32_t B, int32_t n_samples) {
  if (!h || B < 1 || n_samples < kFL || n_samples % kHop) return 0;
  const int F = n_samples / kHop + 1;
  return (size_t)B * kRows * frames_padded(F) * 4;
}

int wg_stft_denoise(wg_stft* h, const float* audio, const float* bias_mag, float strength, float* audio_out,
                    float* mag0_out, int32_t B, int32_t n_samples, void* workspace, size_t workspace_bytes,
                    void* stream) {
  if (!h || !audio || !workspace) return wg_set_error(WG_ERR_INVALID, "null argument");
  if (B < 1 || n_samples < kFL || n_samples % kHop)
    return wg_set_error(WG_ERR_INVALID, "n_samples must be a multiple of 256 and >= 1024");
  const size_t need = wg_stft_workspace_bytes(h, B, n_samples);
  if (workspace_bytes < need) return wg_set_error(WG_ERR_WORKSPACE, "stft workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int F = n_samples / kHop + 1, Fs = frames_padded(F);
  HIP_TRY2(hipMemsetAsync(workspace, 0, need, s));           // zero lead/tail columns and pad rows
  StftArgs a{audio, h->d_fwdA, bias_mag, strength, (float*)workspace, mag0_out, n_samples, F, Fs, nullptr};
  HIP_TRY2(launch_stft(a, B, s));
  if (audio_out) {
    IstftArgs b{(const float*)workspace, h->d_invA, h->d_win, audio_out, n_samples, F, Fs};
    HIP_TRY2(launch_istft(b, B, s));
  }
  return WG_OK;
}

size_t wg_stft_mel_workspace_bytes(const wg_stft* h, int32_t B, int32_t n_samples) {
  if (!h || B < 1 || n_samples < kFL / 2 + 1) return 0;          // reflect padding needs n_samples > filter/2
  return (size_t)B * kCut * (n_samples / kHop + 1) * 4;
}

int wg_stft_mel(wg_stft* h, const float* mel_basis, int32_t n_mel, const float* audio, float* mel_out, int32_t B,
                int32_t n_samples, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !mel_basis || !audio || !mel_out || !workspace) return wg_set_error(WG_ERR_INVALID, "null argument");
  const size_t need = wg_stft_mel_workspace_bytes(h, B, n_samples);
  if (!need || n_mel < 1 || n_mel > 128) return wg_set_error(WG_ERR_INVALID, "bad B / n_samples / n_mel");
  if (workspace_bytes < need) return wg_set_error(WG_ERR_WORKSPACE, "mel workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int F = n_samples / kHop + 1;                             // stft.py:141-152: reflect pad filter/2 both sides
  StftArgs a{audio, h->d_fwdA, nullptr, 0.0f, nullptr, nullptr, n_samples, F, 0, (float*)workspace};
  HIP_TRY2(launch_stft(a, B, s));
  MelArgs m{(const float*)workspace, mel_basis, mel_out, n_mel, F};
  HIP_TRY2(launch_mel(m, B, s));
  return WG_OK;
}

}  // extern "C"
