// Denoiser (src/waveglow/denoiser.py:51-57) on exact-fp32 MFMA: conv-STFT (src/waveglow/stft.py:134-163) as a
// GEMM with the Hann-windowed Fourier basis, spectral subtraction + recombination in the epilogue, inverse STFT
// (stft.py:165-198) as a 4-tap polyphase GEMM with the pseudo-inverse basis, window-sum-square normalisation and
// cropping in its epilogue.  fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32 is bit-for-bit an fp32 fma chain).
// Fixed geometry: filter 1024, hop 256 (TSTFTHParams defaults, taco_stft.py:36-43).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFL = 1024, kHop = 256, kCut = 513;
constexpr int kRows = 1056;            // 2*513 = 1026 interleaved (re_k, im_k) rows, padded to 33 tiles of 32
constexpr int kMT = kRows / 32;        // 33

struct StftArgs {
  const float* audio;      // [B][N]
  const float* fwdA;       // packed A fragments [33 mtile][512 kstep][64 lanes]
  const float* bias;       // [513] magnitude to subtract (device) or null
  float strength;
  float* rec;              // [B][1056][Fs]   Fs = 3 + Fpad (3 zero lead columns), recombined spectrum
  float* mag0;             // optional [B][513]: magnitude of frame 0
  int N, F, Fs;
  float* mag;              // optional [B][513][F]: all magnitudes (mel front-end); rec may then be null
};
struct MelArgs {
  const float* mag;        // [B][513][F]
  const float* basis;      // [n_mel][513]  Slaney mel filterbank (taco_stft.py:66-73)
  float* mel;              // [B][n_mel][F]  log(clamp(basis . mag, 1e-5))   (taco_stft.py:10-16, :99-104)
  int n_mel, F;
};
struct IstftArgs {
  const float* rec;        // [B][1056][Fs]
  const float* invA;       // packed A fragments [8 mtile][4 j][528 kstep][64 lanes]
  const float* win_sq;     // [1024]
  float* out;              // [B][N]
  int N, F, Fs;
};

// LDS index of padded-audio position pos inside a 32-frame segment: one extra word per 256 so that the 32 lanes
// of a B-fragment read (same k, frames 256 apart) hit 32 different banks
__device__ __forceinline__ int seg_idx(int pos) { return pos + (pos >> 8); }

__global__ void __launch_bounds__(512) stft_kernel(const StftArgs a) {
  __shared__ float seg[8960 + 40];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f0 = blockIdx.x * 32, b = blockIdx.y;
  const float* x = a.audio + (size_t)b * a.N;
  for (int i = tid; i < 8960; i += 512) {
    int s = f0 * kHop + i - kFL / 2;                    // audio index of padded position (reflect, stft.py:141-147)
    if (s < 0) s = -s;
    if (s >= a.N) s = 2 * (a.N - 1) - s;
    seg[seg_idx(i)] = (s >= 0 && s < a.N) ? x[s] : 0.0f;
  }
  __syncthreads();
  constexpr int TPWV = 5;                               // M tiles per wave (wave w: w, w+8, ...; 33 tiles)
  f32x16 acc[TPWV];
#pragma unroll
  for (int t = 0; t < TPWV; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int col = lane & 31, kk = lane >> 5;
  const float* ap = a.fwdA + lane;
  for (int ks = 0; ks < kFL / 2; ++ks) {
    const float bv = seg[seg_idx(col * kHop + 2 * ks + kk)];
#pragma unroll
    for (int t = 0; t < TPWV; ++t) {
      const int mt = wave + 8 * t;
      if (mt < kMT) {
        const float av = ap[((size_t)mt * (kFL / 2) + ks) * 64];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  const int f = f0 + col;
#pragma unroll
  for (int t = 0; t < TPWV; ++t) {
    const int mt = wave + 8 * t;
    if (mt >= kMT) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = mt * 32 + 8 * g + 4 * kk;        // rows row0..row0+3 = (re, im) of bins row0/2, row0/2+1
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int bin = row0 / 2 + e;
        float re = acc[t][4 * g + 2 * e], im = acc[t][4 * g + 2 * e + 1];
        if (bin < kCut && f < a.F) {
          const float mag = sqrtf(re * re + im * im);
          if (a.mag0 && f == 0) a.mag0[(size_t)b * kCut + bin] = mag;
          if (a.mag) a.mag[((size_t)b * kCut + bin) * a.F + f] = mag;
          if (!a.rec) continue;
          if (a.bias) {                                 // denoiser.py:54-55, recombined with the original phase
            const float md = fmaxf(mag - a.bias[bin] * a.strength, 0.0f);
            const float sc = mag > 0.0f ? md / mag : 0.0f;
            im = mag > 0.0f ? im * sc : 0.0f;
            re = mag > 0.0f ? re * sc : md;             // phase of (0,0) is 0: cos = 1
          }
          float* rp = a.rec + ((size_t)b * kRows + row0 + 2 * e) * a.Fs + 3 + f;
          rp[0] = re;
          rp[a.Fs] = im;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(512) istft_kernel(const IstftArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // M tile: output phase r in [32w, 32w+32)
  const int q0 = blockIdx.x * 32, b = blockIdx.y;
  const int col = lane & 31, kk = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const float* ap = a.invA + (size_t)wave * 4 * (kRows / 2) * 64 + lane;
  const float* rb = a.rec + (size_t)b * kRows * a.Fs + 3 + q0 + col;
  for (int j = 0; j < 4; ++j) {
    for (int ks = 0; ks < kRows / 2; ++ks) {
      const float av = ap[((size_t)j * (kRows / 2) + ks) * 64];
      const float bv = rb[(size_t)(2 * ks + kk) * a.Fs - j];   // rec[c][q - j]; columns -3..-1 are zero
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  const int q = q0 + col;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = wave * 32 + 8 * g + 4 * kk + e;
      const int n = q * kHop + r;                         // position in the un-cropped inverse transform
      const int o = n - kFL / 2;
      if (o < 0 || o >= a.N) continue;
      float ws = 0.0f;                                    // window_sumsquare at n (stft.py:45-95)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int fr = q - jj;
        if (fr >= 0 && fr < a.F) ws += a.win_sq[r + kHop * jj];
      }
      float v = acc[4 * g + e];
      if (ws > 1.17549435e-38f) v /= ws;                  // tiny(float32)
      a.out[(size_t)b * a.N + o] = v * (float)(kFL / kHop);
    }
  }
}

// mel[b][m][f] = log(max(sum_k basis[m][k] |X|[b][k][f], 1e-5)): 80 x 513 MACs per frame, HBM/VALU-trivial.
// grid (ceil(F/64), B), 256 threads = 64 frames x 4 groups of mel rows.
__global__ void __launch_bounds__(256) mel_kernel(const MelArgs a) {
  const int f = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6, b = blockIdx.y;
  const int per = (a.n_mel + 3) / 4;
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
  const float* mp = a.mag + (size_t)b * kCut * a.F + (f < a.F ? f : a.F - 1);
  for (int k = 0; k < kCut; ++k) {
    const float v = mp[(size_t)k * a.F];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int m = grp * per + i;
      if (i < per && m < a.n_mel) acc[i] = fmaf(a.basis[(size_t)m * kCut + k], v, acc[i]);
    }
  }
  if (f >= a.F) return;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int m = grp * per + i;
    if (i < per && m < a.n_mel) a.mel[((size_t)b * a.n_mel + m) * a.F + f] = logf(fmaxf(acc[i], 1e-5f));
  }
}

hipError_t launch_mel(const MelArgs& a, int B, hipStream_t s) {
  if (a.n_mel < 1 || a.n_mel > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mel_kernel, dim3((a.F + 63) / 64, B), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_stft(const StftArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(stft_kernel, dim3((a.F + 31) / 32, B), dim3(512), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_istft(const IstftArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(istft_kernel, dim3((a.F + 3 + 31) / 32, B), dim3(512), 0, s, a);
  return hipGetLastError();
}

}  // namespace wg
