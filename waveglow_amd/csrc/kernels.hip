// gfx950 (CDNA4 / MI355X) kernels of the WaveGlow hot path.
//
//   upsample_kernel   mel -> squeezed conditioning planes           (reference: src/waveglow/model.py:145-150,
//                                                                    :225-232 / :186-193)
//   infer_flow_kernel affine-coupling inverse + W^-1 mix + early     (model.py:247-271) fused with the NEXT
//                     noise concat, then the next WN's start conv    flow's WN.start (model.py:117)
//   wn_layer_kernel   one WN layer: dilated conv + cond slice as     (model.py:123-135, :13-20, :137)
//                     ONE K-extended MFMA GEMM, gate in registers,
//                     res GEMM + folded end*skip GEMM from LDS
//
// Data layout (see wg_common.h): time-major fp16 planes [chunk][row][64 ch] so that every GEMM K-step's
// B tile is BN contiguous 128-byte rows, fetched HBM/L2 -> LDS by global_load_lds (LDS-DMA), XOR-swizzled
// through the SOURCE address (LDS destination is lane-linear).
#include "wg_common.h"

namespace wg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WG_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float load_io(const void* p, size_t i, int f16) {
  return f16 ? (float)((const _Float16*)p)[i] : ((const float*)p)[i];
}

// tanh(a) * sigmoid(b)  (model.py:17-19) with one reciprocal: (e^{2a}-1) / ((e^{2a}+1)(1+e^{-b}))
__device__ __forceinline__ float gate_act(float a, float b) {
  a = fminf(fmaxf(a, -15.0f), 15.0f);
  const float e2a = __expf(2.0f * a);
  const float emb = __expf(-b);
  return (e2a - 1.0f) * __builtin_amdgcn_rcpf((e2a + 1.0f) * (1.0f + emb));
}

// =============================================================================================
// WN layer
// =============================================================================================
template <int C> struct WnCfg {
  static constexpr int BN = (C >= 512) ? 64 : 128;
};

template <int C, int BN>
__global__ void __launch_bounds__((C / 32) * 64) wn_layer_kernel(const WnLayerArgs a) {
  constexpr int NW = C / 32;             // waves: wave w owns gate channels [32w, 32w+32)
  constexpr int NTHREADS = NW * 64;
  constexpr int NT = BN / 32;            // 32-column MFMA tiles per wave
  constexpr int CC = C / 64;             // 64-channel chunks of x
  constexpr int BT_BYTES = BN * 128;     // one staged B tile: BN rows x 64 fp16
  constexpr int ACT_ROW = 2 * C;         // bytes per acts row
  constexpr int K2 = C / 16;             // k16 steps of GEMM2

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem;                     // 2 x BT_BYTES
  char* const sActs = smem + 2 * BT_BYTES;   // BN x ACT_ROW

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31;
  const int lh = lane >> 5;

  // XCD-aware tile id: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of time tiles -- neighbouring tiles re-read each other's +-dil rows from that L2.
  int tile;
  {
    const int bid = blockIdx.x, nt = a.n_tiles;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int b = tile / a.tiles_per_utt;
  const int jt = tile - b * a.tiles_per_utt;
  const int R = a.g.R;
  const int r0 = b * a.g.Lp + a.g.G + jt * BN;   // first plane row of this tile
  const int t0 = jt * BN;                        // first group-timestep
  const int nK = 3 * CC + a.ns_chunks;

  auto kstep_src = [&](int ks) -> const char* {
    if (ks < 3 * CC) {
      const int tap = ks / CC, cc = ks - tap * CC;
      const int row = r0 + (tap - 1) * a.dil;     // taps t-d, t, t+d (model.py:98-102: padding = dilation)
      return (const char*)(a.x_in + ((size_t)cc * R + row) * 64);
    }
    const int cs = ks - 3 * CC;
    return (const char*)(a.spect + ((size_t)cs * R + r0) * 64);
  };
  // LDS-DMA one B tile: piece idx = row*8 + physical 16-B chunk; logical chunk = phys ^ ((row>>1)&7)
  auto stage_B = [&](int ks, char* dst) {
    const char* src = kstep_src(ks);
#pragma unroll
    for (int i0 = 0; i0 < BN * 8; i0 += NTHREADS) {
      const int base = i0 + wave * 64;
      if (base < BN * 8) {
        const int idx = base + lane;
        const int row = idx >> 3, pc = idx & 7;
        const char* g = src + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4);
        __builtin_amdgcn_global_load_lds(WG_GPTR(g), WG_LPTR(dst + base * 16), 16, 0, 0);
      }
    }
  };
  const int swB = (ln >> 1) & 7;
  auto read_B = [&](const char* buf, int nt, int k16) -> half8 {
    const int n = nt * 32 + ln;
    const int c = (k16 * 2 + lh) ^ swB;
    return *(const half8*)(buf + n * 128 + c * 16);
  };
  const half8* const wA1 = (const half8*)a.wA1;
  auto load_A = [&](int ks, half8 (&dst)[2][4]) {
    const half8* p = wA1 + ((size_t)(ks * NW + wave) * 8) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[mt][k] = p[(mt * 4 + k) * 64];
  };

  // ---- GEMM1 accumulators, initialised with the bias (in_layer bias + cond bias slice)
  f32x16 acc[2][NT];
  {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float* bp = a.bias1 + mt * C + wave * 32 + 4 * lh;
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v;
    }
  }

  half8 aA[2][4], aB[2][4];
  stage_B(0, sB);
  load_A(0, aA);

  auto compute = [&](const char* buf, half8 (&af)[2][4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      half8 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = read_B(buf, nt, k);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][k], bf[nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][k], bf[nt], acc[1][nt], 0, 0, 0);
      }
    }
  };

  // Two K-steps per trip so the A-fragment double buffer needs no register copies.
  int ks = 0;
  for (; ks + 1 < nK; ks += 2) {
    __syncthreads();                       // tile ks landed (vmcnt(0) + barrier); buffer 1 free
    stage_B(ks + 1, sB + BT_BYTES);
    load_A(ks + 1, aB);
    compute(sB, aA);
    __syncthreads();                       // tile ks+1 landed; buffer 0 free
    if (ks + 2 < nK) {
      stage_B(ks + 2, sB);
      load_A(ks + 2, aA);
    }
    compute(sB + BT_BYTES, aB);
  }
  if (ks < nK) {                           // odd nK tail
    __syncthreads();
    compute(sB, aA);
  }

  // ---- residual input (x at this tile, this wave's 32 channels) -> GEMM2 accumulator init
  // Lane (n, h) owns positions [32w+16h, +16) of column n: 32 contiguous bytes.
  constexpr int cc_w_shift = 1;            // two waves per 64-channel chunk
  f32x16 acc2[NT];
  const int my_cc = wave >> cc_w_shift;
  const int my_off = (wave & 1) * 32 + lh * 16;          // fp16 elements inside the 64-wide row
  if (a.has_res) {
    const float* bp = a.bias2 + wave * 32 + 4 * lh;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const size_t row = (size_t)my_cc * R + r0 + nt * 32 + ln;
      const half8* xp = (const half8*)(a.x_in + row * 64 + my_off);
      const half8 x0 = xp[0], x1 = xp[1];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc2[nt][r] = (float)x0[r] + bv[r];
        acc2[nt][8 + r] = (float)x1[r] + bv[8 + r];
      }
    }
  }

  // ---- gate (model.py:13-20) in registers; acts -> LDS as fp16, position-major, XOR-swizzled
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    half8 o0, o1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      o0[r] = (_Float16)gate_act(acc[0][nt][r], acc[1][nt][r]);
      o1[r] = (_Float16)gate_act(acc[0][nt][8 + r], acc[1][nt][8 + r]);
    }
    const int n = nt * 32 + ln;
    const int sw = (ACT_ROW >= 256) ? (n & 15) : ((n >> 1) & 7);
    const int c0 = wave * 4 + lh * 2;
    *(half8*)(sActs + n * ACT_ROW + ((c0 ^ sw) << 4)) = o0;
    *(half8*)(sActs + n * ACT_ROW + (((c0 + 1) ^ sw) << 4)) = o1;
  }
  __syncthreads();

  auto read_acts32 = [&](int nt, int k16) -> half8 {       // B fragment for the 32x32x16 MFMA
    const int n = nt * 32 + ln;
    const int sw = (ACT_ROW >= 256) ? (n & 15) : ((n >> 1) & 7);
    return *(const half8*)(sActs + n * ACT_ROW + (((k16 * 2 + lh) ^ sw) << 4));
  };

  // ---- GEMM2: res rows of this wave (model.py:130-132)
  if (a.has_res) {
    const half8* p2 = (const half8*)a.wA2 + (size_t)wave * K2 * 64 + lane;
    constexpr int PF = 4;
    half8 a2[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) a2[i] = p2[i * 64];
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      const half8 af = a2[k % PF];
      if (k + PF < K2) a2[k % PF] = p2[(k + PF) * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, read_acts32(nt, k), acc2[nt], 0, 0, 0);
    }
  }

  // ---- folded end x skip (model.py:133-137): out[0:8] += (W_end W_skip_i) acts, 16 columns per group,
  // weights split hi+lo fp16 (rows 0-7 / 8-15 of the 16x16x32 MFMA) so the 8 flow outputs keep ~fp32 weights.
  {
    const int l15 = lane & 15, l4 = lane >> 4;
    const half8* pe = (const half8*)a.wEs + lane;
    for (int grp = wave; grp < BN / 16; grp += NW) {
      const int n = grp * 16 + l15;
      const int sw = (ACT_ROW >= 256) ? (n & 15) : ((n >> 1) & 7);
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C / 32; ++s) {
        const half8 bf = *(const half8*)(sActs + n * ACT_ROW + (((s * 4 + l4) ^ sw) << 4));
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(pe[s * 64], bf, d, 0, 0, 0);
      }
      // D: col = lane&15, row = 4*(lane>>4)+reg ; rows 8-15 (lanes 32-63) are the lo parts
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] += __shfl_xor(d[r], 32);
      const int t = t0 + n;
      if (lane < 32 && t < a.g.L) {
        float4* op = (float4*)(a.out + ((size_t)b * a.g.L + t) * 8 + 4 * l4);
        float4 o = *op;
        o.x += d[0]; o.y += d[1]; o.z += d[2]; o.w += d[3];
        *op = o;
      }
    }
  }

  // ---- x_out = fp16(x + res) for valid columns (rows >= L stay zero: they are other tiles' padding)
  if (a.has_res) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (t0 + nt * 32 + ln < a.g.L) {
        half8 o0, o1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          o0[r] = (_Float16)acc2[nt][r];
          o1[r] = (_Float16)acc2[nt][8 + r];
        }
        const size_t row = (size_t)my_cc * R + r0 + nt * 32 + ln;
        half8* xp = (half8*)(a.x_out + row * 64 + my_off);
        xp[0] = o0;
        xp[1] = o1;
      }
    }
  }
}

template <int C>
static hipError_t launch_wn_t(const WnLayerArgs& a, hipStream_t s) {
  constexpr int BN = WnCfg<C>::BN;
  constexpr int smem = 2 * BN * 128 + BN * 2 * C;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)wn_layer_kernel<C, BN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((wn_layer_kernel<C, BN>), dim3(a.n_tiles), dim3((C / 32) * 64), smem, s, a);
  return hipGetLastError();
}

int wn_block_n(int C) {
  switch (C) {
    case 64: return WnCfg<64>::BN;
    case 128: return WnCfg<128>::BN;
    case 256: return WnCfg<256>::BN;
    case 512: return WnCfg<512>::BN;
  }
  return 0;
}

hipError_t launch_wn_layer(const WnLayerArgs& a, int C, hipStream_t s) {
  switch (C) {
    case 64: return launch_wn_t<64>(a, s);
    case 128: return launch_wn_t<128>(a, s);
    case 256: return launch_wn_t<256>(a, s);
    case 512: return launch_wn_t<512>(a, s);
  }
  return hipErrorInvalidValue;
}

// =============================================================================================
// Upsample: ConvTranspose1d(M, M, 1024, stride 256) written straight into the squeezed planes.
// out sample tau = 256 q + 8 t' + g gets taps j = 0..3 from frame q - j with kernel index 8t'+g+256j.
// One workgroup: one t' (phase inside a frame), NF consecutive frames q, all M*8 squeezed channels
// (one thread per channel; a wave = one 64-channel plane chunk => 128-byte coalesced stores).
// =============================================================================================
constexpr int UP_NF = 16;

__global__ void __launch_bounds__(640) upsample_kernel(const UpsampleArgs a) {
  const int M = a.M;
  const int ch = threadIdx.x;            // o*8 + g
  const int tp = blockIdx.x;             // t' in [0, 32)
  const int q0 = blockIdx.y * UP_NF;
  const int b = blockIdx.z;
  __shared__ float smel[80][UP_NF + 4];  // [i][frame q0-3 .. q0+NF-1]
  for (int idx = threadIdx.x; idx < M * (UP_NF + 3); idx += blockDim.x) {
    const int i = idx / (UP_NF + 3), f = idx - i * (UP_NF + 3);
    const int q = q0 - 3 + f;
    smel[i][f] = (q >= 0 && q < a.T) ? load_io(a.mel, ((size_t)b * M + i) * a.T + q, a.io_f16) : 0.0f;
  }
  __syncthreads();
  float acc[UP_NF];
  const float bias = a.bias[ch >> 3];
#pragma unroll
  for (int f = 0; f < UP_NF; ++f) acc[f] = bias;
  const int NCH = M * 8;
  const float* wp = a.w + (size_t)tp * 4 * M * NCH + ch;
  for (int j = 0; j < 4; ++j) {
    for (int i = 0; i < M; ++i) {
      const float w = wp[((size_t)j * M + i) * NCH];
#pragma unroll
      for (int f = 0; f < UP_NF; ++f) acc[f] = fmaf(w, smel[i][f + 3 - j], acc[f]);
    }
  }
  const int cs = ch >> 6, cw = ch & 63;
#pragma unroll
  for (int f = 0; f < UP_NF; ++f) {
    const int t = (q0 + f) * 32 + tp;
    if (t < a.g.L) {
      const size_t row = (size_t)b * a.g.Lp + a.g.G + t;
      a.spect[((size_t)cs * a.g.R + row) * 64 + cw] = (_Float16)acc[f];
    }
  }
}

hipError_t launch_upsample(const UpsampleArgs& a, hipStream_t s) {
  if (a.M * 8 > 640 || a.M > 80) return hipErrorInvalidValue;
  dim3 grid(32, (a.n_q + UP_NF - 1) / UP_NF, a.g.B);
  hipLaunchKernelGGL(upsample_kernel, grid, dim3(a.M * 8), 0, s, a);
  return hipGetLastError();
}

// =============================================================================================
// Flow step (both directions) + next WN start.  64 rows per workgroup, 256 threads.
// =============================================================================================
constexpr int FL_ROWS = 64;

__global__ void __launch_bounds__(256) flow_kernel(const FlowArgs a) {
  __shared__ float s_a0[FL_ROWS][4];
  const int L = a.g.L;
  const size_t nrows = (size_t)a.g.B * L;
  const size_t row0 = (size_t)blockIdx.x * FL_ROWS;
  const int tid = threadIdx.x;

  if (tid < FL_ROWS) {
    const size_t row = row0 + tid;
    if (row < nrows) {
      const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
      float zn[kMaxGroup];
#pragma unroll
      for (int c = 0; c < kMaxGroup; ++c) zn[c] = 0.0f;
      if (a.direction == 0) {
        // ------------------------------------------------ inverse flow (model.py:246-271)
        if (a.first) {
          for (int c = 0; c < a.c_next; ++c)
            zn[c] = a.sigma * load_io(a.z_extra, ((size_t)b * a.c_next + c) * L + t, a.io_f16);   // :243-244
        } else {
          float z[kMaxGroup], o[kMaxGroup], v[kMaxGroup];
          const float4* zp = (const float4*)(a.Z + row * 8);
          const float4* op = (const float4*)(a.out + row * 8);
          float4 z0 = zp[0], z1 = zp[1], o0 = op[0], o1 = op[1];
          z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
          o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
          const int h = a.h_in, c = a.c_in;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < h) {
              v[j] = z[j];
              v[h + j] = (z[h + j] - o[j]) / expf(o[h + j]);                                       // :253-255
            }
          }
          const int ne = a.n_extra;
          for (int r = 0; r < c; ++r) {                                                            // :258 / :59
            float s = 0.0f;
            for (int cc = 0; cc < c; ++cc) s = fmaf(a.winv[r * c + cc], v[cc], s);
            zn[ne + r] = s;
          }
          for (int e = 0; e < ne; ++e)
            zn[e] = a.sigma * load_io(a.z_extra, ((size_t)b * ne + e) * L + t, a.io_f16);          // :260-271
        }
        if (a.last) {                                                                              // :273
          if (a.io_f16) {
            half8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (_Float16)zn[c];
            *(half8*)((_Float16*)a.audio_out + row * 8) = o;
          } else {
            float4* ap = (float4*)((float*)a.audio_out + row * 8);
            ap[0] = make_float4(zn[0], zn[1], zn[2], zn[3]);
            ap[1] = make_float4(zn[4], zn[5], zn[6], zn[7]);
          }
        }
      } else {
        // ------------------------------------------------ forward flow (model.py:200-218)
        float z[kMaxGroup];
        if (a.first) {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            z[c] = load_io(a.audio_in, (size_t)b * L * 8 + (size_t)t * 8 + c, a.io_f16);           // :195
        } else {
          const float4* zp = (const float4*)(a.Z + row * 8);
          const float4* op = (const float4*)(a.out + row * 8);
          float4 z0 = zp[0], z1 = zp[1], o0 = op[0], o1 = op[1];
          float o[kMaxGroup];
          z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
          o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
          const int h = a.h_in;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < h) {
              z[h + j] = expf(o[h + j]) * z[h + j] + o[j];                                         // :213-215
              a.log_s_out[((size_t)b * h + j) * L + t] = o[h + j];                                 // :216
            }
          }
        }
        const int c_in = a.first ? 8 : a.c_in;
        const int np = a.last ? c_in : a.n_peel;                                                   // :201-203, :220
        for (int e = 0; e < np; ++e) a.z_out[((size_t)b * 8 + a.z_out_ch0 + e) * L + t] = z[e];
        if (!a.last) {
          const int c = a.c_next;                                                                  // = c_in - np
          for (int r = 0; r < c; ++r) {                                                            // :64 W z
            float s = 0.0f;
            for (int cc = 0; cc < c; ++cc) s = fmaf(a.winv[r * c + cc], z[np + cc], s);
            zn[r] = s;
          }
        }
      }
      if (!a.last) {
        float4* zp = (float4*)(a.Z + row * 8);
        zp[0] = make_float4(zn[0], zn[1], zn[2], zn[3]);
        zp[1] = make_float4(zn[4], zn[5], zn[6], zn[7]);
        float4* op = (float4*)(a.out + row * 8);
        op[0] = make_float4(a.out_init[0], a.out_init[1], a.out_init[2], a.out_init[3]);
        op[1] = make_float4(a.out_init[4], a.out_init[5], a.out_init[6], a.out_init[7]);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_a0[tid][j] = zn[j];
      }
    }
  }
  if (a.last) return;
  __syncthreads();

  // ---- WN.start of the next flow (model.py:117): x[P] = sum_j Wst[P][j] a0[j] + b[P], fp16, position-major.
  // piece = (chunk cc, row, 8-position group): 16 contiguous bytes; consecutive threads -> consecutive bytes.
  const int C = a.C, h = a.h_next;
  const int pieces = (C / 64) * FL_ROWS * 8;
  for (int idx = tid; idx < pieces; idx += 256) {
    const int cc = idx / (FL_ROWS * 8);
    const int rem = idx - cc * (FL_ROWS * 8);
    const int rl = rem >> 3, g8 = rem & 7;
    const size_t row = row0 + rl;
    if (row >= nrows) continue;
    const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
    const int P0 = cc * 64 + g8 * 8;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = a.bstart[P0 + e];
      for (int j = 0; j < h; ++j) s = fmaf(a.wstart[(P0 + e) * h + j], s_a0[rl][j], s);
      o[e] = (_Float16)s;
    }
    const size_t prow = (size_t)b * a.g.Lp + a.g.G + t;
    *(half8*)(a.x + ((size_t)cc * a.g.R + prow) * 64 + g8 * 8) = o;
  }
}

hipError_t launch_flow(const FlowArgs& a, hipStream_t s) {
  const size_t nrows = (size_t)a.g.B * a.g.L;
  const unsigned grid = (unsigned)((nrows + FL_ROWS - 1) / FL_ROWS);
  hipLaunchKernelGGL(flow_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace wg
