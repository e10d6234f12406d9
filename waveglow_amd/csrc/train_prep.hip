// Training plumbing on the device (round 3): from the module's OWN parameter tensors -- weight-normed (g, v) pairs or dense
// weights, in their native layouts -- to everything wg_train_forward / wg_train_backward consume, and from the packed
// gradients back to one gradient per parameter.  Replaces the torch ops that did this around the library calls
// (torch._weight_norm forward / backward, stack / cat / permute of 277 MB per step, a batched matmul for the
// W_end x W_skip fold and its chain rule, ~350 small copies that gave every parameter its .grad).
//
// Reference modules: WN.start / in_layers / cond_layer / res_skip_layers (weight_norm, model.py:85-113), WN.end
// (model.py:90-92, not weight-normed), Invertible1x1Conv.conv (model.py:29-43), WaveGlow.upsample (model.py:145-150).
//   weight norm (torch.nn.utils.parametrizations.weight_norm, dim 0):  w[r] = g[r] * v[r] / ||v[r]||
//     d v[r] = s (dW[r] - (dW[r] . v[r]) v[r] / ||v[r]||^2),  d g[r] = (dW[r] . v[r]) / ||v[r]||,   s = g[r] / ||v[r]||
//   end x skip fold: Wes_i = W_end . W_skip_i ;  out_init = W_end . sum_i b_skip_i + b_end     
#include "wg_train.h"

namespace wg {

namespace {

__device__ __forceinline__ int p2c(int P) { return pos_to_chan(P); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// where entry fl of a per-layer gradient tensor lives (wg_train_grads: dense or interleaved records)
__device__ __forceinline__ size_t gofs(const PrepArgs& a, int fl, size_t dense) {
  return a.layer_stride ? (size_t)(fl / a.nl) * (size_t)a.flow_stride + (size_t)(fl % a.nl) * (size_t)a.layer_stride
                        : (size_t)fl * dense;
}

// ---- rows of the weight-normed modules, one wave per row.  Class 0: in_layers [FL][2C rows][C*3]; 1: cond_layer
// [nf][2C*nl][M8]; 2: res_skip_layers [FL][2C or C][C]; 3: start [nf][C][h_k].
struct RowRef {
  int cls, mod, row, len, srow;     // module index (fl or k), row inside it, row length, index into the scale arrays
};
__device__ __forceinline__ bool row_ref(const PrepArgs& a, long long r, RowRef& q) {
  const int C = a.C, nl = a.nl;
  const long long nA = (long long)a.FL * 2 * C, nB = (long long)a.nf * 2 * C * nl, nC = (long long)a.FL * 2 * C, nD = (long long)a.nf * C;
  if (r < nA) { q.cls = 0; q.mod = (int)(r / (2 * C)); q.row = (int)(r % (2 * C)); q.len = 3 * C; q.srow = (int)r; return true; }
  r -= nA;
  if (r < nB) { q.cls = 1; q.mod = (int)(r / (2 * C * nl)); q.row = (int)(r % (2 * C * nl)); q.len = a.M8; q.srow = (int)r; return true; }
  r -= nB;
  if (r < nC) {
    q.cls = 2; q.mod = (int)(r / (2 * C)); q.row = (int)(r % (2 * C)); q.len = C; q.srow = (int)r;
    return (q.mod % nl) < nl - 1 || q.row < C;      // the last layer of a flow has C rows (all skip, model.py:106-110)
  }
  r -= nC;
  if (r < nD) { q.cls = 3; q.mod = (int)(r / C); q.row = (int)(r % C); q.len = a.hk[q.mod]; q.srow = (int)r; return true; }
  return false;
}
__device__ __forceinline__ long long total_rows(const PrepArgs& a) {
  return (long long)a.FL * 2 * a.C * 2 + (long long)a.nf * 2 * a.C * a.nl + (long long)a.nf * a.C;
}
__device__ __forceinline__ int sec_v(int cls) { return cls == 0 ? SEC_IN_V : cls == 1 ? SEC_CO_V : cls == 2 ? SEC_RS_V : SEC_ST_V; }
__device__ __forceinline__ float* scale_of(const PrepArgs& a, int cls) {
  return cls == 0 ? a.s_in : cls == 1 ? a.s_co : cls == 2 ? a.s_rs : a.s_st;
}

}  // namespace

// s[row] = g / ||v||, inv[row] = 1 / ||v||   (dense weights: s = 1, inv = 0)
__global__ void __launch_bounds__(256) rownorm_kernel(const PrepArgs a) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  RowRef q;
  if (r >= total_rows(a) || !row_ref(a, r, q)) return;
  const int sv = sec_v(q.cls);
  const float* v = (const float*)a.tab[prep_slot(a, sv, q.mod)] + (size_t)q.row * q.len;
  const float* g = (const float*)a.tab[prep_slot(a, sv + 1, q.mod)];
  float* s = scale_of(a, q.cls);
  float* inv = s + a.n_scale[q.cls];
  if (!g) {
    if (lane == 0) { s[q.srow] = 1.0f; inv[q.srow] = 0.0f; }
    return;
  }
  float acc = 0.0f;
  for (int i = lane; i < q.len; i += 64) { const float x = v[i]; acc += x * x; }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float n = sqrtf(acc);
    s[q.srow] = g[q.row] / n;
    inv[q.srow] = 1.0f / n;
  }
}

// W_end zero-padded to 8 rows, sum_i b_skip_i, out_init; one workgroup per flow
__global__ void __launch_bounds__(256) end_prep_kernel(const PrepArgs a) {
  const int k = blockIdx.x, C = a.C, nl = a.nl, h2 = 2 * a.hk[k];
  const float* we = (const float*)a.tab[prep_slot(a, SEC_EN_W, k)];
  const float* be = (const float*)a.tab[prep_slot(a, SEC_EN_B, k)];
  float* w8 = a.wend8 + (size_t)k * 8 * C;
  float* bs = a.bsum + (size_t)k * C;
  __shared__ float red[8][256];
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < C; j += 256) {
    float sum = 0.0f;
    for (int i = 0; i < nl; ++i) {
      const float* b = (const float*)a.tab[prep_slot(a, SEC_RS_B, k * nl + i)];
      sum += b[i < nl - 1 ? C + j : j];
    }
    bs[j] = sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = e < h2 ? we[(size_t)e * C + j] : 0.0f;
      w8[(size_t)e * C + j] = w;
      part[e] += w * sum;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[e][threadIdx.x] = part[e];
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0.0f;
    for (int i = 0; i < 256; ++i) t += red[threadIdx.x][i];
    ((float*)a.tab[prep_slot(a, SEC_O_OINIT, k)])[threadIdx.x] = t + (threadIdx.x < h2 ? be[threadIdx.x] : 0.0f);
  }
}

// wes[fl][e][c] = sum_j W_end8[k][e][j] * W_skip_fl[j][c]; one workgroup per (flow, layer), thread per c
__global__ void __launch_bounds__(256) wes_fold_kernel(const PrepArgs a) {
  const int fl = blockIdx.x, C = a.C, nl = a.nl, k = fl / nl, i = fl % nl;
  const float* v = (const float*)a.tab[prep_slot(a, SEC_RS_V, fl)];
  const float* s = a.s_rs + (size_t)fl * 2 * C;
  const float* w8 = a.wend8 + (size_t)k * 8 * C;
  const int r0 = i < nl - 1 ? C : 0;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < C; ++j) {
      const float w = v[(size_t)(r0 + j) * C + c] * s[r0 + j];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w8[(size_t)e * C + j] * w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) a.wes[((size_t)fl * 8 + e) * C + c] = acc[e];
  }
}

// b1 (gate pre-scaled), b2, bup, and per flow wstart / bstart (position order) and w1x1
__global__ void __launch_bounds__(256) small_prep_kernel(const PrepArgs a) {
  const int C = a.C, nl = a.nl, FL = a.FL, nf = a.nf, M8 = a.M8;
  const size_t n_b1 = (size_t)FL * 2 * C, n_b2 = (size_t)FL * C, n_up = M8, n_st = (size_t)nf * C * 5, n_w = (size_t)nf * 64;
  const size_t total = n_b1 + n_b2 + n_up + n_st + n_w;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    size_t e = idx;
    if (e < n_b1) {
      const int fl = (int)(e / (2 * C)), m = (int)(e % (2 * C)), k = fl / nl, i = fl % nl;
      const float bi = ((const float*)a.tab[prep_slot(a, SEC_IN_B, fl)])[m];
      const float bc = ((const float*)a.tab[prep_slot(a, SEC_CO_B, k)])[(size_t)i * 2 * C + m];
      a.b1[e] = (bi + bc) * (m < C ? 2.8853900817779268f : -1.4426950408889634f);
      continue;
    }
    e -= n_b1;
    if (e < n_b2) {
      const int fl = (int)(e / C), r = (int)(e % C);
      a.b2[e] = (fl % nl) < nl - 1 ? ((const float*)a.tab[prep_slot(a, SEC_RS_B, fl)])[r] : 0.0f;
      continue;
    }
    e -= n_b2;
    if (e < n_up) {
      a.bup[e] = ((const float*)a.tab[prep_slot(a, SEC_UP_B, 0)])[p2c((int)e) >> 3];
      continue;
    }
    e -= n_up;
    if (e < n_st) {
      const int k = (int)(e / (5 * C)), rest = (int)(e % (5 * C)), P = rest / 5, j = rest % 5, h = a.hk[k], ch = p2c(P);
      if (j == 4) {
        ((float*)a.tab[prep_slot(a, SEC_O_BSTART, k)])[P] = ((const float*)a.tab[prep_slot(a, SEC_ST_B, k)])[ch];
      } else if (j < h) {
        ((float*)a.tab[prep_slot(a, SEC_O_WSTART, k)])[(size_t)P * h + j] =
            ((const float*)a.tab[prep_slot(a, SEC_ST_V, k)])[(size_t)ch * h + j] * a.s_st[(size_t)k * C + ch];
      }
      continue;
    }
    e -= n_st;
    {
      const int k = (int)(e / 64), r = (int)(e % 64) / 8, cc = (int)(e % 8), c = a.ck[k];
      if (r < c && cc < c) ((float*)a.tab[prep_slot(a, SEC_O_W1X1, k)])[r * c + cc] = ((const float*)a.tab[prep_slot(a, SEC_CV_W, k)])[r * c + cc];
    }
  }
}

hipError_t launch_prepare(const PrepArgs& a, hipStream_t s) {
  const long long rows = (long long)a.FL * 2 * a.C * 2 + (long long)a.nf * 2 * a.C * a.nl + (long long)a.nf * a.C;
  hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(end_prep_kernel, dim3(a.nf), dim3(256), 0, s, a);
  hipLaunchKernelGGL(wes_fold_kernel, dim3(a.FL), dim3(256), 0, s, a);
  hipLaunchKernelGGL(small_prep_kernel, dim3(512), dim3(256), 0, s, a);
  return hipGetLastError();
}

// =============================================================================================
// Gradients per parameter.
// =============================================================================================

// Weight-normed rows (and the dense case: d v = dW), one wave per row.  dW of a row comes straight out of the packed
// gradients: in_layers from dw1's tap-major K, cond_layer from dw1's last M8 columns, the res rows from dw2, the skip
// rows through the end x skip fold (W_end^T . dwes), start from dstart.
__global__ void __launch_bounds__(256) wn_grad_kernel(const PrepArgs a) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  RowRef q;
  if (r >= total_rows(a) || !row_ref(a, r, q)) return;
  const int C = a.C, nl = a.nl, K1 = 3 * C + a.M8;
  const int sv = sec_v(q.cls);
  const int slot_v = prep_slot(a, sv, q.mod);
  const float* v = (const float*)a.tab[slot_v] + (size_t)q.row * q.len;
  const bool normed = a.tab[prep_slot(a, sv + 1, q.mod)] != nullptr;
  float* dv = a.flat + a.goff[slot_v] + (size_t)q.row * q.len;
  const float* sarr = scale_of(a, q.cls);
  const float s = sarr[q.srow], inv = sarr[a.n_scale[q.cls] + q.srow];

  // element x of the row's dW
  const float* src = nullptr;       // class-specific base
  int k = 0, w_row = 0;
  bool skip_row = false;
  if (q.cls == 0) src = a.dw1 + gofs(a, q.mod, (size_t)2 * C * K1) + (size_t)q.row * K1;
  else if (q.cls == 1) {
    const int i = q.row / (2 * C), m = q.row % (2 * C);
    src = a.dw1 + gofs(a, q.mod * nl + i, (size_t)2 * C * K1) + (size_t)m * K1 + 3 * C;
  } else if (q.cls == 2) {
    const int i = q.mod % nl;
    k = q.mod / nl;
    skip_row = i == nl - 1 || q.row >= C;
    w_row = i == nl - 1 ? q.row : q.row - C;
    src = skip_row ? a.dwes + gofs(a, q.mod, (size_t)8 * C) : a.dw2 + gofs(a, q.mod, (size_t)C * C) + (size_t)q.row * C;
  } else {
    src = (const float*)a.tab[prep_slot(a, SEC_G_DSTART, q.mod)];
  }
  float w8[8];
  if (skip_row) {
#pragma unroll
    for (int e = 0; e < 8; ++e) w8[e] = a.wend8[((size_t)k * 8 + e) * C + w_row];
  }
  auto dW = [&](int x) -> float {
    if (q.cls == 0) { const int c = x / 3, tap = x - 3 * c; return src[(size_t)tap * C + c]; }     // native [c][tap]
    if (q.cls == 1) return src[x];
    if (q.cls == 2) {
      if (!skip_row) return src[x];
      float t = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) t += w8[e] * src[(size_t)e * C + x];
      return t;
    }
    return src[(size_t)x * C + q.row];       // dstart [5][C]: column j of the start weight
  };
  // one pass over dW and v: the row (<= 12 elements per lane up to 768 columns) stays in registers between the dot product
  // and the write (longer rows -- 512 channels x 3 taps -- take the two-pass form)
  constexpr int kKeep = 12;
  if (q.len <= 64 * kKeep) {
    float dw[kKeep], vv[kKeep];
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
      const int x = lane + 64 * j;
      dw[j] = x < q.len ? dW(x) : 0.0f;
      vv[j] = (normed && x < q.len) ? v[x] : 0.0f;
      dot += dw[j] * vv[j];
    }
    dot = wave_sum(dot);
    const float coef = s * dot * inv * inv;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
      const int x = lane + 64 * j;
      if (x < q.len) dv[x] = normed ? s * dw[j] - coef * vv[j] : dw[j];
    }
    if (normed && lane == 0) a.flat[a.goff[prep_slot(a, sv + 1, q.mod)] + q.row] = dot * inv;
    return;
  }
  float dot = 0.0f;
  if (normed)
    for (int x = lane; x < q.len; x += 64) dot += dW(x) * v[x];
  dot = wave_sum(dot);
  const float coef = s * dot * inv * inv;
  for (int x = lane; x < q.len; x += 64) dv[x] = normed ? s * dW(x) - coef * v[x] : dW(x);
  if (normed && lane == 0) a.flat[a.goff[prep_slot(a, sv + 1, q.mod)] + q.row] = dot * inv;
}

// d W_end[k][e][j] = sum_i sum_c dwes[fl][e][c] * W_skip_fl[j][c] + dout_init[k][e] * bsum[k][j]; one workgroup per (k, j)
__global__ void __launch_bounds__(256) end_grad_kernel(const PrepArgs a) {
  const int C = a.C, nl = a.nl;
  const int k = blockIdx.x / C, j = blockIdx.x % C, h2 = 2 * a.hk[k];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nl; ++i) {
    const int fl = k * nl + i, row = i < nl - 1 ? C + j : j;
    const float* v = (const float*)a.tab[prep_slot(a, SEC_RS_V, fl)] + (size_t)row * C;
    const float sc = a.s_rs[(size_t)fl * 2 * C + row];
    const float* dwes = a.dwes + gofs(a, fl, (size_t)8 * C);
    for (int c = threadIdx.x; c < C; c += 256) {
      const float w = v[c] * sc;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += dwes[(size_t)e * C + c] * w;
    }
  }
  __shared__ float red[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = wave_sum(acc[e]);
    if ((threadIdx.x & 63) == 0) red[e][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < h2) {
    const int e = threadIdx.x;
    const float* doi = (const float*)a.tab[prep_slot(a, SEC_G_DOINIT, k)];
    const float t = (red[e][0] + red[e][1]) + (red[e][2] + red[e][3]) + doi[e] * a.bsum[(size_t)k * C + j];
    a.flat[a.goff[prep_slot(a, SEC_EN_W, k)] + (size_t)e * C + j] = t;
  }
}

// biases, the 1x1 weights and the upsample filter: index arithmetic only
__global__ void __launch_bounds__(256) small_grad_kernel(const PrepArgs a) {
  const int C = a.C, nl = a.nl, FL = a.FL, nf = a.nf, M8 = a.M8, M = M8 / 8;
  const size_t n_in = (size_t)FL * 2 * C, n_co = (size_t)nf * 2 * C * nl, n_rs = (size_t)FL * 2 * C, n_st = (size_t)nf * C,
               n_en = (size_t)nf * 8, n_cv = (size_t)nf * 64, n_uw = (size_t)M * M * 1024, n_ub = M;
  const size_t total = n_in + n_co + n_rs + n_st + n_en + n_cv + n_uw + n_ub;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    size_t e = idx;
    if (e < n_in) {
      const int fl = (int)(e / (2 * C)), m = (int)(e % (2 * C));
      a.flat[a.goff[prep_slot(a, SEC_IN_B, fl)] + m] = a.db1[gofs(a, fl, (size_t)2 * C) + m];
      continue;
    }
    e -= n_in;
    if (e < n_co) {
      const int k = (int)(e / ((size_t)2 * C * nl)), row = (int)(e % ((size_t)2 * C * nl)), i = row / (2 * C), m = row % (2 * C);
      a.flat[a.goff[prep_slot(a, SEC_CO_B, k)] + row] = a.db1[gofs(a, k * nl + i, (size_t)2 * C) + m];
      continue;
    }
    e -= n_co;
    if (e < n_rs) {
      const int fl = (int)(e / (2 * C)), row = (int)(e % (2 * C)), k = fl / nl, i = fl % nl;
      if (i == nl - 1 && row >= C) continue;
      float v;
      if (i < nl - 1 && row < C) {
        v = a.db2[gofs(a, fl, (size_t)C) + row];
      } else {       // skip bias: out_init = W_end . sum_i b_skip_i + b_end
        const int j = i == nl - 1 ? row : row - C;
        const float* doi = (const float*)a.tab[prep_slot(a, SEC_G_DOINIT, k)];
        v = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += a.wend8[((size_t)k * 8 + q) * C + j] * doi[q];
      }
      a.flat[a.goff[prep_slot(a, SEC_RS_B, fl)] + row] = v;
      continue;
    }
    e -= n_rs;
    if (e < n_st) {
      const int k = (int)(e / C), c = (int)(e % C);
      a.flat[a.goff[prep_slot(a, SEC_ST_B, k)] + c] = ((const float*)a.tab[prep_slot(a, SEC_G_DSTART, k)])[(size_t)4 * C + c];
      continue;
    }
    e -= n_st;
    if (e < n_en) {
      const int k = (int)(e / 8), q = (int)(e % 8);
      if (q < 2 * a.hk[k]) a.flat[a.goff[prep_slot(a, SEC_EN_B, k)] + q] = ((const float*)a.tab[prep_slot(a, SEC_G_DOINIT, k)])[q];
      continue;
    }
    e -= n_en;
    if (e < n_cv) {
      const int k = (int)(e / 64), r = (int)(e % 64) / 8, cc = (int)(e % 8), c = a.ck[k];
      if (r < c && cc < c) a.flat[a.goff[prep_slot(a, SEC_CV_W, k)] + r * c + cc] = ((const float*)a.tab[prep_slot(a, SEC_G_DW1X1, k)])[r * 8 + cc];
      continue;
    }
    e -= n_cv;
    if (e < n_uw) {
      // upsample.weight [i][o][tap 1024], tap = 256 j + 8 p + g  <-  dwup[p][8 o + g][128 j + i]
      const int i = (int)(e / ((size_t)M * 1024)), o = (int)((e / 1024) % M), tap = (int)(e % 1024);
      const int j = tap >> 8, p = (tap >> 3) & 31, g = tap & 7;
      a.flat[a.goff[prep_slot(a, SEC_UP_W, 0)] + e] = a.dwup[((size_t)p * M8 + 8 * o + g) * 512 + 128 * j + i];
      continue;
    }
    e -= n_uw;
    {
      float v = 0.0f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += a.dbup[8 * e + g];
      a.flat[a.goff[prep_slot(a, SEC_UP_B, 0)] + e] = v;
    }
  }
}

hipError_t launch_param_grads(const PrepArgs& a, hipStream_t s) {
  const long long rows = (long long)a.FL * 2 * a.C * 2 + (long long)a.nf * 2 * a.C * a.nl + (long long)a.nf * a.C;
  hipLaunchKernelGGL(wn_grad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(end_grad_kernel, dim3(a.nf * a.C), dim3(256), 0, s, a);
  hipLaunchKernelGGL(small_grad_kernel, dim3(2048), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace wg
