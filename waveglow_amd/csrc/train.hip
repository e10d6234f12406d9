// gfx950 kernels of the TRAINING direction: WaveGlow.forward with saved activations and its backward pass
// (reference: src/waveglow/model.py:178-221 under autograd, train.py:190-199 loss.backward()).
//
//   plane_gemm_kernel       D = A . B over fp16 planes (B = runs of 64-channel planes read at tap offsets): the upsample
//                           and the cond_layer dgrad (d spect).  The WN layers themselves run on wn_layer_kernel
//                           (kernels.hip, MODE 1-3).
//   wgrad_kernel            dW = G^T-contracted-over-rows X: both operands are [row][channel] planes, so the MFMA
//                           fragments are fetched with the transposing LDS read ds_read_b64_tr_b16.
//   row kernels             coupling / 1x1 / start backward, column sums, slab reduction, mel planes.
//
// First correct version of this path: compiler-managed waits, register-staged LDS tiles, one barrier per K-step
// (the hand-scheduled inference kernel in kernels.hip is the template for tuning it).
#include "wg_train.h"

#include <cstdlib>
#include <type_traits>

namespace wg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

namespace {

// plane row of tile-local row rr of phase p shifted by dt group-timesteps (RowGeom)
__device__ __forceinline__ size_t shifted_row(const RowGeom& g, int p, int dt) {
  const int pp = p + dt;
  return (size_t)((long long)kRowPad + (long long)(pp & 31) * g.Rp + (pp >> 5));
}

// XCD-aware workgroup order (cdna_hip_programming.md T1, bijective form): blocks are dealt round-robin over the 8 XCDs,
// each with its own L2, so the blocks that share an operand tile are given CONSECUTIVE logical ids on ONE XCD: block
// `bid` of `nwg` becomes logical workgroup (its XCD label's contiguous range) + (its turn on that XCD).  Speed only.
__device__ __forceinline__ unsigned xcd_order(unsigned bid, unsigned nwg) {
  const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// channel stored at position P (wg_common.h: pos_to_chan): inside a 32-block, position 16h + 4g + i holds channel 8g + 4h + i
__device__ __forceinline__ size_t pos_to_natural(size_t P) {
  const unsigned p = (unsigned)P & 31u;
  return (P & ~(size_t)31) + 8 * ((p >> 2) & 3u) + 4 * (p >> 4) + (p & 3u);
}

__device__ __forceinline__ bool column_valid(const RowGeom& g, int p, int rr, int& b, int& t) {
  b = rr / g.Fp;
  const int f = rr - b * g.Fp - g.Gf;
  t = 32 * f + p;
  return b < g.B && f >= 0 && f < g.F && t < g.L;
}

}  // namespace

// =============================================================================================
// plane GEMM  D[M x columns] = A[M x K] . B[K x columns] (+ bias) -> fp16 planes.  Two users per step: the upsample
// (one [640 x 512] matrix per phase) and d spect = W_cond^T . d pre as ONE GEMM with K = (all layers) x 2C.
// Workgroup = 4 waves (one per SIMD, up to 512 registers each) = one tile of 32*CT consecutive rows of one phase x
// 128*MT matrix rows; wave w owns the 32-row blocks w, 4 + w, ... of the group.  MT = 5 covers the 640 spectrogram
// channels (80 mels x 8) in ONE workgroup: the B tile is staged and read from LDS once, not once per 128-row group
// (five groups x one ds_read_b128 per MFMA kept the LDS queues full: 47 % MFMA-busy).  K-step = one 64-channel plane
// chunk: the B tile goes global -> registers -> LDS (XOR-swizzled 16-byte pieces, double buffered, one barrier per
// step); A comes in MFMA-fragment order [K-step][32-row block][sub-step s][lane][8] (wg_train.h), so a wave's load of
// one fragment is one contiguous KiB; element j of lane (r, h) in sub-step s is k = 32h + 8s + j of the step, and the
// B fragment of that sub-step is LDS piece 4h + s of the lane's column -- the K order inside a step is free as long
// as both operands agree.  (Reading fragments straight from a row-major matrix -- 64 lanes on 64 different cache
// lines per load -- ran the whole kernel at the L1's line rate: 2.5x slower.)
// XCD-aware workgroup order: the row groups of one column tile (MT = 1: five of them at M = 640) are consecutive
// workgroups of one XCD, so the tile's K rows come out of HBM once (6.7 -> 5.5 ms for d spect at config 4).
// (Round 1 also ran the forward gate / residual / end x skip GEMMs and the gate derivative through this kernel; they
//  now run on wn_layer_kernel, kernels.hip.)
// =============================================================================================
constexpr int PG_WAVES = 4;
constexpr int PG_THREADS = 64 * PG_WAVES;

template <int CT, int MT>
__global__ void __launch_bounds__(PG_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
plane_gemm_kernel(const PGemmArgs a) {
  constexpr int BN = 32 * CT;    // tile width in rows = B-tile rows: CT 16-byte pieces per thread
  __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const RowGeom& g = a.g;
  const int tpp = g.Rp / BN;
  const unsigned wgid = xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int by = (int)(wgid % gridDim.y), tile = (int)(wgid / gridDim.y);
  const int p = tile / tpp, r0 = (tile - p * tpp) * BN;
  const size_t R64 = (size_t)g.R * 64;

  // 32-row blocks of this wave; blocks past the matrix run the same instruction stream on block 0, nothing stored
  const _Float16* Ap = a.A + (size_t)p * a.a_phase_stride;
  const size_t a_step = (size_t)a.n_blk * 2048;   // elements per K-step: blocks x 4 sub-steps x 64 lanes x 8
  int blk[MT];
  const _Float16* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    blk[mt] = (by * MT + mt) * PG_WAVES + w;
    arow[mt] = Ap + (size_t)(blk[mt] * 32 < a.M ? blk[mt] : 0) * 2048 + lane * 8;
  }

  f32x16 acc[MT][CT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mt][ct][j] = 0.0f;

  int n_steps = 0;
  for (int i = 0; i < a.n_runs; ++i) n_steps += a.run[i].n_chunks;

  // B-tile staging: thread -> pieces tid + 256 q, q < CT (row = idx >> 3 = brow0 + 32 q, piece = idx & 7)
  const int brow0 = tid >> 3, bpc = tid & 7;
  // physical 16-byte piece = logical ^ ((row >> 1) & 7): a ds_read_b128 is served in four 16-lane groups
  // ({0-3,12-15,20-27}, ... MI355X_MICROARCH.md, LDS) over 64 banks = two rows; this key gives every group 16
  // distinct (row parity, piece) pairs.  (Keyed on row & 7 it was 2-way: 43 % of the LDS cycles were conflicts.)
  const int bsw = (brow0 >> 1) & 7;                                    // same key for rows brow0 + 32 q
  const int lds_w0 = brow0 * 64 + ((bpc ^ bsw) << 3);

  // Prefetch distance TWO K-steps (one step of MFMAs is shorter than a loaded L2/HBM round trip when all 256 CUs
  // burst their tiles together): A fragments in a two-step register ring, every fragment refilled for step st+2
  // right after its last MFMA of step st; B in two register stages: fetched for st+2 at the top of step st,
  // committed to the other LDS buffer at the end of step st+1.  Waits are the compiler's (counted vmcnt); the
  // sched_barriers pin the issue order, which is what makes the counts come out as "wait for the oldest only".
  half8 aring[2][MT][4], bst[2][CT];
  int ri = 0, ci = 0;   // run / chunk of the next B tile to fetch (tiles are fetched in step order)
  // (the loop body is kept branch-free: hipcc's vmcnt bookkeeping turns conservative -- vmcnt(0) -- at every
  //  control-flow join, so past the last step the fetches simply repeat the last tile / fragment)
  auto fetch_b = [&](half8* dst) {
    const PRun& R = a.run[ri];
    const _Float16* src = R.base + (size_t)ci * R64 + (shifted_row(g, p, R.dt) + r0) * 64;
#pragma unroll
    for (int q = 0; q < CT; ++q) dst[q] = *(const half8*)(src + (size_t)(brow0 + 32 * q) * 64 + bpc * 8);
    const bool last_chunk = ci + 1 == R.n_chunks, last_run = ri + 1 == a.n_runs;
    ci = last_chunk ? (last_run ? ci : 0) : ci + 1;
    ri = (last_chunk && !last_run) ? ri + 1 : ri;
  };
  auto commit_b = [&](int buf, const half8* srcr) {
#pragma unroll
    for (int q = 0; q < CT; ++q) *(half8*)&sB[buf][lds_w0 + 32 * 64 * q] = srcr[q];
  };
  auto fetch_a = [&](half8 (*dst)[4], int step) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[mt][s] = *(const half8*)(arow[mt] + (size_t)step * a_step + 512 * s);
  };

  const int last = n_steps - 1;
  fetch_b(bst[0]);
  fetch_a(aring[0], 0);
  fetch_b(bst[1]);
  fetch_a(aring[1], last < 1 ? last : 1);
  commit_b(0, bst[0]);
  __syncthreads();

  auto body = [&](auto PAR, int st) {
    constexpr int par = decltype(PAR)::value;    // = st & 1: ring slot, B stage and LDS buffer of this step
    const int st2 = st + 2 < last ? st + 2 : last;
    fetch_b(bst[par]);
    __builtin_amdgcn_sched_barrier(0);
    // B fragments are read one sub-step ahead (one wave per SIMD: nobody else covers the LDS latency)
    half8 bf[2][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      bf[0][ct] = *(const half8*)&sB[par][(ct * 32 + r) * 64 + (((4 * h) ^ ((r >> 1) & 7)) << 3)];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          bf[(s + 1) & 1][ct] = *(const half8*)&sB[par][(ct * 32 + r) * 64 + (((4 * h + s + 1) ^ ((r >> 1) & 7)) << 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[par][mt][s], bf[s & 1][ct], acc[mt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        aring[par][mt][s] = *(const half8*)(arow[mt] + (size_t)st2 * a_step + 512 * s);
      __builtin_amdgcn_sched_barrier(0);
    }
    commit_b(par ^ 1, bst[par ^ 1]);
    __syncthreads();
  };
  int st = 0;
  for (; st + 1 < n_steps; st += 2) {
    body(std::integral_constant<int, 0>{}, st);
    body(std::integral_constant<int, 1>{}, st + 1);
  }
  if (st < n_steps) body(std::integral_constant<int, 0>{}, st);

  // ---- epilogue: lane (column r of column tile ct, half h) holds matrix positions P0 .. P0+15 of each of its 32-blocks
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (blk[mt] * 32 >= a.M) continue;
    const int P0 = blk[mt] * 32 + 16 * h;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int rr = r0 + ct * 32 + r;
      int b, t;
      const bool valid = column_valid(g, p, rr, b, t);
      const size_t prow = (size_t)kRowPad + (size_t)p * g.Rp + rr;
      const size_t addr = ((size_t)(P0 >> 6) * g.R + prow) * 64 + (P0 & 63);
      half8 o0, o1;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = acc[mt][ct][j];
        if (a.bias) v += a.bias[P0 + j];
        if (!valid) v = 0.0f;
        if (j < 8) o0[j] = (_Float16)v; else o1[j - 8] = (_Float16)v;
      }
      *(half8*)(a.o0 + addr) = o0; *(half8*)(a.o0 + addr + 8) = o1;
    }
  }
}

namespace {
int device_cus() {      // of the launch's (current) device
  static int n_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int& n = n_dev[dev];
  if (!n) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n = v;
    if (n <= 0) n = 256;
  }
  return n;
}
}  // namespace

// Tile width: one workgroup per CU, so a launch runs in ceil(workgroups / CUs) rounds; among the widths that divide Rp
// (and fit the registers: MT x CT accumulator tiles of 16 registers) pick the one with the least rounds x width, wider
// first (config 4, M = 640 as one row group: Rp = 2304 -> 768 tiles of 96 columns = exactly 3 rounds on 256 CUs).
int plane_gemm_tile_cols(const RowGeom& g, int m_groups, int mt) {
  const int max_ct = mt >= 5 ? 3 : 6;
  if (const char* e = getenv("WG_TRAIN_CT")) {      // tests: pin the tile width (2, 3, 4 or 6 column tiles of 32)
    const int ct = atoi(e);
    if ((ct == 2 || ct == 3 || ct == 4 || ct == 6) && ct <= max_ct && g.Rp % (32 * ct) == 0) return ct;
  }
  const int cus = device_cus();
  static const int widths[4] = {6, 4, 3, 2};
  int best = 0;
  long long best_cost = 0;
  for (int i = 0; i < 4; ++i) {
    const int ct = widths[i];
    if (ct > max_ct || g.Rp % (32 * ct)) continue;
    const long long wgs = (long long)kPhases * (g.Rp / (32 * ct)) * m_groups;
    // rounds x width, plus 2.5 % per dropped column tile: a narrower tile re-reads the weights more often from L2
    const long long cost = ((wgs + cus - 1) / cus) * ct * (1000 + 25 * (6 - ct));
    if (!best || cost < best_cost) { best = ct; best_cost = cost; }
  }
  return best;
}

hipError_t launch_plane_gemm(const PGemmArgs& a, hipStream_t s) {
  if (a.g.Rp % 128 || a.n_runs < 1 || a.n_runs > kMaxRuns || a.M < 1 || !a.o0) return hipErrorInvalidValue;
  int k = 0;
  for (int i = 0; i < a.n_runs; ++i) k += a.run[i].n_chunks * 64;
  if (k != a.ktot) return hipErrorInvalidValue;
  // all rows in one workgroup when they are 5 x 128 (80 mel channels x 8); WG_TRAIN_MT=1 (tests, A/B): 128-row groups
  int mt = (a.M == 5 * 32 * PG_WAVES) ? 5 : 1;
  if (const char* e = getenv("WG_TRAIN_MT"))
    if (atoi(e) == 1) mt = 1;
  const int m_groups = (a.M + 32 * PG_WAVES * mt - 1) / (32 * PG_WAVES * mt);
  const int ct = plane_gemm_tile_cols(a.g, m_groups, mt);
  dim3 grid(kPhases * (a.g.Rp / (32 * ct)), m_groups);
  if (mt == 5) {
    if (ct == 3) hipLaunchKernelGGL((plane_gemm_kernel<3, 5>), grid, dim3(PG_THREADS), 0, s, a);
    else hipLaunchKernelGGL((plane_gemm_kernel<2, 5>), grid, dim3(PG_THREADS), 0, s, a);
  } else if (ct == 6) hipLaunchKernelGGL((plane_gemm_kernel<6, 1>), grid, dim3(PG_THREADS), 0, s, a);
  else if (ct == 4) hipLaunchKernelGGL((plane_gemm_kernel<4, 1>), grid, dim3(PG_THREADS), 0, s, a);
  else if (ct == 3) hipLaunchKernelGGL((plane_gemm_kernel<3, 1>), grid, dim3(PG_THREADS), 0, s, a);
  else hipLaunchKernelGGL((plane_gemm_kernel<2, 1>), grid, dim3(PG_THREADS), 0, s, a);
  return hipGetLastError();
}

// =============================================================================================
// Slab reduction.  out[i] = scale * sum_s slabs[s][i]   (fixed order: bitwise reproducible).  HBM-bound: 16-byte loads;
// a workgroup is 64 float4 columns x 8 slab groups (thread (x, y) sums slabs y, y+8, ... with two independent partial
// sums), then the eight groups are combined through LDS in a fixed order.  Up to kMaxSlabSegs independent segments per
// launch; a segment's sums can be written in natural channel order, and a segment can be wgrad_kernel's BLOCKED tiles
// (SlabSeg): thread x of a workgroup then stands for lane x of the storing wave, so the reads are the stores' contiguous
// KiB and the (row, 4 columns) of a float4 is decoded from its position (wgrad_tile, wg_train.h).
// =============================================================================================
struct SlabMultiArgs {
  SlabSeg seg[kMaxSlabSegs];
  unsigned first_block[kMaxSlabSegs + 1];    // blocks [first_block[i], first_block[i+1]) work on segment i
  int n_segs;
};
constexpr int SR_GROUPS = 8;                 // 512 threads

__device__ __forceinline__ void slab_reduce_body(const SlabMultiArgs& a, unsigned block, float4 (*part)[64]) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < kMaxSlabSegs; ++i)
    if (i < a.n_segs && block >= a.first_block[i]) si = i;
  const SlabSeg& g = a.seg[si];
  const unsigned nb = a.first_block[si + 1] - a.first_block[si];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const size_t n4 = g.n >> 2;
  const int groups = g.blocked ? g.n_groups : 1, per_group = g.n_slabs / groups;
  const size_t items = n4 * (size_t)groups;        // (group, float4) pairs; n4 is a multiple of 64 when groups > 1
  for (size_t base = (size_t)(block - a.first_block[si]) * 64; base < items; base += (size_t)nb * 64) {
    const size_t it = base + tx;
    const size_t grp = it / n4, i = it - grp * n4;
    const float* sl = g.slabs + grp * (size_t)per_group * g.stride;
    bool live = it < items;
    size_t e = 4 * i;                              // output element of this float4
    if (live && g.blocked) {
      const unsigned lane = (unsigned)i & 63u, q = ((unsigned)i >> 6) & 31u, wv = ((unsigned)i >> 11) & 7u;
      const WgradTile t = wgrad_tile(g.m_chunks, g.k_chunks, (int)(i >> 14));
      const int wm = t.shape ? (int)(wv >> 1) : (int)(wv >> 2), wk = t.shape ? (int)(wv & 1) : (int)(wv & 3);
      const int mch = t.mc0 + 2 * wm + (int)(q >> 4), kch = t.kc0 + wk;
      live = mch < g.m_chunks && kch < g.k_chunks;          // sub-tiles past the matrix are never stored
      size_t m = (size_t)mch * 64 + 16 * ((q >> 2) & 3u) + (lane & 15u);
      size_t k = (size_t)kch * 64 + 16 * (q & 3u) + 4 * (lane >> 4);
      if (g.perm & 1) m = pos_to_natural(m);
      if (g.perm & 2) k = pos_to_natural(k);
      e = m * (size_t)g.row_len + k;
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (live) {
      int k = ty;
      for (; k + SR_GROUPS < per_group; k += 2 * SR_GROUPS) {
        const float4 u = *(const float4*)(sl + (size_t)k * g.stride + 4 * i);
        const float4 v = *(const float4*)(sl + (size_t)(k + SR_GROUPS) * g.stride + 4 * i);
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
      }
      if (k < per_group) {
        const float4 u = *(const float4*)(sl + (size_t)k * g.stride + 4 * i);
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
      }
    }
    part[ty][tx] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (ty == 0 && live) {
      float4 o;
#define WG_SR_COMBINE(c) (((part[0][tx].c + part[1][tx].c) + (part[2][tx].c + part[3][tx].c)) + \
                          ((part[4][tx].c + part[5][tx].c) + (part[6][tx].c + part[7][tx].c))) * g.scale
      o.x = WG_SR_COMBINE(x); o.y = WG_SR_COMBINE(y); o.z = WG_SR_COMBINE(z); o.w = WG_SR_COMBINE(w);
#undef WG_SR_COMBINE
      if (!g.blocked && g.perm) {
        size_t m = e / (size_t)g.row_len, k = e - m * (size_t)g.row_len;
        if (g.perm & 1) m = pos_to_natural(m);
        if (g.perm & 2) k = pos_to_natural(k);
        e = m * (size_t)g.row_len + k;
      }
      *(float4*)(g.out + grp * g.out_group_stride + e) = o;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64 * SR_GROUPS) slab_reduce_multi_kernel(const SlabMultiArgs a) {
  __shared__ float4 part[SR_GROUPS][64];
  slab_reduce_body(a, blockIdx.x, part);
}

namespace {
// fills `a` from the segments; returns the number of workgroups (0: invalid)
unsigned plan_slab_reduce(const SlabSeg* segs, int n_segs, SlabMultiArgs& a) {
  if (n_segs < 1 || n_segs > kMaxSlabSegs) return 0;
  a.n_segs = n_segs;
  unsigned total = 0;
  for (int i = 0; i < n_segs; ++i) {
    const SlabSeg& g = segs[i];
    if ((g.n & 3) || (g.stride & 3) || !g.slabs || !g.out || g.n_slabs < 1) return 0;
    if (g.blocked) {
      if (g.m_chunks < 1 || g.k_chunks < 1 || g.n_groups < 1 || g.n_slabs % g.n_groups || g.row_len != g.k_chunks * 64 ||
          g.n != (size_t)wgrad_tiles(g.m_chunks, g.k_chunks) * kWgradTileFloats)
        return 0;
    } else {
      if (g.perm && (g.row_len < 4 || (g.row_len & 3) || g.n % (size_t)g.row_len)) return 0;
      if ((g.perm & 2) && (g.row_len & 31)) return 0;
      if ((g.perm & 1) && ((g.n / (size_t)g.row_len) & 31)) return 0;
    }
    a.seg[i] = g;
    if (!g.blocked) { a.seg[i].n_groups = 1; a.seg[i].out_group_stride = 0; }
    a.first_block[i] = total;
    size_t blocks = (g.n / 4 * (size_t)(g.blocked ? g.n_groups : 1) + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    total += (unsigned)blocks;
  }
  for (int i = n_segs; i <= kMaxSlabSegs; ++i) a.first_block[i] = total;
  return total;
}
}  // namespace

hipError_t launch_slab_reduce_multi(const SlabSeg* segs, int n_segs, hipStream_t s) {
  SlabMultiArgs a;
  const unsigned total = plan_slab_reduce(segs, n_segs, a);
  if (!total) return hipErrorInvalidValue;
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(total), dim3(64 * SR_GROUPS), 0, s, a);
  return hipGetLastError();
}

// =============================================================================================
// Weight gradient.  dW[m][k'] = sum_rows G[row][m] X[row + shift][k'].  Both operands are [row][channel] planes and the
// contraction runs over ROWS, so an MFMA fragment (8 k = rows, one channel per lane) is a column of the LDS tile:
// fetched with ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a 4-row x 16-column block
// (cdna_hip_programming.md T10).  Guard / invalid rows of G are zero by construction (every producer writes zeros
// there), so no row masking.  v_mfma_f32_16x16x32_f16: one MFMA spans the 32 rows of a step (against 32x32x16 tiles the
// chip holds a higher clock with this shape: MI355X_MICROARCH.md, DVFS item 7).
//
// Round 3: ONE workgroup per CU with a 256 x 256 output tile (half of the CU's register file is accumulators).  The
// round-2 kernel (256 x 128 tiles, two workgroups per CU, one slab per phase) streamed 24 KiB of operands from L2 per
// 2.1 MFLOP: 1.7 GB of L2 requests per 155 us launch and 38 % MFMA-busy -- the operand stream, not the matrix pipe, set
// its pace, and its 704 + 256 workgroups ran in 1.9 rounds.  Now:
//   * tile = 4 G chunks x 4 X chunks (8 waves as 2 x 4, each 128 x 64 = 8 x 4 MFMA tiles) or, over the last 1-3 X chunks
//     of a matrix whose K' is not a multiple of 256, 8 G chunks x 2 X chunks (waves 4 x 2): the same wave shape, so
//     d W1 [512 x 1408] is 10 + 1 tiles with no padding work.  32 KiB per 4.2 MFLOP (40 for the 8 x 2 shape);
//   * the rows (32 phases x Rp) are cut into n_slabs contiguous ranges of 32-row steps -- a range may start and end
//     anywhere, also inside a phase -- with n_slabs = CUs / tiles (wg_train_backward): (tiles x slabs) workgroups are
//     ONE round of the chip, every workgroup equally long, and the slabs to write and reduce are 21 instead of 32;
//   * both jobs of a layer ride in one launch over the same slab partition; d (W_end W_skip) = d out x acts^T needs only
//     16 rows of a fifth G plane: the d W2 tile stages that plane as one more slice and its first wave row spends 4 of
//     its 36 MFMAs per step on it (`extra`) instead of a 256-row tile that would be 94 % padding;
//   * bias gradients (column sums of G) ride on the matrix pipe: one MFMA per fragment against an all-ones operand, the
//     duty spread over the tiles of a row of tiles and over the waves of a tile;
//   * operands swapped (A = X fragment, B = G fragment): a lane's four results are four consecutive K' of one row, and
//     the tile is stored in accumulator order, one contiguous KiB per store instruction (the round-2 epilogue wrote
//     64-byte segments: 2 ms per step); slab_reduce un-blocks.
// Staging: LDS-DMA (global_load_lds_dwordx4) into a three-stage ring, two steps ahead, hand-counted vmcnt.  A DMA
// instruction writes 64 consecutive 16-byte slots, so the LDS tile cannot be padded; a slice is [32 rows][128 B] of one
// chunk with the 32-byte column blocks of row r XOR-ed with (r >> 1) & 3 (applied on the SOURCE side: lane -> which 16
// bytes of its row it fetches): the 8 rows x 32 B a 32-lane half of ds_read_b64_tr_b16 touches then fall on 8 different
// 32-byte bank groups.  Wave w stages the 8-row block w & 3 of slices (w >> 2), (w >> 2) + 2, ...
// Workgroups of one slab (they share its G and X rows) are consecutive on one XCD (xcd_order).
// =============================================================================================
constexpr int WG_STEP = 32;     // rows per step
constexpr int WD_STAGES = 3;
constexpr int WD_MAX_SLICES = 11;                          // 8 + 2 + extra
constexpr int WD_STAGE_BYTES = WD_MAX_SLICES * 4096;       // 44 KiB
constexpr int WD_LDS_BYTES = WD_STAGES * WD_STAGE_BYTES;   // 132 KiB: one workgroup per CU
constexpr int WD_MAX_DMA = (WD_MAX_SLICES + 1) / 2;        // slices per wave

namespace {
__device__ __forceinline__ void tr_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void tr_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// all but the `keep` youngest vector-memory operations of this wave have landed (keep is wave-uniform)
__device__ __forceinline__ void tr_wait_keep(int keep) {
  switch (keep) {
    case 0: tr_wait_vm<0>(); break;
    case 1: tr_wait_vm<1>(); break;
    case 2: tr_wait_vm<2>(); break;
    case 3: tr_wait_vm<3>(); break;
    case 4: tr_wait_vm<4>(); break;
    case 5: tr_wait_vm<5>(); break;
    default: tr_wait_vm<6>(); break;
  }
}
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
struct WgradLaunch {
  WgradJob job[2];
  int tiles[2];           // tiles of each job (the second may be 0)
  RowGeom g;
  int n_slabs[2];
  unsigned long long* stamps;   // diagnostic builds only (-DWGR_STAMPS): [workgroups][8]
};
#ifdef WGR_STAMPS
#define WGR_STAMP(j) do { if (tid == 0 && L.stamps) L.stamps[(size_t)blockIdx.x * 32 + (j)] = __builtin_amdgcn_s_memtime(); } while (0)
// inside step 10: wave 0 (early half) -> slots 8.., wave 4 (late half) -> slots 16..
#define WGR_STEP_STAMP(j) do { if (st == 10 && lane == 0 && (w == 0 || w == 4) && L.stamps) L.stamps[(size_t)blockIdx.x * 32 + 8 + 2 * w + (j)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WGR_STAMP(j) do { } while (0)
#define WGR_STEP_STAMP(j) do { } while (0)
#endif
}  // namespace

// The body, specialised by tile shape and by the extra plane: everything the step loop needs is then a compile-time
// constant or sits in a register.  (The first version of this kernel decided all of it at run time inside the loop --
// which slices to stage, a 7-way switch for the vmcnt immediate, per-fragment bias conditions: 320 scalar instructions
// and 61 branches per step, 1 900 cycles of scalar issue per wave beside 512 cycles of MFMA.)
template <bool SHAPE_B, bool EXTRA>
__device__ __forceinline__ void wgrad_body(const WgradLaunch& L, const WgradJob& a, const WgradTile tl, const int slab,
                                           const int tile, const int job_tiles, const int n_slabs, _Float16* wg_smem) {
  constexpr int MC = SHAPE_B ? 8 : 4, KCW = SHAPE_B ? 2 : 4;
  constexpr int NS = (MC + KCW) / 2;              // staging pieces per wave and step; waves 0-3 one more with EXTRA
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RowGeom& g = L.g;
  const int wm = SHAPE_B ? (w >> 1) : (w >> 2), wk = SHAPE_B ? (w & 1) : (w & 3);
  const int bias_chunks = a.bias_out != nullptr ? tl.bias_chunks : 0;
  const size_t R64 = (size_t)g.R * 64;
  const int m_chunks = a.m_chunks, k_chunks = a.k_chunks;

  // ---- staging duty of this wave: 8-row block rb of slices (w >> 2) + 2 j.  A slice past the matrix is staged from
  // the job's first chunk instead (every wave issues the same number of pieces: the vmcnt immediates depend on it);
  // the accumulators it feeds are never stored.
  const int rb = w & 3;
  const _Float16* sbase[NS + 1];
  int sdt[NS + 1];
#pragma unroll
  for (int j = 0; j < NS + 1; ++j) {
    const int sl = (w >> 2) + 2 * j;
    sbase[j] = a.G;
    sdt[j] = 0;
    if (sl < MC) {
      if (tl.mc0 + sl < m_chunks) sbase[j] = a.G + (size_t)(tl.mc0 + sl) * R64;
    } else if (sl < MC + KCW) {
      int c = tl.kc0 + (sl - MC);
      if (c < k_chunks) {
        int i = 0;
        while (c >= a.run[i].n_chunks) { c -= a.run[i].n_chunks; ++i; }
        sbase[j] = a.run[i].base + (size_t)c * R64;
        sdt[j] = a.run[i].dt;
      }
    } else if (EXTRA) {
      sbase[j] = a.G_extra;
    }
  }
  const bool sixth = EXTRA && w < 4;              // this wave also stages a piece of the extra slice (j = NS)
  // lane -> (row lr of the 8-row block, 16-byte slot): fetches piece slot ^ (key << 1) of its row, key = (row >> 1) & 3
  const int lr = lane >> 3;
  const unsigned voff = (unsigned)(lr * 128 + (((lane & 7) ^ (((lr >> 1) & 3) << 1)) << 4));
  const unsigned smem_addr = (unsigned)(size_t)((__attribute__((address_space(3))) void*)wg_smem);
  const unsigned lds_u = __builtin_amdgcn_readfirstlane(smem_addr + (unsigned)(w >> 2) * 4096u + (unsigned)rb * 1024u);

  // ---- this slab's range of steps (global step = phase * steps_per_phase + step)
  const int spp = g.Rp / WG_STEP;
  const long long total = (long long)kPhases * spp;
  const int gs0 = (int)((long long)slab * total / n_slabs), gs1 = (int)((long long)(slab + 1) * total / n_slabs);
  const int n_steps = gs1 - gs0;

  const char* src[NS + 1];                        // running source pointers: + 4 KiB per step inside a phase
  int ip = gs0 / spp, ist = gs0 - ip * spp;       // cursor of the next step to stage
  auto set_phase = [&](int p, int st0) {
#pragma unroll
    for (int j = 0; j < NS + 1; ++j)
      src[j] = uniform_ptr(sbase[j] + (shifted_row(g, p, sdt[j]) + (size_t)(8 * rb) + (size_t)st0 * WG_STEP) * 64);
  };
  set_phase(ip, ist);
  unsigned lds_w = lds_u;                         // LDS address of this wave's first piece in the stage being filled
  auto issue = [&]() {                            // stage the cursor's step into the next ring stage
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      tr_glds16(src[j], voff, lds_w + (unsigned)j * 8192u);
      src[j] += WG_STEP * 128;
    }
    if (sixth) {
      tr_glds16(src[NS], voff, lds_w + (unsigned)NS * 8192u);
      src[NS] += WG_STEP * 128;
    }
    lds_w = (lds_w == lds_u + 2u * WD_STAGE_BYTES) ? lds_u : lds_w + WD_STAGE_BYTES;
    if (++ist == spp) {
      ist = 0;
      if (++ip < kPhases) set_phase(ip, 0);
    }
  };

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  constexpr int NB = 8 / KCW;      // fragments of its G rows a wave can be responsible for summing (one accumulator each)
  f32x4v acc[8][4], acce[4], accb[NB], accbe;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    accbe[j] = 0.0f;
#pragma unroll
    for (int q = 0; q < NB; ++q) accb[q][j] = 0.0f;
  }
  half8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acce[k][j] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][k][j] = 0.0f;
  }

  // transposing fragment reads: operand lane (u = lane & 15, g16 = lane >> 4) needs 8 k values of channel u of a
  // 16-channel block; k = 8 g16 + j is tile row 4 g16 + j (j < 4) / 16 + 4 g16 + j - 4 (j >= 4) -- both operands use the
  // same map.  The lane's read starts at tile row 4 g16 + (u >> 2), halves 4 (u & 3) of the block, which sits at
  // block position i ^ key(row); the second read is 16 rows further (same key).  Per stage the wave keeps four read
  // bases for its G slices and four for its X slice; everything else is an immediate offset of the read.
  const int g16 = lane >> 4, u = lane & 15;
  const int tr_row = 4 * g16 + (u >> 2);
  const int tr_key = (tr_row >> 1) & 3;
  int gbase[4], xbase[4];   // halves from wg_smem, stage 0
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int off = tr_row * 64 + ((i ^ tr_key) << 4) + 4 * (u & 3);
    gbase[i] = (2 * wm) * 2048 + off;
    xbase[i] = (MC + wk) * 2048 + off;
  }
  auto frag_at = [&](const _Float16* q0) -> half8 {
    const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)q0);
    const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(q0 + 16 * 64));
    const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
    return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // Column sums of the G chunks this tile is responsible for (WgradTile::bias_chunks): fragments i = wk (mod KCW) of this
  // wave's G rows are summed by this wave (bit i), the extra plane's by wave (0, 0) (bit 8).  A fragment of a slice past the
  // matrix is never summed.
  unsigned bias_mask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cl = 2 * wm + (i >> 2);            // chunk of fragment i inside the tile
    if (((bias_chunks >> cl) & 1) && (i & (KCW - 1)) == wk && tl.mc0 + cl < m_chunks) bias_mask |= 1u << i;
  }
  const bool extra_wave = EXTRA && wm == 0;
  if (extra_wave && wk == 0 && a.extra_bias_out) bias_mask |= 256u;
  bias_mask = __builtin_amdgcn_readfirstlane(bias_mask);

  // Two waves share a SIMD (w and w + 4), run the same program and meet at one barrier per step: left alone they read
  // their fragments together (matrix pipe idle) and then queue for the matrix pipe together.  So the second half of the
  // workgroup runs its MFMAs of step st - 1 AFTER the barrier of step st, from fragments it read before that barrier,
  // while the first half reads; then the halves swap (MI355X_MICROARCH.md, "Two waves per SIMD", item 9: split by wave
  // number >= 4).  All 12 fragments of a step live in registers.
  const bool late = w >= 4;
  half8 af[8], bf[4], ef;
  const _Float16* rd = wg_smem;                   // stage being read
  auto read_all = [&]() {
    bf[0] = frag_at(rd + xbase[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = frag_at(rd + gbase[i & 3] + (i >> 2) * 2048);
#pragma unroll
    for (int k = 1; k < 4; ++k) bf[k] = frag_at(rd + xbase[k]);
    if (extra_wave) ef = frag_at(rd + gbase[0] + (MC + KCW) * 2048);     // (wm = 0: gbase[0] is the block-0 offset)
    rd = (rd == wg_smem + 2 * (WD_STAGE_BYTES / 2)) ? wg_smem : rd + WD_STAGE_BYTES / 2;
  };
  auto mfma_all = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef WGR_NOMFMA      // diagnostic builds: one VALU instruction per fragment pair keeps the LDS reads alive
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][k][0] += (float)af[i][0] * (float)bf[k][0];
#else
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[k], af[i], acc[i][k], 0, 0, 0);
#endif
    }
    if (extra_wave) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acce[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[k], ef, acce[k], 0, 0, 0);
    }
    if (bias_mask) {    // one MFMA per fragment this wave sums, against all ones: every row of the result is the column sum
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (bias_mask & (1u << i)) accb[i / KCW] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, af[i], accb[i / KCW], 0, 0, 0);
      if (EXTRA && (bias_mask & 256u)) accbe = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ef, accbe, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  WGR_STAMP(1);
  if (n_steps > 0) issue();
  if (n_steps > 1) issue();
  for (int st = 0; st < n_steps; ++st) {
#ifdef WGR_STAMPS
    if (st == 1) WGR_STAMP(2);
    if (st == n_steps / 2) WGR_STAMP(3);
#endif
    // this wave's pieces of step st have landed: all but the NS youngest (waves with NS + 1 pieces per step wait for the
    // oldest piece of step st + 1 as well, issued a whole step ago)
    WGR_STEP_STAMP(0);
#ifdef WGR_NODMA
    if (st < WD_STAGES) tr_wait_vm<0>();
#else
    if (st + 1 < n_steps) tr_wait_vm<NS>(); else tr_wait_vm<0>();
#endif
    WGR_STEP_STAMP(1);
#ifndef WGR_NOBAR
    __syncthreads();                                                  // everyone's have; stage (st + 2) % 3 is free
#endif
    WGR_STEP_STAMP(2);
    // Staging costs the issuing wave 100+ cycles per piece (in-kernel stamps: 350-530 cycles for a wave's 4-5 pieces when
    // all eight waves issue together right behind the barrier), so the late half runs its MFMAs FIRST -- the matrix pipe
    // starts at the barrier -- and stages afterwards, while the early half stages and reads.
#ifdef WGR_NODMA       // diagnostic builds (A/B timing only, results are garbage): the ring is filled once and never again
    const bool stage_now = st + 2 < n_steps && st + 2 < WD_STAGES;
#else
    const bool stage_now = st + 2 < n_steps;
#endif
    if (!late && stage_now) issue();
    __builtin_amdgcn_sched_barrier(0);
    WGR_STEP_STAMP(3);
    if (late) {
      if (st > 0) mfma_all();            // step st - 1, from the fragments read before this barrier
      if (stage_now) issue();
    }
    __builtin_amdgcn_sched_barrier(0);
    WGR_STEP_STAMP(4);
    read_all();
    WGR_STEP_STAMP(5);
    if (!late) mfma_all();                 // (measured: the early half reading before it stages is 3 % slower)
    WGR_STEP_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (late && n_steps > 0) mfma_all();

  WGR_STAMP(4);
  // ---- epilogue.  D: row = 4 * (lane >> 4) + reg = K' inside a 16-block, col = lane & 15 = m inside a 16-block
  const int Mtot = m_chunks * 64, Ktot = k_chunks * 64;
  const int kch = tl.kc0 + wk;
  float4* dst = (float4*)(a.slabs + ((size_t)slab * job_tiles + tile) * kWgradTileFloats) + (size_t)w * 32 * 64 + lane;
  if (kch < k_chunks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (tl.mc0 + 2 * wm + (i >> 2) >= m_chunks) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[(i * 4 + k) * 64] = make_float4(acc[i][k][0], acc[i][k][1], acc[i][k][2], acc[i][k][3]);
    }
    if (extra_wave) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *(float4*)(a.extra_out + ((size_t)slab * 16 + u) * Ktot + (size_t)kch * 64 + 16 * k + 4 * g16) =
            make_float4(acce[k][0], acce[k][1], acce[k][2], acce[k][3]);
    }
  }
  // column sums: every row of a bias accumulator holds them; lanes 0-15 (row 0) write
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int mch = tl.mc0 + 2 * wm + (i >> 2);
    if ((bias_mask & (1u << i)) && g16 == 0) a.bias_out[(size_t)slab * Mtot + mch * 64 + 16 * (i & 3) + u] = accb[i / KCW][0];
  }
  if (EXTRA && (bias_mask & 256u) && g16 == 0) a.extra_bias_out[(size_t)slab * 16 + u] = accbe[0];
#ifdef WGR_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WGR_STAMP(5);
  if (tid == 0 && L.stamps) {
    L.stamps[(size_t)blockIdx.x * 32 + 6] = ((unsigned long long)(tile + (EXTRA ? 100 : 0)) << 32) | (unsigned)slab;
    L.stamps[(size_t)blockIdx.x * 32 + 7] = ((unsigned long long)n_steps << 32);
  }
#endif
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad_kernel(const WgradLaunch L) {
  extern __shared__ __attribute__((aligned(1024))) _Float16 wg_smem[];
#ifdef WGR_STAMPS
  const int tid = threadIdx.x;
#endif
  WGR_STAMP(0);
  // workgroups of job 0 first (tiles[0] x n_slabs[0]), then job 1's; inside a job the tiles of one slab are consecutive
  const unsigned n0 = (unsigned)(L.tiles[0] * L.n_slabs[0]);
  const bool second = blockIdx.x >= n0;
  const unsigned jb = second ? blockIdx.x - n0 : blockIdx.x, jn = second ? gridDim.x - n0 : n0;
  const unsigned wgid = xcd_order(jb, jn);
  const WgradJob& a = second ? L.job[1] : L.job[0];
  const int job_tiles = second ? L.tiles[1] : L.tiles[0], n_slabs = second ? L.n_slabs[1] : L.n_slabs[0];
  const int slab = (int)(wgid / (unsigned)job_tiles), tile = (int)(wgid - (unsigned)slab * (unsigned)job_tiles);
  const WgradTile tl = wgrad_tile(a.m_chunks, a.k_chunks, tile);
  const bool extra = tl.extra_duty && a.G_extra != nullptr;
  if (tl.shape) {
    if (extra) wgrad_body<true, true>(L, a, tl, slab, tile, job_tiles, n_slabs, wg_smem);
    else wgrad_body<true, false>(L, a, tl, slab, tile, job_tiles, n_slabs, wg_smem);
  } else {
    if (extra) wgrad_body<false, true>(L, a, tl, slab, tile, job_tiles, n_slabs, wg_smem);
    else wgrad_body<false, false>(L, a, tl, slab, tile, job_tiles, n_slabs, wg_smem);
  }
}

namespace {
hipError_t check_wgrad(const WgradJob& a) {
  if (a.n_runs < 1 || a.n_runs > kMaxRuns || a.m_chunks < 1 || !a.G || !a.slabs) return hipErrorInvalidValue;
  int k = 0;
  for (int i = 0; i < a.n_runs; ++i) k += a.run[i].n_chunks;
  if (k != a.k_chunks || k < 1) return hipErrorInvalidValue;
  if (a.G_extra && !a.extra_out) return hipErrorInvalidValue;
  return hipSuccess;
}
}  // namespace

unsigned long long* g_wgrad_stamps = nullptr;     // diagnostic builds: set through wg_debug_set_stamp_buffer

hipError_t launch_wgrad(const WgradJob* jobs, int n_jobs, const RowGeom& g, const int* n_slabs, hipStream_t s) {
  if (n_jobs < 1 || n_jobs > 2 || g.Rp % WG_STEP || !n_slabs) return hipErrorInvalidValue;
  WgradLaunch L;
  unsigned grid = 0;
  for (int j = 0; j < 2; ++j) {
    L.job[j] = jobs[j < n_jobs ? j : 0];
    L.tiles[j] = 0;
    L.n_slabs[j] = 1;
    if (j < n_jobs) {
      const hipError_t e = check_wgrad(jobs[j]);
      if (e != hipSuccess) return e;
      if (n_slabs[j] < 1 || (long long)n_slabs[j] > (long long)kPhases * (g.Rp / WG_STEP)) return hipErrorInvalidValue;
      L.tiles[j] = wgrad_tiles(jobs[j].m_chunks, jobs[j].k_chunks);
      L.n_slabs[j] = n_slabs[j];
      grid += (unsigned)(L.tiles[j] * n_slabs[j]);
    }
  }
  L.g = g;
  L.stamps = g_wgrad_stamps;
  static bool attr_done_dev[64] = {};      // the attribute is per device: keyed by the launch's (current) device
  int cur_dev = 0;
  if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= 64) cur_dev = 0;
  if (!attr_done_dev[cur_dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done_dev[cur_dev] = true;
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(512), WD_LDS_BYTES, s, L);
  return hipGetLastError();
}

// =============================================================================================
// Weight packing (wg_train_pack): natural-order fp32 matrices -> the fp16 MFMA-fragment tensors of wg_train_weights,
// one pass per tensor, one thread per 16-byte output piece (8 consecutive K positions of one row).  Position order is
// a permutation inside 32-blocks (wg_common.h: pos_to_chan): positions p0 .. p0+7 of a block (p0 = 0, 8, 16, 24) are
// natural channels n0 .. n0+3 and n0+8 .. n0+11 with n0 = 16*((p0>>3)&1) + 4*(p0>>4) -- two runs of four.
// =============================================================================================
namespace {
__device__ __forceinline__ int nat0_of(int p0) { return 16 * ((p0 >> 3) & 1) + 4 * (p0 >> 4); }   // p0 in {0, 8, 16, 24}
}  // namespace

// Element accessors of the stacked natural-order matrices.  native = 0: they exist in memory (wg_train_plain).  native = 1
// (wg_train_prepare): read from the module's own tensors with the weight-norm row scale applied on the fly --
//   W1[fl][m][n]: n < 3C: in_layers[fl].weight[m][c = n % C][tap = n / C] (native [2C][C][3]); else cond_layer row
//   (fl % nl) * 2C + m, column n - 3C;   W2[fl][r][c]: res rows of res_skip_layers[fl] (zero in a flow's last layer).
// An fp32 product that is about to be rounded to fp16 must be ROUNDED TO fp32 FIRST (what torch's weight norm followed by
// .half() does): left alone hipcc may fuse multiply and conversion into one v_fma_mix* with a single rounding, which
// differs from the two-step result in the last place of ~3 elements per 100 000.
__device__ __forceinline__ float f32_rounded(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

struct PackSrc {
  const PackArgs& a;
  __device__ __forceinline__ float w1(int fl, int m, int n) const {
    const int C = a.C;
    if (!a.native) return a.w1[((size_t)fl * 2 * C + m) * (3 * C + a.M8) + n];
    const PrepArgs& p = a.prep;
    if (n < 3 * C) {
      const int tap = n / C, c = n - tap * C;
      return f32_rounded(((const float*)p.tab[prep_slot(p, SEC_IN_V, fl)])[((size_t)m * C + c) * 3 + tap] * p.s_in[(size_t)fl * 2 * C + m]);
    }
    const int k = fl / p.nl, row = (fl % p.nl) * 2 * C + m;
    return f32_rounded(((const float*)p.tab[prep_slot(p, SEC_CO_V, k)])[(size_t)row * a.M8 + (n - 3 * C)] * p.s_co[(size_t)k * 2 * C * p.nl + row]);
  }
  __device__ __forceinline__ float w2(int fl, int r, int c) const {
    const int C = a.C;
    if (!a.native) return a.w2[((size_t)fl * C + r) * C + c];
    const PrepArgs& p = a.prep;
    if (fl % p.nl == p.nl - 1) return 0.0f;
    return f32_rounded(((const float*)p.tab[prep_slot(p, SEC_RS_V, fl)])[(size_t)r * C + c] * p.s_rs[(size_t)fl * 2 * C + r]);
  }
  // Wup[p][row = 8 o + g][col = 128 j + i] = upsample.weight[i][o][256 j + 8 p + g], zero for i >= M
  __device__ __forceinline__ float wup(int ph, int row, int col) const {
    if (!a.native) return a.wup[((size_t)ph * a.M8 + row) * 512 + col];
    const int M = a.M8 / 8, j = col >> 7, i = col & 127, o = row >> 3, g = row & 7;
    if (i >= M) return 0.0f;
    return ((const float*)a.prep.tab[prep_slot(a.prep, SEC_UP_W, 0)])[((size_t)i * M + o) * 1024 + 256 * j + 8 * ph + g];
  }
};

__global__ void __launch_bounds__(256) pack_kernel(const PackArgs a) {
  const int C = a.C, M8 = a.M8, NW = a.NW, MB = C / (32 * NW), K1 = 3 * C + M8;
  const PackSrc S{a};
  // 8 K positions of a row that are two runs of four natural columns n0..n0+3, n0+8..n0+11
  auto w1_row_runs = [&](int fl, int m, int n0, float sc) -> half8 {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)f32_rounded(S.w1(fl, m, n0 + (j & 3) + 2 * (j & 4)) * sc);
    return o;
  };
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < a.n_pieces; q += (size_t)gridDim.x * 256) {
    half8 o;
    _Float16* dst = a.dst + q * 8;
    if (a.kind == PACK_A1) {
      // [FL][ks][u1][w][gate][mb][k2][hh][r][8]: row gate*C + 32*(w*MB + mb) + r, K position 64 ks + 32 u1 + 16 k2 + 8 hh
      const size_t per_ks = (size_t)512 * MB * NW, per_fl = per_ks * (K1 / 64);
      const int fl = (int)(q / per_fl);
      size_t e = q - (size_t)fl * per_fl;
      const int ks = (int)(e / per_ks);
      e -= (size_t)ks * per_ks;
      const int r = e & 31, hh = (e >> 5) & 1, k2 = (e >> 6) & 1;
      const int mtq = (int)((e >> 7) % (2 * MB)), rest = (int)((e >> 7) / (2 * MB));
      const int w = rest % NW, u1 = rest / NW;
      const int gate = mtq / MB, mb = mtq - gate * MB;
      const int m = gate * C + 32 * (w * MB + mb) + r;
      const int n0 = 64 * ks + 32 * u1 + nat0_of(16 * k2 + 8 * hh);
      o = w1_row_runs(fl, m, n0, gate ? -1.4426950408889634f : 2.8853900817779268f);
      const int n_tap = 3 * C / 64;
      if (ks < n_tap) dst = a.dst + (((size_t)fl * n_tap + ks) * per_ks + e) * 8;
      else dst = a.dst2 + (((size_t)fl * (K1 / 64 - n_tap) + (ks - n_tap)) * per_ks + e) * 8;
    } else if (a.kind == PACK_A2) {
      // [FL][w][mb][k16][hh][r][8]: W_res[32*(w*MB + mb) + r][position 16 k16 + 8 hh]
      const int r = q & 31, hh = (q >> 5) & 1;
      size_t e = q >> 6;
      const int k16 = (int)(e % (C / 16));
      e /= (C / 16);
      const int blk = (int)(e % (NW * MB)), fl = (int)(e / (NW * MB));
      const int n0 = 32 * (k16 >> 1) + nat0_of(16 * (k16 & 1) + 8 * hh);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (_Float16)S.w2(fl, 32 * blk + r, n0 + (j & 3) + 2 * (j & 4));
    } else if (a.kind == PACK_ES) {
      // [FL][s][l4][row][8]: hi (row < 8) / lo (row >= 8) fp16 half of Wes[row & 7][position 32 s + 8 l4]
      const int row = q & 15, l4 = (q >> 4) & 3;
      const size_t e = q >> 6;
      const int s_ = (int)(e % (C / 32)), fl = (int)(e / (C / 32));
      const float* src = a.wes + ((size_t)fl * 8 + (row & 7)) * C + 32 * s_ + nat0_of(8 * l4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float full = src[(j & 3) + 2 * (j & 4)];
        const _Float16 hi = (_Float16)full;
        o[j] = row < 8 ? hi : (_Float16)(full - (float)hi);
      }
    } else if (a.kind == PACK_WAT || a.kind == PACK_WBT) {
      // plain row blocks [FL][ks][u1][w][mb][k2][hh][r][8]: Mat[32*(w*MB + mb) + r][64 ks + 32 u1 + 16 k2 + 8 hh + j]
      const int Kt = a.kind == PACK_WAT ? C + 64 : 6 * C;
      const size_t per_ks = (size_t)256 * MB * NW, per_fl = per_ks * (Kt / 64);
      const int fl = (int)(q / per_fl);
      size_t e = q - (size_t)fl * per_fl;
      const int ks = (int)(e / per_ks);
      e -= (size_t)ks * per_ks;
      const int r = e & 31, hh = (e >> 5) & 1, k2 = (e >> 6) & 1;
      const int mb = (int)((e >> 7) % MB), rest = (int)((e >> 7) / MB);
      const int w = rest % NW, u1 = rest / NW;
      const int m = 32 * (w * MB + mb) + r;                      // natural channel: a column of the source matrix
      const int k0 = 64 * ks + 32 * u1, p0 = 16 * k2 + 8 * hh;   // K position k0 + p0 + j
      if (a.kind == PACK_WBT) {
        // Mat = W_in[:, :, tap]^T: K = tap * 2C + (position of the d pre row)
        const int tap = k0 / (2 * C), q0 = k0 - tap * 2 * C;
        const int r0 = q0 + nat0_of(p0);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)S.w1(fl, r0 + (j & 3) + 2 * (j & 4), tap * C + m);
      } else if (k0 < C) {
        // Mat = [ W_res^T | ... ]: K = position of the d x row
        const int r0 = k0 + nat0_of(p0);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)S.w2(fl, r0 + (j & 3) + 2 * (j & 4), m);
      } else {
        // ... | (W_end W_skip)^T, 8 channels, padded to 64 ]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = k0 - C + p0 + j;
          o[j] = d < 8 ? (_Float16)a.wes[((size_t)fl * 8 + d) * C + m] : (_Float16)0.0f;
        }
      }
    } else if (a.kind == PACK_WCT) {
      // [t][b][s][h][r][8] of Mat [M8 (pos)][FL*2C (pos)] = cond_layer^T: element = W1[fl][row nat(Q)][3C + 32 b + r]
      const int r = q & 31, h = (q >> 5) & 1, s_ = (q >> 6) & 3;
      const size_t e = q >> 8;
      const int b = (int)(e % (M8 / 32));
      const size_t t = e / (M8 / 32);
      const size_t kc = 64 * t + 32 * h + 8 * s_;
      const int fl = (int)(kc / (2 * C)), q0 = (int)(kc - (size_t)fl * 2 * C);
      const int r0 = (q0 & ~31) + nat0_of(q0 & 31);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (_Float16)S.w1(fl, r0 + (j & 3) + 2 * (j & 4), 3 * C + 32 * b + r);
    } else {
      // PACK_WUP [p][t][b][s][h][r][8] = Wup[p][32 b + r][64 t + 32 h + 8 s + j]   (rows natural: pos(chan_to_pos(r)) = r)
      const int r = q & 31, h = (q >> 5) & 1, s_ = (q >> 6) & 3;
      size_t e = q >> 8;
      const int b = (int)(e % (M8 / 32));
      e /= (M8 / 32);
      const int t = (int)(e % 8), p = (int)(e / 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (_Float16)S.wup(p, 32 * b + r, 64 * t + 32 * h + 8 * s_ + j);
    }
    *(half8*)dst = o;
  }
}

hipError_t launch_pack(const PackArgs& a, hipStream_t s) {
  if (a.n_pieces == 0) return hipSuccess;
  const size_t blocks = (a.n_pieces + 255) / 256;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// mel [B][M][T] -> planes [2 chunks][R][64]: row (p, b, q) = mel[b][:, q] (natural channel order, zero padded to 128),
// identical for every phase; rows of guard frames and frames >= T are zero (transposed-conv padding, model.py:145).
__global__ void __launch_bounds__(256) mel_plane_kernel(const void* __restrict__ mel, int io_f16, int M, RowGeom g,
                                                        _Float16* __restrict__ melp) {
  const size_t n = (size_t)kPhases * g.Rp * 128;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const int i = (int)(idx & 127);
    const size_t row = idx >> 7;                       // p * Rp + rr
    const int rr = (int)(row % g.Rp);
    const int b = rr / g.Fp, f = rr - b * g.Fp - g.Gf;
    float v = 0.0f;
    if (i < M && b < g.B && f >= 0 && f < g.T) {
      const size_t src = ((size_t)b * M + i) * g.T + f;
      v = io_f16 ? (float)((const _Float16*)mel)[src] : ((const float*)mel)[src];
    }
    melp[((size_t)(i >> 6) * g.R + kRowPad + row) * 64 + (i & 63)] = (_Float16)v;
  }
}

hipError_t launch_mel_plane(const void* mel, int io_f16, int M, const RowGeom& g, _Float16* melp, hipStream_t s) {
  if (M > 128) return hipErrorInvalidValue;
  size_t blocks = ((size_t)kPhases * g.Rp * 128 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mel_plane_kernel, dim3((unsigned)blocks), dim3(256), 0, s, mel, io_f16, M, g, melp);
  return hipGetLastError();
}

// =============================================================================================
// Flow backward, part 1 (before the WN backward): affine coupling  a1' = exp(log_s) a1 + b  (model.py:213-216).
//   d b = d a1',  d log_s = d log_s(ext) + d a1' * exp(log_s) * a1,  d a1 = d a1' * exp(log_s)
// (d b | d log_s) is the gradient of the WN output: written as the fp16 K-segment plane of the backward GEMM.
// =============================================================================================
constexpr int FB_ROWS = 256;

__global__ void __launch_bounds__(FB_ROWS) flow_bwd_pre_kernel(const FlowBwdArgs a) {
  const int L = a.g.L;
  const size_t nrows = (size_t)a.g.B * L;
  const size_t row = (size_t)blockIdx.x * FB_ROWS + threadIdx.x;
  if (row >= nrows) return;
  const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
  const int h = a.h, c = a.c;
  float go[kMaxGroup], z[kMaxGroup], o[kMaxGroup];
  {
    const float4* zp = (const float4*)(a.Zpost + row * 8);
    const float4* op = (const float4*)(a.OUT + row * 8);
    const float4 z0 = zp[0], z1 = zp[1], o0 = op[0], o1 = op[1];
    z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
    o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
  }
  if (a.from_z) {
#pragma unroll
    for (int j = 0; j < kMaxGroup; ++j)
      go[j] = (j < c && a.g_z) ? a.scale * a.g_z[((size_t)b * 8 + a.z_ch0 + j) * L + t] : 0.0f;
  } else {
    const float4* gp = (const float4*)(a.GZ + row * 8);
    const float4 g0 = gp[0], g1 = gp[1];
    go[0] = g0.x; go[1] = g0.y; go[2] = g0.z; go[3] = g0.w; go[4] = g1.x; go[5] = g1.y; go[6] = g1.z; go[7] = g1.w;
  }
  float ga[kMaxGroup], gout[kMaxGroup];
#pragma unroll
  for (int j = 0; j < kMaxGroup; ++j) { ga[j] = 0.0f; gout[j] = 0.0f; }
#pragma unroll
  for (int j = 0; j < kMaxGroup; ++j) {
    if (j < h) ga[j] = go[j];                             // d a0 (direct part)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j >= h && j - h == q && q < h) {
        const float e = expf(o[j]);                       // o[h+q] = log_s_q, o[q] = b_q
        const float gy = go[j];
        float gls = gy * e * z[j];
        if (a.g_log_s) gls += a.scale * a.g_log_s[((size_t)b * h + q) * L + t];
        ga[j] = gy * e;                                   // d a1
        gout[q] = gy;                                     // d b
        gout[j] = gls;                                    // d log_s
      }
  }
  float4* gp = (float4*)(a.GZ + row * 8);
  gp[0] = make_float4(ga[0], ga[1], ga[2], ga[3]);
  gp[1] = make_float4(ga[4], ga[5], ga[6], ga[7]);
  half8 o16;
#pragma unroll
  for (int j = 0; j < 8; ++j) o16[j] = (_Float16)gout[j];
  const size_t prow = (size_t)kRowPad + (size_t)(t & 31) * a.g.Rp + (size_t)b * a.g.Fp + a.g.Gf + (t >> 5);
  *(half8*)(a.GO + prow * 64) = o16;
}

// Flow backward, part 2 (after the WN backward): start conv, 1x1 conv, peel.
//   d a0 += Wstart^T d x_0 ;  d(W z)= (d a0 | d a1) ;  d z_in = W^T d(W z) ;  d W += d(W z) z_in^T  (model.py:64, :117)
__global__ void __launch_bounds__(FB_ROWS) flow_bwd_post_kernel(const FlowBwdArgs a) {
  __shared__ float s_g[FB_ROWS][9];
  __shared__ float s_x[FB_ROWS][9];
  const int L = a.g.L;
  const size_t nrows = (size_t)a.g.B * L;
  const size_t row = (size_t)blockIdx.x * FB_ROWS + threadIdx.x;
  const int h = a.h, c = a.c;
  float gy[kMaxGroup], xin[kMaxGroup];
#pragma unroll
  for (int j = 0; j < kMaxGroup; ++j) { gy[j] = 0.0f; xin[j] = 0.0f; }
  // ---- d a0 += Wstart^T d x_0.  A thread owns one row, but a row's 512 bytes of d x_0 sit in four planes whose rows for
  // consecutive t are far apart (phase-major): read per owner, every load instruction touched 64 different 128-byte lines
  // for 16 bytes each (58 us per launch).  Now 8 lanes fetch one row's 128-byte line of a plane together, multiply their
  // 8 channels by their rows of Wstart and the partial sums meet in LDS.
  __shared__ unsigned s_prow[FB_ROWS];
  __shared__ float s_wn[FB_ROWS][4];
  int b = 0, t = 0;
  {
    unsigned pr = 0xffffffffu;
    if (row < nrows) {
      const unsigned r32 = (unsigned)row;
      b = (int)(r32 / (unsigned)L);
      t = (int)(r32 - (unsigned)b * (unsigned)L);
      pr = (unsigned)kRowPad + (unsigned)(t & 31) * (unsigned)a.g.Rp + (unsigned)b * (unsigned)a.g.Fp + (unsigned)a.g.Gf + (unsigned)(t >> 5);
    }
    s_prow[threadIdx.x] = pr;
  }
  __syncthreads();
  {
    const int sub = threadIdx.x & 7, r8 = threadIdx.x >> 3;      // 16-byte piece of a row's line; row inside a pass of 32
    float wacc[FB_ROWS / 32][4];
#pragma unroll
    for (int ps = 0; ps < FB_ROWS / 32; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) wacc[ps][j] = 0.0f;
    for (int cc = 0; cc < a.C / 64; ++cc) {
      float w[8][4];                                             // Wstart rows of this lane's 8 channels of the chunk
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[e][j] = (j < h) ? a.wstart[(cc * 64 + sub * 8 + e) * h + j] : 0.0f;
#pragma unroll
      for (int ps = 0; ps < FB_ROWS / 32; ++ps) {
        const unsigned pr = s_prow[ps * 32 + r8];
        if (pr == 0xffffffffu) continue;
        const half8 x = *(const half8*)(a.GX + ((size_t)cc * a.g.R + pr) * 64 + sub * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (float)x[e];
#pragma unroll
          for (int j = 0; j < 4; ++j) wacc[ps][j] = fmaf(w[e][j], xv, wacc[ps][j]);
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < FB_ROWS / 32; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = wacc[ps][j];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (sub == 0) s_wn[ps * 32 + r8][j] = v;
      }
  }
  __syncthreads();
  if (row < nrows) {
    {
      const float4* gp = (const float4*)(a.GZ + row * 8);
      const float4 g0 = gp[0], g1 = gp[1];
      gy[0] = g0.x; gy[1] = g0.y; gy[2] = g0.z; gy[3] = g0.w; gy[4] = g1.x; gy[5] = g1.y; gy[6] = g1.z; gy[7] = g1.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < h) gy[j] += s_wn[threadIdx.x][j];
    // input of this flow's 1x1 conv
    if (a.Zprev) {
      const float4* zp = (const float4*)(a.Zprev + row * 8);
      const float4* op = (const float4*)(a.OUTprev + row * 8);
      const float4 z0 = zp[0], z1 = zp[1], o0 = op[0], o1 = op[1];
      float z[kMaxGroup], o[kMaxGroup], y[kMaxGroup];
      z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
      o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
      const int hp = a.h_prev;
#pragma unroll
      for (int j = 0; j < kMaxGroup; ++j) {
        y[j] = z[j];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j >= hp && j - hp == q && q < hp) y[j] = expf(o[j]) * z[j] + o[q];
      }
#pragma unroll
      for (int e = 0; e < kMaxGroup; ++e) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxGroup; ++q)
          if (q - a.n_peel == e) v = y[q];
        xin[e] = (e < c) ? v : 0.0f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < kMaxGroup; ++e) xin[e] = a.audio[(size_t)b * L * 8 + (size_t)t * 8 + e];
    }
    if (a.Zprev) {
      // d(previous flow's output) = (d z[peeled channels] | W^T d(W z))
      float gin[kMaxGroup], gz[kMaxGroup];
#pragma unroll
      for (int cc = 0; cc < kMaxGroup; ++cc) {
        float s = 0.0f;
#pragma unroll
        for (int rr = 0; rr < kMaxGroup; ++rr)
          if (rr < c && cc < c) s = fmaf(a.w1x1[rr * c + cc], gy[rr], s);
        gin[cc] = s;
      }
#pragma unroll
      for (int e = 0; e < kMaxGroup; ++e) {
        float v = 0.0f;
        if (e < a.n_peel && a.g_z) v = a.scale * a.g_z[((size_t)b * 8 + a.z_peel_ch0 + e) * L + t];
#pragma unroll
        for (int q = 0; q < kMaxGroup; ++q)
          if (e >= a.n_peel && e - a.n_peel == q && q < c) v = gin[q];
        gz[e] = v;
      }
      float4* gp = (float4*)(a.GZ + row * 8);
      gp[0] = make_float4(gz[0], gz[1], gz[2], gz[3]);
      gp[1] = make_float4(gz[4], gz[5], gz[6], gz[7]);
    }
  }
#pragma unroll
  for (int j = 0; j < kMaxGroup; ++j) {
    s_g[threadIdx.x][j] = (j < c) ? gy[j] : 0.0f;
    s_x[threadIdx.x][j] = xin[j];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int rr = threadIdx.x >> 3, cc = threadIdx.x & 7;
    float s = 0.0f;
    for (int i = 0; i < FB_ROWS; ++i) s = fmaf(s_g[i][rr], s_x[i][cc], s);
    a.dw_partial[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

int flow_bwd_workgroups(const RowGeom& g) { return (int)(((size_t)g.B * g.L + FB_ROWS - 1) / FB_ROWS); }

hipError_t launch_flow_bwd_pre(const FlowBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(flow_bwd_pre_kernel, dim3(flow_bwd_workgroups(a.g)), dim3(FB_ROWS), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_flow_bwd_post(const FlowBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(flow_bwd_post_kernel, dim3(flow_bwd_workgroups(a.g)), dim3(FB_ROWS), 0, s, a);
  return hipGetLastError();
}

// d Wstart[P][j] = sum_rows d x_0[row][P] a0[row][j],  d bstart[P] = sum_rows d x_0[row][P]     (model.py:117)
// grid (Rp/128, 32): 128 plane rows of one phase.  Thread -> (16-byte piece of a row = 8 channel positions, row group):
// 32 pieces x 8 row groups cover 256 positions; every thread walks 16 rows with 16-byte loads (the first version read
// 2 bytes per lane and row, 128 dependent rounds per thread: 85 us per launch for 38 MB, twelve launches on the backward
// chain), sums 5 x 8 accumulators and the eight row groups are combined through a wave shuffle and LDS.
__global__ void __launch_bounds__(256) start_wgrad_kernel(const StartWgradArgs a) {
  __shared__ float4 s_a0[128];
  __shared__ float red[4][5][256];
  const RowGeom& g = a.g;
  const int p = blockIdx.y, r0 = blockIdx.x * 128;
  if (threadIdx.x < 128) {
    int b, t;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (column_valid(g, p, r0 + threadIdx.x, b, t)) {
      const float4 z = *(const float4*)(a.Zpost + ((size_t)b * g.L + t) * 8);
      v.x = z.x;
      v.y = a.h > 1 ? z.y : 0.f;
      v.z = a.h > 2 ? z.z : 0.f;
      v.w = a.h > 3 ? z.w : 0.f;
    }
    s_a0[threadIdx.x] = v;
  }
  __syncthreads();
  const size_t slab = (size_t)p * gridDim.x + blockIdx.x;
  const int piece = threadIdx.x & 31, rg = threadIdx.x >> 5, wv = threadIdx.x >> 6;
  for (int P0 = 0; P0 < a.C; P0 += 256) {
    const int Pp = P0 + 8 * piece;           // first of this thread's 8 channel positions
    const bool live = Pp < a.C;
    const _Float16* src = a.GX + ((size_t)(Pp >> 6) * g.R + kRowPad + (size_t)p * g.Rp + r0) * 64 + (Pp & 63);
    float acc[5][8];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = rg + 8 * i;
      half8 x;
      if (live) x = *(const half8*)(src + (size_t)r * 64);
      else {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = (_Float16)0.0f;
      }
      const float4 a0 = s_a0[r];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float xv = (float)x[c];
        acc[0][c] = fmaf(xv, a0.x, acc[0][c]); acc[1][c] = fmaf(xv, a0.y, acc[1][c]);
        acc[2][c] = fmaf(xv, a0.z, acc[2][c]); acc[3][c] = fmaf(xv, a0.w, acc[3][c]);
        acc[4][c] += xv;
      }
    }
    // row groups 2w and 2w + 1 sit in the two halves of wave w
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] += __shfl_xor(acc[j][c], 32, 64);
    if ((threadIdx.x & 32) == 0) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) red[wv][j][8 * piece + c] = acc[j][c];
    }
    __syncthreads();
    float* o = a.partial + slab * 5 * a.C;
    for (int e = threadIdx.x; e < 5 * 256; e += 256) {
      const int j = e >> 8, c = e & 255;
      if (P0 + c < a.C) o[j * a.C + P0 + c] = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
    }
    __syncthreads();
  }
}

int start_wgrad_workgroups(const RowGeom& g) { return kPhases * (g.Rp / 128); }

hipError_t launch_start_wgrad(const StartWgradArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(start_wgrad_kernel, dim3(a.g.Rp / 128, kPhases), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace wg
