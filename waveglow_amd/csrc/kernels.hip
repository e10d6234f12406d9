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

#include <type_traits>

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

// tanh(a) * sigmoid(b)  (model.py:17-19).  The host pre-scales the tanh rows of the GEMM-1 weights/bias by
// 2*log2(e) and the sigmoid rows by -log2(e), so with u = 2a*log2e, v = -b*log2e, E1 = 2^u, E2 = 2^v:
//   tanh(a) sigmoid(b) = (E1 - 1) / ((E1 + 1)(1 + E2))        -- two exp2 and ONE reciprocal (transcendentals are
// quarter rate and dominate the gate).  u is clamped to +-60 (tanh is +-1 to fp32 there) so E1 stays finite and
// non-zero: E2 = inf then gives 1/inf = 0 and E2 = 0 gives plain tanh, no NaN path.
__device__ __forceinline__ float gate_act(float u, float v) {
  const float e1 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(u, -60.0f, 60.0f));
  const float t = 1.0f + __builtin_amdgcn_exp2f(v);
  return (e1 - 1.0f) * __builtin_amdgcn_rcpf(fmaf(e1, t, t));
}

// ---- hand-counted VMEM in the GEMM main loop (cdna_hip_programming.md 5.7): hipcc drains an LDS-DMA before
// any later LDS read and sinks register loads next to their use; both serialise the K loop on memory latency.
// These loads are invisible to the compiler's s_waitcnt bookkeeping; every consumer sits behind wait_vm0*.
// LDS-DMA: 64 lanes x 16 B from (sbase + voff) to LDS [lds_addr + lane*16] (M0 saved/restored in-statement).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload16(half8& dst, const void* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}

// =============================================================================================
// WN layer
// =============================================================================================
// Diagnostic build (-DWG_STAMPS, tools/stamp_phases.py): wave 0 of every workgroup records s_memtime at phase
// boundaries into args.stamps, a buffer nothing else reads.  The shipped library contains no stamp.
#ifdef WG_STAMPS
#define WG_STAMP(i)                                                                              \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    unsigned long long _t;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                   \
    if (a.stamps && HAS_RES && tid == 0) a.stamps[(size_t)tile * 8 + (i)] = _t;                        \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define WG_STAMP(i) do {} while (0)
#endif

constexpr bool kPersistent = true;    // cross-tile prefetch inside a workgroup (see the tile loop of wn_layer_kernel)
#ifdef WG_DBG_NO_XTILE_DMA
constexpr bool kXTileDMA = false;
#else
constexpr bool kXTileDMA = kPersistent;
#endif
// The next tile's A fragments are NOT prefetched under the epilogue: between such an inline-asm load and its wait
// lies a long stretch of compiler-scheduled code, and hipcc moved the still-in-flight destination registers
// there (wrong results).  They are loaded at the tile top instead (~1 L2 latency exposed per tile).
constexpr bool kXTileA = false;
constexpr int kTilesPerWG = 2;        // tiles per workgroup, fully unrolled
// Experiment (-DWG_REGSTAGE_B): stage the B tiles of K-steps >= 1 global -> VGPR -> LDS (loaded two steps ahead,
// ds_write one step ahead, right after the barrier) instead of by LDS-DMA.  Measured SLOWER on MI355X (K loop 59.3k
// vs 57.6k cycles per tile, launch 0.687 vs 0.670 ms), so LDS-DMA stays the default.
#ifdef WG_REGSTAGE_B
constexpr bool kRegStageB = true;
#else
constexpr bool kRegStageB = false;
#endif

template <int C> struct WnCfg {
  static constexpr int NW = (C >= 256) ? 8 : C / 32;   // waves per workgroup
  static constexpr int BN = (C >= 512) ? 64 : 128;     // columns (group-timesteps) per workgroup
};

// "wait until at most N VMEM ops are outstanding" for the hand-issued loads.  Deliberately NOT tied to the
// destination registers: a "+v" tie lets the register allocator copy a fragment BEFORE the wait (observed:
// v_mov of in-flight registers).  Order is pinned instead by the sched_barrier(0) that follows every wait and
// precedes every MFMA group (cdna_hip_programming.md 5.7 form (iii), 5.4 rule 18).
template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// One WN layer for one tile of BN group-timesteps of one utterance.
//   GEMM 1  [2C x (3C + NS)] . [(3C + NS) x BN]: the three dilated taps of in_layers[i] and the layer's slice of
//           cond_layer as ONE K-extended MFMA GEMM (the [B,16C,L] cond tensor of model.py:121 never exists).
//   gate    tanh * sigmoid in the accumulator registers (model.py:13-20) -> fp16 acts tile in LDS.
//   GEMM 2  res rows of res_skip_layers[i] . acts, accumulated onto x (model.py:130-132) -> x_out.
//   GEMM 3  (W_end . W_skip_i) . acts, 8 rows: the skip path folded through WN.end (model.py:133-137), added to out.
// Wave w owns gate channels [32*MB*w, 32*MB*(w+1)): its A (weight) fragments are private, so they go L2 -> VGPR
// directly (pre-packed in fragment order, 1 KiB per wave-load); the B tile (activations) is shared by all waves
// and goes HBM/L2 -> LDS by LDS-DMA.  All VMEM of the K loop is hand-counted (see glds16).
template <int C, int NW, int BN, bool HAS_RES>
__global__ void __launch_bounds__(NW * 64) wn_layer_kernel(const WnLayerArgs a) {
  constexpr int MB = C / (32 * NW);      // 32-channel blocks per wave
  constexpr int MT = 2 * MB;             // M tiles per wave: MB tanh blocks, then MB sigmoid blocks
  constexpr int NTHREADS = NW * 64;
  constexpr int NT = BN / 32;            // 32-column MFMA tiles per wave
  constexpr int CC = C / 64;             // 64-channel chunks of x
  constexpr int BT_BYTES = BN * 128;     // one staged B tile: BN rows x 64 fp16
  constexpr int ACT_ROW = 2 * C + 16;    // bytes per acts row: +16 B pad => conflict-free b128 reads/writes with
                                         // immediate-offset addressing (one base VGPR per 32-column tile)
  constexpr int K2 = C / 16;             // k16 steps of GEMM2
  constexpr int NG = BN * 8 / NTHREADS;  // LDS-DMA instructions per wave per B tile
  constexpr int NAH = MT * 2;            // A fragments per half K-step (packing unit)
  constexpr bool DEFER = (MB == 1);      // defer a step's last sub-step past the barrier (needs spare registers)
  static_assert(BN * 8 % NTHREADS == 0 && MB >= 1 && MB <= 2, "tile geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem;                     // 2 x BT_BYTES
  char* const sActs = smem + 2 * BT_BYTES;   // BN x ACT_ROW

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Lane-derived values are re-derived from an opaque copy of tid at the top of every tile (set_lane_ids), so
  // hipcc cannot share address arithmetic between the unrolled tile bodies and keep it live across a whole tile.
  int lane = tid & 63, ln = lane & 31, lh = lane >> 5;

  // LDS-DMA one B tile: piece idx = row*8 + physical 16-B chunk; logical chunk = phys ^ ((row>>1)&7)
  // (LDS destination is lane-linear, so the bank swizzle is applied to the SOURCE address).
  // Per-lane source offsets are the same for every K-step; only the scalar tile base moves.
  unsigned pvoff[NG];
  int swB;
  unsigned a_voff;
  auto set_lane_ids = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63;
    ln = lane & 31;
    lh = lane >> 5;
    swB = (ln >> 1) & 7;
    a_voff = lane * 16;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int idx = i * NTHREADS + t;
      const int row = idx >> 3, pc = idx & 7;
      pvoff[i] = row * 128 + ((pc ^ ((row >> 1) & 7)) << 4);
    }
  };
  set_lane_ids();
  const unsigned sB_addr = (unsigned)(size_t)WG_LPTR(sB);
  const int R = a.g.R;
  const int nK = 3 * CC + a.ns_chunks;
  auto kstep_src = [&](int r0, int ks) -> const char* {      // scalar: B-tile source of K-step ks
    if (ks < 3 * CC) {
      const int tap = ks / CC, cc = ks - tap * CC;
      const int row = r0 + (tap - 1) * a.dil;     // taps t-d, t, t+d (model.py:98-102: padding = dilation)
      return (const char*)(a.x_in + ((size_t)cc * R + row) * 64);
    }
    return (const char*)(a.spect + ((size_t)(ks - 3 * CC) * R + r0) * 64);
  };
  auto stage_B_piece = [&](int r0, int ks, int bufsel, int i) {
#ifndef WG_DBG_NO_DMA
    glds16(kstep_src(r0, ks), pvoff[i],
           __builtin_amdgcn_readfirstlane(sB_addr + bufsel * BT_BYTES + (i * NTHREADS + wave * 64) * 16));
#endif
  };
  auto read_B = [&](const char* buf, int nt, int k16) -> half8 {
#ifdef WG_DBG_NO_LDSREAD
    half8 z; asm volatile("" : "=v"(z)); return z;
#endif
    const int n = nt * 32 + ln;
    const int c = (k16 * 2 + lh) ^ swB;
    return *(const half8*)(buf + n * 128 + c * 16);
  };
  // A fragments: packed [2*nK half K-steps][wave][MT][2 k16][64 lanes][8].  q[g][mt] holds the fragment of
  // k16 sub-step g (0..3) of the current K-step; one fragment = one 1 KiB wave-load straight from L2.
  auto load_Aq = [&](int ks, int g, int mt, half8& dst) {
#ifdef WG_DBG_A_SAME   // timing experiment only: every step re-reads fragment block 0 (L1-resident)
    ks = 0;
#endif
    const char* p = (const char*)a.wA1 + ((size_t)(2 * ks + (g >> 1)) * NW + wave) * (NAH * 1024) +
                    (mt * 2 + (g & 1)) * 1024;                                      // wave-uniform
#ifndef WG_DBG_NO_ALOAD
    gload16<0>(dst, p, a_voff);
#else
    asm volatile("" : "=v"(dst));
#endif
  };
  auto tile_row0 = [&](int tile) -> int {
    const int b = tile / a.tiles_per_utt;
    return b * a.g.Lp + a.g.G + (tile - b * a.tiles_per_utt) * BN;
  };

  // ---- tile walk.  Blocks b and b+8 share an XCD (round-robin dispatch; speed only), so XCD label x owns a
  // contiguous run of time tiles; its blocks take tiles start+idx, start+idx+step, ... (step = blocks on that
  // label): tiles that run at the same time are neighbours and re-read each other's +-dil halo rows from that L2.
  // A workgroup processes kTilesPerWG tiles in a FULLY UNROLLED loop: the next tile's first B tile (LDS-DMA) and A
  // fragments are issued under the current tile's gate / GEMM2 / epilogue, and the workgroup launch cost is paid
  // once per kTilesPerWG tiles.  (A real persistent loop makes hipcc spill 100+ VGPRs around the back edge.)
  int tile, tile_end;
  const int tile_step = gridDim.x >> 3;        // grid is a multiple of 8
  {
    const int bid = blockIdx.x, ntl = a.n_tiles;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntl >> 3, r = ntl & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile_end = start + (xcd < r ? q + 1 : q);
    tile = start + idx;
  }

  // bias of GEMM1 (pre-scaled in_layer bias + cond bias slice), fp32 [2C], kept in LDS for the whole launch
  float* const sBias = (float*)(sActs + BN * ACT_ROW);
  for (int i = tid; i < 2 * C; i += NTHREADS) sBias[i] = a.bias1[i];

  half8 q[4][MT];
  half8 breg[NG];                            // register-staged B tile (kRegStageB)
  int par = 0;                               // LDS buffer of K-step ks is (ks + par) & 1
  if (tile < tile_end) {
#pragma unroll
    for (int i = 0; i < NG; ++i) stage_B_piece(tile_row0(tile), 0, 0, i);
  }
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) load_Aq(0, g, mt, q[g][mt]);
  __syncthreads();

#pragma unroll
  for (int it = 0; it < kTilesPerWG; ++it, tile += tile_step) {
    if (tile >= tile_end) break;
    if (it > 0) {
      set_lane_ids();
      if constexpr (!kXTileDMA) {
#pragma unroll
        for (int i = 0; i < NG; ++i) stage_B_piece(tile_row0(tile), 0, par, i);
      }
      if constexpr (!kXTileA) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) load_Aq(0, g, mt, q[g][mt]);
      }
    }
    const int b = tile / a.tiles_per_utt;
    const int jt = tile - b * a.tiles_per_utt;
    const int r0 = b * a.g.Lp + a.g.G + jt * BN;   // first plane row of this tile
    const int t0 = jt * BN;                        // first group-timestep
    const int next_tile = tile + tile_step;
    WG_STAMP(0);
    // ---- GEMM1 accumulators start from the bias
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row0 = (mt < MB ? 0 : C) + (wave * MB + (mt < MB ? mt : mt - MB)) * 32;
      const float4* bp = (const float4*)(sBias + row0 + 4 * lh);
      f32x16 v;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q4 = bp[2 * g];
        v[4 * g] = q4.x; v[4 * g + 1] = q4.y; v[4 * g + 2] = q4.z; v[4 * g + 3] = q4.w;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v;
    }

    // ---- K loop (GEMM 1).  One K-step = 4 k16 sub-steps g = 0..3, each MT*NT MFMAs on fragments q[g][.]
    // (weights) x bf[g&1][.] (activations, read from LDS one sub-step ahead).  Every VMEM / LDS instruction is
    // placed by hand BETWEEN MFMAs (one "slot" after each column tile's MFMAs; sched_barrier pins the order): an
    // LDS-DMA or 1-KiB load costs its wave ~60-180 issue cycles, and the two waves of a SIMD run this loop in
    // lockstep (one barrier per K-step), so clustered loads leave the matrix pipe idle in both at once.
    //   after the barrier : read bf[0] <- sub-step 0 fragments
    //   D  (deferred g=3 of the previous step; operands already in registers, covers the LDS latency above)
    //        slots: LDS-DMA pieces of tile ks+1
    //   g=0  slots: read bf[1] <- sub-step 1 ; reload q[3] <- A(ks, 3)
    //   g=1  slots: read bf[0] <- sub-step 2 ; reload q[0] <- A(ks+1, 0)       (wait q[1] first)
    //   g=2  slots: read bf[1] <- sub-step 3 ; reload q[1] <- A(ks+1, 1)       (wait q[2] first)
    //   then reload q[2] <- A(ks+1, 2); vmcnt(2*MT): DMA and q[0] landed; lgkmcnt(0); ONE s_barrier.
    // VMEM issue order per step: DMA xNG, q3 xMT, q0 xMT, q1 xMT, q2 xMT -- every wait is a counted vmcnt.
    if constexpr (kRegStageB) {
      const char* src1 = kstep_src(r0, 1);
#pragma unroll
      for (int i = 0; i < NG; ++i) gload16<0>(breg[i], src1, pvoff[i]);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WG_STAMP(1);
    constexpr int LPS = (MT + NT - 1) / NT;      // A-fragment reloads per slot
    constexpr int GPS = (NG + NT - 1) / NT;      // LDS-DMA pieces per slot
    half8 bf[2][NT];
    auto mfma_col = [&](int g, int nt) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[g][mt], bf[g & 1][nt], acc[mt][nt], 0, 0, 0);
    };
    if constexpr (kPersistent) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)        // bf[1] carries nothing into a tile (the ks > 0 test below): tell
        asm volatile("" : "=v"(bf[1][nt]));  // the register allocator, or it keeps stale fragments alive
    }
    // One K-step; MORE = "a next step exists" is a compile-time flag (the last step is peeled) so the body has
    // no branches (so are the first step's missing deferred MFMAs), and the next tile's DMA source is computed
    // once per step in scalar registers.
    auto kstep = [&](auto more_tag, auto more2_tag, auto first_tag, auto hi_tag, int ks) {
      constexpr bool more = decltype(more_tag)::value;      // step ks+1 exists
      constexpr bool more2 = decltype(more2_tag)::value;    // step ks+2 exists
      constexpr bool first = decltype(first_tag)::value;
      // Stagger: the two waves of a SIMD (w, w + NW/2) run this loop in lockstep; the upper half takes its
      // VMEM slots half a sub-step later, so one wave's load issue sits beside its partner's MFMAs.
      constexpr int SH = decltype(hi_tag)::value ? NT / 2 : 0;
      // VMEM ops this step issues for the B operand ahead of the q loads: LDS-DMA pieces of tile ks+1, or
      // (register staging) the loads of tile ks+2
      constexpr int NB = kRegStageB ? (more2 ? NG : 0) : (more ? NG : 0);
      const char* buf = sB + ((ks + par) & 1) * BT_BYTES;
      const char* src_next = nullptr;
      if constexpr (kRegStageB) { if constexpr (more2) src_next = kstep_src(r0, ks + 2); }
      else { if constexpr (more) src_next = kstep_src(r0, ks + 1); }
      const unsigned lds_next = sB_addr + ((ks + 1 + par) & 1) * BT_BYTES + wave * 1024;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[0][nt] = read_B(buf, nt, 0);
      if constexpr (kRegStageB && more) {
        // tile ks+1 (loaded during step ks-1, landed before the barrier) -> its LDS buffer, free since the barrier
        char* wp = sB + ((ks + 1 + par) & 1) * BT_BYTES + tid * 16;
#pragma unroll
        for (int i = 0; i < NG; ++i) *(half8*)(wp + i * NTHREADS * 16) = breg[i];
      }
      __builtin_amdgcn_sched_barrier(0);
      auto dma_slot = [&](int nt) {
#pragma unroll
        for (int i = nt * GPS; i < (nt + 1) * GPS && i < NG; ++i) {
          if constexpr (kRegStageB) {
            if constexpr (more2) gload16<0>(breg[i], src_next, pvoff[i]);
          } else {
            if constexpr (more) glds16(src_next, pvoff[i], __builtin_amdgcn_readfirstlane(lds_next + i * NTHREADS * 16));
          }
        }
      };
      if constexpr (DEFER) {
        // ---- D: deferred sub-step 3 of step ks-1 + B-operand loads
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (!first) mfma_col(3, nt);
          __builtin_amdgcn_sched_barrier(0);
          dma_slot((nt + NT - SH) % NT);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int g = 0; g < (DEFER ? 3 : 4); ++g) {
        if (g == 1) wait_vm<2 * MT + NB>();                          // q[1] landed
        if (g == 2) wait_vm<NB + MT + (more ? MT : 0)>();            // q[2] landed
        if (g == 3) wait_vm<more ? 2 * MT : 0>();                    // q[3] landed (no deferral)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          mfma_col(g, nt);
          __builtin_amdgcn_sched_barrier(0);
          if (g < 3) bf[(g + 1) & 1][nt] = read_B(buf, nt, g + 1);
          const int snt = (nt + NT - SH) % NT;
          if (!DEFER && g == 0) dma_slot(snt);
          if (g == 0 || more) {
#pragma unroll
            for (int mt = snt * LPS; mt < (snt + 1) * LPS && mt < MT; ++mt)
              load_Aq(g == 0 ? ks : ks + 1, (g + 3) & 3, mt, q[(g + 3) & 3][mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (more) {
        if constexpr (DEFER) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) load_Aq(ks + 1, 2, mt, q[2][mt]);
        }
        // B operand of the next steps and q[0] landed (only q[1], q[2] reloads may be outstanding); own LDS
        // reads of this buffer and LDS writes of the next one done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(2 * MT) : "memory");
#ifndef WG_DBG_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto kloop = [&](auto hi_tag) {
      using T = std::true_type;
      using F = std::false_type;
      kstep(T{}, T{}, T{}, hi_tag, 0);            // nK >= 13: first, middle, second-last and last steps all exist
#pragma clang loop unroll(disable)
      for (int ks = 1; ks < nK - 2; ++ks) kstep(T{}, T{}, F{}, hi_tag, ks);
      kstep(T{}, F{}, F{}, hi_tag, nK - 2);
      kstep(F{}, F{}, F{}, hi_tag, nK - 1);
    };
#ifdef WG_STAGGER   // experiment: two copies of the loop push hipcc into spilling (26 VGPRs @C=256) -- off
    if (wave >= NW / 2) kloop(std::true_type{}); else kloop(std::false_type{});
#else
    kloop(std::false_type{});
#endif
    if constexpr (DEFER) {
      wait_vm<0>();                          // q[3] of the last step
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) mfma_col(3, nt);
    }
    __builtin_amdgcn_sched_barrier(0);
    par = (par + nK) & 1;
    // Next tile's first B tile goes out now, into the LDS buffer the last step did not use (slow waves may
    // still be reading that one); its A fragments follow after the gate -- all land under the phases below.
    if constexpr (kXTileDMA) {
      if (next_tile < tile_end) {
#pragma unroll
        for (int i = 0; i < NG; ++i) stage_B_piece(tile_row0(next_tile), 0, par, i);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    WG_STAMP(2);
    // Per-tile opaque copies of the lane ids: every address / weight load of the phases below depends on them,
    // so hipcc cannot hoist those (tile-invariant) values out of the persistent loop and spill them around it
    // (a spill reload is a VMEM op whose compiler-inserted vmcnt(0) would drain the hand-placed prefetches).
    int lno = ln, lho = lh, laneo = lane;
    asm volatile("" : "+v"(lno), "+v"(lho), "+v"(laneo));
    char* const acts_lane = sActs + lno * ACT_ROW + lho * 32;      // this lane's write slot in row n = lno
    const char* const acts_rd = sActs + lno * ACT_ROW + lho * 16;  // B-fragment read base (k16 = 0)
    // ---- issue the loads the post-gate phases need now, so their latency hides under the gate's VALU work:
    // residual input x (this tile, this wave's channels: lane (n, h) owns positions [32*blk + 16h, +16) of
    // column n = 32 contiguous bytes) and the first GEMM2 weight fragments.
    constexpr int PF = 8 / MB;                // GEMM2 A-fragment prefetch depth (per 32-channel block)
    half8 xres[MB][NT][2];
    half8 a2[MB][PF];
    const half8* const p2 = (const half8*)a.wA2 + (size_t)wave * MB * K2 * 64 + laneo;
    if constexpr (HAS_RES) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int blk = wave * MB + mb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const size_t row = (size_t)(blk >> 1) * R + r0 + nt * 32 + lno;
          const half8* xp = (const half8*)(a.x_in + row * 64 + (blk & 1) * 32 + lho * 16);
          xres[mb][nt][0] = xp[0];
          xres[mb][nt][1] = xp[1];
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) a2[mb][i] = p2[((size_t)mb * K2 + i) * 64];
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- gate (model.py:13-20) in registers; acts -> LDS as fp16, position-major, XOR-swizzled
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        half8 o0, o1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          o0[r] = (_Float16)gate_act(acc[mb][nt][r], acc[MB + mb][nt][r]);
          o1[r] = (_Float16)gate_act(acc[mb][nt][8 + r], acc[MB + mb][nt][8 + r]);
        }
        char* ap = acts_lane + nt * 32 * ACT_ROW + (wave * MB + mb) * 64;   // positions [32*blk + 16h, +16)
        *(half8*)(ap) = o0;
        *(half8*)(ap + 16) = o1;
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    WG_STAMP(3);
    // The accumulators are dead now: room for the next tile's first A fragments.  Unconditional (the weights
    // are the same for every tile) so h0/h1 are plainly dead across the gate above, not "maybe still needed".
    if constexpr (kXTileA) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_Aq(0, g, mt, q[g][mt]);
    }
    __builtin_amdgcn_sched_barrier(0);

    // GEMM2 accumulators start from x + b_res (residual add for free, model.py:132)
    f32x16 acc2[MB][NT];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float* bp = a.bias2 + (wave * MB + mb) * 32 + 4 * lho;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            acc2[mb][nt][r] = (float)xres[mb][nt][0][r] + bv[r];
            acc2[mb][nt][8 + r] = (float)xres[mb][nt][1][r] + bv[8 + r];
          }
      }
    }

    auto read_acts32 = [&](int nt, int k16) -> half8 {       // B fragment for the 32x32x16 MFMA
      return *(const half8*)(acts_rd + nt * 32 * ACT_ROW + k16 * 32);
    };

    const int l15 = laneo & 15, l4 = laneo >> 4;
    constexpr int NGRP = (BN / 16 + NW - 1) / NW;            // 16-column groups per wave
    constexpr bool kWesEarly = (C / 32) * 4 <= 32;           // folded-end weight fragments fit beside GEMM2's registers
    half8 wes[C / 32];
    auto load_wes = [&]() {
      const half8* pe = (const half8*)a.wEs + laneo;
#pragma unroll
      for (int s = 0; s < C / 32; ++s) wes[s] = pe[s * 64];
    };
    if constexpr (kWesEarly) load_wes();                     // issue now, consume after GEMM2

    // ---- GEMM2: res rows of this wave (model.py:130-132); acts fragments read one k16 step ahead
    if constexpr (HAS_RES) {
      half8 bq[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bq[0][nt] = read_acts32(nt, 0);
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        if (k + 1 < K2) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bq[(k + 1) & 1][nt] = read_acts32(nt, k + 1);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const half8 af = a2[mb][k % PF];
          if (k + PF < K2) a2[mb][k % PF] = p2[((size_t)mb * K2 + k + PF) * 64];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc2[mb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bq[k & 1][nt], acc2[mb][nt], 0, 0, 0);
        }
      }
    }

    WG_STAMP(4);
    if constexpr (!kWesEarly) load_wes();
    // ---- folded end x skip (model.py:133-137): out[0:8] += (W_end W_skip_i) acts, 16 columns per group,
    // weights split hi+lo fp16 (rows 0-7 / 8-15 of the 16x16x32 MFMA) so the 8 flow outputs keep ~fp32 weights.
#pragma unroll
    for (int gi = 0; gi < NGRP; ++gi) {
      const int grp = wave + gi * NW;
      if (grp < BN / 16) {
        const int n = grp * 16 + l15;
        const int t = t0 + n;
        const bool valid = laneo < 32 && t < a.g.L;
        float4* op = (float4*)(a.out + ((size_t)b * a.g.L + t) * 8 + 4 * l4);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) o = *op;
        const char* ep = sActs + n * ACT_ROW + l4 * 16;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < C / 32; ++s) {
          const half8 bfe = *(const half8*)(ep + s * 64);
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wes[s], bfe, d, 0, 0, 0);
        }
        // D: col = lane&15, row = 4*(lane>>4)+reg ; rows 8-15 (lanes 32-63) are the lo parts
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] += __shfl_xor(d[r], 32);
        if (valid) {
          o.x += d[0]; o.y += d[1]; o.z += d[2]; o.w += d[3];
          *op = o;
        }
      }
    }

    WG_STAMP(5);
    // ---- x_out = fp16(x + res) for valid columns (rows >= L stay zero: they are other tiles' padding)
    if constexpr (HAS_RES) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int blk = wave * MB + mb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (t0 + nt * 32 + lno < a.g.L) {
            half8 o0, o1;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              o0[r] = (_Float16)acc2[mb][nt][r];
              o1[r] = (_Float16)acc2[mb][nt][8 + r];
            }
            const size_t row = (size_t)(blk >> 1) * R + r0 + nt * 32 + lno;
            half8* xp = (half8*)(a.x_out + row * 64 + (blk & 1) * 32 + lho * 16);
            xp[0] = o0;
            xp[1] = o1;
          }
        }
      }
    }
    WG_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);   // keep the next tile's prologue (128 accumulator inits) out of this epilogue
  }
}

template <int C, int BN, bool HAS_RES>
static hipError_t launch_wn_tt(const WnLayerArgs& a, hipStream_t s) {
  constexpr int NW = WnCfg<C>::NW;
  constexpr int smem = 2 * BN * 128 + BN * (2 * C + 16) + 2 * C * 4;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)wn_layer_kernel<C, NW, BN, HAS_RES>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  // kTilesPerWG tiles per workgroup: per XCD label ceil(tiles_on_label / kTilesPerWG) blocks
  const int per_label = ((a.n_tiles + 7) / 8 + kTilesPerWG - 1) / kTilesPerWG;
  const int grid = 8 * per_label;
  hipLaunchKernelGGL((wn_layer_kernel<C, NW, BN, HAS_RES>), dim3(grid), dim3(NW * 64), smem, s, a);
  return hipGetLastError();
}
template <int C>
static hipError_t launch_wn_t(const WnLayerArgs& a, int bn, hipStream_t s) {
  // the last layer of a WN has no residual output (model.py:106-110): separate instantiation.
  // bn = 64: small-batch variant (twice the tiles; used when 128-column tiles would leave CUs idle).
  if (bn == 64) return a.has_res ? launch_wn_tt<C, 64, true>(a, s) : launch_wn_tt<C, 64, false>(a, s);
  if constexpr (WnCfg<C>::BN == 128) {
    if (bn == 128) return a.has_res ? launch_wn_tt<C, 128, true>(a, s) : launch_wn_tt<C, 128, false>(a, s);
  }
  return hipErrorInvalidValue;
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
int wn_waves(int C) {
  switch (C) {
    case 64: return WnCfg<64>::NW;
    case 128: return WnCfg<128>::NW;
    case 256: return WnCfg<256>::NW;
    case 512: return WnCfg<512>::NW;
  }
  return 0;
}

hipError_t launch_wn_layer(const WnLayerArgs& a, int C, int bn, hipStream_t s) {
  switch (C) {
    case 64: return launch_wn_t<64>(a, bn, s);
    case 128: return launch_wn_t<128>(a, bn, s);
    case 256: return launch_wn_t<256>(a, bn, s);
    case 512: return launch_wn_t<512>(a, bn, s);
  }
  return hipErrorInvalidValue;
}

// =============================================================================================
// Upsample: ConvTranspose1d(M, M, 1024, stride 256) (model.py:145-150) written straight into the squeezed
// conditioning planes (model.py:230-232), as an MFMA GEMM per frame phase t' (= group-timestep inside a frame):
//   S[ch = o*8+g][frame q] = bias[o] + sum_{j<4, i<M} W[i][o][8t'+g+256j] * mel[i][q-j]
// i.e. [NS x 4M] . [4M x frames].  One workgroup: one t', 128 consecutive frames of one utterance, all NS
// channels; 5 waves, wave w owns the 64-channel plane chunks w, w+5, ...  The mel window is transposed once
// into LDS as fp16 [frame][i] (row stride M+8 halfs => conflict-free ds_read_b128 B fragments); weights come
// pre-packed in A-fragment order straight from L2.  Output lanes hold 16 consecutive storage positions of one
// row => two 16-byte stores (spect planes use the same position-major channel order as x, wg_common.h).
// =============================================================================================
constexpr int UP_FRAMES = 128;
constexpr int UP_WAVES = 5;

__global__ void __launch_bounds__(UP_WAVES * 64) upsample_kernel(const UpsampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = a.M;
  const int RS = (M + 8) * 2;                 // melT row stride in bytes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31, lh = lane >> 5;
  const int tp = blockIdx.x;                  // t'
  const int q0 = blockIdx.y * UP_FRAMES;
  const int b = blockIdx.z;
  const int NF = UP_FRAMES + 3;               // frames q0-3 .. q0+127
  for (int idx = tid; idx < M * NF; idx += UP_WAVES * 64) {
    const int i = idx / NF, f = idx - i * NF;
    const int q = q0 - 3 + f;
    const float v = (q >= 0 && q < a.T) ? load_io(a.mel, ((size_t)b * M + i) * a.T + q, a.io_f16) : 0.0f;
    *(_Float16*)(smem + f * RS + i * 2) = (_Float16)v;
  }
  __syncthreads();
  const int KS = M / 4;                       // k16 steps (K = 4M)
  const int KPJ = M / 16;                     // k16 steps per tap
  const int NSC = M / 8;                      // 64-channel chunks (NS/64)
  const half8* const wp = (const half8*)a.w;  // [32 t'][NSC][2 mb][KS][64 lanes][8]
  for (int cs = wave; cs < NSC; cs += UP_WAVES) {
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = cs * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        v[r] = a.bias[ch >> 3];
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mb][nt] = v;
    }
    const half8* pa = wp + ((size_t)(tp * NSC + cs) * 2 * KS) * 64 + lane;
    for (int k = 0; k < KS; ++k) {
      const int j = k / KPJ, i0 = (k - j * KPJ) * 16;
      const half8 a0 = pa[(size_t)k * 64];
      const half8 a1 = pa[(size_t)(KS + k) * 64];
      half8 bf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        bf[nt] = *(const half8*)(smem + (nt * 32 + ln + 3 - j) * RS + (i0 + 8 * lh) * 2);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf[nt], acc[1][nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int q = q0 + nt * 32 + ln;
      const int t = q * 32 + tp;
      if (t < a.g.L) {
        const size_t row = (size_t)b * a.g.Lp + a.g.G + t;
        _Float16* dst = a.spect + ((size_t)cs * a.g.R + row) * 64 + lh * 16;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          half8 o0, o1;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            o0[r] = (_Float16)acc[mb][nt][r];
            o1[r] = (_Float16)acc[mb][nt][8 + r];
          }
          *(half8*)(dst + mb * 32) = o0;
          *(half8*)(dst + mb * 32 + 8) = o1;
        }
      }
    }
  }
}

hipError_t launch_upsample(const UpsampleArgs& a, hipStream_t s) {
  if (a.M % 16 != 0 || a.M > 80) return hipErrorInvalidValue;
  dim3 grid(32, (a.n_q + UP_FRAMES - 1) / UP_FRAMES, a.g.B);
  const int smem = (UP_FRAMES + 3) * (a.M + 8) * 2;
  hipLaunchKernelGGL(upsample_kernel, grid, dim3(UP_WAVES * 64), smem, s, a);
  return hipGetLastError();
}

// =============================================================================================
// Flow step (both directions) + next WN start.  64 rows per workgroup, 256 threads.
// =============================================================================================
constexpr int FL_ROWS = 256;   // rows per workgroup = threads per workgroup

__global__ void __launch_bounds__(FL_ROWS) flow_kernel(const FlowArgs a) {
  __shared__ float4 s_a0[FL_ROWS];
  const int L = a.g.L;
  const size_t nrows = (size_t)a.g.B * L;
  const size_t row0 = (size_t)blockIdx.x * FL_ROWS;
  const int tid = threadIdx.x;

  {
    const size_t row = row0 + tid;
    if (row < nrows) {
      const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
      float zn[kMaxGroup];
#pragma unroll
      for (int c = 0; c < kMaxGroup; ++c) zn[c] = 0.0f;
      if (a.direction == 0) {
        // ------------------------------------------------ inverse flow (model.py:246-271)
        if (a.first) {
#pragma unroll
          for (int c = 0; c < kMaxGroup; ++c)
            if (c < a.c_next)
              zn[c] = a.sigma * load_io(a.z_extra, ((size_t)b * a.c_next + c) * L + t, a.io_f16);   // :243-244
        } else {
          float z[kMaxGroup], o[kMaxGroup], v[kMaxGroup], y[kMaxGroup];
          const float4* zp = (const float4*)(a.Z + row * 8);
          const float4* op = (const float4*)(a.out + row * 8);
          float4 z0 = zp[0], z1 = zp[1], o0 = op[0], o1 = op[1];
          z[0] = z0.x; z[1] = z0.y; z[2] = z0.z; z[3] = z0.w; z[4] = z1.x; z[5] = z1.y; z[6] = z1.z; z[7] = z1.w;
          o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
          const int h = a.h_in, c = a.c_in;
          // v = [a0 ; (a1 - b) / exp(s)]  with b = o[0:h], s = o[h:2h]                               :253-255
#pragma unroll
          for (int j = 0; j < kMaxGroup; ++j) {
            float num = z[j], den = 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j >= h && j - h == q) { num = z[j] - o[q]; den = expf(o[j]); }
            v[j] = (j < c) ? num / den : 0.0f;
          }
#pragma unroll
          for (int r = 0; r < kMaxGroup; ++r) {                                                     // :258 / :59
            float s = 0.0f;
#pragma unroll
            for (int cc = 0; cc < kMaxGroup; ++cc)
              if (r < c && cc < c) s = fmaf(a.winv[r * c + cc], v[cc], s);
            y[r] = s;
          }
          const int ne = a.n_extra;
#pragma unroll
          for (int e = 0; e < kMaxGroup; ++e) {                                                     // :260-271
            float val = 0.0f;
            if (e < ne) val = a.sigma * load_io(a.z_extra, ((size_t)b * ne + e) * L + t, a.io_f16);
#pragma unroll
            for (int r = 0; r < kMaxGroup; ++r)
              if (e >= ne && e - ne == r) val = y[r];
            zn[e] = val;
          }
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
          for (int j = 0; j < kMaxGroup; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j >= h && j - h == q && q < h) {
                z[j] = expf(o[j]) * z[j] + o[q];                                                   // :213-215
                a.log_s_out[((size_t)b * h + q) * L + t] = o[j];                                   // :216
              }
          }
        }
        const int c_in = a.first ? 8 : a.c_in;
        const int np = a.last ? c_in : a.n_peel;                                                   // :201-203, :220
#pragma unroll
        for (int e = 0; e < kMaxGroup; ++e)
          if (e < np) a.z_out[((size_t)b * 8 + a.z_out_ch0 + e) * L + t] = z[e];
        if (!a.last) {
          const int c = a.c_next;                                                                  // = c_in - np
          float zs[kMaxGroup];                                                                     // z[np:]
#pragma unroll
          for (int e = 0; e < kMaxGroup; ++e) {
            float val = 0.0f;
#pragma unroll
            for (int q = 0; q < kMaxGroup; ++q)
              if (q - np == e) val = z[q];
            zs[e] = val;
          }
#pragma unroll
          for (int r = 0; r < kMaxGroup; ++r) {                                                    // :64  W z
            float s = 0.0f;
#pragma unroll
            for (int cc = 0; cc < kMaxGroup; ++cc)
              if (r < c && cc < c) s = fmaf(a.winv[r * c + cc], zs[cc], s);
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
        s_a0[tid] = make_float4(zn[0], zn[1], zn[2], zn[3]);
      }
    }
  }
  if (a.last) return;
  __syncthreads();

  // ---- WN.start of the next flow (model.py:117): x[P] = sum_j Wst[P][j] a0[j] + b[P], fp16, position-major.
  // piece = (chunk cc, row, 8-position group g8) = 16 contiguous bytes; idx = cc*2048 + row*8 + g8, so
  // consecutive threads write consecutive bytes and a thread's g8 is fixed: its 8x4 weights sit in registers.
  const int C = a.C, h = a.h_next;
  const int g8 = tid & 7;
  for (int cc = 0; cc < C / 64; ++cc) {
    const int P0 = cc * 64 + g8 * 8;
    float w[8][4], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bs[e] = a.bstart[P0 + e];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[e][j] = (j < h) ? a.wstart[(P0 + e) * h + j] : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < FL_ROWS * 8 / FL_ROWS; ++it) {
      const int rl = it * (FL_ROWS / 8) + (tid >> 3);
      const size_t row = row0 + rl;
      if (row >= nrows) continue;
      const float4 a0 = s_a0[rl];
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = (_Float16)fmaf(w[e][3], a0.w, fmaf(w[e][2], a0.z, fmaf(w[e][1], a0.y, fmaf(w[e][0], a0.x, bs[e]))));
      const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
      const size_t prow = (size_t)b * a.g.Lp + a.g.G + t;
      *(half8*)(a.x + ((size_t)cc * a.g.R + prow) * 64 + g8 * 8) = o;
    }
  }
}

hipError_t launch_flow(const FlowArgs& a, hipStream_t s) {
  const size_t nrows = (size_t)a.g.B * a.g.L;
  const unsigned grid = (unsigned)((nrows + FL_ROWS - 1) / FL_ROWS);
  hipLaunchKernelGGL(flow_kernel, dim3(grid), dim3(FL_ROWS), 0, s, a);
  return hipGetLastError();
}

// =============================================================================================
// WaveGlowLoss (src/waveglow/train.py:31-45): sum z^2 and sum log_s on the device (fp64 accumulation: block tree +
// one atomic per block), then loss = (sum z^2 / (2 sigma^2) - sum log_s - sum log_det_W) / (B * 8 * L).
// =============================================================================================
template <bool SQUARE>
__global__ void __launch_bounds__(256) reduce_sum_kernel(const float* __restrict__ x, size_t n, double* acc) {
  double s = 0.0;
  const size_t n4 = n / 4;
  const float4* x4 = (const float4*)x;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = x4[i];
    if (SQUARE) s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    else s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[n4 * 4 + threadIdx.x];
    s += SQUARE ? (double)v * v : (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ void loss_final_kernel(const double* acc, double log_det_total, float sigma, double denom, float* out) {
  *out = (float)((acc[0] / (2.0 * (double)sigma * (double)sigma) - acc[1] - log_det_total) / denom);
}

hipError_t launch_reduce_sum(const float* x, size_t n, int square, double* acc, hipStream_t s) {
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  if (square) hipLaunchKernelGGL(reduce_sum_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, n, acc);
  else hipLaunchKernelGGL(reduce_sum_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, n, acc);
  return hipGetLastError();
}
hipError_t launch_loss_final(const double* acc, double log_det_total, float sigma, double denom, float* out, hipStream_t s) {
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, s, acc, log_det_total, sigma, denom, out);
  return hipGetLastError();
}

}  // namespace wg
