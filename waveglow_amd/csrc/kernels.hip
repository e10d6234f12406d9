// gfx950 (CDNA4 / MI355X) kernels of the WaveGlow hot path (inference direction and the no-grad forward).
//
//   mel_pack_kernel   mel [B,M,T] -> frame-major fp16 rows, the B operand of the folded conditioning K-steps
//                     (reference: src/waveglow/model.py:145-150, :225-232 / :186-193 upsample + squeeze, folded away)
//   cond_fold_kernel  load time: cond_layer o upsample -> one [2C x 4M] matrix per layer and phase, in A-fragment order
//   flow_kernel       affine coupling (inverse or forward) + 1x1 mix + early      (model.py:247-271 / :200-218) fused
//                     noise concat / early output, then the next WN's start conv   with the NEXT flow's WN.start
//                     (x_0 planes and the a0 plane of the folded first layer)      (model.py:117)
//   wn_layer_kernel   one WN layer: dilated conv + cond slice as ONE K-extended    (model.py:123-135, :13-20, :137)
//                     MFMA GEMM, gate in registers, res GEMM + folded end*skip GEMM from LDS
//   reduce_sum_kernel / loss_final_kernel   WaveGlowLoss (train.py:31-45)
//
// Data layout (see wg_common.h): phase-major fp16 planes [chunk][row][64 ch] so that every GEMM K-step's B tile is BN
// contiguous 128-byte rows, fetched HBM/L2 -> LDS by global_load_lds (LDS-DMA), XOR-swizzled through the SOURCE
// address (LDS destination is lane-linear).
#include "wg_common.h"

#include <cstdlib>
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

// Training forward: the backward pass needs tanh and sigmoid themselves (wn_layer_kernel MODE 3), so all three come out of
// the same three transcendentals:  r = 1 / ((E1 + 1)(1 + E2)),  acts = (E1 - 1) r,  sigmoid = (E1 + 1) r,
// tanh = acts (1 + E2).  Both exponents are clamped to +-60 so that every factor stays finite (E2 = inf would make the
// last product 0 * inf).
__device__ __forceinline__ void gate_act3(float u, float v, float& th, float& sg, float& ac) {
  const float e1 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(u, -60.0f, 60.0f));
  const float t = 1.0f + __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(v, -60.0f, 60.0f));
  const float r = __builtin_amdgcn_rcpf(fmaf(e1, t, t));
  ac = fmaf(e1, r, -r);
  sg = fmaf(e1, r, r);
  th = ac * t;
}

// ---- hand-counted VMEM in the GEMM main loop (cdna_hip_programming.md 5.7): hipcc drains an LDS-DMA before
// any later LDS read and sinks register loads next to their use; both serialise the K loop on memory latency.
// These loads are invisible to the compiler's s_waitcnt bookkeeping; every consumer sits behind wait_vm0*.
// LDS-DMA: 64 lanes x 16 B from (sbase + voff) to LDS [lds_addr + lane*16] (M0 saved/restored in-statement).
// Wait states INSIDE the strings (hipcc pads nothing for inline asm): an SGPR base that a VALU instruction wrote
// just before -- v_readlane of a spilled SGPR, v_readfirstlane -- needs 5 wait states before a VMEM instruction
// reads it (observed: memory access fault from a stale base right after an SGPR-spill reload); M0 needs 1.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload16(half8& dst, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
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
// (The next tile's A fragments are NOT prefetched under the epilogue: between such an inline-asm load and its
// wait lies a long stretch of compiler-scheduled code, and hipcc moved the still-in-flight destination registers
// there -- wrong results.  They are loaded at the tile top: ~1 L2 latency exposed per tile.)
constexpr int kTilesPerWG = 2;        // tiles per workgroup (template TPW), fully unrolled; 1 for small workloads
#ifdef WG_NO_PIPE_EPI
constexpr bool kPipeEpi = false;      // A/B builds: the round-1 epilogue (gate, barrier, GEMM 2, end x skip, stores in sequence)
#else
constexpr bool kPipeEpi = true;       // gate of column chunk c overlapped with GEMM 2 of chunk c-1 (see "pipelined epilogue")
#endif
// Tried and measured slower on MI355X, kept out of the source: staging B tiles global -> VGPR -> ds_write instead
// of LDS-DMA (K loop 59.3k vs 57.6k cycles per tile); offsetting the VMEM slots of the two waves of a SIMD
// (two copies of the loop made hipcc spill).

// Alternating issue priority in the K loop.  The SIMD arbiter favours the older of its two waves: one wave runs a whole
// K-step ahead of its partner, then idles at the barrier while the partner finishes alone with its load-issue stalls
// uncovered (with the barrier removed, wave 0 is through the K loop in 28 k cycles and its partner in 44 k).  With kPrio the
// two waves of a SIMD take turns at s_setprio 1: waves 4-7 during the deferred sub-step and sub-step 0, waves 0-3 during
// sub-steps 1 and 2.  Same-box A/B: +0.4 ... +1.2 % on the inference kernel (box dependent), -0.4 ms on the training
// forward.  Measured and dropped: the other phase (-1.1 %), either group always ahead (-0.1 / -0.3 %), a 1/4 : 3/4 split
// (-0.4 %); switches in front of sub-step 2 (quarters, 3/4 : 1/4) and any switch in the first-layer variant make hipcc spill.
#ifdef WG_NO_PRIO
constexpr bool kPrio = false;         // A/B builds
#else
constexpr bool kPrio = true;
#endif
// (Also measured and dropped, round 2: waves 4-7 sleeping 64-192 cycles after every K-loop barrier to de-phase the two
// groups' loads: -0.5 %; the GEMM-2 weight fragments fetched inside the last K-step instead of phase 0 of the epilogue:
// phase 0 -1.2 k cycles, K loop +0.9 k, +-0.1 % end to end; both waves' loads moved to different slots by branches inside
// the load statements: the "+v" ties make hipcc copy fragment registers, and the K loop spills.)
// q[2] of a K-step is fetched in the load-free slots of the step's own deferred sub-step (behind the DMA pieces) instead of
// as two loads back to back at the end of the step before, where nothing covers their issue; the step then ends with the
// counted wait and the barrier alone.  Same-box A/B +0.25 % (0.6143 -> 0.6158), training neutral.
#ifdef WG_NO_TILE_INTERLEAVE
constexpr bool kTileInterleave = false;   // A/B builds
#else
constexpr bool kTileInterleave = true;
#endif
#ifdef WG_NO_Q2_LATE
constexpr bool kQ2Late = false;       // A/B builds
#else
constexpr bool kQ2Late = true;
#endif
// DEEP rings: quarter 2 of step ks+1 is fetched in the last slot of step ks's deferred sub-step (its ring slot was freed in
// sub-step 2 of step ks-1) instead of as two uncovered loads at the end of step ks-1 (-0.3 % on configs[0], same-box).
#ifdef WG_NO_DEEP_Q2
constexpr bool kDeepQ2 = false;       // A/B builds
#else
constexpr bool kDeepQ2 = true;
#endif
// The folded end x skip weight fragments (hi + lo rows, C/32 KiB) are the same for every wave and every tile of a launch:
// they are staged in LDS once per workgroup and read from there right before their MFMAs, instead of C/32 global 1-KiB
// loads per wave and tile (8 x redundant through the vector memory path).  Same-box A/B: fetching them one per two slots
// of the second-to-last phase instead of as a burst gave +0.5 %, the LDS copy another +0.25 % (0.6180 -> 0.6195).
// (The priority alternation in the backward dgrad variants: no measurable change, left out.)
#ifdef WG_NO_WES_LDS
constexpr bool kWesLds = false;       // A/B builds: global loads at the top of the second-to-last phase
#else
constexpr bool kWesLds = true;
#endif
#ifdef WG_NO_DEEP
constexpr bool kDeep = false;         // A/B builds: the one-step ring for small workloads too
#else
constexpr bool kDeep = true;
#endif

// 16x16x32 MFMA of the M16 K loop.  (Measured: the same instruction as an asm statement accumulating in place -- hipcc
// cannot then give a result a fresh destination tuple -- -0.16 % same-box, and hipcc no longer sees the MFMA hazards.)
__device__ __forceinline__ void mfma16_acc(f32x4& c, const half8& a, const half8& b) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

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
// CX = 64-channel chunks per dilated tap of GEMM 1: C/64 normally; 1 for the FIRST layer of a WN, whose input
// x_0 = start(a0) (model.py:117) is itself linear in the <= 4 coupling channels a0, so in_layers[0] o start collapses
// to a [2C x (3 taps x 5)] matrix on the rows of the a0 plane (a0 | 1) that flow_kernel writes INSTEAD of x_0:
// ONE gathered K-step (NTAPS = 1, see A0G) instead of 3C/64 (the bias of start rides on the constant-1 channel, which is 0
// in the guard rows, so the convolution's zero padding of x_0 stays exact); the residual input x_0 itself is rebuilt from
// the same plane by one MFMA step of the epilogue (RES_A0).
// TR = training forward (model.py:178-221 under autograd): the conditioning K-steps read the upsampled spectrogram
// planes a.sp (weights change every optimiser step, so the per-phase cond_layer o upsample fold would have to be rebuilt
// every step), and the gate also writes tanh, sigmoid and acts as fp16 planes for the backward pass (train.hip).
// MODE 2 / 3 = the two dgrad GEMMs of the backward pass on the same K loop (train_api.cpp): plain row blocks instead of
// (tanh, sigmoid) pairs, no bias, and a register epilogue instead of gate / GEMM 2 / end x skip:
//   2  d x_i = d x_{i+1} + sum_tap W_in[tap]^T d pre(t -+ d)        (NTAPS = 3 taps over the 2C d-pre planes, no cond part)
//   3  d acts = W_res^T d x_{i+1} + (W_end W_skip)^T d out, then the gate derivative -> d pre   (NTAPS = 1: the d x planes,
//      "cond" part = the d out plane; the last layer of a flow has no d x: one K-step on the d out plane alone)
// DEEP = two-step-deep weight prefetch for small workloads (one 64-column tile per CU: every CU streams the layer's whole
// 1.1 MB of A fragments from L2 for 64 columns, and a K-step is only 1024 MFMA cycles per SIMD, so the ring's prefetch
// distance of 3/4 step -- ~400 ns -- no longer covers the L2 latency: measured 2086 cycles per K-step, 1196 with the A loads
// removed).  Two rings Q[parity of the step][quarter]; the quarter consumed in sub-step g of step ks is reloaded with the
// fragment of step ks + 2.  Every fragment a step needs was issued more than a step earlier, so the step's only wait is
// the counted one in front of its barrier.  The steps are written out for nK = NKX + 5 (80 mel channels).
template <int C, int NW, int BN, bool HAS_RES, int TPW, int CX, int MODE = 0, int NTAPS = 3, bool HAS_COND = true, bool DEEP = false, bool M16 = false>
__global__ void __launch_bounds__(NW * 64) wn_layer_kernel(const WnLayerArgs a) {
  constexpr bool TR = MODE == 1;         // training forward
  constexpr bool PLAIN = MODE >= 2;      // backward dgrad GEMMs
  constexpr int MB = C / (32 * NW);      // 32-channel blocks per wave
  constexpr int MT = PLAIN ? MB : 2 * MB;   // M tiles per wave: MB tanh blocks, then MB sigmoid blocks (PLAIN: MB row blocks)
  constexpr int NTHREADS = NW * 64;
  constexpr int NT = BN / 32;            // 32-column MFMA tiles per wave
  constexpr int CC = CX;                 // 64-channel chunks per tap of the GEMM-1 B operand (a.x_tap)
  constexpr int NKX = NTAPS * CC;        // K-steps of the dilated taps
  constexpr int BT_BYTES = BN * 128;     // one staged B tile: BN rows x 64 fp16
  constexpr int ACT_ROW = 2 * C + 16;    // bytes per acts row: +16 B pad => conflict-free b128 reads/writes with
                                         // immediate-offset addressing (one base VGPR per 32-column tile)
  constexpr int K2 = C / 16;             // k16 steps of GEMM2
  constexpr int NG = BN * 8 / NTHREADS;  // LDS-DMA instructions per wave per B tile
  constexpr int NAH = MT * 2;            // A fragments per half K-step (packing unit)
  constexpr bool DEFER = (MB == 1);      // defer a step's last sub-step past the barrier (needs spare registers)
  static_assert(BN * 8 % NTHREADS == 0 && MB >= 1 && MB <= 2 && NKX >= 1 && (NTAPS == 1 || NTAPS == 3), "tile geometry");
  constexpr bool Q2L = !M16 && kQ2Late && DEFER && !DEEP && (NG + NT - 1) / NT == 1 && NT >= NG + MT;   // free slots behind the DMA pieces
  // M16: GEMM 1 on 16x16x32 MFMAs (same tile, same loads, same FLOPs per K-step; the matrix pipe holds a higher clock on
  // them -- DESIGN.md section 3).  A K-step is two 32-deep sub-steps s; quarter g = (s = g >> 1, column half g & 1) runs
  // the wave's four 16-row tiles m (tanh rows 0-15, 16-31, sigmoid rows 0-15, 16-31 of its channel block) against the
  // half's NT 16-column tiles.  Accumulator acc[mt][c][4 v + r], v = 2 (m & 1) + (column tile & 1): lane (j = lane & 15,
  // g4 = lane >> 4) holds channel 16 (m & 1) + 4 g4 + r of column 32 c + 16 (tile & 1) + j.
  static_assert(!M16 || (MODE == 0 && MB == 1 && DEFER && !DEEP && NT >= 2), "16x16x32 variant");
  // A0G: first layer of a WN with the start fold, ONE tap K-step instead of three: the a0 plane row carries 5 live values
  // (a0 | 1), so the three taps' rows fit side by side in one 64-wide B tile -- 16-byte chunk t of tile row n = chunk 0 of
  // the plane row of tap t (the LDS-DMA takes a per-lane source: stage_B_piece), chunks 3-7 from the rows' own zero tails --
  // against in_layers[0] o start packed as [tap][8] along K (api.cpp wA1fx).  8 -> 6 K-steps for that launch.
  constexpr bool A0G = MODE == 0 && CX == 1 && NTAPS == 1;
  static_assert(!DEEP || (MODE == 0 && DEFER && HAS_COND && MT <= NT && NG <= NT && ((NKX >= 4 && NKX % 2 == 0) || NKX == 1)), "deep prefetch variant");
  static_assert(MODE == 0 || (TPW == 1 && CX == ((MODE == 2 || MODE == 4) ? 2 : 1) * (C / 64)) || (MODE == 3 && CX == 1), "training variants");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem;                     // 2 x BT_BYTES
  char* const sActs = MODE == 4 ? smem : smem + 2 * BT_BYTES;   // BN x ACT_ROW (MODE 4: over the B tiles, after the K loop --
                                                                  // 66 KB per workgroup keeps two workgroups on a CU)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Lane-derived values are re-derived from an opaque copy of tid at the top of every tile (set_lane_ids), so
  // hipcc cannot share address arithmetic between the unrolled tile bodies and keep it live across a whole tile.
  int lane = tid & 63, ln = lane & 31, lh = lane >> 5;

  // LDS-DMA of one B tile: piece idx = row*8 + physical 16-B chunk; logical chunk = phys ^ ((row>>1)&7)
  // (the LDS destination is lane-linear, so the bank swizzle is applied to the SOURCE address).
  unsigned pvoff[NG];     // x planes: byte offset of this lane's piece inside a tile of BN contiguous rows
  int pchunk[NG];         // logical 16-byte chunk (0..7) this lane fetches
  int swB;
  int l15 = lane & 15, l4 = lane >> 4;   // M16: column / K group of the 16x16x32 operands
  unsigned a_voff;
  auto set_lane_ids = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63;
    ln = lane & 31;
    lh = lane >> 5;
    swB = M16 ? (lane >> 1) & 7 : (ln >> 1) & 7;
    l15 = lane & 15;
    l4 = lane >> 4;
    a_voff = lane * 16;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int idx = i * NTHREADS + t;
      const int row = idx >> 3, pc = idx & 7;
      pchunk[i] = pc ^ ((row >> 1) & 7);
      pvoff[i] = row * 128 + (pchunk[i] << 4);
    }
  };
  set_lane_ids();
  const unsigned sB_addr = (unsigned)(size_t)WG_LPTR(sB);
  const int R = a.g.R, Rp = a.g.Rp, Fp = a.g.Fp, M = a.M;
  const int nK = NKX + (HAS_COND ? a.n_cond_steps : 0);
  const int mel_rows_per_utt = a.g.T + 6;

  // Tile = (phase p, BN consecutive rows of that phase's block).  Row of column n: kRowPad + p*Rp + jt*BN + n.
  // B-tile source of tap K-step ks (scalar): tap offset delta = (tap-1)*dil group-timesteps moves phase p to
  // (p+delta)&31 and the frame by (p+delta)>>5 -- again BN contiguous rows (model.py:98-102: padding = dilation).
  auto xstep_src = [&](int p, int jt, int ks) -> const char* {
    const int tap = ks / CC, cc = ks - tap * CC;
    const int pp = p + (tap - NTAPS / 2) * a.dil;
    const int row = kRowPad + (pp & 31) * Rp + a.row0 + jt * BN + (pp >> 5);
    return (const char*)(a.x_tap + ((size_t)cc * R + row) * 64);
  };
  // training forward: conditioning K-step s = chunk s of the spectrogram planes, same rows as the centre tap
  auto sp_src = [&](int p, int jt, int s_) -> const char* {
    const int row = kRowPad + p * Rp + a.row0 + jt * BN;
    return (const char*)(a.sp + ((size_t)s_ * R + row) * 64);
  };
  // Conditioning K-step s (folded cond_layer o upsample, K = 4 taps x M mel channels): column n needs the mel
  // frames q-j, j = 0..3; k index = j*M + i.  Per-lane gather from the frame-major mel (mrow = melT row of the
  // column's frame; 3 for columns outside any utterance: melT rows 0..3 are zero and mrow - j stays in bounds).
  int mrow[NG];
  auto cond_voff = [&](int s, int i) -> unsigned {
    const int k0 = s * 64 + pchunk[i] * 8;
    const int j = (k0 >= M) + (k0 >= 2 * M) + (k0 >= 3 * M);
    return (unsigned)(((mrow[i] - j) * M + (k0 - j * M)) * 2);
  };
  auto stage_B_piece = [&](int p, int jt, int ks, int bufsel, int i) {
#ifndef WG_DBG_NO_DMA
    const unsigned lds = __builtin_amdgcn_readfirstlane(sB_addr + bufsel * BT_BYTES + (i * NTHREADS + wave * 64) * 16);
    if constexpr (A0G) {
      if (ks < NKX) {
        const int row = (i * NTHREADS + tid) >> 3, lc = pchunk[i];
        const int pp = p + ((lc < 3 ? lc : 1) - 1) * a.dil;
        const unsigned prow = (unsigned)(kRowPad + (pp & 31) * Rp + a.row0 + jt * BN + (pp >> 5) + row);
        glds16(a.x_tap, prow * 128u + (lc < 3 ? 0u : (unsigned)lc * 16u), lds);
        return;
      }
    }
    if (ks < NKX) glds16(xstep_src(p, jt, ks), pvoff[i], lds);
    else if constexpr (MODE != 0) glds16(sp_src(p, jt, ks - NKX), pvoff[i], lds);
    else glds16(a.melT, cond_voff(ks - NKX, i), lds);
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
  // M16: fragment of quarter g, tile nt: columns 16 (NT (g & 1) + nt) + j, K = 32 (g >> 1) + 8 g4 .. + 7
  auto read_B16 = [&](const char* buf, int g, int nt) -> half8 {
    const int n = (NT * (g & 1) + nt) * 16 + l15;
    const int c = (4 * (g >> 1) + l4) ^ swB;
    return *(const half8*)(buf + n * 128 + c * 16);
  };
  // A fragments, packed [half K-step][wave][MT][2 k16][64 lanes][8]: tap steps from wA1, conditioning steps from
  // this tile's phase block of wA1c.  q[g][mt] holds the fragment of k16 sub-step g (0..3) of the current K-step;
  // one fragment = one 1 KiB wave-load straight from L2.
  const char* wA1c_p = nullptr;   // set per tile
  auto load_Aq = [&](int ks, int g, int mt, half8& dst) {
    const char* base = ks < NKX ? (const char*)a.wA1 : wA1c_p;
    const int kl = ks < NKX ? ks : ks - NKX;
    const char* p = base + ((size_t)(2 * kl + (g >> 1)) * NW + wave) * (NAH * 1024) + (mt * 2 + (g & 1)) * 1024;
#ifndef WG_DBG_NO_ALOAD
    gload16<0>(dst, p, a_voff);
#else
    asm volatile("" : "=v"(dst));
#endif
  };
  // M16: [half K-step = sub-step s][wave][tile m][64 lanes = (row i, K group)][8]; fragment (s, m) lives in q[2 s + (m >> 1)][m & 1]
  auto load_A16 = [&](int ks, int s_, int m, half8& dst) {
    const char* base = ks < NKX ? (const char*)a.wA1 : wA1c_p;
    const int kl = ks < NKX ? ks : ks - NKX;
#ifndef WG_DBG_NO_ALOAD
    gload16<0>(dst, base + ((size_t)(2 * kl + s_) * NW + wave) * (NAH * 1024) + m * 1024, a_voff);
#else
    asm volatile("" : "=v"(dst));
#endif
  };
  // column -> (utterance, frame): rr = row inside the phase block, one of the BN consecutive rows of the current tile.
  // The utterance of the tile's first row (tile_b0, wave-uniform: one scalar division per tile) is at most one
  // utterance behind every other row's when Fp >= BN, so the per-lane integer division (~25 VALU instructions incl. a
  // transcendental, in phases that are VALU-bound) reduces to a compare and a subtract.
  int tile_b0 = 0;
  const bool fp_ge_bn = Fp >= BN;
  auto column_of = [&](int rr, int p, int& b, int& t) -> bool {
    int rem = rr - tile_b0 * Fp;
    b = tile_b0;
    if (fp_ge_bn) {
      if (rem >= Fp) { rem -= Fp; ++b; }
    } else {
      const int qd = rem / Fp;
      b += qd;
      rem -= qd * Fp;
    }
    const int fq = rem - a.g.Gf;
    t = fq * 32 + p;
    if (!(b < a.g.B && fq >= 0 && fq < a.g.F && t < a.g.L)) return false;
    return a.g.frames == nullptr || t < 32 * a.g.frames[b];
  };

  // ---- tile walk.  Blocks b and b+8 share an XCD (round-robin dispatch; speed only), so XCD label x owns a
  // contiguous run of tiles (= a few phases: their folded conditioning weights stay in that XCD's L2); its blocks
  // take tiles start+idx, start+idx+step, ... so tiles that run at the same time are neighbours.
  // A workgroup processes kTilesPerWG tiles in a FULLY UNROLLED loop: the next tile's first B tile (LDS-DMA) is
  // issued under the current tile's gate / GEMM2 / epilogue, and the workgroup launch cost is paid once per
  // kTilesPerWG tiles.  (A real persistent loop makes hipcc spill 100+ VGPRs around the back edge.)
  int tile, tile_end;
  const int tile_step = gridDim.x >> 3;        // grid is a multiple of 8
  int run_start, run_quarter = 0;              // this XCD's run of tiles; a quarter of it (0: run not divisible by 4)
  {
    const int bid = blockIdx.x, ntl = a.n_tiles;
    const int xcd = bid & 7, idx = bid >> 3;
    const int qq = ntl >> 3, r = ntl & 7;
    const int start = xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq;
    tile_end = start + (xcd < r ? qq + 1 : qq);
    tile = start + idx;
    run_start = start;
    if (kTileInterleave && ((tile_end - start) & 3) == 0) run_quarter = (tile_end - start) >> 2;
  }
  // Walk order inside the run: index i takes tile (i & 3) * quarter + (i >> 2), so the tiles an XCD works on at one time are
  // the same few rows of its (typically four) phases instead of 32 consecutive tiles of one phase -- with every XCD doing the
  // same, a plane row's three uses (centre tap of its own tile, the outer taps of the tiles one dilation away, mostly on other
  // XCDs) fall close together in time instead of a quarter of the launch apart.
  auto tile_of = [&](int t) { return run_quarter ? run_start + ((t - run_start) & 3) * run_quarter + ((t - run_start) >> 2) : t; };

  // (Measured and dropped: issuing the first tile's step-0 A fragments here, ahead of the bias loads, so that a cold
  // launch pays one memory round trip instead of two -- no gain at batch 1, -0.6 % at config 1.)
  half8 q[4][MT];
  int par = 0;                               // LDS buffer of K-step ks is (ks + par) & 1
  if (tile < tile_end) {
    const int t0 = tile_of(tile), p0 = t0 / a.tiles_per_phase;
#pragma unroll
    for (int i = 0; i < NG; ++i) stage_B_piece(p0, t0 - p0 * a.tiles_per_phase, 0, 0, i);
  }
  // bias of GEMM1 (pre-scaled: in_layer bias + cond_layer bias slice + W_cond . upsample bias), fp32 [2C], in LDS
  float* const sBias = (float*)(sActs + BN * ACT_ROW);
  if constexpr (!PLAIN) {
    for (int i = tid; i < 2 * C; i += NTHREADS) sBias[i] = a.bias1[i];
    if constexpr (HAS_RES)
      for (int i = tid; i < C; i += NTHREADS) sBias[2 * C + i] = a.bias2[i];   // b_res, read by the pipelined epilogue
    if constexpr (kWesLds) {
      uint4* const sW = (uint4*)(sBias + 3 * C);                 // [C/32][64 lanes] x 16 B, behind the biases
      for (int i = tid; i < (C / 32) * 64; i += NTHREADS) sW[i] = ((const uint4*)a.wEs)[i];
    }
  }
  __syncthreads();

#pragma unroll
  for (int it = 0; it < TPW; ++it, tile += tile_step) {
    if (tile >= tile_end) break;
    if (it > 0) set_lane_ids();
    const int tile_m = tile_of(tile);
    const int p = tile_m / a.tiles_per_phase;         // phase of every column of this tile
    const int jt = tile_m - p * a.tiles_per_phase;
    const int rr0 = a.row0 + jt * BN;                 // first row inside the phase block
    tile_b0 = rr0 / Fp;
    const int r0 = kRowPad + p * Rp + rr0;            // first plane row of this tile
    const int next_tile = tile + tile_step;
    wA1c_p = (const char*)a.wA1c + (MODE != 0 ? (size_t)0 : (size_t)p * (2 * a.n_cond_steps) * NW * (NAH * 1024));
    if constexpr (!kXTileDMA) {
      if (it > 0) {
#pragma unroll
        for (int i = 0; i < NG; ++i) stage_B_piece(p, jt, 0, par, i);
      }
    }
    half8 Q[DEEP ? 2 : 1][4][MT];               // DEEP: fragment rings of the even / odd K-steps
    if constexpr (DEEP) {
      // issue order A(0, 0..3), A(1, 0..2): the steady state's (see kstep_d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_Aq(0, g, mt, Q[0][g][mt]);
#pragma unroll
      for (int g = 0; g < (kDeepQ2 ? 2 : 3); ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_Aq(1, g, mt, Q[DEEP ? 1 : 0][g][mt]);
    } else {
#pragma unroll
      for (int g = 0; g < (Q2L || M16 ? 2 : 3); ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (M16) load_A16(0, 0, 2 * g + mt, q[g][mt]);
          else load_Aq(0, g, mt, q[g][mt]);
        }
    }
    // mel rows of the frames this lane gathers for the conditioning K-steps
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        int b, t;
        const int rr = rr0 + ((i * NTHREADS + tid) >> 3);
        const bool ok = column_of(rr, p, b, t);
        mrow[i] = ok ? 3 + b * mel_rows_per_utt + 3 + (t >> 5) : 3;   // 3: rows 0..3 are zero, and mrow - j >= 0
      }
    }
    // Which of this lane's columns (one per N tile: row rr0 + nt*32 + ln) are real columns -- needed by the epilogue's
    // plane stores; evaluated here, where the VALU idles behind the first loads, and carried through the K loop in one
    // register (the epilogue phases are VALU-issue bound: ~8.5 cycles of phase time per instruction and wave).
    unsigned vmask = 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int cb, ct;
      vmask |= column_of(rr0 + nt * 32 + ln, p, cb, ct) ? (1u << nt) : 0u;
    }
    WG_STAMP(0);
    // ---- GEMM1 accumulators start from the bias
    f32x16 acc[MT][NT];
    f32x4 acc4[M16 ? MT : 1][M16 ? NT : 1][4];     // M16: the same accumulators as four independent 4-register tuples
    auto acc_at = [&](int mt, int nt, int e) -> float {
      if constexpr (M16) return acc4[mt][nt][e >> 2][e & 3];
      else return acc[mt][nt][e];
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x16 v;
      if constexpr (PLAIN) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = 0.0f;
      } else {
        const int row0 = (mt < MB ? 0 : C) + (wave * MB + (mt < MB ? mt : mt - MB)) * 32;
        const float4* bp = (const float4*)(sBias + row0 + 4 * (M16 ? l4 : lh));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 q4 = bp[M16 ? 4 * (g >> 1) : 2 * g];
          v[4 * g] = q4.x; v[4 * g + 1] = q4.y; v[4 * g + 2] = q4.z; v[4 * g + 3] = q4.w;
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (M16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc4[mt][nt][e] = f32x4{v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]};
        } else {
          acc[mt][nt] = v;
        }
      }
    }

    constexpr bool PIPE = !PLAIN && kPipeEpi && HAS_RES && MB == 1;     // pipelined epilogue (below); else the sequential one
    constexpr bool RES_A0 = A0G && wn_res_a0(C);                        // first layer of a WN: x_0 rebuilt from the a0 plane
    half8 a2r[PIPE ? K2 : 1];                                 // GEMM-2 weight fragments of this wave (pipelined epilogue)
    // ---- K loop (GEMM 1).  One K-step = 4 k16 sub-steps g = 0..3, each MT*NT MFMAs on fragments q[g][.]
    // (weights) x bf[g&1][.] (activations, read from LDS one sub-step ahead).  Every VMEM / LDS instruction is
    // placed by hand BETWEEN MFMAs (one "slot" after each column tile's MFMAs; sched_barrier pins the order): an
    // LDS-DMA or 1-KiB load costs its wave ~60-180 issue cycles, and the two waves of a SIMD run this loop in
    // lockstep (one barrier per K-step), so clustered loads leave the matrix pipe idle in both at once.
    //   after the barrier : read bf[0] <- sub-step 0 fragments
    //   D  (deferred g=3 of the previous step; operands already in registers, covers the LDS latency above)
    //        slots: LDS-DMA pieces of tile ks+1, then (Q2L) q[2] <- A(ks, 2) of THIS step
    //   g=0  slots: read bf[1] <- sub-step 1 ; reload q[3] <- A(ks, 3)
    //   g=1  slots: read bf[0] <- sub-step 2 ; reload q[0] <- A(ks+1, 0)       (wait q[1] first)
    //   g=2  slots: read bf[1] <- sub-step 3 ; reload q[1] <- A(ks+1, 1)       (wait q[2] first)
    //   then vmcnt(MT): DMA, q[3] and q[0] landed; lgkmcnt(0); ONE s_barrier.
    //   (without Q2L -- tiles with no free slot behind the DMA pieces: q[2] <- A(ks+1, 2) here, vmcnt(2*MT))
    // VMEM issue order per step: DMA xNG, q2 xMT, q3 xMT, q0 xMT, q1 xMT -- every wait is a counted vmcnt.
    // The first and last steps are peeled and "next step is a conditioning step" is a compile-time flag, so the
    // loop bodies are branch-free.
    wait_vm<DEEP ? (kDeepQ2 ? 3 : 4) * MT : 0>();   // DEEP: A(0, 3) and A(1, .) may still be in flight (steady-state invariant)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WG_STAMP(1);
    constexpr int LPS = (MT + NT - 1) / NT;      // A-fragment reloads per slot
    constexpr int GPS = (NG + NT - 1) / NT;      // LDS-DMA pieces per slot
    half8 bf[2][NT];
    // kPrio (see the knob): waves 4-7 hold the raised priority through the deferred sub-step and sub-step 0, waves 0-3
    // through sub-steps 1 and 2; the switch sits inside the last slot of the sub-step before.  The training forward and the
    // 512-channel kernel (+0.8 % at configs[2]) have it too; the first-layer variants (CX = 1) spill with it and go without.
    constexpr bool PRIO = kPrio && NW == 8 && ((BN == 128 && ((MODE == 0 && CX == C / 64) || MODE == 1)) || (C == 512 && MODE == 0 && CX == C / 64));
    auto prio_set = [&](int hi_grp) {
      if (hi_grp == 1)
        asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 .Lpa%=\n\ts_setprio 1\n\ts_branch .Lpb%=\n.Lpa%=:\n\ts_setprio 0\n.Lpb%=:" :: "s"(wave) : "memory", "scc");
      else
        asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 .Lpa%=\n\ts_setprio 0\n\ts_branch .Lpb%=\n.Lpa%=:\n\ts_setprio 1\n.Lpb%=:" :: "s"(wave) : "memory", "scc");
    };
    auto mfma_col = [&](int g, int nt) {
      if constexpr (M16) {
        const int t16 = NT * (g & 1) + nt;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          mfma16_acc(acc4[m >> 1][t16 >> 1][2 * (m & 1) + (t16 & 1)], q[2 * (g >> 1) + (m >> 1)][m & 1], bf[g & 1][nt]);
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#ifdef WG_DBG_MFMA16   // timing experiment only (results are garbage): the 32x32x16 loop's loads and schedule, two 16x16x32 MFMAs each
          f32x4 lo = __builtin_shufflevector(acc[mt][nt], acc[mt][nt], 0, 1, 2, 3);
          f32x4 hi = __builtin_shufflevector(acc[mt][nt], acc[mt][nt], 4, 5, 6, 7);
          lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(q[g][mt], bf[g & 1][nt], lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(q[g][mt], bf[g & 1][nt], hi, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[mt][nt][j] = lo[j]; acc[mt][nt][4 + j] = hi[j]; }
#else
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[g][mt], bf[g & 1][nt], acc[mt][nt], 0, 0, 0);
#endif
        }
      }
    };
    // more: step ks+1 exists; ncond: step ks+1 is a conditioning step; first: no deferred MFMAs pending
    auto kstep = [&](auto more_tag, auto ncond_tag, auto first_tag, int ks) {
      constexpr bool more = decltype(more_tag)::value;
      constexpr bool ncond = decltype(ncond_tag)::value;
      constexpr bool first = decltype(first_tag)::value;
      const char* buf = sB + ((ks + par) & 1) * BT_BYTES;
      const char* src_next = nullptr;
      if constexpr (more && !ncond) src_next = xstep_src(p, jt, ks + 1);
      if constexpr (more && ncond && MODE != 0) src_next = sp_src(p, jt, ks + 1 - NKX);
      const unsigned lds_next = sB_addr + ((ks + 1 + par) & 1) * BT_BYTES + wave * 1024;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[0][nt] = M16 ? read_B16(buf, 0, nt) : read_B(buf, nt, 0);
      __builtin_amdgcn_sched_barrier(0);
      auto dma_slot = [&](int nt) {
        if constexpr (more) {
#pragma unroll
          for (int i = nt * GPS; i < (nt + 1) * GPS && i < NG; ++i) {
#ifndef WG_DBG_NO_DMA
            const unsigned lds = __builtin_amdgcn_readfirstlane(lds_next + i * NTHREADS * 16);
            if constexpr (ncond && MODE == 0) glds16(a.melT, cond_voff(ks + 1 - NKX, i), lds);
            else glds16(src_next, pvoff[i], lds);
#endif
          }
        }
      };
      if constexpr (PRIO) {
        prio_set(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DEFER) {
        // ---- D: deferred sub-step 3 of step ks-1 + DMA of tile ks+1
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (!first) mfma_col(3, nt);
          __builtin_amdgcn_sched_barrier(0);
          dma_slot(nt);
          if constexpr (Q2L) {
            if (nt >= NG && nt - NG < MT) load_Aq(ks, 2, nt - NG, q[2][nt - NG]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int g = 0; g < (DEFER ? 3 : 4); ++g) {
        if constexpr (M16) {
          // VMEM issue order per step: DMA x NG (D slots), A(ks, sub-step 1) x 4 (quarter 0 slots), A(ks+1, sub-step 0) x 4
          // (quarter 2 slots).  Sub-step 0's fragments are used by quarters 0 and 1, sub-step 1's by quarter 2 and D.
          // (Measured and dropped: quarter 1 and D fragment-major -- slot m = tile m against all columns -- so that fragment
          // m is dead after slot m and its successor goes out two quarters before its first use: -0.6 % same-box.)
          if (g == 0) wait_vm<more ? NG : 0>();                  // A(ks, sub-step 0) landed
          if (g == 2) wait_vm<0>();                              // A(ks, sub-step 1) landed (and the B tile of step ks+1)
        } else {
        if (g == 1) wait_vm<more ? 2 * MT + NG : 2 * MT>();      // q[1] landed
        if (g == 2) wait_vm<Q2L ? (more ? 2 * MT : MT) : (more ? NG + 2 * MT : MT)>();          // q[2] landed
        if (g == 3) wait_vm<more ? 2 * MT : 0>();                // q[3] landed (no deferral)
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          mfma_col(g, nt);
          __builtin_amdgcn_sched_barrier(0);
          if (g < 3) bf[(g + 1) & 1][nt] = M16 ? read_B16(buf, g + 1, nt) : read_B(buf, nt, g + 1);
          if (!DEFER && g == 0) dma_slot(nt);
          if constexpr (M16) {
            constexpr int LP16 = (4 + NT - 1) / NT;
            if (g == 0 || (g == 2 && more)) {
#pragma unroll
              for (int m = nt * LP16; m < (nt + 1) * LP16 && m < 4; ++m)
                load_A16(g == 0 ? ks : ks + 1, g == 0 ? 1 : 0, m, q[(g == 0 ? 2 : 0) + (m >> 1)][m & 1]);
            }
          } else if (g == 0 || more) {
#pragma unroll
            for (int mt = nt * LPS; mt < (nt + 1) * LPS && mt < MT; ++mt)
              load_Aq(g == 0 ? ks : ks + 1, (g + 3) & 3, mt, q[(g + 3) & 3][mt]);
          }
          if constexpr (PRIO) {
            if (g == 0 && nt == NT - 1) prio_set(0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (more) {
        if constexpr (DEFER && !Q2L && !M16) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) load_Aq(ks + 1, 2, mt, q[2][mt]);
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(M16 ? 4 : Q2L ? MT : 2 * MT) : "memory");   // DMA, q[0] landed; reads done
#ifndef WG_DBG_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---- DEEP K-step.  VMEM issue order per step: DMA(ks+1) x NG, A(ks+1, 2) x MT, A(ks+1, 3) x MT, A(ks+2, 0), A(ks+2, 1);
    // the wait in front of the barrier -- at most A(ks+1, 3), A(ks+2, 0) and A(ks+2, 1) outstanding -- covers the B tile and
    // every fragment step ks+1 uses.  P = parity of ks; sub-step 3 of step ks-1 (ring P^1) is the deferred one.
    auto kstep_d = [&](auto par_tag, auto more_tag, auto more2_tag, auto ncond_tag, auto first_tag, int ks) {
      constexpr int P = DEEP ? decltype(par_tag)::value : 0, PO = DEEP ? 1 - P : 0;
      constexpr bool more = decltype(more_tag)::value, more2 = decltype(more2_tag)::value;
      constexpr bool ncond = decltype(ncond_tag)::value, first = decltype(first_tag)::value;
      const char* buf = sB + ((ks + par) & 1) * BT_BYTES;
      const char* src_next = nullptr;
      if constexpr (more && !ncond) src_next = xstep_src(p, jt, ks + 1);
      const unsigned lds_next = sB_addr + ((ks + 1 + par) & 1) * BT_BYTES + wave * 1024;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[0][nt] = read_B(buf, nt, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (!first) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Q[PO][3][mt], bf[1][nt], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {
          if (nt < NG) {
            const unsigned lds = __builtin_amdgcn_readfirstlane(lds_next + nt * NTHREADS * 16);
            if constexpr (ncond) glds16(a.melT, cond_voff(ks + 1 - NKX, nt), lds);
            else glds16(src_next, pvoff[nt], lds);
          }
          if constexpr (kDeepQ2) {            // quarter 2 of the NEXT step (its ring slot was freed in sub-step 2 of step ks-1)
            if (nt == NT - 1) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) load_Aq(ks + 1, 2, mt, Q[PO][2][mt]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Q[P][g][mt], bf[g & 1][nt], acc[mt][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bf[(g + 1) & 1][nt] = read_B(buf, nt, g + 1);
          if (nt < MT) {
            if (g == 0 && more) load_Aq(ks + 1, 3, nt, Q[PO][3][nt]);
            if (g >= 1 && more2) load_Aq(ks + 2, g - 1, nt, Q[P][g - 1][nt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (more2 && !kDeepQ2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_Aq(ks + 2, 2, mt, Q[P][2][mt]);
      }
      if constexpr (more) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(more2 ? (kDeepQ2 ? 3 : 4) * MT : MT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (DEEP) {
      using T_ = std::true_type;
      using F_ = std::false_type;
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, 1>;
      //        parity more  more2 ncond first
      if constexpr (NKX == 1) {                                         // first layer of a WN: one gathered a0-plane step (A0G)
        kstep_d(P0{}, T_{}, T_{}, T_{}, T_{}, 0);                       // the next B tile is a conditioning tile
        kstep_d(P1{}, T_{}, T_{}, T_{}, F_{}, 1);                       // five conditioning steps (the host checks n_cond_steps)
        kstep_d(P0{}, T_{}, T_{}, T_{}, F_{}, 2);
        kstep_d(P1{}, T_{}, T_{}, T_{}, F_{}, 3);
        kstep_d(P0{}, T_{}, F_{}, T_{}, F_{}, 4);
        kstep_d(P1{}, F_{}, F_{}, F_{}, F_{}, 5);
      } else {
        kstep_d(P0{}, T_{}, T_{}, F_{}, T_{}, 0);
#pragma clang loop unroll(disable)
        for (int ks = 1; ks < NKX - 1; ks += 2) {                       // NKX even: pairs (odd, even)
          kstep_d(P1{}, T_{}, T_{}, F_{}, F_{}, ks);
          kstep_d(P0{}, T_{}, T_{}, F_{}, F_{}, ks + 1);
        }
        kstep_d(P1{}, T_{}, T_{}, T_{}, F_{}, NKX - 1);                 // last tap step: the next B tile is a conditioning tile
        kstep_d(P0{}, T_{}, T_{}, T_{}, F_{}, NKX);                     // five conditioning steps (the host checks n_cond_steps)
        kstep_d(P1{}, T_{}, T_{}, T_{}, F_{}, NKX + 1);
        kstep_d(P0{}, T_{}, T_{}, T_{}, F_{}, NKX + 2);
        kstep_d(P1{}, T_{}, F_{}, T_{}, F_{}, NKX + 3);
        kstep_d(P0{}, F_{}, F_{}, F_{}, F_{}, NKX + 4);
      }
      constexpr int PL = (NKX + 4) & 1;                                 // ring of the last step
      wait_vm<0>();                                                     // A(nK-1, 3)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Q[DEEP ? PL : 0][3][mt], bf[1][nt], acc[mt][nt], 0, 0, 0);
    }
    if constexpr (!DEEP) {
      using T_ = std::true_type;
      using F_ = std::false_type;
      if constexpr (NKX == 1) {
        if constexpr (HAS_COND) kstep(T_{}, T_{}, T_{}, 0);         // the only tap step: next is conditioning
        else kstep(F_{}, F_{}, T_{}, 0);                            // the only step
      } else {
        kstep(T_{}, F_{}, T_{}, 0);                                 // tap steps 0 .. NKX-2: next is a tap step
#pragma clang loop unroll(disable)
        for (int ks = 1; ks < NKX - 1; ++ks) kstep(T_{}, F_{}, F_{}, ks);
        if constexpr (HAS_COND) kstep(T_{}, T_{}, F_{}, NKX - 1);   // last tap step: next is conditioning
        else kstep(F_{}, F_{}, F_{}, NKX - 1);                      // last step
      }
      if constexpr (HAS_COND) {
#pragma clang loop unroll(disable)
        for (int ks = NKX; ks < nK - 1; ++ks) kstep(T_{}, T_{}, F_{}, ks);
        kstep(F_{}, F_{}, F_{}, nK - 1);                            // last conditioning step
      }
    }
    if constexpr (DEFER && !DEEP) {
      wait_vm<0>();                            // q[3] of the last step
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) mfma_col(3, nt);
    }
    if constexpr (PRIO) asm volatile("s_setprio 0" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    par = (par + nK) & 1;
    // Next tile's first B tile goes out now, into the LDS buffer the last step did not use (slow waves may
    // still be reading that one) -- it lands under the phases below.
    if constexpr (kXTileDMA) {
      if (next_tile < tile_end) {
        const int tn = tile_of(next_tile), pn = tn / a.tiles_per_phase;
#pragma unroll
        for (int i = 0; i < NG; ++i) stage_B_piece(pn, tn - pn * a.tiles_per_phase, 0, par, i);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    WG_STAMP(2);
    if constexpr (PLAIN) {
      // ---- backward dgrad epilogues: lane (column n = ln of N tile nt, half h) holds rows (= storage positions)
      // [32 blk + 16 h, +16) of its column, 32 contiguous bytes of every fp16 plane.  Only real columns are written:
      // rows of padding columns stay zero (cleared once per workspace geometry), the weight-gradient kernels sum over them.
      // gate derivative (model.py:17-19): d u = g S (1 - T^2), d v = g T S (1 - S) -> the 2C d-pre planes
      auto gate_derivative = [&](const f32x16& g, const _Float16* tp, const _Float16* sp_, _Float16* dst, size_t off) {
        const half8 t0 = *(const half8*)(tp + off), t1 = *(const half8*)(tp + off + 8);
        const half8 s0 = *(const half8*)(sp_ + off), s1 = *(const half8*)(sp_ + off + 8);
        half8 o0, o1, p0, p1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float ta = (float)t0[r], sa = (float)s0[r], ga = g[r];
          const float tb = (float)t1[r], sb = (float)s1[r], gb = g[8 + r];
          o0[r] = (_Float16)(ga * sa * (1.0f - ta * ta));
          o1[r] = (_Float16)(gb * sb * (1.0f - tb * tb));
          p0[r] = (_Float16)(ga * ta * sa * (1.0f - sa));
          p1[r] = (_Float16)(gb * tb * sb * (1.0f - sb));
        }
        const size_t off_s = off + (size_t)(C / 64) * R * 64;      // the sigmoid half: chunks C/64 .. 2C/64-1
        *(half8*)(dst + off) = o0;
        *(half8*)(dst + off + 8) = o1;
        *(half8*)(dst + off_s) = p0;
        *(half8*)(dst + off_s + 8) = p1;
      };
      int lno = ln, lho = lh;
      if constexpr (MODE == 4) {
        asm volatile("" : "+v"(lno), "+v"(lho));
        __syncthreads();            // every wave is through its last B-tile reads: the d x tile goes over the B tiles in LDS
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int cb, ct;
        const bool col_ok = column_of(rr0 + nt * 32 + ln, p, cb, ct);
        if (MODE != 4 && !col_ok) continue;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int blk = wave * MB + mb;
          const size_t off = ((size_t)(blk >> 1) * R + r0 + nt * 32 + ln) * 64 + (blk & 1) * 32 + lh * 16;
          if constexpr (MODE == 2 || MODE == 4) {
            // d x_i = d x_{i+1} (residual path, model.py:131-132) + the three taps' contributions
            half8 i0, i1, o0, o1;
            if (a.in0) { i0 = *(const half8*)(a.in0 + off); i1 = *(const half8*)(a.in0 + off + 8); }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              o0[r] = (_Float16)(acc[mb][nt][r] + (a.in0 ? (float)i0[r] : 0.0f));
              o1[r] = (_Float16)(acc[mb][nt][8 + r] + (a.in0 ? (float)i1[r] : 0.0f));
            }
            if (col_ok) {
              *(half8*)(a.out0 + off) = o0;
              *(half8*)(a.out0 + off + 8) = o1;
            }
            if constexpr (MODE == 4) {
              // ... and into LDS as the B operand of the next GEMM (fp16, position-major rows like the forward's acts tile;
              // padding columns: zeros, as the planes hold them)
              if (!col_ok) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { o0[r] = (_Float16)0.0f; o1[r] = (_Float16)0.0f; }
              }
              char* ap = sActs + lno * ACT_ROW + lho * 32 + nt * 32 * ACT_ROW + blk * 64;
              *(half8*)(ap) = o0;
              *(half8*)(ap + 16) = o1;
            }
          } else {
            gate_derivative(acc[mb][nt], a.in0, a.in1, a.out0, off);
          }
        }
      }
      if constexpr (MODE == 4) {
        // ---- fused: d acts_{i-1} = W_res_{i-1}^T d x_i + (W_end W_skip_{i-1})^T d out  on the tile just computed (it would
        // otherwise be written, and read back by a launch of its own: wn_layer_kernel MODE 3), then the gate derivative
        // -> d pre_{i-1}.  A fragments straight from the K-loop layout of `wat` ([K-step][half][wave][mb][k16 & 1][64][8]):
        // k16 step k of the d x part sits at K-step k >> 2, half (k >> 1) & 1, k & 1; the d out plane's 16 live positions
        // are k16 step 0 of K-step C/64.
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int KX = C / 16;                             // k16 steps over the d x tile
        auto a_frag = [&](int mb, int k) -> half8 {
          const size_t e = ((((size_t)(k >> 2) * 2 + ((k >> 1) & 1)) * NW + wave) * MB + mb) * 2 + (k & 1);
          return *((const half8*)a.wat_prev + e * 64 + lane);
        };
        f32x16 acc2[MB][NT];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[mb][nt][j] = 0.0f;
        constexpr int PF4 = 4;
        half8 af[MB][PF4];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int i = 0; i < PF4; ++i) af[mb][i] = a_frag(mb, i);
        // d out fragments (k16 step 0 of the d out plane): positions 8 lh .. 8 lh + 7 of this lane's columns
        half8 go[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) go[nt] = *(const half8*)(a.gout + ((size_t)r0 + nt * 32 + lno) * 64 + lho * 8);
        const char* const acts_rd = sActs + lno * ACT_ROW + lho * 16;
        half8 bq[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[0][nt] = *(const half8*)(acts_rd + nt * 32 * ACT_ROW);
#pragma unroll
        for (int k = 0; k < KX; ++k) {
          if (k + 1 < KX) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[(k + 1) & 1][nt] = *(const half8*)(acts_rd + nt * 32 * ACT_ROW + (k + 1) * 32);
          }
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const half8 afk = af[mb][k % PF4];
            if (k + PF4 < KX) af[mb][k % PF4] = a_frag(mb, k + PF4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc2[mb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afk, bq[k & 1][nt], acc2[mb][nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const half8 afo = a_frag(mb, 4 * (C / 64));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc2[mb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afo, go[nt], acc2[mb][nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int cb, ct;
          if (!column_of(rr0 + nt * 32 + lno, p, cb, ct)) continue;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int blk = wave * MB + mb;
            const size_t off = ((size_t)(blk >> 1) * R + r0 + nt * 32 + lno) * 64 + (blk & 1) * 32 + lho * 16;
            gate_derivative(acc2[mb][nt], a.t_prev, a.s_prev, a.dpre_prev, off);
          }
        }
      }
    }
    if constexpr (!PLAIN) {
    // Per-tile opaque copies of the lane ids: every address / weight load of the phases below depends on them,
    // so hipcc cannot hoist those (tile-invariant) values out of the tile bodies and spill them
    // (a spill reload is a VMEM op whose compiler-inserted vmcnt(0) would drain the hand-placed prefetches).
    int lno = ln, lho = lh, laneo = lane;
    asm volatile("" : "+v"(lno), "+v"(lho), "+v"(laneo));
    char* const acts_lane = sActs + lno * ACT_ROW + lho * 32;      // this lane's write slot in row n = lno
    const char* const acts_rd = sActs + lno * ACT_ROW + lho * 16;  // B-fragment read base (k16 = 0)
    // ---- folded end x skip (model.py:133-137): out[0:8] += (W_end W_skip_i) acts, 16 columns per group, weights split
    // hi+lo fp16 (rows 0-7 / 8-15 of the 16x16x32 MFMA) so the 8 flow outputs keep ~fp32 weights.  Two parts: the
    // loads (weight fragments, the out rows to update) are issued early, the MFMAs run once every acts row is in LDS.
    const int l15 = laneo & 15, l4 = laneo >> 4;
    // M16: the gate's lane holds 4 consecutive channels (16 mh + 4 g4 + r) of its column per 16-row tile: positions
    // 16 (g4 & 1) + 8 mh + 4 (g4 >> 1) + r of the position-major acts row (wg_common.h chan_to_pos), 8 bytes per write
    char* const acts_lane16 = sActs + l15 * ACT_ROW + (l4 & 1) * 32 + (l4 >> 1) * 8;
    auto write_acts16 = [&](int c, int blk_, int mh, const half8& o) {
      typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
      const half4_t lo = {o[0], o[1], o[2], o[3]}, hi = {o[4], o[5], o[6], o[7]};
      char* ap = acts_lane16 + c * 32 * ACT_ROW + blk_ * 64 + mh * 16;
      *(half4_t*)(ap) = lo;                       // column tile 2 c
      *(half4_t*)(ap + 16 * ACT_ROW) = hi;        // column tile 2 c + 1
    };
    constexpr int NGRP = (BN / 16 + NW - 1) / NW;            // 16-column groups per wave
    constexpr bool kWesEarly = (C / 32) * 4 <= 32;           // folded-end weight fragments fit beside GEMM2's registers
    half8 wes[C / 32];
    float4 es_o[NGRP];
    float4* es_op[NGRP];
    bool es_valid[NGRP];
    auto es_prefetch_w = [&](int s_) {
      if constexpr (kWesLds) wes[s_] = ((const half8*)(sBias + 3 * C) + laneo)[s_ * 64];
      else wes[s_] = ((const half8*)a.wEs + laneo)[s_ * 64];
    };
    auto es_prefetch = [&](bool with_w = true) {
      const half8* pe = (const half8*)a.wEs + laneo;
      if (with_w && !kWesLds) {
#pragma unroll
        for (int s = 0; s < C / 32; ++s) wes[s] = pe[s * 64];
      }
#pragma unroll
      for (int gi = 0; gi < NGRP; ++gi) {
        const int grp = wave + gi * NW;
        es_valid[gi] = false;
        es_o[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
        es_op[gi] = nullptr;
        if (grp < BN / 16) {
          int cb, ct;
          es_valid[gi] = column_of(rr0 + grp * 16 + l15, p, cb, ct) && laneo < 32;
          es_op[gi] = (float4*)(a.out + ((size_t)cb * a.g.L + ct) * 8 + 4 * l4);
          if (es_valid[gi]) es_o[gi] = *es_op[gi];
        }
      }
    };
    auto es_compute = [&]() {
      if constexpr (kWesLds) {
#pragma unroll
        for (int s = 0; s < C / 32; ++s) es_prefetch_w(s);
      }
#pragma unroll
      for (int gi = 0; gi < NGRP; ++gi) {
        const int grp = wave + gi * NW;
        if (grp < BN / 16) {
          const char* ep = sActs + (grp * 16 + l15) * ACT_ROW + l4 * 16;
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < C / 32; ++s) {
            const half8 bfe = *(const half8*)(ep + s * 64);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wes[s], bfe, d, 0, 0, 0);
          }
          // D: col = lane&15, row = 4*(lane>>4)+reg ; rows 8-15 (lanes 32-63) are the lo parts
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] += __shfl_xor(d[r], 32);
          if (es_valid[gi]) {
            float4 o = es_o[gi];
            o.x += d[0]; o.y += d[1]; o.z += d[2]; o.w += d[3];
            *es_op[gi] = o;
          }
        }
      }
    };
    auto read_acts32 = [&](int nt, int k16) -> half8 {       // B fragment for the 32x32x16 MFMA
      return *(const half8*)(acts_rd + nt * 32 * ACT_ROW + k16 * 32);
    };
    constexpr int PF = 8 / MB;                // GEMM2 A-fragment prefetch depth (per 32-channel block)
    half8 xres[RES_A0 ? 1 : MB][RES_A0 ? 1 : NT][2];
    half8 a0f[RES_A0 ? NT : 1], wst[RES_A0 ? MB : 1];     // sequential epilogue, RES_A0: a0-plane row fragments, start weights
    half8 a2[MB][PF];
    const half8* const p2 = (const half8*)a.wA2 + (size_t)wave * MB * K2 * 64 + laneo;
    f32x16 acc2[MB][NT];
    if constexpr (PIPE) {
      // ---- pipelined epilogue.  The gate is VALU-bound (2 exp2 + 1 rcp per element at quarter rate) and GEMM 2 is
      // MFMA-bound; run back to back they leave the matrix pipe idle for the whole gate and the VALU idle for the
      // whole GEMM 2.  Here the tile's NT column chunks (32 columns = one N tile each) go through a software pipeline:
      //     phase c :   gate(chunk c) -> acts rows of chunk c in LDS      ||      GEMM 2 (chunk c-1) -> x_out rows of c-1
      // with ONE barrier between phases (GEMM 2 of a chunk contracts over the channels of all waves).  Inside a phase
      // the two instruction streams are interleaved by hand, one MFMA (+ its LDS fragment read, two slots ahead) and
      // one gate element per slot; the gate itself is software-pipelined over three slots (A: clamp + 2 exp2,
      // B: denominator + rcp, C: numerator, product, fp16 pack) so that no slot waits on a transcendental.
      // The wave's GEMM-2 weight fragments (K2 x 1 KiB) stay in registers for all chunks: the accumulators of the
      // chunks already gated are dead by then, and a chunk's acc2 is only 16 registers.
      constexpr int SL = 16;                                  // slots per phase = gate elements per lane and chunk
      const int blk = wave;                                   // MB == 1
      // a2r: loaded in the slots of phase 0 (first use: phase 1)
      const float* const sBias2 = sBias + 2 * C;              // b_res, fp32 [C], staged next to the GEMM-1 bias
      // Residual add (model.py:131-132) on the matrix pipe: x_out = b_res + P x + W_res acts, where P selects this
      // wave's 32 channels of x -- two more k16 steps whose A fragments are a constant 0/1 matrix (row r = natural
      // channel r of the block, k = storage position chan_to_pos(r)) and whose B fragments are the x rows themselves,
      // loaded straight in B-fragment order.  x (fp16) times 1.0 accumulated in fp32 is exact, and it replaces 32 VALU
      // instructions per chunk (16 cvt + 16 add) in phases that are VALU-bound while the matrix pipe has room.
      // First layer of a WN (RES_A0): x_0 = W_start a0 + b_start was never stored -- ONE k16 step against the a0 plane row
      // (a0 | 1 | 0 0 0, twice) with the start weights as A fragment: lanes 0-31 the fp16 hi parts (k = 0..4), lanes 32-63 the
      // lo parts (k = 8..12), so the sum carries ~fp32 weights and is never rounded to fp16 on its own.
      half8 idA[RES_A0 ? 1 : 2];
      if constexpr (RES_A0) {
        idA[0] = ((const half8*)a.wStA)[blk * 64 + laneo];
      } else {
        const int pr = 16 * ((lno >> 2) & 1) + 4 * (lno >> 3) + (lno & 3);     // chan_to_pos of row r = lno inside a 32-block
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int j = 0; j < 8; ++j) idA[sx][j] = (_Float16)((16 * sx + 8 * lho + j) == pr ? 1.0f : 0.0f);
      }
      half8 xr[RES_A0 ? 1 : 2];                               // residual x of the chunk whose GEMM 2 runs next (B fragments)
      auto load_xr = [&](int nt) {
        if constexpr (RES_A0) {
          xr[0] = *(const half8*)(a.x_tap + ((size_t)r0 + nt * 32 + lno) * 64 + lho * 8);
        } else {
          const size_t row = (size_t)(blk >> 1) * R + r0 + nt * 32 + lno;
          const half8* xp = (const half8*)(a.x_in + row * 64 + (blk & 1) * 32 + lho * 8);
          xr[0] = xp[0];                                        // positions 8h .. 8h+7       (k16 step 0)
          xr[1] = xp[2];                                        // positions 16 + 8h .. +7    (k16 step 1)
        }
      };
      load_xr(0);
      // (Measured and dropped: storing a chunk's x_out rows one phase late, inside the next phase's slots, to keep the wait
      // for the last MFMA out of the phase's tail -- the exec-masked store blocks in the middle of the slots cost 1 %.)
      half8* const xo_base = (half8*)(a.x_out + ((size_t)(blk >> 1) * R + r0 + lno) * 64 + (blk & 1) * 32 + lho * 16);
      auto store_xout_half = [&](const f32x16& d, int nt, int hf) {
        if ((vmask >> nt) & 1u) {
          half8 qv;
#pragma unroll
          for (int r = 0; r < 8; ++r) qv[r] = (_Float16)d[8 * hf + r];
          xo_base[(size_t)nt * 32 * 8 + hf] = qv;               // row + 32 nt: 32 rows x 64 halves = 256 half8
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      // one phase; c is a compile-time constant (a plain unrolled loop over c left acc[..][c] dynamically indexed -- in
      // scratch -- in some instantiations)
      auto phase = [&](auto c_tag) {
        constexpr int c = decltype(c_tag)::value;
        constexpr bool do_gate = c < NT, do_mm = c >= 1;
        f32x16 d2;                                            // GEMM 2 accumulator of chunk c-1: x + b_res + W_res acts
        if (do_mm) {
          const float4* bp = (const float4*)(sBias2 + blk * 32 + 4 * lho);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = bp[2 * g];
            d2[4 * g + 0] = b4.x;
            d2[4 * g + 1] = b4.y;
            d2[4 * g + 2] = b4.z;
            d2[4 * g + 3] = b4.w;
          }
          d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(idA[0], xr[0], d2, 0, 0, 0);
          if constexpr (!RES_A0) d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(idA[1], xr[1], d2, 0, 0, 0);
        }
        if (do_mm && c < NT) load_xr(c);                      // residual of chunk c: phase c+1 starts from it
        if (c == NT - 1) es_prefetch();                       // end x skip out rows (and weights, without kWesLds): consumed in the last phase
        if (c == NT) es_compute();                            // every acts row is in LDS (barrier of phase NT-1)
        // gate pipeline state (static indices after unrolling)
        float e1[SL], den[SL], rc[SL];
        half8 o0, o1, th0, th1, sg0, sg1;                     // th / sg: training forward only (saved tanh, sigmoid)
        auto stageA = [&](int e) {
          const int nt = c < NT ? c : 0;
          const float u = __builtin_amdgcn_fmed3f(acc_at(0, nt, e), -60.0f, 60.0f);
          e1[e] = __builtin_amdgcn_exp2f(u);
          den[e] = __builtin_amdgcn_exp2f(TR ? __builtin_amdgcn_fmed3f(acc_at(MB, nt, e), -60.0f, 60.0f) : acc_at(MB, nt, e));
        };
        auto stageB = [&](int e) {
          const float t = 1.0f + den[e];
          rc[e] = __builtin_amdgcn_rcpf(fmaf(e1[e], t, t));
          if constexpr (TR) den[e] = t;
        };
        auto stageC = [&](int e) {
          // (E1 - 1) * rc as ONE fma (exactly rounded once); kept scalar: v_pk_*_f32 costs more per element than a
          // plain VALU instruction here (tools/ubench/valu_costs.hip)
          float v = fmaf(e1[e], rc[e], -rc[e]);
          asm volatile("" : "+v"(v));
          if (e < 8) o0[e] = (_Float16)v; else o1[e - 8] = (_Float16)v;
          if constexpr (TR) {                                 // gate_act3's sigmoid and tanh from the same factors
            float sg = fmaf(e1[e], rc[e], rc[e]), th = v * den[e];
            asm volatile("" : "+v"(sg), "+v"(th));
            if (e < 8) { sg0[e] = (_Float16)sg; th0[e] = (_Float16)th; } else { sg1[e - 8] = (_Float16)sg; th1[e - 8] = (_Float16)th; }
          }
        };
        bool save_ok = false;                                 // training forward: this lane's column of chunk c is a real column
        size_t save_off = 0;
        if constexpr (TR) {
          if (do_gate) {
            save_ok = (vmask >> c) & 1u;
            save_off = ((size_t)(blk >> 1) * R + r0 + c * 32 + lno) * 64 + (blk & 1) * 32 + lho * 16;
          }
        }
        // K2 <= SL (MB == 1: C <= 256): slot i carries the MFMA of k16 step k when k = i*K2/SL changes at i+1
        static_assert(K2 <= SL, "one GEMM-2 MFMA per slot at most");
        half8 bq[K2];
        const int ntm = c >= 1 ? c - 1 : 0;
        auto k_of = [](int i) { return i * K2 / SL; };
        auto has_k = [&](int i) { return i >= 0 && i < SL && k_of(i) != k_of(i + 1); };
        if (do_gate) { stageA(0); stageA(1); stageB(0); }
        if (do_mm) {
          if (has_k(0)) bq[k_of(0)] = read_acts32(ntm, k_of(0));
          if (has_k(1)) bq[k_of(1)] = read_acts32(ntm, k_of(1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SL; ++i) {
          if (do_mm) {
            if (has_k(i + 2)) bq[k_of(i + 2)] = read_acts32(ntm, k_of(i + 2));   // fragment read two slots ahead
            if (has_k(i)) d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2r[k_of(i)], bq[k_of(i)], d2, 0, 0, 0);
          }
          if (c == 0 && i < K2) a2r[i] = p2[(size_t)i * 64];
          if (do_gate) {
            if (i + 2 < SL) stageA(i + 2);
            if (i + 1 < SL) stageB(i + 1);
            if (i & 1) { stageC(i - 1); stageC(i); }
            if constexpr (M16) {
              if (i == 7) write_acts16(c, blk, 0, o0);
              if (i == SL - 1) write_acts16(c, blk, 1, o1);
            } else {
              if (i == 7) *(half8*)(acts_lane + c * 32 * ACT_ROW + blk * 64) = o0;          // positions [32 blk + 16 h, +8)
              if (i == SL - 1) *(half8*)(acts_lane + c * 32 * ACT_ROW + blk * 64 + 16) = o1;
            }
            if constexpr (TR) {
              // saved activations: rows of padding columns stay zero (cleared once per workspace geometry), the
              // weight-gradient kernels sum over every row of a phase
              if (i == 7 && save_ok) {
                *(half8*)(a.save_a + save_off) = o0;
                *(half8*)(a.save_t + save_off) = th0;
                *(half8*)(a.save_s + save_off) = sg0;
              }
              if (i == SL - 1 && save_ok) {
                *(half8*)(a.save_a + save_off + 8) = o1;
                *(half8*)(a.save_t + save_off + 8) = th1;
                *(half8*)(a.save_s + save_off + 8) = sg1;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (do_mm) {
          // x_out rows of chunk c-1 = fp16(x + b_res + W_res acts) for valid columns (model.py:130-132); every other row
          // of the plane stays zero: it is the convolution padding of other tiles
          store_xout_half(d2, ntm, 0);
          store_xout_half(d2, ntm, 1);
        }
        if (do_gate) {
          // acts of chunk c complete in LDS for every wave before anyone's GEMM 2 reads them.  Raw barrier: the x_out
          // stores and the next tile's LDS-DMA stay in flight across it (a __syncthreads() fence would drain them).
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (c == 0) WG_STAMP(3);
          if (c == 1) WG_STAMP(5);
          if (c == 2) WG_STAMP(6);
          if (c == 3) WG_STAMP(7);
        }
      };
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
      if constexpr (NT >= 2) phase(std::integral_constant<int, 2>{});
      if constexpr (NT >= 3) phase(std::integral_constant<int, 3>{});
      if constexpr (NT >= 4) phase(std::integral_constant<int, 4>{});
      static_assert(NT <= 4, "phases are instantiated explicitly");
      WG_STAMP(4);
    }
    if constexpr (!PIPE) {
    // ---- issue the loads the post-gate phases need now, so their latency hides under the gate's VALU work:
    // residual input x (this tile, this wave's channels: lane (n, h) owns positions [32*blk + 16h, +16) of
    // column n = 32 contiguous bytes) and the first GEMM2 weight fragments.
    if constexpr (HAS_RES) {
      if constexpr (RES_A0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a0f[nt] = *(const half8*)(a.x_tap + ((size_t)r0 + nt * 32 + lno) * 64 + lho * 8);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) wst[mb] = ((const half8*)a.wStA)[(wave * MB + mb) * 64 + laneo];
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int blk = wave * MB + mb;
        if constexpr (!RES_A0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const size_t row = (size_t)(blk >> 1) * R + r0 + nt * 32 + lno;
            const half8* xp = (const half8*)(a.x_in + row * 64 + (blk & 1) * 32 + lho * 16);
            xres[mb][nt][0] = xp[0];
            xres[mb][nt][1] = xp[1];
          }
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) a2[mb][i] = p2[((size_t)mb * K2 + i) * 64];
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- gate (model.py:13-20) in registers; acts -> LDS as fp16, position-major, padded rows
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        half8 o0, o1;
        if constexpr (TR) {
          half8 th0, th1, sg0, sg1;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float th, sg, ac;
            gate_act3(acc[mb][nt][r], acc[MB + mb][nt][r], th, sg, ac);
            o0[r] = (_Float16)ac; th0[r] = (_Float16)th; sg0[r] = (_Float16)sg;
            gate_act3(acc[mb][nt][8 + r], acc[MB + mb][nt][8 + r], th, sg, ac);
            o1[r] = (_Float16)ac; th1[r] = (_Float16)th; sg1[r] = (_Float16)sg;
          }
          int cb, ct;
          if (column_of(rr0 + nt * 32 + lno, p, cb, ct)) {     // padding columns: rows stay zero (see the pipelined path)
            const int blk = wave * MB + mb;
            const size_t off = ((size_t)(blk >> 1) * R + r0 + nt * 32 + lno) * 64 + (blk & 1) * 32 + lho * 16;
            *(half8*)(a.save_a + off) = o0; *(half8*)(a.save_a + off + 8) = o1;
            *(half8*)(a.save_t + off) = th0; *(half8*)(a.save_t + off + 8) = th1;
            *(half8*)(a.save_s + off) = sg0; *(half8*)(a.save_s + off + 8) = sg1;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            o0[r] = (_Float16)gate_act(acc_at(mb, nt, r), acc_at(MB + mb, nt, r));
            o1[r] = (_Float16)gate_act(acc_at(mb, nt, 8 + r), acc_at(MB + mb, nt, 8 + r));
          }
        }
        if constexpr (M16) {
          write_acts16(nt, wave * MB + mb, 0, o0);
          write_acts16(nt, wave * MB + mb, 1, o1);
        } else {
          char* ap = acts_lane + nt * 32 * ACT_ROW + (wave * MB + mb) * 64;   // positions [32*blk + 16h, +16)
          *(half8*)(ap) = o0;
          *(half8*)(ap + 16) = o1;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    WG_STAMP(3);

    // GEMM2 accumulators start from x + b_res (residual add for free, model.py:132)
    if constexpr (HAS_RES) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float* bp = a.bias2 + (wave * MB + mb) * 32 + 4 * lho;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (RES_A0) {                    // x_0 = W_start a0 + b_start, one MFMA step (see the pipelined epilogue)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mb][nt][r] = bv[r];
            acc2[mb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wst[mb], a0f[nt], acc2[mb][nt], 0, 0, 0);
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              acc2[mb][nt][r] = (float)xres[mb][nt][0][r] + bv[r];
              acc2[mb][nt][8 + r] = (float)xres[mb][nt][1][r] + bv[8 + r];
            }
          }
        }
      }
    }

    }   // !PIPE


    if constexpr (kWesEarly && !PIPE) es_prefetch();          // issue now, consume after GEMM2

    // ---- GEMM2: res rows of this wave (model.py:130-132); acts fragments read one k16 step ahead
    if constexpr (HAS_RES && !PIPE) {
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

    if constexpr (!PIPE) {
      WG_STAMP(4);
      if constexpr (!kWesEarly) es_prefetch();
      es_compute();
    }

    if constexpr (!PIPE) WG_STAMP(5);
    // ---- x_out = fp16(x + res) for valid columns (all other rows stay zero: they are the padding of other tiles)
    if constexpr (HAS_RES && !PIPE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int cb, ct;
        if (column_of(rr0 + nt * 32 + lno, p, cb, ct)) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int blk = wave * MB + mb;
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
    if constexpr (!PIPE) WG_STAMP(6);
    }   // !PLAIN
    __builtin_amdgcn_sched_barrier(0);   // keep the next tile's prologue (128 accumulator inits) out of this epilogue
  }
}

template <int C, int BN, bool HAS_RES, int TPW, int CX, int MODE = 0, int NTAPS = 3, bool HAS_COND = true, bool DEEP = false>
static hipError_t launch_wn_tttt(const WnLayerArgs& a, hipStream_t s) {
  constexpr int NW = WnCfg<C>::NW;
  constexpr bool M16 = wn_frag16(C, BN) && MODE == 0 && !DEEP;   // the host packs GEMM-1 weights for it (api.cpp)
  if (M16 != (a.frag16 != 0)) return hipErrorInvalidValue;
  if (MODE == 0 && CX == 1 && NTAPS == 1 && wn_res_a0(C) && HAS_RES && !a.wStA) return hipErrorInvalidValue;
  constexpr int smem = MODE == 4 ? (BN * (2 * C + 16) > 2 * BN * 128 ? BN * (2 * C + 16) : 2 * BN * 128)
                                 : 2 * BN * 128 + (MODE >= 2 ? 0 : BN * (2 * C + 16) + 3 * C * 4 + (kWesLds ? (C / 32) * 1024 : 0));
  static bool attr_done_dev[64] = {};      // the attribute is per device: keyed by the launch's (current) device
  int cur_dev = 0;
  if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= 64) cur_dev = 0;
  bool& attr_done = attr_done_dev[cur_dev];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)wn_layer_kernel<C, NW, BN, HAS_RES, TPW, CX, MODE, NTAPS, HAS_COND, DEEP, M16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  // TPW tiles per workgroup: per XCD label ceil(tiles_on_label / TPW) blocks
  const int per_label = ((a.n_tiles + 7) / 8 + TPW - 1) / TPW;
  const int grid = 8 * per_label;
  hipLaunchKernelGGL((wn_layer_kernel<C, NW, BN, HAS_RES, TPW, CX, MODE, NTAPS, HAS_COND, DEEP, M16>), dim3(grid), dim3(NW * 64), smem, s, a);
  return hipGetLastError();
}
template <int C, int BN, bool HAS_RES, int TPW>
static hipError_t launch_wn_ttt(const WnLayerArgs& a, hipStream_t s) {
  // small workloads (one tile per workgroup, 64 columns) at 256 channels / 80 mel channels: two-step-deep weight prefetch.
  // WG_DISABLE_DEEP=1 (read per launch, tests only) takes the one-step ring instead: the two must agree bit for bit.
  if constexpr (kDeep && C == 256 && BN == 64 && TPW == 1) {
    const char* e = getenv("WG_DISABLE_DEEP");
    if (a.n_cond_steps == 5 && !(e && *e == '1')) {
      if (a.a0_fold) return launch_wn_tttt<C, BN, HAS_RES, TPW, 1, 0, 1, true, true>(a, s);
      return launch_wn_tttt<C, BN, HAS_RES, TPW, C / 64, 0, 3, true, true>(a, s);
    }
  }
  // first layer with the start fold: the three taps of the a0 plane in one gathered K-step (A0G; wA1 is packed for it)
  if (a.a0_fold) return a.x_chunks_per_tap == 1 ? launch_wn_tttt<C, BN, HAS_RES, TPW, 1, 0, 1>(a, s) : hipErrorInvalidValue;
  return launch_wn_tttt<C, BN, HAS_RES, TPW, C / 64>(a, s);
}
template <int C, int BN, bool HAS_RES>
static hipError_t launch_wn_tt(const WnLayerArgs& a, hipStream_t s) {
  // two tiles per workgroup amortise the launch and prefetch across the tile seam, but need >= 2 tiles per CU -- and must not
  // cost a round: 1120 tiles on 256 CUs are 5 rounds of single tiles but 3 rounds of pairs = 6 tile times (batch 5 x 80x864)
  const int ncu = a.n_cu > 0 ? a.n_cu : 1;
  const int rounds1 = (a.n_tiles + ncu - 1) / ncu;
  const int rounds2 = ((a.n_tiles + kTilesPerWG - 1) / kTilesPerWG + ncu - 1) / ncu * kTilesPerWG;
  if (a.n_tiles >= 2 * kTilesPerWG * ncu && rounds2 <= rounds1) return launch_wn_ttt<C, BN, HAS_RES, kTilesPerWG>(a, s);
  return launch_wn_ttt<C, BN, HAS_RES, 1>(a, s);
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

// training forward: one tile per workgroup, x_0 planes (no a0 fold), spectrogram planes as the conditioning operand
template <int C>
static hipError_t launch_wn_train_t(const WnLayerArgs& a, int bn, hipStream_t s) {
  if (bn == 64) return a.has_res ? launch_wn_tttt<C, 64, true, 1, C / 64, 1>(a, s) : launch_wn_tttt<C, 64, false, 1, C / 64, 1>(a, s);
  if constexpr (WnCfg<C>::BN == 128) {
    if (bn == 128) return a.has_res ? launch_wn_tttt<C, 128, true, 1, C / 64, 1>(a, s) : launch_wn_tttt<C, 128, false, 1, C / 64, 1>(a, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_wn_layer_train(const WnLayerArgs& a, int C, int bn, hipStream_t s) {
  if (!a.sp || !a.save_t || !a.save_s || !a.save_a || a.x_chunks_per_tap != C / 64) return hipErrorInvalidValue;
  switch (C) {
    case 64: return launch_wn_train_t<64>(a, bn, s);
    case 128: return launch_wn_train_t<128>(a, bn, s);
    case 256: return launch_wn_train_t<256>(a, bn, s);
    case 512: return launch_wn_train_t<512>(a, bn, s);
  }
  return hipErrorInvalidValue;
}

// backward dgrad GEMMs on the WN-layer K loop.  kind 2: d x (3 taps over the 2C d-pre planes); kind 3: d acts + gate
// derivative (a.x_tap = d x planes + a.sp = d out plane, or -- a.x_tap == a.sp's plane alone -- the last layer of a flow);
// kind 4: kind 2 of layer i with kind 3 of layer i-1 fused behind it (wat_prev, gout, t_prev, s_prev, dpre_prev)
template <int C>
static hipError_t launch_wn_plain_t(const WnLayerArgs& a, int kind, int bn, hipStream_t s) {
  constexpr int CCH = C / 64;
  if constexpr (WnCfg<C>::BN == 128) {
    if (bn == 128) {
      if (kind == 2) return launch_wn_tttt<C, 128, false, 1, 2 * CCH, 2, 3, false>(a, s);
      if (kind == 4) return launch_wn_tttt<C, 128, false, 1, 2 * CCH, 4, 3, false>(a, s);
      if (a.x_chunks_per_tap == 1 && a.n_cond_steps == 0) return launch_wn_tttt<C, 128, false, 1, 1, 3, 1, false>(a, s);
      return launch_wn_tttt<C, 128, false, 1, CCH, 3, 1, true>(a, s);
    }
  }
  if (bn != 64) return hipErrorInvalidValue;
  if (kind == 2) return launch_wn_tttt<C, 64, false, 1, 2 * CCH, 2, 3, false>(a, s);
  if (kind == 4) return launch_wn_tttt<C, 64, false, 1, 2 * CCH, 4, 3, false>(a, s);
  if (a.x_chunks_per_tap == 1 && a.n_cond_steps == 0) return launch_wn_tttt<C, 64, false, 1, 1, 3, 1, false>(a, s);
  return launch_wn_tttt<C, 64, false, 1, CCH, 3, 1, true>(a, s);
}
hipError_t launch_wn_plain(const WnLayerArgs& a, int C, int kind, int bn, hipStream_t s) {
  if ((kind != 2 && kind != 3 && kind != 4) || !a.out0 || !a.x_tap || (kind == 3 && (!a.in0 || !a.in1))) return hipErrorInvalidValue;
  if ((kind == 2 || kind == 4) && (a.x_chunks_per_tap != 2 * (C / 64) || a.n_cond_steps != 0)) return hipErrorInvalidValue;
  if (kind == 4 && (!a.wat_prev || !a.gout || !a.t_prev || !a.s_prev || !a.dpre_prev)) return hipErrorInvalidValue;
  if (kind == 3 && !((a.x_chunks_per_tap == 1 && a.n_cond_steps == 0) || (a.x_chunks_per_tap == C / 64 && a.n_cond_steps == 1 && a.sp)))
    return hipErrorInvalidValue;
  switch (C) {
    case 64: return launch_wn_plain_t<64>(a, kind, bn, s);
    case 128: return launch_wn_plain_t<128>(a, kind, bn, s);
    case 256: return launch_wn_plain_t<256>(a, kind, bn, s);
    case 512: return launch_wn_plain_t<512>(a, kind, bn, s);
  }
  return hipErrorInvalidValue;
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
// Conditioning input.  The reference upsamples the mel with ConvTranspose1d(M, M, 1024, stride 256)
// (model.py:145-150, :225-232) and feeds the squeezed result to every WN's cond_layer (model.py:121).  Both are
// linear, so per phase p (group-timestep inside a frame) they compose into ONE matrix acting on the 4 mel frames
// q..q-3:   cond[m][t = 32q+p] = sum_{j<4, i<M} Wc_p[m][j*M+i] * mel[i][q-j] + const,
//           Wc_p[m][j*M+i] = sum_{o<M, g<8} W_cond[m][o*8+g] * W_up[i][o][8p+g+256j]
// -- K = 4M = 320 instead of 8M = 640 per cond slice, and the upsampled [B,640,L] tensor never exists.
// mel_pack_kernel transposes the mel to frame-major fp16 rows (the B operand of those K-steps);
// cond_fold_kernel builds Wc_p for every (layer, phase) in MFMA A-fragment order at weight-load time.
// =============================================================================================
__global__ void __launch_bounds__(256) mel_pack_kernel(const MelPackArgs a) {
  // rows: 3 zero | per utterance: 3 zero, T frames, 3 zero
  const int rows_per_utt = a.T + 6;
  const size_t n = (size_t)(3 + a.B * rows_per_utt) * a.M;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const int i = (int)(idx % a.M);
    const int row = (int)(idx / a.M) - 3;
    float v = 0.0f;
    if (row >= 0) {
      const int b = row / rows_per_utt, f = row - b * rows_per_utt - 3;
      if (f >= 0 && f < (a.frames ? a.frames[b] : a.T)) v = load_io(a.mel, ((size_t)b * a.M + i) * a.T + f, a.io_f16);
    }
    a.melT[idx] = (_Float16)v;
  }
}

// zero fill as a KERNEL: a hipMemsetAsync issued while the stream is being captured into a hipGraph did not end up as
// a node of the graph here (first replay right, later replays read dirty planes), a kernel launch always does
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t s) {
  if (((size_t)p & 15) || (bytes & 15)) return hipErrorInvalidValue;
  size_t blocks = (bytes / 16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)p, bytes / 16);
  return hipGetLastError();
}

hipError_t launch_mel_pack(const MelPackArgs& a, hipStream_t s) {
  const size_t n = (size_t)(3 + a.B * (a.T + 6)) * a.M;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mel_pack_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// One block: one (layer l, phase p, tap j) and 64 gate rows m; computes the [64 x M] block
//   Wc[m][i] = sum_{s = o*8+g} W_cond[2C*l + m][s] * W_up[i][o][8p + g + 256 j]
// in fp32 through LDS tiles and scatters it, gate pre-scaled and rounded to fp16, into A-fragment order
// [l][p][half K-step][wave][MT][2 k16][64 lanes][8]   (K index kk = j*M + i; see wn_layer_kernel::load_Aq).
__global__ void __launch_bounds__(256) cond_fold_kernel(const float* __restrict__ w_cond, const float* __restrict__ w_up,
                                                        _Float16* __restrict__ out, int C, int NW, int M, int up_kernel,
                                                        float tanh_scale, float sigm_scale, int frag16) {
  __shared__ float sW[64][33];    // W_cond rows m0..m0+63, s-chunk of 32
  __shared__ float sU[32][81];    // U[s][i] = W_up[i][o][8p+g+256j]
  const int p = blockIdx.x >> 2, j = blockIdx.x & 3;
  const int m0 = blockIdx.y * 64;
  const int l = blockIdx.z;
  const int NS = M * 8, MB = C / (32 * NW), MT = 2 * MB;
  const int tid = threadIdx.x;
  float acc[20];                                   // 64 x 80 outputs / 256 threads (M <= 80)
#pragma unroll
  for (int e = 0; e < 20; ++e) acc[e] = 0.0f;
  const int per_thread = (64 * M + 255) / 256;
  for (int s0 = 0; s0 < NS; s0 += 32) {
    for (int e = tid; e < 64 * 32; e += 256) {
      const int mm = e >> 5, ss = e & 31;
      sW[mm][ss] = w_cond[((size_t)(2 * C * l + m0 + mm)) * NS + s0 + ss];
    }
    for (int e = tid; e < 32 * M; e += 256) {
      const int ss = e / M, i = e - ss * M;
      const int sch = s0 + ss, o = sch >> 3, g = sch & 7;
      sU[ss][i] = w_up[((size_t)i * M + o) * up_kernel + 8 * p + g + 256 * j];
    }
    __syncthreads();
    for (int e = 0; e < per_thread; ++e) {
      const int idx = e * 256 + tid;
      if (idx < 64 * M) {
        const int mm = idx / M, i = idx - mm * M;
        float v = acc[e];
#pragma unroll 8
        for (int ss = 0; ss < 32; ++ss) v = fmaf(sW[mm][ss], sU[ss][i], v);
        acc[e] = v;
      }
    }
    __syncthreads();
  }
  const int n_half = 2 * (M / 16);                 // half K-steps of the conditioning part
  for (int e = 0; e < per_thread; ++e) {
    const int idx = e * 256 + tid;
    if (idx >= 64 * M) continue;
    const int mm = idx / M, i = idx - mm * M;
    const int m = m0 + mm;                         // gate row in [0, 2C)
    const bool tanh_row = m < C;
    const int ch = tanh_row ? m : m - C;           // gate channel
    const int blk = ch >> 5, r = ch & 31;
    const int w = blk / MB, mt = (tanh_row ? 0 : MB) + (blk - w * MB);
    const int kk = j * M + i;
    const int u = kk >> 5, k2 = (kk >> 4) & 1, hh = (kk >> 3) & 1, jj = kk & 7;
    const _Float16 val = (_Float16)(acc[e] * (tanh_row ? tanh_scale : sigm_scale));
    if (frag16) {        // 16x16x32 A fragments (wn_layer_kernel M16): [..][half K-step][wave][tile m][(row i, K group)][8]
      const int m16 = 2 * (tanh_row ? 0 : 1) + (r >> 4), kg = (kk >> 3) & 3;
      const size_t f16 = (((size_t)(l * kPhases + p) * n_half + u) * NW + w) * 4 + m16;
      out[(f16 * 64 + kg * 16 + (r & 15)) * 8 + jj] = val;
      continue;
    }
    const size_t frag = ((((size_t)(l * kPhases + p) * n_half + u) * NW + w) * MT + mt) * 2 + k2;
    out[(frag * 64 + hh * 32 + r) * 8 + jj] = val;
  }
}

hipError_t launch_cond_fold(const float* w_cond, const float* w_up, _Float16* out, int C, int NW, int M, int n_layers,
                            int up_kernel, float tanh_scale, float sigm_scale, int frag16, hipStream_t s) {
  if (M > 80 || M % 16 || (frag16 && C != 32 * NW)) return hipErrorInvalidValue;
  dim3 grid(kPhases * 4, 2 * C / 64, n_layers);
  hipLaunchKernelGGL(cond_fold_kernel, grid, dim3(256), 0, s, w_cond, w_up, out, C, NW, M, up_kernel, tanh_scale,
                     sigm_scale, frag16);
  return hipGetLastError();
}

// =============================================================================================
// Flow step (both directions) + next WN start.  64 rows per workgroup, 256 threads.
// =============================================================================================
// FL_ROWS rows per workgroup = threads per workgroup: 256, or 64 for small workloads (a single utterance of 500 frames
// is 16 000 rows: 63 workgroups of 256 left three quarters of the chip idle, 21 us per launch x 13 launches).
template <int FL_ROWS>
__global__ void __launch_bounds__(FL_ROWS) flow_kernel(const FlowArgs a) {
  __shared__ float4 s_a0[FL_ROWS];
  const int L = a.g.L;
  const size_t nrows = (size_t)a.g.B * L;
  const size_t row0 = (size_t)blockIdx.x * FL_ROWS;
  const int tid = threadIdx.x;

  {
    const size_t row = row0 + tid;
    if (row < nrows) {
      // 32-bit row arithmetic (launch_flow checks the range): a 64-bit division is ~150 instructions, and this kernel
      // did 33 of them per thread -- it was ALU-bound on them, not HBM-bound
      const unsigned r32 = (unsigned)row;
      const int b = (int)(r32 / (unsigned)L), t = (int)(r32 - (unsigned)b * (unsigned)L);
      // ragged batch: columns behind an utterance's own length are padding -- their state is carried along (finite,
      // never read by a valid column) but they are never written into the x / a0 planes, and their audio is zero
      const bool pad_row = a.g.frames != nullptr && t >= 32 * a.g.frames[b];
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
          if (pad_row) {
#pragma unroll
            for (int c = 0; c < 8; ++c) zn[c] = 0.0f;
          }
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
        float4* zp = (float4*)(a.Z_w + row * 8);
        zp[0] = make_float4(zn[0], zn[1], zn[2], zn[3]);
        zp[1] = make_float4(zn[4], zn[5], zn[6], zn[7]);
        float4* op = (float4*)(a.out_w + row * 8);
        op[0] = make_float4(a.out_init[0], a.out_init[1], a.out_init[2], a.out_init[3]);
        op[1] = make_float4(a.out_init[4], a.out_init[5], a.out_init[6], a.out_init[7]);
        s_a0[tid] = make_float4(zn[0], zn[1], zn[2], zn[3]);
        if (a.a0p && !pad_row) {   // a0 plane for the folded first WN layer: (a0_0..a0_3 | 1 | 0 0 0), see wn_layer_kernel CX = 1
          half8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)0.0f;
          const int hn = a.h_next;
          o[0] = (_Float16)zn[0];
          if (hn > 1) o[1] = (_Float16)zn[1];
          if (hn > 2) o[2] = (_Float16)zn[2];
          if (hn > 3) o[3] = (_Float16)zn[3];
          o[4] = (_Float16)1.0f;
          const size_t prow = (size_t)kRowPad + (size_t)(t & 31) * a.g.Rp + (size_t)b * a.g.Fp + a.g.Gf + (t >> 5);
          *(half8*)(a.a0p + prow * 64) = o;
          *(half8*)(a.a0p + prow * 64 + 8) = o;     // again at 8..12: the lo weight parts of the first layer's residual step
        }
      }
    }
  }
  if (a.last || a.skip_x) return;
  __syncthreads();

  // ---- WN.start of the next flow (model.py:117): x[P] = sum_j Wst[P][j] a0[j] + b[P], fp16, position-major.
  // piece = (chunk cc, row, 8-position group g8) = 16 contiguous bytes; idx = cc*2048 + row*8 + g8, so
  // consecutive threads write consecutive bytes and a thread's g8 is fixed: its 8x4 weights sit in registers.
  const int C = a.C, h = a.h_next;
  const int g8 = tid & 7;
  // plane rows of this thread's 8 rows (one per pass), once for all chunks: phase-major row of (b, t = 32q + p), see RowGeom
  unsigned prow[8];
  unsigned ok = 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int rl = it * (FL_ROWS / 8) + (tid >> 3);
    const size_t row = row0 + rl;
    prow[it] = 0;
    if (row >= nrows) continue;
    const unsigned r32 = (unsigned)row;
    const int b = (int)(r32 / (unsigned)L), t = (int)(r32 - (unsigned)b * (unsigned)L);
    if (a.g.frames != nullptr && t >= 32 * a.g.frames[b]) continue;      // padding column of a ragged batch
    prow[it] = (unsigned)kRowPad + (unsigned)(t & 31) * (unsigned)a.g.Rp + (unsigned)b * (unsigned)a.g.Fp + (unsigned)a.g.Gf + (unsigned)(t >> 5);
    ok |= 1u << it;
  }
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
    for (int it = 0; it < 8; ++it) {
      if (!((ok >> it) & 1u)) continue;
      const int rl = it * (FL_ROWS / 8) + (tid >> 3);
      const float4 a0 = s_a0[rl];
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = (_Float16)fmaf(w[e][3], a0.w, fmaf(w[e][2], a0.z, fmaf(w[e][1], a0.y, fmaf(w[e][0], a0.x, bs[e]))));
      *(half8*)(a.x + ((size_t)cc * a.g.R + prow[it]) * 64 + g8 * 8) = o;
    }
  }
}

hipError_t launch_flow(const FlowArgs& a, hipStream_t s) {
  const size_t nrows = (size_t)a.g.B * a.g.L;
  if (nrows >= (1ull << 31) || (size_t)a.g.R >= (1ull << 31)) return hipErrorInvalidValue;   // 32-bit row arithmetic in the kernel
  if (nrows < 256 * 256) {
    hipLaunchKernelGGL(flow_kernel<64>, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, s, a);
  } else {
    hipLaunchKernelGGL(flow_kernel<256>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, a);
  }
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

__global__ void loss_final_kernel(const double* acc, double log_det_total, const float* log_det_dev, int n_dev, float sigma,
                                  double denom, float* out) {
  for (int k = 0; k < n_dev; ++k) log_det_total += (double)log_det_dev[k];     // same order as the host sum (train.py:37,41)
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
hipError_t launch_loss_final(const double* acc, double log_det_total, const float* log_det_dev, int n_dev, float sigma, double denom,
                             float* out, hipStream_t s) {
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, s, acc, log_det_total, log_det_dev, n_dev, sigma, denom, out);
  return hipGetLastError();
}

}  // namespace wg
