// Micro-benchmark (diagnostic, GPU box only): issue cost of the gate's VALU instruction mix on gfx950, for one and
// two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_costs.hip -o valu_costs && ./valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

template <int KIND>
__global__ void __launch_bounds__(512) bench(unsigned long long* out, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.5f, b1 = 2.5f, b2 = 3.5f, b3 = 4.5f, b4 = .5f, b5 = .25f, b6 = .75f, b7 = .125f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {   // 8 independent v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 1) {   // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 2) {   // 8 independent v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 3) {   // 4 v_pk_mul_f32 (8 elements)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));
    } else if constexpr (KIND == 4) {   // 4 v_cvt_pk_f16_f32
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_pk_f16_f32 %3, %4, %5\n\tv_cvt_pk_f16_f32 %6, %7, %1\n\tv_cvt_pk_f16_f32 %8, %2, %4"
                   : "=v"(b0), "+v"(a0), "+v"(a1), "=v"(b1), "+v"(a2), "+v"(a3), "=v"(b2), "+v"(a4), "=v"(b3));
    } else if constexpr (KIND == 5) {   // the gate mix for 2 elements: 2 med3, 4 exp, 2 add, 2 fmac, 2 rcp, pk_add, pk_mul, cvt_pk
      asm volatile(
        "v_med3_f32 %0, %0, %8, %9\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %2, %2\n\t"
        "v_med3_f32 %1, %1, %8, %9\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %3, %3\n\t"
        "v_add_f32 %4, 1.0, %2\n\tv_fmac_f32 %4, %0, %4\n\tv_rcp_f32 %4, %4\n\t"
        "v_add_f32 %5, 1.0, %3\n\tv_fmac_f32 %5, %1, %5\n\tv_rcp_f32 %5, %5\n\t"
        "v_pk_add_f32 %6, %6, %7\n\tv_pk_mul_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %2, %0, %1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(*(double*)&b2), "+v"(*(double*)&b4) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 6) {   // same, scalar instead of packed ops
      asm volatile(
        "v_med3_f32 %0, %0, %8, %9\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %2, %2\n\t"
        "v_med3_f32 %1, %1, %8, %9\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %3, %3\n\t"
        "v_add_f32 %4, 1.0, %2\n\tv_fmac_f32 %4, %0, %4\n\tv_rcp_f32 %4, %4\n\t"
        "v_add_f32 %5, 1.0, %3\n\tv_fmac_f32 %5, %1, %5\n\tv_rcp_f32 %5, %5\n\t"
        "v_add_f32 %6, -1.0, %0\n\tv_add_f32 %7, -1.0, %1\n\tv_mul_f32 %6, %6, %4\n\tv_mul_f32 %7, %7, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 7) {   // 8 independent v_exp_f16
      asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\tv_exp_f16 %4, %4\n\tv_exp_f16 %5, %5\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 8) {   // 4 exp + 4 fma alternating
      asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %4, %4, %8, %9\n\tv_exp_f32 %1, %1\n\tv_fma_f32 %5, %5, %8, %9\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %6, %6, %8, %9\n\tv_exp_f32 %3, %3\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 9) {   // 4 pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 == 12345.678f) out[63] = 1;
}

template <int KIND>
void run(const char* name, int n_instr, unsigned long long* d) {
  for (int waves : {4, 8}) {
    unsigned long long h[8];
    const int iters = 2000;
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(waves * 64), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-44s %d wave/SIMD: %7.1f cycles per block of %2d instr = %6.2f cyc/instr/wave  (SIMD throughput %6.2f cyc/instr)\n", name, waves / 4,
           mx / iters, n_instr, mx / iters / n_instr, mx / iters / n_instr / (waves / 4));
  }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  run<0>("8 x v_exp_f32", 8, d);
  run<2>("8 x v_rcp_f32", 8, d);
  run<7>("8 x v_exp_f16", 8, d);
  run<1>("8 x v_fma_f32", 8, d);
  run<3>("4 x v_pk_mul_f32", 4, d);
  run<9>("4 x v_pk_fma_f32", 4, d);
  run<4>("4 x v_cvt_pk_f16_f32", 4, d);
  run<8>("4 x (v_exp_f32 + v_fma_f32)", 8, d);
  run<5>("gate mix, 2 elements, packed tail (15 instr)", 15, d);
  run<6>("gate mix, 2 elements, scalar tail (17 instr)", 17, d);
  return 0;
}
