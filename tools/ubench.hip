// tools/ubench.hip -- instruction-rate micro-benchmarks on gfx950 used to choose the hot-path instruction mix.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o tools/ubench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define NACC 8
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[NACC]; double f[NACC];
  uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9E3779B9u;
  for (int i = 0; i < NACC; i++) { a[i] = x + i; f[i] = (double)(x & 1023) + i; }
  unsigned long long q[NACC]; for (int i = 0; i < NACC; i++) q[i] = 0;
  double g = 1.000001 + (seed & 3), hh = 0.5;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (OP == 0) a[i] = a[i] + y;                                        // v_add_u32
      if (OP == 1) a[i] = __builtin_amdgcn_sad_u16(x, y + i, a[i]);        // v_sad_u16
      if (OP == 2) a[i] = __builtin_amdgcn_sad_u8(x, y + i, a[i]);         // v_sad_u8
      if (OP == 3) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)x << 32) | (y + i), x, q[i]);   // v_qsad_pk_u16_u8
      if (OP == 4) f[i] = f[i] * g;                                        // v_mul_f64
      if (OP == 5) f[i] = f[i] + g;                                        // v_add_f64
      if (OP == 6) f[i] = __builtin_fma(f[i], g, hh);                      // v_fma_f64
      if (OP == 7) a[i] = (uint32_t)(int)(f[i] + (double)a[i]);            // cvt_f64_i32 + add + cvt_i32_f64
      if (OP == 8) a[i] = __builtin_amdgcn_update_dpp(a[i], a[i], 0xB1, 0xF, 0xF, false) + y;   // dpp mov + add
      if (OP == 9) a[i] = __builtin_amdgcn_alignbit(a[i], y, 16);          // v_alignbit
      if (OP == 10) a[i] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 1) << 2, a[i]) + 1;     // ds_bpermute
      if (OP == 11) a[i] = a[i] * 3 + y;                                   // v_mad_u32_u24 / mul_lo
    }
  }
  uint32_t r = 0; for (int i = 0; i < NACC; i++) r += a[i] + (uint32_t)f[i] + (uint32_t)q[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, uint32_t* d, double ops_per) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid(256 * 8), blk(256);
  hipLaunchKernelGGL(k<OP>, grid, blk, 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, grid, blk, 0, 0, d, 2u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double n = (double)grid.x * blk.x * ITER * NACC * ops_per;
  printf("%-28s %8.3f ms  %8.2f T lane-op/s  (x64 = wave instr: %6.1f G/s)\n", name, ms, n / ms / 1e9, n / 64 / ms / 1e6);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", d, 1); run<1>("v_sad_u16", d, 1); run<2>("v_sad_u8", d, 1); run<3>("v_qsad_pk_u16_u8", d, 1);
  run<4>("v_mul_f64", d, 1); run<5>("v_add_f64", d, 1); run<6>("v_fma_f64", d, 1); run<7>("cvt f64<->i32 + add_f64 (3)", d, 3);
  run<8>("dpp mov + add (2)", d, 2); run<9>("v_alignbit", d, 1); run<10>("ds_bpermute + add (2)", d, 2); run<11>("mul+add (2)", d, 2);
  return 0;
}
