// tools/ubench_mfma.hip -- is the 8x8 Hadamard SATD (TComRdCost::xCalcHADs8x8, the cost of the fractional / GT searches and of the intra rough search) faster as +-1 matrix
// products on the matrix cores?  For 8-bit samples it would be exact: |diff| <= 255 and first-stage sums <= 2040 are integers an f16 holds exactly, the f32 accumulators
// hold every sum exactly, and sum |H D H^T| does not depend on the order of the Hadamard rows.  One v_mfma_f32_16x16x16_f16 with H16 = diag(H8, H8) transforms the rows
// of FOUR 8x8 blocks laid out as the quadrants of a 16x16 tile, a second one the columns: 2 MFMAs (+ the f32 -> f16 repack through LDS, since the second product sums over
// the first one's lane index) per 4 blocks, against 6 butterfly stages of DPP / ds_swizzle adds per block in hopd_satd8x8_wave (hevc-hop_amd/csrc/hop_dev.h).
// Build: hipcc --offload-arch=gfx950 -O3 -I hevc-hop_amd/csrc -I include tools/ubench_mfma.hip -o tools/ubench_mfma ; run on an MI355X.  Prints exactness and both rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "hop_dev.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ static inline float hsign(int k, int n) { return (__builtin_popcount(k & n) & 1) ? -1.0f : 1.0f; }   // Sylvester Hadamard H8[k][n]

// blocks: nblk x 64 int16 differences; out[b] = sum |H8 D_b H8^T| (the reference's (sum + 2) >> 2 is applied by the caller); ITER passes over the same data for timing
__global__ __launch_bounds__(256) void k_satd_mfma(const int16_t* __restrict__ blocks, int ngroups, int iter, uint32_t* __restrict__ out) {
  __shared__ _Float16 t[4][16][16 + 8];                              // per wave: the first product, [row][col], padded rows
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
  // H16 = diag(H8, H8) as the A operand (rows r, k = 4 kq + j) and, transposed, as the B operand of the second product (k = 4 kq + j, col r): the same values (H symmetric)
  h4 hfrag;
  for (int j = 0; j < 4; j++) { const int k = 4 * kq + j; hfrag[j] = ((r >> 3) == (k >> 3)) ? (_Float16)hsign(r & 7, k & 7) : (_Float16)0.0f; }
  for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
    uint32_t acc = 0;
    for (int it = 0; it < iter; it++) {
      // X (16 x 16): quadrant q = block 4 g + q; as the B operand lane holds X[k = 4 kq + j][col r]
      h4 x;
      for (int j = 0; j < 4; j++) { const int row = 4 * kq + j, col = r, q = (row >> 3) * 2 + (col >> 3); x[j] = (_Float16)(float)(blocks[((size_t)(4 * g + q)) * 64 + (row & 7) * 8 + (col & 7)] + (it & 1)); }   // (+ it & 1: keeps the pass inside the loop)
      f4 y = { 0, 0, 0, 0 };
      y = __builtin_amdgcn_mfma_f32_16x16x16f16(hfrag, x, y, 0, 0, 0);          // Y = H16 X : lane holds Y[row 4 kq + i][col r]
      for (int i = 0; i < 4; i++) t[wave][4 * kq + i][r] = (_Float16)y[i];
      __builtin_amdgcn_wave_barrier();
      h4 ya;                                                                    // Y as the A operand: lane holds Y[row r][k = 4 kq + j]
      for (int j = 0; j < 4; j++) ya[j] = t[wave][r][4 * kq + j];
      f4 z = { 0, 0, 0, 0 };
      z = __builtin_amdgcn_mfma_f32_16x16x16f16(ya, hfrag, z, 0, 0, 0);         // Z = Y H16^T : lane holds Z[row 4 kq + i][col r]
      __builtin_amdgcn_wave_barrier();
      // per-quadrant sums of |Z|: rows 4 kq .. 4 kq + 3 lie in quadrant row kq >> 1, column r in quadrant column r >> 3
      int s = (int)(__builtin_fabsf(z[0]) + __builtin_fabsf(z[1]) + __builtin_fabsf(z[2]) + __builtin_fabsf(z[3]));
      // reduce over the 8 lanes of a quadrant column block and the two kq of a quadrant row
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 16);
      if ((lane & 7) == 0 && (kq & 1) == 0) { if (it == 0) out[4 * g + (kq >> 1) * 2 + (r >> 3)] = (uint32_t)s; acc += (uint32_t)s; }
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;                                       // keeps the loop
  }
}

__global__ __launch_bounds__(256) void k_satd_valu(const int16_t* __restrict__ blocks, int nblk, int iter, uint32_t* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = blockIdx.x * 4 + wave; b < nblk; b += gridDim.x * 4) {
    uint32_t acc = 0;
    for (int it = 0; it < iter; it++) {
      const int d = blocks[(size_t)b * 64 + lane] + (it & 1);
      const int s = hopd_satd8x8_wave(d, lane);                                 // (sum + 2) >> 2 inside
      if (lane == 0) { if (it == 0) out[b] = (uint32_t)s; acc += (uint32_t)s; }
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
  }
}

static uint32_t cpu_satd8(const int16_t* d) {                                   // sum |H8 D H8^T| with the Sylvester matrix
  int m[64], t2[64];
  for (int k = 0; k < 8; k++) for (int c = 0; c < 8; c++) { int s = 0; for (int n = 0; n < 8; n++) s += ((__builtin_popcount(k & n) & 1) ? -1 : 1) * d[n * 8 + c]; m[k * 8 + c] = s; }
  for (int k = 0; k < 8; k++) for (int c = 0; c < 8; c++) { int s = 0; for (int n = 0; n < 8; n++) s += m[k * 8 + n] * ((__builtin_popcount(c & n) & 1) ? -1 : 1); t2[k * 8 + c] = s; }
  uint32_t a = 0; for (int i = 0; i < 64; i++) a += (uint32_t)abs(t2[i]);
  return a;
}

int main() {
  const int nblk = 1 << 20, iter = 16;
  std::vector<int16_t> h((size_t)nblk * 64);
  uint32_t x = 12345u;
  for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; h[i] = (int16_t)((int)((x >> 8) % 511) - 255); }   // the full 8-bit difference range
  int16_t* d; uint32_t *o1, *o2;
  (void)hipMalloc(&d, h.size() * 2); (void)hipMalloc(&o1, nblk * 4); (void)hipMalloc(&o2, nblk * 4);
  (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms1 = 0, ms2 = 0;
  hipLaunchKernelGGL(k_satd_mfma, dim3(2048), dim3(256), 0, 0, d, nblk / 4, 1, o1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); hipLaunchKernelGGL(k_satd_mfma, dim3(2048), dim3(256), 0, 0, d, nblk / 4, iter, o1); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms1, a, b);
  hipLaunchKernelGGL(k_satd_valu, dim3(2048), dim3(256), 0, 0, d, nblk, 1, o2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); hipLaunchKernelGGL(k_satd_valu, dim3(2048), dim3(256), 0, 0, d, nblk, iter, o2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms2, a, b);
  std::vector<uint32_t> r1(nblk), r2(nblk);
  (void)hipMemcpy(r1.data(), o1, nblk * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(r2.data(), o2, nblk * 4, hipMemcpyDeviceToHost);
  long bad1 = 0, bad2 = 0;
  for (int i = 0; i < nblk; i++) { const uint32_t w = cpu_satd8(&h[(size_t)i * 64]); bad1 += r1[i] != w; bad2 += r2[i] != ((w + 2) >> 2); }
  printf("{\"blocks\": %d, \"iter\": %d, \"mfma_mismatches\": %ld, \"valu_mismatches\": %ld, \"mfma_ms\": %.3f, \"valu_ms\": %.3f, \"mfma_Gblocks_per_s\": %.2f, \"valu_Gblocks_per_s\": %.2f}\n",
         nblk, iter, bad1, bad2, ms1, ms2, (double)nblk * iter / ms1 / 1e6, (double)nblk * iter / ms2 / 1e6);
  return (bad1 || bad2) ? 1 : 0;
}
