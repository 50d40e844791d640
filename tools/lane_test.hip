// developer check: cross-lane exchange primitives used by k_gt_search (DPP row_mirror, permlane16/32 swap)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ static inline int dpp_get(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
  int lane = threadIdx.x, v = lane * 3 + 1;
  out[lane] = dpp_get<0x140>(v);
  out[64 + lane] = dpp_get<0x141>(v);
  { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); out[128 + lane] = r[0]; out[192 + lane] = r[1]; }
  { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); out[256 + lane] = r[0]; out[320 + lane] = r[1]; }
}
int main() {
  int* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); int h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[6] = {"row_mirror", "half_mirror", "pl16 r0", "pl16 r1", "pl32 r0", "pl32 r1"};
  for (int t = 0; t < 6; t++) { printf("%s:", nm[t]); for (int l = 0; l < 64; l++) printf(" %d", (h[t * 64 + l] - 1) / 3); printf("\n"); }
  return 0;
}
