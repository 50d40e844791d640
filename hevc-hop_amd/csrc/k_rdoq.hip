// k_rdoq.hip -- rate-distortion optimised quantisation of one transform unit.
// Replaces TComTrQuant::xRateDistOptQuant (TLibCommon/TComTrQuant.cpp:1489-1999) with xGetCodedLevel (:2123-2173),
// xGetICRate (:2182-2240), xGetRateLast (:2253-2268), calcPatternSigCtx (:2008-2026), getSigCtxInc (:2038-2092),
// getSigCoeffGroupCtxInc (:2291-2313), the flat-list error scale of setErrScaleCoeff (:2360-2379) and the scans of
// TComRom.cpp:355-481.  Every cost is a double evaluated in the reference's operation order (-ffp-contract=off):
// the decisions are comparisons of such sums, so the order is part of the result.
//
// Mapping: one LANE per TU.  The level decision walks the scan backwards with a context state (c1, c2, Rice parameter,
// context set) that every coefficient updates, so a TU is a serial loop; the parallelism of this row is across TUs (an RD
// search tests thousands of TUs per CTU).  The TUs are bucketed by size class, 64 of a class share a wave and walk in
// lock step; the per-coefficient state a TU must keep for the later passes (last position, sign hiding) lives in a work
// area in HBM laid out [scan position][lane], so every step's loads and stores are consecutive words across the wave.
#include "hop_dev.h"

#include "k_rdoq_dev.inl"

template <int LOG2>
__global__ __launch_bounds__(64) void k_rdoq(const hop_rdoq_job* __restrict__ jobs, const hop_estbits* __restrict__ tables, const uint16_t* __restrict__ scans,
                                             const int32_t* __restrict__ src_all, int32_t* __restrict__ dst_all, uint32_t* __restrict__ abs_sum_out,
                                             const int* __restrict__ list, const int* __restrict__ count_ptr, char* __restrict__ work) {
  constexpr int N2 = 1 << (2 * LOG2), CGN = N2 >> 4;
  __shared__ uint16_t s_scan[3][N2];
  __shared__ uint16_t s_scanCG[3][CGN];
  __shared__ double s_cgSig[CGN][64];                                     // cost of the coded_sub_block_flag of each group, one column per lane
  const int lane = threadIdx.x;
  const int count = *count_ptr;
  if (blockIdx.x * 64 >= count) return;                                   // nothing of this size class for this block (uniform)
  for (int i = lane; i < 3 * N2; i += 64) s_scan[i / N2][i % N2] = rq_scan(scans, i / N2, LOG2)[i % N2];
  for (int i = lane; i < 3 * CGN; i += 64) s_scanCG[i / CGN][i % CGN] = rq_scan_cg(scans, i / CGN, LOG2)[i % CGN];
  __syncthreads();
  // this block's slice of the work area: arrays [scan position][lane], so that the lanes of a step touch consecutive words
  double* const wd = (double*)(work + (size_t)blockIdx.x * ((size_t)N2 * 64 * RQ_WORK_PER_COEF));
  for (int g = blockIdx.x; g * 64 < count; g += gridDim.x) {
    const int idx = g * 64 + lane;
    if (idx >= count) continue;                                           // no barrier below: a lane only ever reads what it wrote itself
    const int ti = list[idx];
    const hop_rdoq_job jb = jobs[ti];
    rdoq_tu<LOG2>(jb, tables + jb.estbits_index, s_scan[jb.scan_idx], s_scanCG[jb.scan_idx], &s_cgSig[0][lane], 64, src_all + jb.coeff_offset, dst_all + jb.coeff_offset, abs_sum_out + ti, wd, 64, lane);
  }
}

// TUs by size class: list[c * n + k] = index of the k-th TU of class c (log2_size - 2), counts[c] of them (any order: a TU's result does not depend on it)
__global__ __launch_bounds__(256) void k_rdoq_classify(const hop_rdoq_job* __restrict__ jobs, int n, int* __restrict__ counts, int* __restrict__ list) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int cls = i < n ? jobs[i].log2_size - 2 : -1;
  const int lane = threadIdx.x & 63;
  for (int c = 0; c < 4; c++) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&counts[c], __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (cls == c) list[(size_t)c * n + base + __popcll(m & ((1ull << lane) - 1))] = i;
  }
}

// ---- scan tables (host), TComRom.cpp:355-481: 3 scans x sides 4..32 + the coefficient-group scans the RDOQ walks ----
static void rq_diag(uint16_t* T, int w) {                                 // up-right diagonal scan of a w x w grid
  int next = 0;
  for (int line = 0; next < w * w; line++) {
    int prim = line, scnd = 0;
    while (prim >= w) { scnd++; prim--; }
    while (prim >= 0 && scnd < w) { T[next++] = (uint16_t)(prim * w + scnd); scnd++; prim--; }
  }
}
void hop_rdoq_build_scans(uint16_t* tabs /* 4080 + 255 entries */) {
  uint16_t d2[4], d4[16], d8[64];
  rq_diag(d2, 2); rq_diag(d4, 4); rq_diag(d8, 8);
  const int offs[4] = { 0, 16, 80, 336 }, cgo[4] = { 0, 1, 5, 21 };
  for (int l = 2; l <= 5; l++) {
    const int w = 1 << l, ns = w >> 2;
    uint16_t* D = tabs + 0 * 1360 + offs[l - 2]; uint16_t* Hs = tabs + 1 * 1360 + offs[l - 2]; uint16_t* V = tabs + 2 * 1360 + offs[l - 2];
    const uint16_t* cgd = (ns == 1) ? nullptr : (ns == 2) ? d2 : (ns == 4) ? d4 : d8;   // diagonal scan of the group grid
    if (l == 2) rq_diag(D, 4);
    else
      for (int blk = 0; blk < ns * ns; blk++) {
        const int init = cgd[blk], oy = init / ns, ox = init - oy * ns, offD = 4 * (ox + oy * w);
        for (int k = 0; k < 16; k++) D[16 * blk + k] = (uint16_t)((d4[k] >> 2) * w + (d4[k] & 3) + offD);
      }
    int cnt = 0;
    for (int by = 0; by < ns; by++) for (int bx = 0; bx < ns; bx++)
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) Hs[cnt++] = (uint16_t)((by * 4 + y) * w + bx * 4 + x);
    cnt = 0;
    for (int bx = 0; bx < ns; bx++) for (int by = 0; by < ns; by++)
      for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) V[cnt++] = (uint16_t)((by * 4 + y) * w + bx * 4 + x);
    // group scans: 4x4 -> {0}; 8x8 -> g_sigLastScan8x8[scan]; 16x16 -> the 4x4 scans; 32x32 -> the 8x8 diagonal for every scan
    for (int s = 0; s < 3; s++) {
      uint16_t* C = tabs + 4080 + s * 85 + cgo[l - 2];
      if (l == 2) C[0] = 0;
      else if (l == 3) { const uint16_t t8[3][4] = { {0, 2, 1, 3}, {0, 1, 2, 3}, {0, 2, 1, 3} }; for (int k = 0; k < 4; k++) C[k] = t8[s][k]; }
      else if (l == 4) {
        if (s == 0) for (int k = 0; k < 16; k++) C[k] = d4[k];
        else if (s == 1) for (int k = 0; k < 16; k++) C[k] = (uint16_t)k;
        else for (int k = 0; k < 16; k++) C[k] = (uint16_t)((k & 3) * 4 + (k >> 2));
      } else for (int k = 0; k < 64; k++) C[k] = d8[k];
    }
  }
}

static const int rq_grid_cap[4] = { 8192, 4096, 2048, 1024 };             // blocks per size class: bounds the work area (about 5 GB at the caps)
static int rq_blocks(int n, int cls) { const int b = (n + 63) / 64; return b < rq_grid_cap[cls] ? b : rq_grid_cap[cls]; }
size_t hop_rdoq_work_bytes(int n) {
  size_t b = 256 + (((size_t)4 * n * sizeof(int)) + 255 & ~(size_t)255);
  for (int cls = 0; cls < 4; cls++) b += (size_t)rq_blocks(n, cls) * 64 * ((size_t)16 << (2 * cls)) * RQ_WORK_PER_COEF;
  return b;
}

int hop_launch_rdoq(hop_ctx* c, int n, const hop_rdoq_job* d_jobs, const hop_estbits* d_tables, const int32_t* d_src, int32_t* d_dst, uint32_t* d_abs_sum,
                    void* d_work) {
  char* w = (char*)d_work;
  int* counts = (int*)w; int* list = (int*)(w + 256);
  char* area = w + 256 + (((size_t)4 * n * sizeof(int)) + 255 & ~(size_t)255);
  const int pr = hop_prof_begin(c, HOP_K_RDOQ, (uint64_t)n);
  (void)hipMemsetAsync(counts, 0, 256, c->stream);
  hipLaunchKernelGGL(k_rdoq_classify, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, counts, list);
  // one launch per TU size class, one lane per TU; a block walks the wave-groups blockIdx, blockIdx + grid, ... of its class
  char* a2 = area; char* a3 = a2 + (size_t)rq_blocks(n, 0) * 64 * 16 * RQ_WORK_PER_COEF; char* a4 = a3 + (size_t)rq_blocks(n, 1) * 64 * 64 * RQ_WORK_PER_COEF;
  char* a5 = a4 + (size_t)rq_blocks(n, 2) * 64 * 256 * RQ_WORK_PER_COEF;
  hipLaunchKernelGGL((k_rdoq<5>), dim3(rq_blocks(n, 3)), dim3(64), 0, c->stream, d_jobs, d_tables, c->rdoq_scans, d_src, d_dst, d_abs_sum, list + (size_t)3 * n, counts + 3, a5);
  hipLaunchKernelGGL((k_rdoq<4>), dim3(rq_blocks(n, 2)), dim3(64), 0, c->stream, d_jobs, d_tables, c->rdoq_scans, d_src, d_dst, d_abs_sum, list + (size_t)2 * n, counts + 2, a4);
  hipLaunchKernelGGL((k_rdoq<3>), dim3(rq_blocks(n, 1)), dim3(64), 0, c->stream, d_jobs, d_tables, c->rdoq_scans, d_src, d_dst, d_abs_sum, list + (size_t)1 * n, counts + 1, a3);
  hipLaunchKernelGGL((k_rdoq<2>), dim3(rq_blocks(n, 0)), dim3(64), 0, c->stream, d_jobs, d_tables, c->rdoq_scans, d_src, d_dst, d_abs_sum, list, counts, a2);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "rdoq launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
