// k_tq.hip -- transform unit round trip: residual -> forward DCT/DST -> flat
// quantisation -> dequantisation -> inverse transform -> reconstruction clip -> SSE.
// Replaces, for RDOQ off / flat scaling list / no sign-bit hiding, the chain of TEncSearch::xIntraCodingLumaBlk
// (TLibEncoder/TEncSearch.cpp:1082-1160) and xEstimateResidualQT (:6912-7015):
//   TComTrQuant::transformNxN (TLibCommon/TComTrQuant.cpp:1204-1258) = xT (:1341, xTrMxN :786-822 = partialButterfly4/8/16/32
//   :400,490,563,661 / fastForwardDst :426) or xTransformSkip (:1402) + xQuant flat branch (:1071-1107);
//   invtransformNxN (:1260-1283) = xDeQuant (:1124-1183) + xIT (:1370, xITrMxN :829-863) or xITransformSkip (:1442).
// The partial butterflies are exact refactorings of the integer matrix products (no intermediate rounding), so
// each stage is evaluated as sum_n T[k][n] x[n] followed by the stage's rounding shift -- bit-exact.
// One workgroup per TU (<= 32x32); all stages through LDS; ~2 x 2 x N^3 integer MACs, far below 1 % of a step,
// so plain VALU (an exact MFMA form would be v_mfma_f64 or an i8 split).
#include "hop_dev.h"

#include "k_turd_dev.inl"
__constant__ int c_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // TComRom.cpp:164-167

struct TqShared { int16_t a[32 * 32]; int16_t b[32 * 32]; int32_t lv[32 * 32]; int16_t T[32 * 32]; unsigned int acc[2]; int32_t q[32 * 32]; int32_t du[32 * 32]; };   // q / du: levels and the quantiser's remainders (sign-bit hiding)

__global__ __launch_bounds__(256) void k_tu_roundtrip(const hop_tu_job* __restrict__ jobs, hop_pics pic, const uint16_t* __restrict__ scans, int16_t* __restrict__ rec_y,
                                                      int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr,
                                                      hop_tu_result* __restrict__ res, int32_t* __restrict__ levels, const int64_t* __restrict__ level_off) {
  __shared__ TqShared sh;
  const hop_tu_job jb = jobs[blockIdx.x];
  const int tid = threadIdx.x, N = 1 << jb.log2_size, NN = N * N, log2N = jb.log2_size;
  const bool chroma = jb.comp != 0;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  int16_t* rec = (jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0;
  const bool dst = jb.use_dst && N == 4;
  if (tid < 2) sh.acc[tid] = 0;
  // transform matrix of this size into LDS
  for (int i = tid; i < NN; i += 256) { int k = i >> log2N, n = i & (N - 1); sh.T[i] = (int16_t)(dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n)); }
  // residual (TEncSearch.cpp:1082-1096)
  for (int i = tid; i < NN; i += 256) { int r = i >> log2N, c = i & (N - 1); sh.a[i] = (int16_t)(org[(size_t)r * pitch + c] - prd[(size_t)r * pitch + c]); }
  __syncthreads();
  const int transformShift = 15 - bd - log2N;                        // MAX_TR_DYNAMIC_RANGE - bitDepth - log2
  if (jb.transform_skip) {                                           // xTransformSkip :1402-1420
    for (int i = tid; i < NN; i += 256) sh.lv[i] = (int)sh.a[i] * (1 << transformShift);
  } else {
    const int s1 = log2N - 1 + bd - 8, s2 = log2N + 6;               // xTrMxN :788-789
    for (int i = tid; i < NN; i += 256) {                            // stage 1: b[k][j] = (sum_n T[k][n] a[j][n] + add) >> s1
      int k = i >> log2N, j = i & (N - 1), sum = 0;
      for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.a[j * N + n];
      sh.b[k * N + j] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
    }
    __syncthreads();
    for (int i = tid; i < NN; i += 256) {                            // stage 2: coeff[k][j] = (sum_n T[k][n] b[j][n] + add) >> s2
      int k = i >> log2N, j = i & (N - 1), sum = 0;
      for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.b[j * N + n];
      sh.lv[k * N + j] = (int)(int16_t)((sum + (1 << (s2 - 1))) >> s2);
    }
  }
  __syncthreads();
  // flat quantiser :1079-1107 (with the remainders deltaU :1100), sign-bit hiding :868-990 if asked for, dequantiser :1171-1182
  {
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6;
    const int qBits = 14 + per + transformShift;
    const long long add = (long long)(jb.is_i_slice ? 171 : 85) << (qBits - 9);
    const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;
    unsigned int part = 0;
    for (int i = tid; i < NN; i += 256) {
      int c = sh.lv[i], sign = c < 0 ? -1 : 1;
      long long t = (long long)(c < 0 ? -c : c) * c_quant_scales[rem];
      int lv = (int)((t + add) >> qBits);
      sh.du[i] = (int)((t - ((long long)lv << qBits)) >> (qBits - 8));
      part += (unsigned)lv;
      sh.q[i] = clip16(lv * sign);
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if ((tid & 63) == 0) atomicAdd(&sh.acc[0], part);
    __syncthreads();
    if (jb.sign_hide && sh.acc[0] >= 2 && tid == 0) {                  // signBitHidingHDQ: a serial walk over the coefficient groups of the scan (rows a10's rarely used corner)
      const uint16_t* scan = scans + jb.scan_idx * 1360 + (log2N == 2 ? 0 : log2N == 3 ? 16 : log2N == 4 ? 80 : 336);
      int lastCG = -1;
      for (int subSet = (NN - 1) >> 4; subSet >= 0; subSet--) {
        const int subPos = subSet << 4;
        int firstNZ = 16, lastNZ = -1, absSum = 0, n;
        for (n = 15; n >= 0; --n) if (sh.q[scan[n + subPos]]) { lastNZ = n; break; }
        for (n = 0; n < 16; n++) if (sh.q[scan[n + subPos]]) { firstNZ = n; break; }
        for (n = firstNZ; n <= lastNZ; n++) absSum += sh.q[scan[n + subPos]];
        if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
        if (lastNZ - firstNZ >= 4) {
          const unsigned signbit = sh.q[scan[subPos + firstNZ]] > 0 ? 0 : 1;
          if (signbit != (unsigned)(absSum & 1)) {
            int minCostInc = 0x7FFFFFFF, minPos = -1, finalChange = 0, curCost = 0x7FFFFFFF, curChange = 0;
            for (n = (lastCG == 1 ? lastNZ : 15); n >= 0; --n) {
              const int blkPos = scan[n + subPos];
              const int qv = sh.q[blkPos], du = sh.du[blkPos];
              if (qv != 0) {
                if (du > 0) { curCost = -du; curChange = 1; }
                else if (n == firstNZ && (qv == 1 || qv == -1)) curCost = 0x7FFFFFFF;
                else { curCost = du; curChange = -1; }
              } else if (n < firstNZ) {
                const unsigned thisSign = sh.lv[blkPos] >= 0 ? 0 : 1;
                if (thisSign != signbit) curCost = 0x7FFFFFFF;
                else { curCost = -du; curChange = 1; }
              } else { curCost = -du; curChange = 1; }
              if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = blkPos; }
            }
            if (sh.q[minPos] == 32767 || sh.q[minPos] == -32768) finalChange = -1;
            if (sh.lv[minPos] >= 0) sh.q[minPos] += finalChange; else sh.q[minPos] -= finalChange;
          }
        }
        if (lastCG == 1) lastCG = 0;
      }
    }
    __syncthreads();
    for (int i = tid; i < NN; i += 256) {
      const int lv = sh.q[i];
      if (levels) levels[level_off[blockIdx.x] + i] = lv;
      sh.lv[i] = clip16((lv * scale + dadd) >> dshift);              // dequantised coefficient
    }
  }
  __syncthreads();
  if (jb.transform_skip) {                                           // xITransformSkip :1442-1460
    for (int i = tid; i < NN; i += 256) sh.a[i] = (int16_t)((sh.lv[i] + (1 << (transformShift - 1))) >> transformShift);
  } else {
    const int s1 = 7, s2 = 12 - (bd - 8);                            // SHIFT_INV_1ST / SHIFT_INV_2ND, TComRom.h:99-100
    for (int i = tid; i < NN; i += 256) {                            // b[j][n] = clip((sum_k T[k][n] c[k][j] + add) >> 7)
      int j = i >> log2N, n = i & (N - 1), sum = 0;
      for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * (int)(int16_t)sh.lv[k * N + j];
      sh.b[j * N + n] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
    }
    __syncthreads();
    for (int i = tid; i < NN; i += 256) {
      int j = i >> log2N, n = i & (N - 1), sum = 0;
      for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * sh.b[k * N + j];
      sh.a[j * N + n] = (int16_t)clip16((sum + (1 << (s2 - 1))) >> s2);
    }
  }
  __syncthreads();
  // reconstruction + SSE (TEncSearch.cpp:1128-1160)
  {
    const int maxVal = (1 << bd) - 1;
    const unsigned sshift = (unsigned)((bd - 8) << 1);
    unsigned int part = 0;
    for (int i = tid; i < NN; i += 256) {
      int r = i >> log2N, c = i & (N - 1);
      int v = min(maxVal, max(0, (int)prd[(size_t)r * pitch + c] + (int)sh.a[i]));
      rec[(size_t)r * pitch + c] = (int16_t)v;
      int e = (int)org[(size_t)r * pitch + c] - v;
      part += (unsigned)(e * e) >> sshift;
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if ((tid & 63) == 0) atomicAdd(&sh.acc[1], part);
  }
  __syncthreads();
  if (tid == 0) { res[blockIdx.x].abs_sum = sh.acc[0]; res[blockIdx.x].sse = sh.acc[1]; }
}

int hop_launch_tu(hop_ctx* c, int n, const hop_tu_job* d_jobs, hop_tu_result* d_res, int32_t* d_levels, const int64_t* d_level_off) {
  const int pr = hop_prof_begin(c, HOP_K_TQ, (uint64_t)n);
  hipLaunchKernelGGL(k_tu_roundtrip, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), c->rdoq_scans, c->rec[0], c->rec[1], c->rec[2], d_res, d_levels, d_level_off);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "tu_roundtrip launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// Row a8b, leaf step: the evaluation of one component TU inside TEncSearch::xEstimateResidualQT
// (TLibEncoder/TEncSearch.cpp:6896-7200) as a pipeline of kernels over a batch of TUs:
//   k_turd_forward   residual = original - prediction picture, xT (:6912 first half), zero-residual distortion (:6984)
//   k_turd_setup     TEncSbac::estBit from the TU's context snapshot (:6901-6904) + the job records of the next two stages
//   k_rdoq           xRateDistOptQuant (second half of transformNxN)                                   [k_rdoq.hip]
//   k_coeff_bits     cbf flag + levels through the counting coder, from the snapshot (:6957-6962)        [k_cabac.hip]
//   k_turd_inverse   xDeQuant + xIT of the levels (:6998), distortion against the residual (:7000)
//   k_turd_decide    integer bits, calcRdCost of coding vs. cbf = 0 (:7008-7032), the choice; levels zeroed if cbf = 0 wins
// The default transform only: the 4x4 transform-skip retry (:7210-7440), the chroma-of-4x4-luma merging and the split
// recursion stay with the caller.
// =====================================================================================================================

// ---- the kernels of the staged pipeline (batches of thousands of TUs: hop_launch_tu_rd with n above the fused kernel's range) ----
__global__ __launch_bounds__(256) void k_turd_forward(const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,                                                       int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) { __shared__ TurdFwdShared sh; turd_forward_body(sh, blockIdx.x, jobs, pic, coef_off, coef, zero_sse); }
__global__ __launch_bounds__(256) void k_turd_inverse(const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,                                                       const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,                                                       int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) { __shared__ TurdFwdShared sh; turd_inverse_body(sh, blockIdx.x, jobs, pic, coef_off, levels, abs_sum, nz_sse, rec_y, rec_cb, rec_cr); }
__global__ __launch_bounds__(256) void k_turd_forward_small(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,                                                             int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) { __shared__ TurdSmallShared sh; turd_forward_small_body(sh, threadIdx.x >> 6, threadIdx.x & 63, blockIdx.x * 4 + (threadIdx.x >> 6), jobs, n, pic, coef_off, coef, zero_sse); }
__global__ __launch_bounds__(256) void k_turd_inverse_small(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,                                                             const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,                                                             int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) { __shared__ TurdSmallShared sh; turd_inverse_small_body(sh, threadIdx.x >> 6, threadIdx.x & 63, blockIdx.x * 4 + (threadIdx.x >> 6), jobs, n, pic, coef_off, levels, abs_sum, nz_sse, rec_y, rec_cb, rec_cr); }
__global__ void k_turd_setup(const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,                              const int32_t* __restrict__ entropy_bits, hop_estbits* __restrict__ tables, hop_rdoq_job* __restrict__ rq, hop_coeff_bits_job* __restrict__ cb) { turd_setup_body(blockIdx.x * blockDim.x + threadIdx.x, jobs, n, ctx_in, coef_off, entropy_bits, tables + (blockIdx.x * blockDim.x + threadIdx.x), rq, cb); }
__global__ void k_turd_decide(const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,                               const int32_t* __restrict__ entropy_bits, const uint32_t* __restrict__ abs_sum, const unsigned long long* __restrict__ frac,                               const uint32_t* __restrict__ zero_sse, const uint32_t* __restrict__ nz_sse, int32_t* __restrict__ levels,                               hop_tu_rd_result* __restrict__ res) { turd_decide_body(blockIdx.x * blockDim.x + threadIdx.x, jobs, n, ctx_in, coef_off, entropy_bits, abs_sum, frac, zero_sse, nz_sse, levels, res); }

// size_hint: 0 = transform sizes unknown / mixed, 1 = every TU is 4x4 or 8x8 (one wave per TU), 2 = every TU is 16x16 or 32x32
int hop_launch_tu_rd(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_res, int size_hint) {
  // (a wave per small TU stays the better form far beyond the workgroup-per-TU one's range: 8 x as many fit a compute unit, and the lanes of the staged form diverge)
  if (n <= c->fused_leaf_max || (size_hint == 1 && c->fused_leaf_max > 0 && n <= 8 * c->fused_leaf_max)) return hop_launch_tu_rd_fused(c, n, d_jobs, d_ctx, d_coef_off, n_coeff, d_levels, d_res, size_hint == 1);   // launch-bound batches: one kernel
  // scratch: coefficients, zero / non-zero SSE, abs sums, counted bits, the bit-estimate tables and the job records of the inner stages
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_coef = 0, o_zs = al(o_coef + n_coeff * 4), o_ns = al(o_zs + (size_t)n * 4), o_as = al(o_ns + (size_t)n * 4), o_fr = al(o_as + (size_t)n * 4);
  const size_t o_tab = al(o_fr + (size_t)n * 8), o_rq = al(o_tab + (size_t)n * sizeof(hop_estbits)), o_cb = al(o_rq + (size_t)n * sizeof(hop_rdoq_job));
  const size_t o_wk = al(o_cb + (size_t)n * sizeof(hop_coeff_bits_job));
  void* sc; int r = hop_scratch(c, o_wk + hop_rdoq_work_bytes(n) + 256, &sc); if (r) return r;
  char* b = (char*)sc;
  int32_t* coef = (int32_t*)(b + o_coef); uint32_t* zs = (uint32_t*)(b + o_zs); uint32_t* ns = (uint32_t*)(b + o_ns); uint32_t* as = (uint32_t*)(b + o_as);
  unsigned long long* fr = (unsigned long long*)(b + o_fr); hop_estbits* tab = (hop_estbits*)(b + o_tab);
  hop_rdoq_job* rq = (hop_rdoq_job*)(b + o_rq); hop_coeff_bits_job* cb = (hop_coeff_bits_job*)(b + o_cb);
  hop_pics pic = hop_make_pics(c);
  const int pr = hop_prof_begin(c, HOP_K_TQ, (uint64_t)n);
  if (size_hint != 1) hipLaunchKernelGGL(k_turd_forward, dim3(n), dim3(256), 0, c->stream, d_jobs, pic, d_coef_off, coef, zs);
  if (size_hint != 2) hipLaunchKernelGGL(k_turd_forward_small, dim3((n + 3) / 4), dim3(256), 0, c->stream, d_jobs, n, pic, d_coef_off, coef, zs);
  hipLaunchKernelGGL(k_turd_setup, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_jobs, n, d_ctx, d_coef_off, hop_entropy_bits_device(c), tab, rq, cb);
  hop_prof_end(c, pr);
  r = hop_launch_rdoq(c, n, rq, tab, coef, d_levels, as, b + o_wk); if (r) return r;
  r = hop_launch_coeff_bits(c, n, cb, d_ctx, d_levels, fr, nullptr); if (r) return r;
  const int pr2 = hop_prof_begin(c, HOP_K_TQ, 0);
  if (size_hint != 1) hipLaunchKernelGGL(k_turd_inverse, dim3(n), dim3(256), 0, c->stream, d_jobs, pic, d_coef_off, d_levels, as, ns, c->rec[0], c->rec[1], c->rec[2]);
  if (size_hint != 2) hipLaunchKernelGGL(k_turd_inverse_small, dim3((n + 3) / 4), dim3(256), 0, c->stream, d_jobs, n, pic, d_coef_off, d_levels, as, ns, c->rec[0], c->rec[1], c->rec[2]);
  hipLaunchKernelGGL(k_turd_decide, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_jobs, n, d_ctx, d_coef_off, hop_entropy_bits_device(c), as, fr, zs, ns, d_levels, d_res);
  hop_prof_end(c, pr2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "tu_rd launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// the inverse path alone in reconstruction mode (is_intra jobs: Clip(prediction + residual) into the reconstruction picture, SSE against the
// original); slots with log2_size < 2 are empty
int hop_launch_tu_recon(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const int64_t* d_coef_off, const int32_t* d_levels, const uint32_t* d_abs_sum, uint32_t* d_sse) {
  hop_pics pic = hop_make_pics(c);
  const int pr = hop_prof_begin(c, HOP_K_TQ, (uint64_t)n);
  hipLaunchKernelGGL(k_turd_inverse, dim3(n), dim3(256), 0, c->stream, d_jobs, pic, d_coef_off, d_levels, d_abs_sum, d_sse, c->rec[0], c->rec[1], c->rec[2]);
  hipLaunchKernelGGL(k_turd_inverse_small, dim3((n + 3) / 4), dim3(256), 0, c->stream, d_jobs, n, pic, d_coef_off, d_levels, d_abs_sum, d_sse, c->rec[0], c->rec[1], c->rec[2]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "tu recon launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
