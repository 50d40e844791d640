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

// first column of the 32-point core transform; every entry of every size follows from the cosine symmetries
__constant__ int16_t c_dct_a[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                     61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
__constant__ int16_t c_dst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
__constant__ int c_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // TComRom.cpp:164-167
__constant__ int c_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                    // TComRom.cpp:169-172

__device__ static inline int dct_coef(int step32, int k, int n) {     // T_N[k][n] = T32[k * 32/N][n]
  const int k32 = k * step32;
  if (k32 == 0) return 64;
  const int th = ((2 * n + 1) * k32) & 127;
  if (th <= 32) return c_dct_a[th];
  if (th <= 64) return -c_dct_a[64 - th];
  if (th <= 96) return -c_dct_a[th - 64];
  return c_dct_a[128 - th];
}
__device__ static inline int clip16(int v) { return min(32767, max(-32768, v)); }

struct TqShared { int16_t a[32 * 32]; int16_t b[32 * 32]; int32_t lv[32 * 32]; int16_t T[32 * 32]; unsigned int acc[2]; };

__global__ __launch_bounds__(256) void k_tu_roundtrip(const hop_tu_job* __restrict__ jobs, hop_pics pic, int16_t* __restrict__ rec_y,
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
  // flat quantiser :1079-1107 and dequantiser :1171-1182
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
      part += (unsigned)lv;
      lv = clip16(lv * sign);
      if (levels) levels[level_off[blockIdx.x] + i] = lv;
      sh.lv[i] = clip16((lv * scale + dadd) >> dshift);              // dequantised coefficient
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if ((tid & 63) == 0) atomicAdd(&sh.acc[0], part);
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
  hipLaunchKernelGGL(k_tu_roundtrip, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), c->rec[0], c->rec[1], c->rec[2], d_res, d_levels, d_level_off);
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
struct TurdFwdShared { int16_t a[32 * 32]; int16_t b[32 * 32]; int16_t T[32 * 32]; unsigned int acc; };

__global__ __launch_bounds__(256) void k_turd_forward(const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                      int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) {
  __shared__ TurdFwdShared sh;
  const hop_tu_rd_job jb = jobs[blockIdx.x];
  if (jb.log2_size <= 3) return;                                   // k_turd_forward_small takes it
  const int tid = threadIdx.x, log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  if (tid == 0) sh.acc = 0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  for (int i = tid; i < NN; i += 256) { int k = i >> log2N, n = i & (N - 1); sh.T[i] = (int16_t)(dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n)); }
  __syncthreads();
  unsigned part = 0;
  const unsigned sshift = (unsigned)((bd - 8) << 1);
  for (int i = tid; i < NN; i += 256) {
    int r = i >> log2N, c = i & (N - 1);
    const int e = (int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c];
    sh.a[i] = (int16_t)e;
    part += (unsigned)(e * e) >> sshift;                          // getDistPart(zero block, residual), SSE
  }
  part = (unsigned)hopd_wave_sum((int)part);
  if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
  __syncthreads();
  int32_t* out = coef + coef_off[blockIdx.x];
  if (jb.flags & HOP_TU_RD_TS) {                                     // xTransformSkip, TComTrQuant.cpp:1402-1420 (shift >= 0 for bit depths <= 13)
    const int shift = 15 - bd - log2N;
    for (int i = tid; i < NN; i += 256) out[i] = (int)sh.a[i] * (1 << shift);
    if (tid == 0) zero_sse[blockIdx.x] = sh.acc;
    return;
  }
  const int s1 = log2N - 1 + bd - 8, s2 = log2N + 6;               // xTrMxN :788-789
  for (int i = tid; i < NN; i += 256) {
    int k = i >> log2N, j = i & (N - 1), sum = 0;
    for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.a[j * N + n];
    sh.b[k * N + j] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
  }
  __syncthreads();
  for (int i = tid; i < NN; i += 256) {
    int k = i >> log2N, j = i & (N - 1), sum = 0;
    for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.b[j * N + n];
    out[k * N + j] = (int)(int16_t)((sum + (1 << (s2 - 1))) >> s2);
  }
  if (tid == 0) zero_sse[blockIdx.x] = sh.acc;
}

__global__ __launch_bounds__(256) void k_turd_inverse(const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                      const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,
                                                      int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  __shared__ TurdFwdShared sh;                                     // a: dequantised / residual, b: intermediate
  const hop_tu_rd_job jb = jobs[blockIdx.x];
  if (jb.log2_size <= 3) return;                                   // k_turd_inverse_small takes it
  if (abs_sum[blockIdx.x] == 0) {
    if (!jb.is_intra) { if (threadIdx.x == 0) nz_sse[blockIdx.x] = 0; return; }
    // reconstruction = prediction (TEncSearch.cpp:1118-1127,1133-1152); its distortion against the original is reported like a coded block's
    const bool ch = jb.comp != 0; const int pt = ch ? pic.pic_w >> 1 : pic.pic_w, xx = ch ? jb.x >> 1 : jb.x, yy = ch ? jb.y >> 1 : jb.y, NNz = 1 << jb.log2_size;
    const int16_t* pp = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)yy * pt + xx;
    const int16_t* oo = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)yy * pt + xx;
    int16_t* rr = (jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)yy * pt + xx;
    if (threadIdx.x == 0) sh.acc = 0;
    __syncthreads();
    unsigned part0 = 0; const unsigned ss0 = (unsigned)(((ch ? pic.bd_c : pic.bd_y) - 8) << 1);
    for (int i = threadIdx.x; i < NNz * NNz; i += 256) {
      const int r = i >> jb.log2_size, c2 = i & (NNz - 1);
      const int v = pp[(size_t)r * pt + c2], e = v - (int)oo[(size_t)r * pt + c2];
      rr[(size_t)r * pt + c2] = (int16_t)v; part0 += (unsigned)(e * e) >> ss0;
    }
    part0 = (unsigned)hopd_wave_sum((int)part0);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh.acc, part0);
    __syncthreads();
    if (threadIdx.x == 0) nz_sse[blockIdx.x] = sh.acc;
    return;
  }
  const int tid = threadIdx.x, log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  if (tid == 0) sh.acc = 0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  for (int i = tid; i < NN; i += 256) { int k = i >> log2N, n = i & (N - 1); sh.T[i] = (int16_t)(dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n)); }
  {
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
    const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;     // xDeQuant :1171-1182
    const int32_t* lv = levels + coef_off[blockIdx.x];
    for (int i = tid; i < NN; i += 256) sh.a[i] = (int16_t)clip16((clip16(lv[i]) * scale + dadd) >> dshift);
  }
  __syncthreads();
  if ((jb.flags & HOP_TU_RD_TS) && !jb.is_intra) {                   // xITransformSkip, TComTrQuant.cpp:1442-1460; the dequantised value is an Int here (no 16-bit clip in between)
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
    const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;
    const int32_t* lv = levels + coef_off[blockIdx.x];
    unsigned part = 0; const unsigned sshift = (unsigned)((bd - 8) << 1);
    for (int i = tid; i < NN; i += 256) {
      const int j = i >> log2N, n = i & (N - 1);
      const int dq = clip16((clip16(lv[i]) * scale + dadd) >> dshift);
      const int rr = (int)(int16_t)((dq + (1 << (transformShift - 1))) >> transformShift);
      const int e = rr - ((int)org[(size_t)j * pitch + n] - (int)prd[(size_t)j * pitch + n]);
      part += (unsigned)(e * e) >> sshift;
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
    __syncthreads();
    if (tid == 0) nz_sse[blockIdx.x] = sh.acc;
    return;
  }
  const int s1 = 7, s2 = 12 - (bd - 8);                            // SHIFT_INV_1ST / SHIFT_INV_2ND
  for (int i = tid; i < NN; i += 256) {
    int j = i >> log2N, n = i & (N - 1), sum = 0;
    for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * sh.a[k * N + j];
    sh.b[j * N + n] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
  }
  __syncthreads();
  unsigned part = 0;
  const unsigned sshift = (unsigned)((bd - 8) << 1);
  for (int i = tid; i < NN; i += 256) {
    int j = i >> log2N, n = i & (N - 1), sum = 0;
    for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * sh.b[k * N + j];
    const int rr = clip16((sum + (1 << (s2 - 1))) >> s2);          // reconstructed residual sample (j, n)
    int e;
    if (jb.is_intra) {                                             // ClipY / ClipC(prediction + residual), distortion against the original (:1133-1158)
      const int v = min((1 << bd) - 1, max(0, (int)prd[(size_t)j * pitch + n] + rr));
      ((jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0)[(size_t)j * pitch + n] = (int16_t)v;
      e = v - (int)org[(size_t)j * pitch + n];
    } else e = rr - ((int)org[(size_t)j * pitch + n] - (int)prd[(size_t)j * pitch + n]);
    part += (unsigned)(e * e) >> sshift;
  }
  part = (unsigned)hopd_wave_sum((int)part);
  if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
  __syncthreads();
  if (tid == 0) nz_sse[blockIdx.x] = sh.acc;
}

// ---- the same two stages for 4x4 and 8x8 TUs: one WAVE per TU (a lane per sample), four TUs per workgroup, no workgroup barrier ----
// (a 256-thread workgroup per 16-sample TU spends its time being scheduled: the residual quadtree of 8x8 CUs is 2 M such TUs per frame)
struct TurdSmallShared { int16_t a[4][64]; int16_t b[4][64]; };
__device__ static inline int turd_t(bool dst, int log2N, int k, int n) { return dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n); }

__global__ __launch_bounds__(256) void k_turd_forward_small(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                            int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) {
  __shared__ TurdSmallShared sh;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * 4 + w;
  if (j >= n) return;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size > 3 || jb.log2_size < 2) return;                // < 2: an empty slot of a job table (k_rqt.inl)
  const int log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0, live = lane < NN;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  const int r = lane >> log2N, c = lane & (N - 1);                 // as sample: row r, column c; as coefficient: k = r, j = c
  int e = 0;
  if (live) e = (int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c];
  sh.a[w][lane] = (int16_t)e;
  const unsigned zs = (unsigned)hopd_wave_sum((int)((unsigned)(e * e) >> (unsigned)((bd - 8) << 1)));
  if (lane == 0) zero_sse[j] = zs;
  int32_t* out = coef + coef_off[j];
  if (jb.flags & HOP_TU_RD_TS) { if (live) out[lane] = e * (1 << (15 - bd - log2N)); return; }      // xTransformSkip
  __builtin_amdgcn_wave_barrier();
  const int s1 = log2N - 1 + bd - 8, s2 = log2N + 6;
  int t[8];
  for (int q = 0; q < N; q++) t[q] = turd_t(dst, log2N, r, q);     // row k = r of the transform matrix
  int sum = 0;
  if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.a[w][c * N + q];
  sh.b[w][r * N + c] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
  __builtin_amdgcn_wave_barrier();
  sum = 0;
  if (live) { for (int q = 0; q < N; q++) sum += t[q] * sh.b[w][c * N + q]; out[r * N + c] = (int)(int16_t)((sum + (1 << (s2 - 1))) >> s2); }
}

__global__ __launch_bounds__(256) void k_turd_inverse_small(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                            const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,
                                                            int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  __shared__ TurdSmallShared sh;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * 4 + w;
  if (j >= n) return;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size > 3 || jb.log2_size < 2) return;                // < 2: an empty slot of a job table (k_rqt.inl)
  const int log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0, live = lane < NN;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  int16_t* rec = (jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0;
  const int r = lane >> log2N, c = lane & (N - 1);
  if (abs_sum[j] == 0) {
    if (!jb.is_intra) { if (lane == 0) nz_sse[j] = 0; return; }
    int e0 = 0;                                                    // reconstruction = prediction; its distortion against the original
    if (live) { const int v = prd[(size_t)r * pitch + c]; rec[(size_t)r * pitch + c] = (int16_t)v; e0 = v - (int)org[(size_t)r * pitch + c]; }
    const unsigned z0 = (unsigned)hopd_wave_sum((int)((unsigned)(e0 * e0) >> (unsigned)((bd - 8) << 1)));
    if (lane == 0) nz_sse[j] = z0;
    return;
  }
  const bool dst = jb.use_dst && N == 4 && !chroma;
  const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
  const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;     // xDeQuant :1171-1182
  const int32_t* lv = levels + coef_off[j];
  const int dq = live ? clip16((clip16(lv[lane]) * scale + dadd) >> dshift) : 0;
  int rr;
  if (jb.flags & HOP_TU_RD_TS) rr = (int)(int16_t)((dq + (1 << (transformShift - 1))) >> transformShift);      // xITransformSkip
  else {
    sh.a[w][lane] = (int16_t)dq;
    __builtin_amdgcn_wave_barrier();
    const int s1 = 7, s2 = 12 - (bd - 8);
    // lane (j2 = r, n2 = c): b[j2][n2] = sum_k T[k][n2] a[k][j2], then rr(j2, n2) = sum_k T[k][n2] b[k][j2]
    int t[8];
    for (int q = 0; q < N; q++) t[q] = turd_t(dst, log2N, q, c);   // column n = c of the transform matrix
    int sum = 0;
    if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.a[w][q * N + r];
    sh.b[w][r * N + c] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
    __builtin_amdgcn_wave_barrier();
    sum = 0;
    if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.b[w][q * N + r];
    rr = clip16((sum + (1 << (s2 - 1))) >> s2);                    // reconstructed residual sample (row r, column c)
  }
  int e = 0;
  if (live) {
    if (jb.is_intra) {
      const int v = min((1 << bd) - 1, max(0, (int)prd[(size_t)r * pitch + c] + rr));
      rec[(size_t)r * pitch + c] = (int16_t)v;
      e = v - (int)org[(size_t)r * pitch + c];
    } else e = rr - ((int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c]);
  }
  const unsigned ns = (unsigned)hopd_wave_sum((int)((unsigned)(e * e) >> (unsigned)((bd - 8) << 1)));
  if (lane == 0) nz_sse[j] = ns;
}

// one thread per TU: the bit-estimate table of its snapshot (TEncSbac::estBit as hop_cabac_est_bits) and the job records
__global__ void k_turd_setup(const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,
                             const int32_t* __restrict__ entropy_bits, hop_estbits* __restrict__ tables, hop_rdoq_job* __restrict__ rq, hop_coeff_bits_job* __restrict__ cb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const hop_tu_rd_job jb = jobs[i];
  const uint8_t* s = ctx_in[jb.ctx_index].state;
  hop_estbits* eb = tables + i;
  const int width = 1 << jb.log2_size, chroma = jb.comp != 0;
  int32_t* w = (int32_t*)eb;
  for (int k = 0; k < (int)(sizeof(hop_estbits) / 4); k++) w[k] = 0;
  for (int k = 0; k < 12; k++) { eb->blockCbpBits[k][0] = entropy_bits[s[k] ^ 0]; eb->blockCbpBits[k][1] = entropy_bits[s[k] ^ 1]; }
  for (int k = 0; k < 4; k++) { eb->blockRootCbpBits[k][0] = entropy_bits[s[11 + k] ^ 0]; eb->blockRootCbpBits[k][1] = entropy_bits[s[11 + k] ^ 1]; }
  for (int k = 0; k < 2; k++) for (int b = 0; b < 2; b++) eb->significantCoeffGroupBits[k][b] = entropy_bits[s[12 + 2 * chroma + k] ^ b];
  int firstCtx = 1, numCtx = 8;
  if (width >= 16) { firstCtx = chroma ? 12 : 21; numCtx = chroma ? 3 : 6; }
  else if (width == 8) { firstCtx = 9; numCtx = chroma ? 3 : 12; }
  const int base = 16 + (chroma ? 27 : 0);
  for (int b = 0; b < 2; b++) eb->significantBits[0][b] = entropy_bits[s[base] ^ b];
  for (int k = firstCtx; k < firstCtx + numCtx; k++) for (int b = 0; b < 2; b++) eb->significantBits[k][b] = entropy_bits[s[base + k] ^ b];
  const int cbt = jb.log2_size - 2;
  const int off = chroma ? 0 : (cbt * 3 + ((cbt + 1) >> 2)), shf = chroma ? cbt : ((cbt + 3) >> 2);
  const uint8_t* px = s + 58 + 15 * chroma; const uint8_t* py = s + 88 + 15 * chroma;
  const int gmax = (width == 4) ? 3 : (width == 8) ? 5 : (width == 16) ? 7 : 9;       // g_uiGroupIdx[width - 1]
  int bitsX = 0, bitsY = 0, c;
  for (c = 0; c < gmax; c++) { const int o = off + (c >> shf); eb->lastXBits[c] = bitsX + entropy_bits[px[o] ^ 0]; bitsX += entropy_bits[px[o] ^ 1]; }
  eb->lastXBits[c] = bitsX;
  for (c = 0; c < gmax; c++) { const int o = off + (c >> shf); eb->lastYBits[c] = bitsY + entropy_bits[py[o] ^ 0]; bitsY += entropy_bits[py[o] ^ 1]; }
  eb->lastYBits[c] = bitsY;
  const int no = chroma ? 8 : 16, na = chroma ? 2 : 4, oo = 118 + (chroma ? 16 : 0), oa = 142 + (chroma ? 4 : 0);
  for (int k = 0; k < no; k++) { eb->greaterOneBits[k][0] = entropy_bits[s[oo + k] ^ 0]; eb->greaterOneBits[k][1] = entropy_bits[s[oo + k] ^ 1]; }
  for (int k = 0; k < na; k++) { eb->levelAbsBits[k][0] = entropy_bits[s[oa + k] ^ 0]; eb->levelAbsBits[k][1] = entropy_bits[s[oa + k] ^ 1]; }
  hop_rdoq_job r;
  r.log2_size = jb.log2_size; r.comp = jb.comp; r.is_intra = jb.is_intra; r.scan_idx = jb.scan_idx; r.tr_depth = jb.tr_depth; r.qp_scaled = jb.qp_scaled;
  r.bit_depth = jb.bit_depth; r.sign_hide = jb.sign_hide; r.lambda = jb.lambda_rdoq; r.coeff_offset = coef_off[i]; r.estbits_index = i; r.reserved = 0;
  rq[i] = r;
  hop_coeff_bits_job b;
  b.log2_size = jb.log2_size; b.comp = jb.comp; b.scan_idx = jb.scan_idx; b.sign_hide = jb.sign_hide; b.use_ts = jb.use_ts; b.ts_flag = (jb.flags & HOP_TU_RD_TS) ? 1 : 0; b.ctx_index = jb.ctx_index;
  b.cbf_ctx_plus1 = 1 + 4 * chroma + (chroma ? jb.tr_depth : (jb.tr_depth == 0 ? 1 : 0));           // getCtxQtCbf, TComDataCU.cpp:1848-1859
  b.coeff_offset = coef_off[i];
  cb[i] = b;
}

// one thread per TU: integer bits, the two RD costs, the choice (TEncSearch.cpp:7004-7032; calcRdCost TComRdCost.cpp:59-111)
__global__ void k_turd_decide(const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,
                              const int32_t* __restrict__ entropy_bits, const uint32_t* __restrict__ abs_sum, const unsigned long long* __restrict__ frac,
                              const uint32_t* __restrict__ zero_sse, const uint32_t* __restrict__ nz_sse, int32_t* __restrict__ levels,
                              hop_tu_rd_result* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const hop_tu_rd_job jb = jobs[i];
  const uint8_t* s = ctx_in[jb.ctx_index].state;
  const uint32_t left = (uint32_t)s[150] | ((uint32_t)s[151] << 8);                                  // fraction below one bit the coder carries
  const int chroma = jb.comp != 0;
  const uint32_t zeroDist = chroma ? (uint32_t)(int)(jb.dist_weight * zero_sse[i]) : zero_sse[i];   // getDistPart, TComRdCost.cpp:493-502
  const uint32_t nzDist = chroma ? (uint32_t)(int)(jb.dist_weight * nz_sse[i]) : nz_sse[i];
  const uint32_t singleBits = (uint32_t)((left + frac[i]) >> 15);
  hop_tu_rd_result r;
  r.abs_sum = abs_sum[i]; r.zero_dist = zeroDist; r.nonzero_dist = 0; r.bits = singleBits; r.null_bits = 0; r.dist = zeroDist; r.pad = 0;
  if (jb.is_intra) {                                               // xIntraCodingLumaBlk has no cbf-zero test: the block is what RDOQ made of it
    r.dist = r.abs_sum ? nzDist : zeroDist;
    r.nonzero_dist = r.abs_sum ? nzDist : 0;
    r.cost = (double)(uint32_t)floor((double)r.dist + (double)((int)(singleBits * jb.lambda_rd + .5)));
  } else if (r.abs_sum) {
    r.nonzero_dist = nzDist;
    const double singleCost = (double)(uint32_t)floor((double)nzDist + (double)((int)(singleBits * jb.lambda_rd + .5)));
    const int cbfCtx = 4 * chroma + (chroma ? jb.tr_depth : (jb.tr_depth == 0 ? 1 : 0));
    r.null_bits = (uint32_t)((left + (unsigned long long)entropy_bits[s[cbfCtx] ^ 0]) >> 15);       // encodeQtCbfZero from the snapshot
    const double nullCost = (double)(uint32_t)floor((double)zeroDist + (double)((int)(r.null_bits * jb.lambda_rd + .5)));
    if (jb.flags & HOP_TU_RD_KEEP) { r.dist = nzDist; r.cost = singleCost; }       // the transform-skip retry compares this cost itself (:7258-7262)
    else if (nullCost < singleCost) {
      r.abs_sum = 0; r.cost = nullCost;
      int32_t* lv = levels + coef_off[i];
      for (int k = 0; k < (1 << (2 * jb.log2_size)); k++) lv[k] = 0;
    } else { r.dist = nzDist; r.cost = singleCost; }
  } else {
    r.cost = (double)(uint32_t)floor((double)zeroDist + (double)((int)(singleBits * jb.lambda_rd + .5)));
  }
  r.cbf = r.abs_sum != 0;
  res[i] = r;
}

// size_hint: 0 = transform sizes unknown / mixed, 1 = every TU is 4x4 or 8x8 (one wave per TU), 2 = every TU is 16x16 or 32x32
int hop_launch_tu_rd(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_res, int size_hint) {
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
