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
