/*
 * hop_oracle_tq.c -- CPU restatement of the transform / flat-quantisation part of the hot path
 * (SURVEY.md section 8(a) rows a9, a10).  TEST INFRASTRUCTURE ONLY -- same rules as hop_oracle.c.
 *
 * Reference: zinsayon/HEVC-HOP (HM-15.0 fork), paths relative to /root/reference/source/Lib.
 * Pinned against the reference's own xTrMxN / xITrMxN (TLibCommon/TComTrQuant.cpp:786,829, free functions with
 * external linkage, called through oracle/_ref/libref_harness.so), TComTrQuant::xDeQuant (:1124-1183) and, for the
 * forward flat quantiser, TComTrQuant::xQuant with RDOQ switched off (:1022-1119, ref_quant_flat of the harness).
 * The transform-skip pair and every other function here also stand in for the reference's members inside the
 * reference encoder (oracle/enc_shim.cpp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hop_oracle.h"

/* The HEVC core transform matrix (TLibCommon/TComRom.cpp:174-249).  All four sizes are sub-sampled rows of the
 * 32-point matrix, and the 32-point matrix is fixed by its first column through the cosine symmetries:
 * T32[k][n] = +-a[theta], theta = (2n+1)k mod 128 folded into 0..32. */
static const int16_t kA[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
int hop_o_dct_coef(int N, int k, int n)
{
  int k32 = k * (32 / N);
  int th = ((2 * n + 1) * k32) & 127;
  if (k32 == 0) return 64;
  if (th <= 32) return kA[th];
  if (th <= 64) return -kA[64 - th];
  if (th <= 96) return -kA[th - 64];
  return kA[128 - th];
}
/* DST-VII 4x4 (TLibCommon/TComRom.cpp g_as_DST_MAT_4; fastForwardDst/fastInverseDst :426-462 equal the matrix product) */
static const int16_t kDst[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

static int tcoef(int N, int dst, int k, int n) { return dst ? kDst[k][n] : hop_o_dct_coef(N, k, n); }
static int clip16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

/* xTrMxN, TComTrQuant.cpp:786-822 (partialButterfly4/8/16/32 :400,490,563,661; fastForwardDst :426):
 * stage 1: tmp[k][j] = (sum_n T[k][n] block[j][n] + add) >> (log2N - 1 + bd - 8), stored as Short;
 * stage 2: coeff[k2][k] = (sum_j T[k2][j] tmp[k][j] + add) >> (log2N + 6), stored as Short. */
void hop_o_fwd_transform(int bitDepth, const int16_t* block, int16_t* coeff, int N, int useDst)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int s1 = log2N - 1 + bitDepth - 8, s2 = log2N + 6;
  int16_t tmp[32 * 32];
  for (int j = 0; j < N; j++)
    for (int k = 0; k < N; k++) {
      int sum = 0;
      for (int n = 0; n < N; n++) sum += tcoef(N, useDst, k, n) * block[j * N + n];
      tmp[k * N + j] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
    }
  for (int j = 0; j < N; j++)
    for (int k = 0; k < N; k++) {
      int sum = 0;
      for (int n = 0; n < N; n++) sum += tcoef(N, useDst, k, n) * tmp[j * N + n];
      coeff[k * N + j] = (int16_t)((sum + (1 << (s2 - 1))) >> s2);
    }
}

/* xITrMxN, TComTrQuant.cpp:829-863 (partialButterflyInverse* :464,527,614,722; fastInverseDst :445):
 * out[j][n] = Clip3(-32768, 32767, (sum_k T[k][n] in[k][j] + add) >> shift), shifts 7 and 12 - (bd - 8). */
void hop_o_inv_transform(int bitDepth, const int16_t* coeff, int16_t* block, int N, int useDst)
{
  int s1 = 7, s2 = 12 - (bitDepth - 8);
  int16_t tmp[32 * 32];
  for (int j = 0; j < N; j++)
    for (int n = 0; n < N; n++) {
      int sum = 0;
      for (int k = 0; k < N; k++) sum += tcoef(N, useDst, k, n) * coeff[k * N + j];
      tmp[j * N + n] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
    }
  for (int j = 0; j < N; j++)
    for (int n = 0; n < N; n++) {
      int sum = 0;
      for (int k = 0; k < N; k++) sum += tcoef(N, useDst, k, n) * tmp[k * N + j];
      block[j * N + n] = (int16_t)clip16((sum + (1 << (s2 - 1))) >> s2);
    }
}

static const int kQuantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };     /* TComRom.cpp:164-167 */
static const int kInvQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                   /* TComRom.cpp:169-172 */

/* xQuant, flat (non-RDOQ, flat scaling list, no sign-bit hiding) branch, TComTrQuant.cpp:1071-1107 with
 * MaxDeltaQP 0 (cQpBase == m_cQP).  qpScaled = the value setQPforQuant hands to setQpParam (:192-214).
 * Returns uiAcSum. */
uint32_t hop_o_quant_flat(int bitDepth, int qpScaled, int isISlice, const int32_t* coef, int32_t* level, int N)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int per = qpScaled / 6, rem = qpScaled % 6;
  int transformShift = 15 - bitDepth - log2N;
  int qBits = 14 + per + transformShift;
  int add = (isISlice ? 171 : 85) << (qBits - 9);
  uint32_t acSum = 0;
  for (int n = 0; n < N * N; n++) {
    int c = coef[n], sign = c < 0 ? -1 : 1;
    int64_t t = (int64_t)abs(c) * kQuantScales[rem];
    int lv = (int)((t + add) >> qBits);
    acSum += (uint32_t)lv;
    lv *= sign;
    level[n] = clip16(lv);
  }
  return acSum;
}

/* TComTrQuant::signBitHidingHDQ (TComTrQuant.cpp:868-990): per coefficient group of the scan whose first and last non-zero level lie at least SBH_THRESHOLD (4)
 * positions apart, the parity of the group's level sum has to equal the sign of its first non-zero level; where it does not, the level whose change costs least in
 * distortion (deltaU, the quantiser's remainder in 1/256 steps) moves by one.  No rate is considered. */
const uint32_t* hop_o_scan(int scan_idx, int log2_size);
static void sign_bit_hiding_hdq(int32_t* q, const int32_t* coef, const uint32_t* scan, const int* deltaU, int N)
{
  int lastCG = -1;
  for (int subSet = (N * N - 1) >> 4; subSet >= 0; subSet--) {
    const int subPos = subSet << 4;
    int firstNZ = 16, lastNZ = -1, absSum = 0, n;
    for (n = 15; n >= 0; --n) if (q[scan[n + subPos]]) { lastNZ = n; break; }
    for (n = 0; n < 16; n++) if (q[scan[n + subPos]]) { firstNZ = n; break; }
    for (n = firstNZ; n <= lastNZ; n++) absSum += q[scan[n + subPos]];
    if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
    if (lastNZ - firstNZ >= 4) {
      const unsigned signbit = q[scan[subPos + firstNZ]] > 0 ? 0 : 1;
      if (signbit != (unsigned)(absSum & 1)) {
        int minCostInc = 0x7FFFFFFF, minPos = -1, finalChange = 0, curCost = 0x7FFFFFFF, curChange = 0;
        for (n = (lastCG == 1 ? lastNZ : 15); n >= 0; --n) {
          const uint32_t blkPos = scan[n + subPos];
          if (q[blkPos] != 0) {
            if (deltaU[blkPos] > 0) { curCost = -deltaU[blkPos]; curChange = 1; }
            else if (n == firstNZ && abs(q[blkPos]) == 1) curCost = 0x7FFFFFFF;
            else { curCost = deltaU[blkPos]; curChange = -1; }
          } else if (n < firstNZ) {
            const unsigned thisSign = coef[blkPos] >= 0 ? 0 : 1;
            if (thisSign != signbit) curCost = 0x7FFFFFFF;
            else { curCost = -deltaU[blkPos]; curChange = 1; }
          } else { curCost = -deltaU[blkPos]; curChange = 1; }
          if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = (int)blkPos; }
        }
        if (q[minPos] == 32767 || q[minPos] == -32768) finalChange = -1;
        if (coef[minPos] >= 0) q[minPos] += finalChange; else q[minPos] -= finalChange;
      }
    }
    if (lastCG == 1) lastCG = 0;
  }
}

/* xQuant, the non-RDOQ branch with sign-bit hiding (TComTrQuant.cpp:1071-1116): the flat quantiser above, the remainders deltaU (:1100), and signBitHidingHDQ along
 * the TU's scan when the level sum is at least 2.  Returns uiAcSum (the sum before the hiding, as the reference reports it). */
uint32_t hop_o_quant_flat_sbh(int bitDepth, int qpScaled, int isISlice, const int32_t* coef, int32_t* level, int N, int scan_idx)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int per = qpScaled / 6, rem = qpScaled % 6;
  int transformShift = 15 - bitDepth - log2N;
  int qBits = 14 + per + transformShift, qBits8 = qBits - 8;
  int add = (isISlice ? 171 : 85) << (qBits - 9);
  uint32_t acSum = 0;
  int deltaU[32 * 32];
  for (int n = 0; n < N * N; n++) {
    int c = coef[n], sign = c < 0 ? -1 : 1;
    int64_t t = (int64_t)abs(c) * kQuantScales[rem];
    int lv = (int)((t + add) >> qBits);
    deltaU[n] = (int)((t - ((int64_t)lv << qBits)) >> qBits8);   /* the reference shifts an Int: lv << qBits stays below 2^31 for 16-bit levels and qBits <= 29 only in its own use; kept 64-bit exact here, equal wherever the reference's does not overflow */
    acSum += (uint32_t)lv;
    lv *= sign;
    level[n] = clip16(lv);
  }
  if (acSum >= 2) sign_bit_hiding_hdq(level, coef, hop_o_scan(scan_idx, log2N), deltaU, N);
  return acSum;
}

/* xDeQuant, flat scaling list branch, TComTrQuant.cpp:1171-1182 */
void hop_o_dequant_flat(int bitDepth, int qpScaled, const int32_t* level, int32_t* coef, int N)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int per = qpScaled / 6, rem = qpScaled % 6;
  int transformShift = 15 - bitDepth - log2N;
  int shift = 20 - 14 - transformShift;
  int add = 1 << (shift - 1);
  int scale = kInvQuantScales[rem] << per;
  for (int n = 0; n < N * N; n++) {
    int q = clip16(level[n]);
    coef[n] = clip16((q * scale + add) >> shift);
  }
}

/* xTransformSkip / xITransformSkip (shift >= 0 branches, bit depth <= 13), TComTrQuant.cpp:1402-1420, :1442-1460 */
void hop_o_transform_skip(int bitDepth, const int16_t* resi, int32_t* coef, int N)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int shift = 15 - bitDepth - log2N;
  for (int n = 0; n < N * N; n++) coef[n] = resi[n] * (1 << shift);
}
void hop_o_inv_transform_skip(int bitDepth, const int32_t* coef, int16_t* resi, int N)
{
  int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int shift = 15 - bitDepth - log2N;
  for (int n = 0; n < N * N; n++) resi[n] = (int16_t)((coef[n] + (1 << (shift - 1))) >> shift);
}

/* One transform unit through the chain of xIntraCodingLumaBlk / xEstimateResidualQT with RDOQ off:
 * residual = org - pred (TEncSearch.cpp:1082-1096 / :6688), transformNxN (xT + flat xQuant, TComTrQuant.cpp:1204-1258),
 * invtransformNxN (xDeQuant + xIT, :1260-1283), recon = Clip(pred + resi) (TEncSearch.cpp:1128-1151),
 * SSE(org, recon) (:1160).  All blocks contiguous NxN.  Returns uiAbsSum; *sse receives the distortion. */
uint32_t hop_o_tu_roundtrip_sbh(int bitDepth, int qpScaled, int isISlice, int useDst, int transformSkip, int N, int signHide, int scanIdx,
                                const int16_t* org, const int16_t* pred, int32_t* level, int16_t* recon, uint32_t* sse);
uint32_t hop_o_tu_roundtrip(int bitDepth, int qpScaled, int isISlice, int useDst, int transformSkip, int N,
                            const int16_t* org, const int16_t* pred, int32_t* level, int16_t* recon, uint32_t* sse)
{ return hop_o_tu_roundtrip_sbh(bitDepth, qpScaled, isISlice, useDst, transformSkip, N, 0, 0, org, pred, level, recon, sse); }
/* the same with the PPS's sign_data_hiding flag (signHide) along scan scanIdx */
uint32_t hop_o_tu_roundtrip_sbh(int bitDepth, int qpScaled, int isISlice, int useDst, int transformSkip, int N, int signHide, int scanIdx,
                                const int16_t* org, const int16_t* pred, int32_t* level, int16_t* recon, uint32_t* sse)
{
  int16_t resi[32 * 32] = {0}, c16[32 * 32], r2[32 * 32];
  int32_t c32[32 * 32], dq[32 * 32];
  for (int i = 0; i < N * N; i++) resi[i] = (int16_t)(org[i] - pred[i]);
  if (transformSkip) hop_o_transform_skip(bitDepth, resi, c32, N);
  else { hop_o_fwd_transform(bitDepth, resi, c16, N, useDst); for (int i = 0; i < N * N; i++) c32[i] = c16[i]; }
  uint32_t absSum = signHide ? hop_o_quant_flat_sbh(bitDepth, qpScaled, isISlice, c32, level, N, scanIdx) : hop_o_quant_flat(bitDepth, qpScaled, isISlice, c32, level, N);
  hop_o_dequant_flat(bitDepth, qpScaled, level, dq, N);
  if (transformSkip) hop_o_inv_transform_skip(bitDepth, dq, r2, N);
  else { for (int i = 0; i < N * N; i++) c16[i] = (int16_t)dq[i]; hop_o_inv_transform(bitDepth, c16, r2, N, useDst); }
  int maxVal = (1 << bitDepth) - 1;
  uint32_t d = 0, sh = (uint32_t)((bitDepth - 8) << 1);
  for (int i = 0; i < N * N; i++) {
    int v = pred[i] + r2[i];
    v = v < 0 ? 0 : v > maxVal ? maxVal : v;
    recon[i] = (int16_t)v;
    int e = org[i] - v;
    d += (uint32_t)(e * e) >> sh;
  }
  *sse = d;
  return absSum;
}
