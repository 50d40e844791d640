/*
 * hop_oracle_intra.c -- CPU restatement of the 35-mode intra rough search.
 * TEST INFRASTRUCTURE ONLY -- same rules as hop_oracle.c.
 *
 * Reference: zinsayon/HEVC-HOP (HM-15.0 fork), paths relative to /root/reference/source/Lib.
 * Pinned through oracle/_ref/libref_harness.so against the reference's own TComPattern::fillReferenceSamples
 * (TLibCommon/TComPattern.cpp:374-558), TComPattern::getPredictorPtr (:583-607), TComPrediction::predIntraLumaAng
 * (TLibCommon/TComPrediction.cpp:340-372: xPredIntraAng :192-338, xPredIntraPlanar :1468-1505, xDCPredFiltering
 * :1521-1541, predIntraGetPredValDC :130-167) and TComRdCost::calcHAD (TComRdCost.cpp:391-425).  The reference-sample
 * smoothing sits inline in TComPattern::initAdiPattern (:237-312), which needs a TComDataCU/TComPic graph; the harness
 * restates those 30 lines too, so the smoothing filter alone is checked restatement-against-restatement.
 *
 * Reference line L[0..4N]: L[2N] = corner p[-1][-1]; L[2N-1-i] = left column sample of row i (i = 0..2N-1, downwards);
 * L[2N+1+i] = top row sample of column i (i = 0..2N-1).  Availability flags are per 4-sample unit in the same order
 * as the reference's bNeighborFlags (index 0 = bottom-most below-left unit, 2U = corner, then above, above-right).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "hop_oracle.h"

uint32_t hop_o_calc_had(const int16_t* a, int sa, const int16_t* b, int sb, int w, int h, int bitDepth);

/* fillReferenceSamples, TComPattern.cpp:374-558 (unit size 4 = g_uiMaxCUWidth >> g_uiMaxCUDepth) */
/* unit: samples per availability flag -- 4 for luma, 2 for the chroma planes (g_uiMaxCUWidth >> g_uiMaxCUDepth, halved: TComPattern.cpp:325-331) */
void hop_o_intra_fill_refs_u(const int16_t* rec, int stride, int x, int y, int N, int unit, const uint8_t* flags, int bitDepth, int* L)
{
  const int U = N / unit, units = 4 * U + 1, dc = 1 << (bitDepth - 1);
  const int16_t* org = rec + (ptrdiff_t)y * stride + x;
  int navail = 0;
  for (int u = 0; u < units; u++) navail += flags[u] ? 1 : 0;
  if (navail == 0) { for (int i = 0; i <= 4 * N; i++) L[i] = dc; return; }
  /* line buffer in unit order; the corner owns a whole unit */
  int line[5 * 64 + 8], ok[4 * 16 + 1];
  for (int i = 0; i < units * unit; i++) line[i] = dc;
  for (int u = 0; u < units; u++) ok[u] = flags[u] != 0;
  if (ok[2 * U]) for (int i = 0; i < unit; i++) line[2 * U * unit + i] = org[-stride - 1];
  for (int u = 0; u < 2 * U; u++)                     /* left + below-left: unit 2U-1-j holds rows 4j..4j+3, stored upwards */
    if (ok[u]) { int j = 2 * U - 1 - u; for (int i = 0; i < unit; i++) line[u * unit + unit - 1 - i] = org[(ptrdiff_t)(unit * j + i) * stride - 1]; }
  for (int u = 2 * U + 1; u < units; u++)             /* above + above-right */
    if (ok[u]) { int j = u - 2 * U - 1; for (int i = 0; i < unit; i++) line[u * unit + i] = org[-stride + unit * j + i]; }
  /* substitution, :505-545 */
  int cur = 0;
  while (cur < units) {
    if (!ok[cur]) {
      if (cur == 0) {
        int nxt = 1;
        while (nxt < units && !ok[nxt]) nxt++;
        int ref = line[nxt * unit];
        while (cur < nxt) { for (int i = 0; i < unit; i++) line[cur * unit + i] = ref; cur++; }
      } else {
        int ref = line[cur * unit - 1];
        for (int i = 0; i < unit; i++) line[cur * unit + i] = ref;
        cur++;
      }
    } else cur++;
  }
  /* copy out, :547-556: left part as is, corner once, top part */
  for (int i = 0; i < 2 * N; i++) L[i] = line[i];
  L[2 * N] = line[2 * U * unit];
  for (int i = 0; i < 2 * N; i++) L[2 * N + 1 + i] = line[(2 * U + 1) * unit + i];
}
void hop_o_intra_fill_refs(const int16_t* rec, int stride, int x, int y, int N, const uint8_t* flags, int bitDepth, int* L)
{
  hop_o_intra_fill_refs_u(rec, stride, x, y, N, 4, flags, bitDepth, L);
}


/* reference smoothing, TComPattern::initAdiPattern :237-299: [1 2 1], or the bilinear "strong" filter for 32x32 */
void hop_o_intra_smooth(const int* L, int N, int bitDepth, int strong, int* F)
{
  const int n = 4 * N + 1;
  if (strong && N >= 32) {
    int bl = L[0], tl = L[2 * N], tr = L[n - 1], thr = 1 << (bitDepth - 5);
    int bilLeft = abs(bl + tl - 2 * L[N]) < thr, bilAbove = abs(tl + tr - 2 * L[2 * N + N]) < thr;
    if (bilLeft && bilAbove) {
      int shift = (N == 32 ? 3 : 4) + 3;             /* g_aucConvertToBit[N] + 3 = log2(2N) */
      F[0] = L[0]; F[2 * N] = L[2 * N]; F[n - 1] = L[n - 1];
      for (int i = 1; i < 2 * N; i++) F[i] = ((2 * N - i) * bl + i * tl + N) >> shift;
      for (int i = 1; i < 2 * N; i++) F[2 * N + i] = ((2 * N - i) * tl + i * tr + N) >> shift;
      return;
    }
  }
  F[0] = L[0]; F[n - 1] = L[n - 1];
  for (int i = 1; i < n - 1; i++) F[i] = (L[i - 1] + 2 * L[i] + L[i + 1] + 2) >> 2;
}

static const uint8_t kIntraFilter[5] = { 10, 7, 1, 0, 10 };          /* TComPattern.cpp:49-56 */
static const int kAng[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 }, kInvAng[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

/* predIntraLumaAng, TComPrediction.cpp:340-372, with bAbove = bLeft = true (initAdiPattern :213-214).
 * pred is N x N contiguous. */
static void intra_pred_core(const int* Lunf, const int* Lfil, int N, int mode, int bitDepth, int16_t* pred, int luma)
{
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  int diff = abs(mode - 10) < abs(mode - 26) ? abs(mode - 10) : abs(mode - 26);
  int filt = luma && diff > kIntraFilter[log2N - 2];
  if (mode == 1) filt = 0;
  const int* L = filt ? Lfil : Lunf;
  const int* top = L + 2 * N + 1;                      /* top[i], i = -1 .. 2N-1 (top[-1] = corner) */
#define LEFT(i) (L[2 * N - 1 - (i)])                   /* left[i], i = -1 .. 2N-1 (LEFT(-1) = corner) */
  const int maxVal = (1 << bitDepth) - 1;
  if (mode == 0) {                                     /* planar, :1468-1505 */
    int bottomLeft = LEFT(N), topRight = top[N], shift1 = log2N, shift2 = log2N + 1;
    int topRow[64], bottomRow[64];
    for (int k = 0; k < N; k++) { bottomRow[k] = bottomLeft - top[k]; topRow[k] = top[k] << shift1; }
    for (int k = 0; k < N; k++) {
      int hor = (LEFT(k) << shift1) + N, right = topRight - LEFT(k);
      for (int l = 0; l < N; l++) { hor += right; topRow[l] += bottomRow[l]; pred[k * N + l] = (int16_t)((hor + topRow[l]) >> shift2); }
    }
    return;
  }
  const int useEdge = luma && N <= 16;                 /* bFilter of xPredIntraAng, :358-366; the chroma planes never filter (:375-390) */
  if (mode == 1) {                                     /* DC, :130-167 + :1521-1541 */
    int sum = 0;
    for (int i = 0; i < N; i++) sum += top[i] + LEFT(i);
    int dcv = (sum + N) / (2 * N);
    for (int i = 0; i < N * N; i++) pred[i] = (int16_t)dcv;
    if (useEdge) {
      pred[0] = (int16_t)((top[0] + LEFT(0) + 2 * pred[0] + 2) >> 2);
      for (int xx = 1; xx < N; xx++) pred[xx] = (int16_t)((top[xx] + 3 * pred[xx] + 2) >> 2);
      for (int yy = 1; yy < N; yy++) pred[yy * N] = (int16_t)((LEFT(yy) + 3 * pred[yy * N] + 2) >> 2);
    }
    return;
  }
  /* angular, :192-338 */
  const int modeHor = mode < 18, modeVer = !modeHor;
  int ang = modeVer ? mode - 26 : -(mode - 10);
  const int sign = ang < 0 ? -1 : 1, aabs = abs(ang);
  const int invAngle = kInvAng[aabs];
  ang = sign * kAng[aabs];
  int refA[2 * 64 + 1 + 64], refL[2 * 64 + 1 + 64];
  int *refMain, *refSide;
  if (ang < 0) {
    for (int k = 0; k < N + 1; k++) { refA[k + N - 1] = top[k - 1]; refL[k + N - 1] = LEFT(k - 1); }
    refMain = (modeVer ? refA : refL) + (N - 1);
    refSide = (modeVer ? refL : refA) + (N - 1);
    int invSum = 128;
    for (int k = -1; k > (N * ang) >> 5; k--) { invSum += invAngle; refMain[k] = refSide[invSum >> 8]; }
  } else {
    for (int k = 0; k < 2 * N + 1; k++) { refA[k] = top[k - 1]; refL[k] = LEFT(k - 1); }
    refMain = modeVer ? refA : refL;
    refSide = modeVer ? refL : refA;
  }
  int16_t tmp[64 * 64];
  if (ang == 0) {
    for (int k = 0; k < N; k++) for (int l = 0; l < N; l++) tmp[k * N + l] = (int16_t)refMain[l + 1];
    if (useEdge)
      for (int k = 0; k < N; k++) {
        int v = tmp[k * N] + ((refSide[k + 1] - refSide[0]) >> 1);
        tmp[k * N] = (int16_t)(v < 0 ? 0 : v > maxVal ? maxVal : v);
      }
  } else {
    int deltaPos = 0;
    for (int k = 0; k < N; k++) {
      deltaPos += ang;
      int di = deltaPos >> 5, df = deltaPos & 31;
      for (int l = 0; l < N; l++) {
        int idx = l + di + 1;
        tmp[k * N + l] = df ? (int16_t)(((32 - df) * refMain[idx] + df * refMain[idx + 1] + 16) >> 5) : (int16_t)refMain[idx];
      }
    }
  }
  if (modeHor) { for (int k = 0; k < N; k++) for (int l = 0; l < N; l++) pred[k * N + l] = tmp[l * N + k]; }
  else memcpy(pred, tmp, (size_t)N * N * sizeof(int16_t));
#undef LEFT
}

void hop_o_intra_pred(const int* Lunf, const int* Lfil, int N, int mode, int bitDepth, int16_t* pred) { intra_pred_core(Lunf, Lfil, N, mode, bitDepth, pred, 1); }
/* predIntraChromaAng, TComPrediction.cpp:375-390: the unfiltered line, no edge filters, no DC filtering */
void hop_o_intra_pred_chroma(const int* L, int N, int mode, int bitDepth, int16_t* pred) { intra_pred_core(L, L, N, mode, bitDepth, pred, 0); }

/* The distortion half of the rough mode search of estIntraPredQT, TEncSearch.cpp:2430-2461: reference samples,
 * smoothing, 35 predictions, calcHAD against the original.  rec = reconstruction picture (sample (0,0)), org =
 * original picture.  satd[35] out.  (The caller adds xModeBitsIntra * sqrt(lambda), :2460-2461.) */
void hop_o_intra_rough(const int16_t* rec, int recStride, const int16_t* org, int orgStride, int x, int y, int N,
                       const uint8_t* flags, int bitDepth, int strong, uint32_t satd[35])
{
  int L[4 * 64 + 1], F[4 * 64 + 1];
  int16_t pred[64 * 64];
  hop_o_intra_fill_refs(rec, recStride, x, y, N, flags, bitDepth, L);
  hop_o_intra_smooth(L, N, bitDepth, strong, F);
  for (int m = 0; m < 35; m++) {
    hop_o_intra_pred(L, F, N, m, bitDepth, pred);
    satd[m] = hop_o_calc_had(org + (ptrdiff_t)y * orgStride + x, orgStride, pred, N, N, N, bitDepth);
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * The mode-decision half of the rough search of TEncSearch::estIntraPredQT (TLibEncoder/TEncSearch.cpp:2440-2493):
 * xModeBitsIntra (:7734-7745) = bits of the luma intra direction through the counting coder from the CI_CURR_BEST state
 * (TEncSbac::codeIntraDirLumaAng, TEncSbac.cpp:770-831: prev_intra_luma_pred_flag on its context, then the MPM index or the
 * 5-bit remainder in bypass), cost = SATD + bits * sqrt(lambda) in double (:2461), xUpdateCandList (:7747-7767) and the MPMs
 * appended if missing (:2466-2488). */
uint8_t hop_o_ctx_next(uint8_t state, int bin);
int32_t hop_o_ctx_bits(uint8_t state, int bin);

uint32_t hop_o_intra_mode_bits(uint8_t* ctx_state, uint64_t* frac, int mode, const int preds[3], int pred_num)
{
  int idx = -1;
  for (int i = 0; i < pred_num; i++) if (mode == preds[i]) idx = i;
  *frac &= 32767;                                                       /* resetBits */
  *frac += (uint64_t)hop_o_ctx_bits(*ctx_state, idx != -1); *ctx_state = hop_o_ctx_next(*ctx_state, idx != -1);
  *frac += (uint64_t)32768 * (uint64_t)(idx == -1 ? 5 : (idx ? 2 : 1));
  return (uint32_t)(*frac >> 15);
}

int hop_o_cand_update(int mode, double cost, int n, uint32_t* modes, double* costs)
{
  int shift = 0;
  while (shift < n && cost < costs[n - 1 - shift]) shift++;
  if (!shift) return 0;
  for (int i = 1; i < shift; i++) { modes[n - i] = modes[n - 1 - i]; costs[n - i] = costs[n - 1 - i]; }
  modes[n - shift] = (uint32_t)mode; costs[n - shift] = cost;
  return 1;
}

/* satd[35] from the rough search; ctx_state / frac_left: prev_intra_luma_pred_flag context and the coder's fraction at CI_CURR_BEST;
 * preds / pred_num: getIntraDirLumaPredictor; mpm_cand: how many of them the list must contain (numCand, :2468-2473).
 * out_modes[num_full_rd + 3], out_costs[num_full_rd]; returns the number of modes for the full RD. */
int hop_o_intra_cand_list(const uint32_t satd[35], uint8_t ctx_state, uint32_t frac_left, double sqrt_lambda, const int preds[3], int pred_num, int mpm_cand,
                          int num_full_rd, uint32_t* out_modes, double* out_costs)
{
  for (int i = 0; i < num_full_rd; i++) { out_costs[i] = 1.7e+308; out_modes[i] = 0; }
  for (int mode = 0; mode < 35; mode++) {
    uint8_t s = ctx_state; uint64_t f = frac_left;
    const uint32_t bits = hop_o_intra_mode_bits(&s, &f, mode, preds, pred_num);
    const double cost = (double)satd[mode] + (double)bits * sqrt_lambda;
    hop_o_cand_update(mode, cost, num_full_rd, out_modes, out_costs);
  }
  int n = num_full_rd;
  for (int j = 0; j < mpm_cand; j++) {
    int inc = 0;
    for (int i = 0; i < n; i++) inc |= (preds[j] == (int)out_modes[i]);
    if (!inc) out_modes[n++] = (uint32_t)preds[j];
  }
  return n;
}
