/*
 * hop_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product (hevc-hop_amd/csrc, libhophip.so) never links, imports or calls it.
 *
 * Reference: zinsayon/HEVC-HOP (HM-15.0 fork).  Every function cites the reference file:line
 * it restates (paths relative to /root/reference/source/Lib).  Parity is PINNED: each function is
 * checked against golden vectors (the npz files of tests/golden, replayed by tests/test_oracle_golden*.py) that
 * oracle/make_golden*.py generated from the reference's own code compiled into
 * oracle/_ref/libref_harness.so, and it runs inside the reference encoder in place of the reference's
 * members with an unchanged bitstream (tests/test_encoder_shim.py, where /root/reference exists).
 *
 * Plain C99, integer arithmetic except the GT warp which is IEEE double evaluated in the
 * reference's operation order; build with -ffp-contract=off (oracle/Makefile).
 *
 * Conventions: Pel = int16_t, strides in elements, "ref" pointers address sample (0,0) of a plane
 * that has the reference's margins (80 luma / 40 chroma, TLibCommon/TComPicYuv.cpp:82-85).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
static int trunc_x86(double v) { return (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1); }   /* cvttsd2si */
#include "hop_oracle.h"

#define HOP_NOT_VALID (-1)           /* TLibCommon/CommonDef.h:126 */
#define HOP_MAX_UINT 0xFFFFFFFFu

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------ */
/* a12: distortion                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* TLibCommon/TComRdCost.cpp:541-1011 (xGetSAD4/8/16/32/64/12/24/48): rows stepped by 1<<subShift,
 * sum <<= subShift, then >> (bitDepth-8) (DISTORTION_PRECISION_ADJUSTMENT, TypeDef.h:162-167). */
uint32_t hop_o_sad(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth, int subShift)
{
  uint32_t sum = 0;
  int step = 1 << subShift;
  for (int r = 0; r < h; r += step)
    for (int c = 0; c < w; c++)
      sum += (uint32_t)iabs(org[r * so + c] - cur[r * sc + c]);
  sum <<= subShift;
  return sum >> (bitDepth - 8);
}

/* TLibCommon/TComRdCost.cpp:1018-1360 (xGetSSE*): per-sample (d*d) >> ((bitDepth-8)<<1) */
uint32_t hop_o_sse(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth)
{
  uint32_t sum = 0, shift = (uint32_t)((bitDepth - 8) << 1);
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      int d = org[r * so + c] - cur[r * sc + c];
      sum += (uint32_t)(d * d) >> shift;
    }
  return sum;
}

/* TLibCommon/TComRdCost.cpp:1366-1385 */
static uint32_t had2x2(const int16_t* o, const int16_t* c, int so, int sc)
{
  int d0 = o[0] - c[0], d1 = o[1] - c[1], d2 = o[so] - c[sc], d3 = o[so + 1] - c[sc + 1];
  int m0 = d0 + d2, m1 = d1 + d3, m2 = d0 - d2, m3 = d1 - d3;
  return (uint32_t)(iabs(m0 + m1) + iabs(m0 - m1) + iabs(m2 + m3) + iabs(m2 - m3));
}

/* TLibCommon/TComRdCost.cpp:1387-1479: 4x4 Hadamard, satd = (sum|coef| + 1) >> 1.
 * The sum of absolute values is invariant to the butterfly ordering, so a plain
 * separable +/-1 transform is used. */
static uint32_t had4x4(const int16_t* o, const int16_t* c, int so, int sc)
{
  int d[4][4], t[4][4];
  for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) d[r][k] = o[r * so + k] - c[r * sc + k];
  for (int r = 0; r < 4; r++) {
    int a = d[r][0] + d[r][2], b = d[r][1] + d[r][3], e = d[r][0] - d[r][2], f = d[r][1] - d[r][3];
    t[r][0] = a + b; t[r][1] = a - b; t[r][2] = e + f; t[r][3] = e - f;
  }
  int satd = 0;
  for (int k = 0; k < 4; k++) {
    int a = t[0][k] + t[2][k], b = t[1][k] + t[3][k], e = t[0][k] - t[2][k], f = t[1][k] - t[3][k];
    satd += iabs(a + b) + iabs(a - b) + iabs(e + f) + iabs(e - f);
  }
  return (uint32_t)((satd + 1) >> 1);
}

/* TLibCommon/TComRdCost.cpp:1481-1575: 8x8 Hadamard, sad = (sum|coef| + 2) >> 2 */
static uint32_t had8x8(const int16_t* o, const int16_t* c, int so, int sc)
{
  int m[8][8];
  for (int r = 0; r < 8; r++) for (int k = 0; k < 8; k++) m[r][k] = o[r * so + k] - c[r * sc + k];
  for (int r = 0; r < 8; r++)
    for (int len = 1; len < 8; len <<= 1)
      for (int i = 0; i < 8; i += len << 1)
        for (int j = i; j < i + len; j++) { int a = m[r][j], b = m[r][j + len]; m[r][j] = a + b; m[r][j + len] = a - b; }
  for (int k = 0; k < 8; k++)
    for (int len = 1; len < 8; len <<= 1)
      for (int i = 0; i < 8; i += len << 1)
        for (int j = i; j < i + len; j++) { int a = m[j][k], b = m[j + len][k]; m[j][k] = a + b; m[j + len][k] = a - b; }
  int sad = 0;
  for (int r = 0; r < 8; r++) for (int k = 0; k < 8; k++) sad += iabs(m[r][k]);
  return (uint32_t)((sad + 2) >> 2);
}

/* TLibCommon/TComRdCost.cpp:1641-1708 (xGetHADs, iStep == 1) */
uint32_t hop_o_hads(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth)
{
  uint32_t sum = 0;
  if ((h % 8 == 0) && (w % 8 == 0)) {
    for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) sum += had8x8(org + y * so + x, cur + y * sc + x, so, sc);
  } else if ((h % 4 == 0) && (w % 4 == 0)) {
    for (int y = 0; y < h; y += 4) for (int x = 0; x < w; x += 4) sum += had4x4(org + y * so + x, cur + y * sc + x, so, sc);
  } else {
    for (int y = 0; y < h; y += 2) for (int x = 0; x < w; x += 2) sum += had2x2(org + y * so + x, cur + y * sc + x, so, sc);
  }
  return sum >> (bitDepth - 8);
}

/* TLibCommon/TComRdCost.cpp:391-425 (calcHAD): 8x8 blocks if both dims %8==0 else 4x4 */
uint32_t hop_o_calc_had(const int16_t* a, int sa, const int16_t* b, int sb, int w, int h, int bitDepth)
{
  uint32_t sum = 0;
  if ((w % 8 == 0) && (h % 8 == 0)) {
    for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) sum += had8x8(a + y * sa + x, b + y * sb + x, sa, sb);
  } else {
    for (int y = 0; y < h; y += 4) for (int x = 0; x < w; x += 4) sum += had4x4(a + y * sa + x, b + y * sb + x, sa, sb);
  }
  return sum >> (bitDepth - 8);
}

/* TLibCommon/TComRdCost.cpp:270-284: exp-Golomb length */
uint32_t hop_o_component_bits(int v)
{
  uint32_t len = 1, t = (v <= 0) ? (uint32_t)((-v << 1) + 1) : (uint32_t)(v << 1);
  while (t != 1) { t >>= 1; len += 2; }
  return len;
}

/* TLibCommon/TComRdCost.h:205-215 (IT_GT_CODING 0, IT_GT_AFFINE 1, W_GT 1): corners 0..2 only */
uint32_t hop_o_bits_gt(const int v[8])
{
  uint32_t b = 0;
  for (int i = 0; i < 6; i++) b += hop_o_component_bits(v[i]);
  return b;
}

/* TLibCommon/TComRdCost.h:185-202 with FIX203 (TComRdCost.h:52): (cost * bits) >> 16 in UInt */
static inline uint32_t mv_bits(int x, int y, int scale, int predX, int predY)
{
  return hop_o_component_bits((x * (1 << scale)) - predX) + hop_o_component_bits((y * (1 << scale)) - predY);
}
static inline uint32_t mv_cost(uint32_t lambdaCost, int x, int y, int scale, int predX, int predY)
{
  return (lambdaCost * mv_bits(x, y, scale, predX, predY)) >> 16;
}

/* ------------------------------------------------------------------------------------------ */
/* a13: SS reference upkeep                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* TLibCommon/TComSlice.cpp:241-255 -> TComPicYuv::setPicPel(NOT_VALID) TComPicYuv.cpp:199-207:
 * byte memset of 0xFF => every Pel of every padded plane = -1 */
void hop_o_ssref_reset(int16_t* bufY, int16_t* bufCb, int16_t* bufCr, int picW, int picH)
{
  size_t ny = (size_t)(picW + 160) * (size_t)(picH + 160);
  size_t nc = (size_t)((picW >> 1) + 80) * (size_t)((picH >> 1) + 80);
  memset(bufY, 0xFF, ny * sizeof(int16_t));
  memset(bufCb, 0xFF, nc * sizeof(int16_t));
  memset(bufCr, 0xFF, nc * sizeof(int16_t));
}

/* TLibCommon/TComPicYuv.cpp:247-275 (xExtendPicCompBorder) */
static void extend_border(int16_t* p, int stride, int w, int h, int mx, int my)
{
  int16_t* pi = p;
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < mx; x++) { pi[-mx + x] = pi[0]; pi[w + x] = pi[w - 1]; }
    pi += stride;
  }
  pi -= (stride + mx);
  for (int y = 0; y < my; y++) memcpy(pi + (y + 1) * stride, pi, sizeof(int16_t) * (size_t)(w + (mx << 1)));
  pi -= ((h - 1) * stride);
  for (int y = 0; y < my; y++) memcpy(pi - (y + 1) * stride, pi, sizeof(int16_t) * (size_t)(w + (mx << 1)));
}

/* TLibEncoder/TEncCu.cpp:1677-1697 (xCopyYuv2SSRef, leaf branch): copy the finalised CU's
 * reconstruction into the SS reference, then re-extend ALL borders of all three planes.
 * Planes are addressed at sample (0,0); recY is size x size contiguous, recCb/Cr (size/2)^2. */
void hop_o_ssref_commit_cu(int16_t* y, int16_t* cb, int16_t* cr, int picW, int picH,
                           int x0, int y0, int size, const int16_t* recY, const int16_t* recCb, const int16_t* recCr)
{
  int sy = picW + 160, sc = (picW >> 1) + 80;
  for (int r = 0; r < size; r++) memcpy(y + (size_t)(y0 + r) * sy + x0, recY + r * size, (size_t)size * sizeof(int16_t));
  int cs = size >> 1;
  for (int r = 0; r < cs; r++) {
    memcpy(cb + (size_t)((y0 >> 1) + r) * sc + (x0 >> 1), recCb + r * cs, (size_t)cs * sizeof(int16_t));
    memcpy(cr + (size_t)((y0 >> 1) + r) * sc + (x0 >> 1), recCr + r * cs, (size_t)cs * sizeof(int16_t));
  }
  extend_border(y, sy, picW, picH, 80, 80);
  extend_border(cb, sc, picW >> 1, picH >> 1, 40, 40);
  extend_border(cr, sc, picW >> 1, picH >> 1, 40, 40);
}

/* ------------------------------------------------------------------------------------------ */
/* a1: search range + SS integer full search                                                  */
/* ------------------------------------------------------------------------------------------ */

/* TLibCommon/TComDataCU.cpp:3492-3504 */
static void clip_mv(int picW, int picH, int cuX, int cuY, int* hor, int* ver)
{
  int sh = 2, off = 8;
  int hmax = (picW + off - cuX - 1) << sh, hmin = (-64 - off - cuX + 1) * (1 << sh);
  int vmax = (picH + off - cuY - 1) << sh, vmin = (-64 - off - cuY + 1) * (1 << sh);
  *hor = imin(hmax, imax(hmin, *hor));
  *ver = imin(vmax, imax(vmin, *ver));
}

/* TLibEncoder/TEncSearch.cpp:6204-6220 then :6224-6259.  TComMv components are Short, hence the casts.
 * out = {left, right, top, bottom, offX', offY'} */
void hop_o_set_search_range(int picW, int picH, int cuX, int cuY, int cuSize, int ctuAddr, int frameWidthInCtu,
                            int predX, int predY, int srchRng, int offX, int offY, int firstRow, int firstCol, int out[6])
{
  int sh = 2;
  int ph = predX, pv = predY;
  clip_mv(picW, picH, cuX, cuY, &ph, &pv);
  int lh = (int16_t)(ph - (srchRng << sh)), lv = (int16_t)(pv - (srchRng << sh));
  int rh = (int16_t)(ph + (srchRng << sh)), rv = (int16_t)(pv + (srchRng << sh));
  clip_mv(picW, picH, cuX, cuY, &lh, &lv);
  clip_mv(picW, picH, cuX, cuY, &rh, &rv);
  int left = lh >> sh, top = lv >> sh, right = rh >> sh, bottom = rv >> sh;
  if (firstCol && firstRow) {
    right = left + 1;
    top = bottom + 1;
  } else {
    bottom = (bottom > (-offY - 4)) ? (-offY - 4) : bottom;
    offX = -offX - cuSize - 4;
    offY = -offY - cuSize - 4;
    bottom = (firstCol && (bottom > offY)) ? offY : bottom;
    right = (firstRow && (right > offX)) ? offX : right;
    right = (!firstRow && (ctuAddr < frameWidthInCtu) && (right > (offX + (cuSize << 1)))) ? (offX + (cuSize << 1)) : right;
  }
  lh = (int16_t)(left * 4); lv = (int16_t)(top * 4); rh = (int16_t)(right * 4); rv = (int16_t)(bottom * 4);
  clip_mv(picW, picH, cuX, cuY, &lh, &lv);
  clip_mv(picW, picH, cuX, cuY, &rh, &rv);
  out[0] = lh >> sh; out[1] = rh >> sh; out[2] = lv >> sh; out[3] = rv >> sh; out[4] = offX; out[5] = offY;
}

/* TLibEncoder/TEncSearch.cpp:6262-6371 (xPatternSearch, isSSE) with
 * TLibCommon/TComRdCost.cpp:444-458 (isValidPattern(DistParam*, patternSize)).
 * refPU addresses the SS-ref sample co-located with the PU's top-left.
 * Cost scale is 2 here (TEncSearch.cpp:4560).  Returns 1 if at least one candidate was valid. */
int hop_o_ss_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                    int left, int right, int top, int bottom, int offX, int offY,
                    int predX, int predY, uint32_t lambdaCost, int fen, int bitDepth,
                    int* bestX, int* bestY, uint32_t* sadOut)
{
  uint32_t best = HOP_MAX_UINT;
  int bx = 0, by = 0, valid = 0;
  int subShift = (fen && h > 8) ? 1 : 0;          /* :6303-6309 */
  for (int y = top; y <= bottom; y++) {
    for (int x = left; x <= right; x++) {
      const int16_t* cur = refPU + (ptrdiff_t)y * refStride + x;
      if ((x >= offX) && (y > offY)) continue;      /* :6328 */
      /* isValidPattern: rows = h + 4, patternSize = w + 4 */
      const int16_t* pLB = cur + (ptrdiff_t)(h + 4) * refStride;
      if (pLB[0] == HOP_NOT_VALID || pLB[w + 4] == HOP_NOT_VALID) continue;
      valid = 1;
      uint32_t sad = hop_o_sad(org, orgStride, cur, refStride, w, h, bitDepth, subShift);
      sad += mv_cost(lambdaCost, x, y, 2, predX, predY);
      if (sad < best) { best = sad; bx = x; by = y; }
    }
  }
  if (!valid) { *sadOut = HOP_MAX_UINT; *bestX = 0; *bestY = 0; return 0; }
  *bestX = bx; *bestY = by;
  *sadOut = best - mv_cost(lambdaCost, bx, by, 2, predX, predY);
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* DCT-IF (TLibCommon/TComInterpolationFilter.cpp:55-75 taps, :92-152 filterCopy, :170-245 filter)*/
/* ------------------------------------------------------------------------------------------ */
static const int16_t kLuma[4][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int16_t kChroma[8][4] = {
  { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
  { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

/* one sample of the reference's 2-stage separable interpolation at integer position (x,y) of
 * `src` displaced by fractional phase (xf,yf); ntaps = 8 (luma, phases 0..3) or 4 (chroma, 0..7).
 * mode 0: single call of filterHor/filterVer with isLast=true when one phase is zero, both stages
 *         otherwise  -- TComPrediction.cpp:662-678 (uni-prediction, bi = false)
 * mode 1: always horizontal (isFirst, !isLast) then vertical (!isFirst, isLast), including
 *         phase 0 via filterCopy -- the encoder's xExtDIFUpSamplingH/Q, TEncSearch.cpp:7818-8011 */
static int interp_sample(const int16_t* src, int stride, int x, int y, int xf, int yf, int ntaps, int bitDepth, int mode)
{
  const int16_t* cx = ntaps == 8 ? kLuma[xf] : kChroma[xf];
  const int16_t* cy = ntaps == 8 ? kLuma[yf] : kChroma[yf];
  int half = ntaps / 2 - 1;
  int headRoom = 14 - bitDepth;                /* IF_INTERNAL_PREC - bitDepth */
  int maxVal = (1 << bitDepth) - 1;
  const int16_t* p = src + (ptrdiff_t)y * stride + x;
  if (mode == 0) {
    if (xf == 0 && yf == 0) return p[0];       /* filterCopy(isFirst == isLast) */
    if (yf == 0 || xf == 0) {                  /* filter<N,*,true,true>: shift 6, offset 32, clip */
      int sum = 0;
      if (yf == 0) for (int k = 0; k < ntaps; k++) sum += p[k - half] * cx[k];
      else         for (int k = 0; k < ntaps; k++) sum += p[(ptrdiff_t)(k - half) * stride] * cy[k];
      int16_t val = (int16_t)((sum + 32) >> 6);
      if (val < 0) val = 0;
      if (val > maxVal) val = (int16_t)maxVal;
      return val;
    }
  }
  /* stage 1: horizontal into 14-bit intermediate, rows y-half .. y-half+ntaps-1 */
  int16_t tmp[8];
  for (int k = 0; k < ntaps; k++) {
    const int16_t* q = p + (ptrdiff_t)(k - half) * stride;
    if (xf == 0) {
      int16_t val = (int16_t)(q[0] << headRoom);          /* filterCopy isFirst: :120-121 */
      tmp[k] = (int16_t)(val - (int16_t)8192);
    } else {
      int sum = 0;
      for (int j = 0; j < ntaps; j++) sum += q[j - half] * cx[j];
      int shift = 6 - headRoom;                            /* isFirst, !isLast */
      int offset = -8192 * (1 << shift);
      tmp[k] = (int16_t)((sum + offset) >> shift);
    }
  }
  /* stage 2: vertical, !isFirst, isLast */
  if (yf == 0) {                                           /* filterCopy else-branch :130-150 */
    int shift = headRoom;
    int16_t offset = (int16_t)(8192 + (shift ? (1 << (shift - 1)) : 0));
    int16_t val = tmp[half];
    val = (int16_t)((val + offset) >> shift);
    if (val < 0) val = 0;
    if (val > maxVal) val = (int16_t)maxVal;
    return val;
  } else {
    int sum = 0;
    for (int k = 0; k < ntaps; k++) sum += tmp[k] * cy[k];
    int shift = 6 + headRoom;
    int offset = (1 << (shift - 1)) + (8192 << 6);
    int16_t val = (int16_t)((sum + offset) >> shift);
    if (val < 0) val = 0;
    if (val > maxVal) val = (int16_t)maxVal;
    return val;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* a2: half/quarter-pel refinement                                                            */
/* ------------------------------------------------------------------------------------------ */
static const int8_t kRefineH[9][2] = { {0,0},{0,-1},{0,1},{-1,0},{1,0},{-1,-1},{1,-1},{-1,1},{1,1} };   /* TEncSearch.cpp:46-57 */
static const int8_t kRefineQ[9][2] = { {0,0},{0,-1},{0,1},{-1,-1},{1,-1},{-1,0},{1,0},{-1,1},{1,1} };   /* TEncSearch.cpp:59-70 */

/* block of the reference interpolated at quarter-pel offset (fx,fy) from integer position (encoder planes) */
static void frac_block(const int16_t* ref, int stride, int w, int h, int fx, int fy, int bitDepth, int16_t* dst)
{
  int xi = fx >> 2, xf = fx & 3, yi = fy >> 2, yf = fy & 3;
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++)
      dst[r * w + c] = (int16_t)interp_sample(ref, stride, c + xi, r + yi, xf, yf, 8, bitDepth, 1);
}

/* TLibEncoder/TEncSearch.cpp:6564-6610 (xPatternSearchFracDIF) + :709-761 (xPatternRefinement) +
 * :7818-8011 (the 16 phase planes).  refPU = SS-ref at the PU; (mvX,mvY) integer MV.
 * useHad selects xGetHADs (HadamardME=1) or SAD.  Returns the cost after the quarter-pel step. */
uint32_t hop_o_frac_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                           int mvX, int mvY, int predX, int predY, uint32_t lambdaCost, int useHad, int bitDepth,
                           int half[2], int qter[2])
{
  const int16_t* ref = refPU + (ptrdiff_t)mvY * refStride + mvX;
  int16_t* blk = (int16_t*)malloc((size_t)w * h * sizeof(int16_t));
  uint32_t best = HOP_MAX_UINT; int bi = 0;
  for (int i = 0; i < 9; i++) {                       /* half-pel, cost scale 1 (:4615) */
    int hx = kRefineH[i][0], hy = kRefineH[i][1];
    frac_block(ref, refStride, w, h, 2 * hx, 2 * hy, bitDepth, blk);
    uint32_t d = useHad ? hop_o_hads(org, orgStride, blk, w, w, h, bitDepth) : hop_o_sad(org, orgStride, blk, w, w, h, bitDepth, 0);
    d += mv_cost(lambdaCost, hx + (mvX * 2), hy + (mvY * 2), 1, predX, predY);
    if (d < best) { best = d; bi = i; }
  }
  half[0] = kRefineH[bi][0]; half[1] = kRefineH[bi][1];
  best = HOP_MAX_UINT; bi = 0;
  for (int i = 0; i < 9; i++) {                       /* quarter-pel, cost scale 0 (:6599) */
    int qx = kRefineQ[i][0], qy = kRefineQ[i][1];
    frac_block(ref, refStride, w, h, 2 * half[0] + qx, 2 * half[1] + qy, bitDepth, blk);
    uint32_t d = useHad ? hop_o_hads(org, orgStride, blk, w, w, h, bitDepth) : hop_o_sad(org, orgStride, blk, w, w, h, bitDepth, 0);
    d += mv_cost(lambdaCost, qx + (((mvX * 2) + half[0]) * 2), qy + (((mvY * 2) + half[1]) * 2), 0, predX, predY);
    if (d < best) { best = d; bi = i; }
  }
  qter[0] = kRefineQ[bi][0]; qter[1] = kRefineQ[bi][1];
  free(blk);
  return best;
}

/* ------------------------------------------------------------------------------------------ */
/* a4/a5: homography + warp                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* TLibCommon/TComPrediction.cpp:807-832 */
void hop_o_calc_param_projective(const int x[4], const int y[4], double h[9], int Width, int Height)
{
  double H, W, dx1, dx2, dx3, dy1, dy2, dy3;
  W = (double)Width - 1.0;
  H = (double)Height - 1.0;
  dx1 = (double)x[1] - x[2];
  dx2 = (double)x[3] - x[2];
  dx3 = (double)x[0] - x[1] + x[2] - x[3];
  dy1 = (double)y[1] - y[2];
  dy2 = (double)y[3] - y[2];
  dy3 = (double)y[0] - y[1] + y[2] - y[3];
  h[2] = ((dx3 * dy2 - dx2 * dy3) / (dx1 * dy2 - dx2 * dy1)) / W;
  h[5] = ((dx1 * dy3 - dx3 * dy1) / (dx1 * dy2 - dx2 * dy1)) / H;
  h[0] = (double)(x[1] - x[0]) / W + h[2] * x[1];
  h[3] = (double)(x[3] - x[0]) / H + h[5] * x[3];
  h[6] = (double)x[0];
  h[1] = (double)(y[1] - y[0]) / W + h[2] * y[1];
  h[4] = (double)(y[3] - y[0]) / H + h[5] * y[3];
  h[7] = (double)y[0];
  h[8] = 1.0;
}

/* TLibCommon/TComPrediction.cpp:834-859 (chroma: corners are doubles) */
void hop_o_calc_param_projective_c(const double x[4], const double y[4], double h[9], int Width, int Height)
{
  double H, W, dx1, dx2, dx3, dy1, dy2, dy3;
  W = (double)Width - 1.0;
  H = (double)Height - 1.0;
  dx1 = x[1] - x[2];
  dx2 = x[3] - x[2];
  dx3 = x[0] - x[1] + x[2] - x[3];
  dy1 = y[1] - y[2];
  dy2 = y[3] - y[2];
  dy3 = y[0] - y[1] + y[2] - y[3];
  h[2] = ((dx3 * dy2 - dx2 * dy3) / (dx1 * dy2 - dx2 * dy1)) / W;
  h[5] = ((dx1 * dy3 - dx3 * dy1) / (dx1 * dy2 - dx2 * dy1)) / H;
  h[0] = (x[1] - x[0]) / W + h[2] * x[1];
  h[3] = (x[3] - x[0]) / H + h[5] * x[3];
  h[6] = x[0];
  h[1] = (y[1] - y[0]) / W + h[2] * y[1];
  h[4] = (y[3] - y[0]) / H + h[5] * y[3];
  h[7] = y[0];
  h[8] = 1.0;
}

/* TLibCommon/TComPrediction.cpp:904-1030, IT_GT_GRID_SIZE 2 / IT_GT_Interpolation_Filter 0 branch.
 * W,H are the DOUBLED block dimensions; refCentre addresses the patch sample co-located with the
 * block's top-left; `clipPatch`: 0 = read the source as is, 1 = clamp each source sample to
 * [0,(1<<bitDepth)-1] first (what the encoder's m_filteredBlock[0][0] holds, see hop_o_gt_search). */
/* candidate warps evaluated since the last reset (tests: how many the GT searches of a run went through) */
static long g_warp_count = 0;
long hop_o_warp_counter(int reset) { long v = g_warp_count; if (reset) g_warp_count = 0; return v; }

static void projective_transform(const int16_t* refCentre, int16_t* aux, const double h[9], int W, int H, int stride,
                                 int nssWindow, int clipPatch, int bitDepth)
{
  g_warp_count++;
  int offsetX = W / 2 - (W / 2 / 2);
  int offsetY = H / 2 - (H / 2 / 2);
  int m = nssWindow / 2, wv = W / 2, hv = H / 2;
  int maxVal = (1 << bitDepth) - 1;
  for (int y = offsetY; y < offsetY + hv; y++) {
    for (int x = offsetX; x < offsetX + wv; x++) {
      double Fx = (h[0] * x + h[3] * y + h[6]) / (h[2] * x + h[5] * y + h[8]);
      double Fy = (h[1] * x + h[4] * y + h[7]) / (h[2] * x + h[5] * y + h[8]);
      /* (Int)Fy - offset as the reference's x86 build computes it: cvttsd2si gives INT_MIN for NaN / out-of-range quotients, the subtraction wraps */
      int Y = (int)((unsigned)trunc_x86(Fy) - (unsigned)offsetY);
      int X = (int)((unsigned)trunc_x86(Fx) - (unsigned)offsetX);
      double q = (Fy - offsetY - (double)Y);
      double p = (Fx - offsetX - (double)X);
      if (Y < -m) Y = -m;
      if (X < -m) X = -m;
      if (Y > m + hv - 1) Y = m + hv - 1;
      if (X > m + wv - 1) X = m + wv - 1;
      if (Y + 1 > m + hv - 1) Y = m + hv - 2;
      if (X + 1 > m + wv - 1) X = m + wv - 2;
      const int16_t* pa = refCentre + (ptrdiff_t)Y * stride;
      int a = pa[X], b = pa[X + 1], c = pa[stride + X], d = pa[stride + X + 1];
      if (clipPatch) {
        a = imin(maxVal, imax(0, a)); b = imin(maxVal, imax(0, b));
        c = imin(maxVal, imax(0, c)); d = imin(maxVal, imax(0, d));
      }
      double v = (1.0 - q) * ((1.0 - p) * (double)a + p * (double)b);
      v += q * ((1.0 - p) * (double)c + p * (double)d);
      if (v > 255) v = 255;                    /* hard-coded 8-bit clip, :969-972 */
      if (v < 0) v = 0;
      aux[(y - offsetY) * wv + (x - offsetX)] = (int16_t)(v + 0.5);
    }
  }
}

void hop_o_projective_transform(const int16_t* refCentre, int16_t* aux, const double h[9], int W, int H, int stride, int nssWindow)
{
  projective_transform(refCentre, aux, h, W, H, stride, nssWindow, 0, 8);
}

/* ------------------------------------------------------------------------------------------ */
/* a3: GT / HOP 4-corner diamond search                                                       */
/* ------------------------------------------------------------------------------------------ */

/* TLibEncoder/TEncSearch.cpp:4686-4790 (set-up) + :5093-5467 (IT_GT_SEARCH == 2).
 * in : org PU, SS-ref at the PU, integer MV + half + quarter from the previous stages (only used
 *      when the search fails to improve), start vectors = ssBestCand (integer) + nAmvp AMVP
 *      candidates (quarter-pel), incumbent cost, predictor, lambda cost, useHad.
 * out: gt[8] = GT0..GT3 (x,y), return gtFlag; *cost, mv/half/qter updated as the reference does.
 * The search patch is the reference's m_filteredBlock[0][0]: the 2Wx2H SS-ref region pushed through
 * filterCopy twice (TEncSearch.cpp:5161-5165 -> :7832,:7837; TComInterpolationFilter.cpp:92-152),
 * i.e. each sample clamped to [0, maxVal] (so sentinel -1 reads as 0). */
int hop_o_gt_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                    int mvInt[2], int half[2], int qter[2], const int ssBest[2], int nAmvp, const int* amvpXY,
                    int predX, int predY, uint32_t lambdaCost, int useHad, int bitDepth,
                    uint32_t* cost, int gt[8])
{
  int iRows = h, iCols = w;
  int16_t* aux = (int16_t*)malloc((size_t)w * h * sizeof(int16_t));
  double hp[9];
  int bestCX[4] = {0,0,0,0}, bestCY[4] = {0,0,0,0}, curCX[4], curCY[4];
  int bestNX[4], bestNY[4], curNX[4], curNY[4];
  int maxIter = 6;                                              /* IT_MAX_NSS_Iteration, TypeDef.h:211 */
  int nssWindow = (imin(iRows, iCols) >> 1) * 2;                /* :4756-4759 */
  int lastStep = nssWindow >> maxIter;                          /* :4763-4765 */
  if (lastStep == 0) lastStep = 1;
  uint32_t distBest = *cost;                                    /* :4769 */
  int bestSSX = 0, bestSSY = 0;
  bestNX[0] = 0;             bestNY[0] = 0;
  bestNX[1] = iCols * 2 - 1; bestNY[1] = 0;
  bestNX[2] = iCols * 2 - 1; bestNY[2] = iRows * 2 - 1;
  bestNX[3] = 0;             bestNY[3] = iRows * 2 - 1;
  for (int k = 0; k < 4; k++) { curNX[k] = bestNX[k]; curNY[k] = bestNY[k]; }

  for (int b = 0; b < 1 + nAmvp; b++) {                         /* :5106-5178 */
    int16_t Hor, Ver; int sx, sy;
    if (b < 1) {
      if (ssBest[0] == 0 && ssBest[1] == 0) continue;
      sx = ssBest[0]; sy = ssBest[1];
    } else {
      int ax = amvpXY[2 * (b - 1)], ay = amvpXY[2 * (b - 1) + 1];
      if (ax == 0 && ay == 0) continue;
      sx = (int16_t)ax >> 2; sy = (int16_t)ay >> 2;
    }
    Hor = (int16_t)(sx * 4); Ver = (int16_t)(sy * 4);
    const int16_t* centre = refPU + (ptrdiff_t)sy * refStride + sx;   /* patch centre block == PU displaced by the start vector */
    int iter = 1;
    for (int j0 = nssWindow; (j0 > 1) && (iter <= maxIter); j0 /= 2) {
      iter++;
      if (j0 == nssWindow) {                                    /* :5183-5203 */
        curNX[0] = bestNX[0] = 0;             curNY[0] = bestNY[0] = 0;
        curNX[1] = bestNX[1] = iCols * 2 - 1; curNY[1] = bestNY[1] = 0;
        curNX[2] = bestNX[2] = iCols * 2 - 1; curNY[2] = bestNY[2] = iRows * 2 - 1;
        curNX[3] = bestNX[3] = 0;             curNY[3] = bestNY[3] = iRows * 2 - 1;
      } else {
        for (int k = 0; k < 4; k++) { curNX[k] = bestNX[k]; curNY[k] = bestNY[k]; }
      }
      int s = j0 / 2;
      for (int y0 = s; y0 >= -s; y0 -= s) { curCY[0] = curNY[0] + y0;
      for (int x0 = s; x0 >= -s; x0 -= s) { if (y0 != 0 && x0 != 0) continue; curCX[0] = curNX[0] + x0;
      for (int y1 = s; y1 >= -s; y1 -= s) { curCY[1] = curNY[1] + y1;
      for (int x1 = s; x1 >= -s; x1 -= s) { if (y1 != 0 && x1 != 0) continue; curCX[1] = curNX[1] + x1;
      for (int y2 = s; y2 >= -s; y2 -= s) { curCY[2] = curNY[2] + y2;
      for (int x2 = s; x2 >= -s; x2 -= s) { if (y2 != 0 && x2 != 0) continue; curCX[2] = curNX[2] + x2;
      for (int y3 = s; y3 >= -s; y3 -= s) { curCY[3] = curNY[3] + y3;
      for (int x3 = s; x3 >= -s; x3 -= s) { if (y3 != 0 && x3 != 0) continue; curCX[3] = curNX[3] + x3;
        if (x0 == x1 && x0 == x2 && x0 == x3 && y0 == y1 && y0 == y2 && y0 == y3) continue;   /* :5289 */
        hop_o_calc_param_projective(curCX, curCY, hp, iCols * 2, iRows * 2);                    /* :5316 */
        if (!(hp[2] == 0.0 && hp[5] == 0.0)) continue;                                          /* :5323 */
        projective_transform(centre, aux, hp, iCols * 2, iRows * 2, refStride, nssWindow, 1, bitDepth);  /* :5336 */
        uint32_t d = useHad ? hop_o_hads(org, orgStride, aux, iCols, iCols, iRows, bitDepth)
                            : hop_o_sad(org, orgStride, aux, iCols, iCols, iRows, bitDepth, 0);
        d += mv_cost(lambdaCost, Hor, Ver, 0, predX, predY);                                    /* :5345 */
        int v[8] = { curCX[0] / lastStep, curCY[0] / lastStep,
                     (curCX[1] - iCols * 2 + 1) / lastStep, curCY[1] / lastStep,
                     (curCX[2] - iCols * 2 + 1) / lastStep, (curCY[2] - iRows * 2 + 1) / lastStep,
                     curCX[3] / lastStep, (curCY[3] - iRows * 2 + 1) / lastStep };
        d += (lambdaCost * hop_o_bits_gt(v)) >> 16;                                             /* :5346-5358 */
        if (d < distBest) {                                                                     /* :5361-5383 */
          distBest = d;
          for (int k = 0; k < 4; k++) { bestCX[k] = curCX[k]; bestCY[k] = curCY[k]; bestNX[k] = curCX[k]; bestNY[k] = curCY[k]; }
          bestSSX = Hor; bestSSY = Ver;
        }
      }}}}}}}}
    }
  }
  free(aux);
  int flag = 0;
  for (int k = 0; k < 4; k++) if (bestCX[k] != 0 || bestCY[k] != 0) flag = 1;                  /* :5436-5439 */
  if (flag) {
    gt[0] = bestCX[0] / lastStep;                   gt[1] = bestCY[0] / lastStep;
    gt[2] = (bestCX[1] - iCols * 2 + 1) / lastStep; gt[3] = bestCY[1] / lastStep;
    gt[4] = (bestCX[2] - iCols * 2 + 1) / lastStep; gt[5] = (bestCY[2] - iRows * 2 + 1) / lastStep;
    gt[6] = bestCX[3] / lastStep;                   gt[7] = (bestCY[3] - iRows * 2 + 1) / lastStep;
    *cost = distBest;
    mvInt[0] = bestSSX >> 2; mvInt[1] = bestSSY >> 2;          /* :5455-5457 */
    half[0] = half[1] = 0; qter[0] = qter[1] = 0;
  } else {
    for (int k = 0; k < 8; k++) gt[k] = 0;
  }
  return flag;
}

/* ------------------------------------------------------------------------------------------ */
/* a6: final (normative) predictor                                                            */
/* ------------------------------------------------------------------------------------------ */

/* TLibCommon/TComPrediction.cpp:723-805 (xPredGTLuma) / :1351-1420 (xPredGTChroma), after the
 * 2Wx2H patch has been produced.  patch is contiguous, stride 2*bw. isChroma selects GT/2 corners. */
static void pred_gt_block(const int16_t* patch, int bw, int bh, const int gt[8], int isChroma, int16_t* dst, int dstStride)
{
  double hp[9];
  int nssWindow = (imin(bh, bw) >> 1) * 2;
  int lastStep = nssWindow >> 6; if (lastStep == 0) lastStep = 1;
  if (!isChroma) {
    int cx[4], cy[4];
    cx[0] = gt[0] * lastStep;              cy[0] = gt[1] * lastStep;
    cx[1] = gt[2] * lastStep + bw * 2 - 1; cy[1] = gt[3] * lastStep;
    cx[2] = gt[4] * lastStep + bw * 2 - 1; cy[2] = gt[5] * lastStep + bh * 2 - 1;
    cx[3] = gt[6] * lastStep;              cy[3] = gt[7] * lastStep + bh * 2 - 1;
    hop_o_calc_param_projective(cx, cy, hp, bw * 2, bh * 2);
  } else {
    double ls = (double)lastStep, cx[4], cy[4];
    cx[0] = ((double)gt[0] / 2) * ls;                cy[0] = ((double)gt[1] / 2) * ls;
    cx[1] = (((double)gt[2] / 2) * ls) + bw * 2 - 1; cy[1] = ((double)gt[3] / 2) * ls;
    cx[2] = (((double)gt[4] / 2) * ls) + bw * 2 - 1; cy[2] = (((double)gt[5] / 2) * ls) + bh * 2 - 1;
    cx[3] = ((double)gt[6] / 2) * ls;                cy[3] = (((double)gt[7] / 2) * ls) + bh * 2 - 1;
    hop_o_calc_param_projective_c(cx, cy, hp, bw * 2, bh * 2);
  }
  int16_t* aux = (int16_t*)malloc((size_t)bw * bh * sizeof(int16_t));
  const int16_t* centre = patch + bw / 2 + (bh / 2) * (bw * 2);
  projective_transform(centre, aux, hp, bw * 2, bh * 2, bw * 2, nssWindow, 0, 8);
  for (int r = 0; r < bh; r++) memcpy(dst + r * dstStride, aux + r * bw, (size_t)bw * sizeof(int16_t));
  free(aux);
}

/* TLibCommon/TComPrediction.cpp:639-720 (xPredInterLumaBlk) and :1235-1347 (xPredInterChromaBlk),
 * uni-prediction (bi = false), both the plain and the GT branch.
 * refY/refCb/refCr address sample (0,0) of the padded SS-ref planes; mv in quarter-pel. */
void hop_o_pred_inter(const int16_t* refY, int strideY, const int16_t* refCb, const int16_t* refCr, int strideC,
                      int puX, int puY, int w, int h, int mvx, int mvy, int useGT, const int gt[8],
                      int bitDepthY, int bitDepthC, int16_t* predY, int16_t* predCb, int16_t* predCr)
{
  int anyGT = 0;
  for (int k = 0; k < 8; k++) if (gt[k] != 0) anyGT = 1;
  int cw = w >> 1, ch = h >> 1;
  if (!useGT || !anyGT) {
    const int16_t* r = refY + (ptrdiff_t)(puY + (mvy >> 2)) * strideY + puX + (mvx >> 2);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
      predY[y * w + x] = (int16_t)interp_sample(r, strideY, x, y, mvx & 3, mvy & 3, 8, bitDepthY, 0);
    const int16_t* rb = refCb + (ptrdiff_t)((puY >> 1) + (mvy >> 3)) * strideC + (puX >> 1) + (mvx >> 3);
    const int16_t* rr = refCr + (ptrdiff_t)((puY >> 1) + (mvy >> 3)) * strideC + (puX >> 1) + (mvx >> 3);
    for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
      predCb[y * cw + x] = (int16_t)interp_sample(rb, strideC, x, y, mvx & 7, mvy & 7, 4, bitDepthC, 0);
      predCr[y * cw + x] = (int16_t)interp_sample(rr, strideC, x, y, mvx & 7, mvy & 7, 4, bitDepthC, 0);
    }
    return;
  }
  /* GT branch: 2Wx2H patch at (mv>>2) - (W/2,H/2), interpolated at the MV's phase (:683-713) */
  int16_t* patch = (int16_t*)malloc((size_t)w * h * 4 * sizeof(int16_t));
  const int16_t* r = refY + (ptrdiff_t)(puY + (mvy >> 2) - h / 2) * strideY + puX + (mvx >> 2) - w / 2;
  for (int y = 0; y < 2 * h; y++) for (int x = 0; x < 2 * w; x++)
    patch[y * 2 * w + x] = (int16_t)interp_sample(r, strideY, x, y, mvx & 3, mvy & 3, 8, bitDepthY, 0);
  pred_gt_block(patch, w, h, gt, 0, predY, w);
  /* chroma: patch (2cw x 2ch) at (mv>>3) - (W/4,H/4) (:1295) */
  const int16_t* planes[2] = { refCb, refCr };
  int16_t* outs[2] = { predCb, predCr };
  for (int pl = 0; pl < 2; pl++) {
    const int16_t* rc = planes[pl] + (ptrdiff_t)((puY >> 1) + (mvy >> 3) - h / 4) * strideC + (puX >> 1) + (mvx >> 3) - w / 4;
    for (int y = 0; y < 2 * ch; y++) for (int x = 0; x < 2 * cw; x++)
      patch[y * 2 * cw + x] = (int16_t)interp_sample(rc, strideC, x, y, mvx & 7, mvy & 7, 4, bitDepthC, 0);
    pred_gt_block(patch, cw, ch, gt, 1, outs[pl], cw);
  }
  free(patch);
}

/* ------------------------------------------------------------------------------------------ */
/* One PU through the ME chain of xMotionEstimation, TEncSearch.cpp:4552-4656 (see also          */
/* oracle/ref_harness.cpp:ref_me_pu which drives the reference's own functions the same way).    */
/* out layout identical to ref_me_pu.                                                            */
/* ------------------------------------------------------------------------------------------ */
int hop_o_me_pu(const int16_t* org, int orgStride, const int16_t* refY00, int refStride, int puX, int puY, int w, int h,
                int rngL, int rngR, int rngT, int rngB, int offX, int offY,
                int predX, int predY, int nAmvp, const int* amvpXY, uint32_t lambdaCost,
                int fen, int useHad, int bitDepth, int stage, int64_t* out)
{
  const int16_t* refPU = refY00 + (ptrdiff_t)puY * refStride + puX;
  int mv[2] = {0, 0}; uint32_t cost = 0;
  hop_o_ss_search(org, orgStride, refPU, refStride, w, h, rngL, rngR, rngT, rngB, offX, offY,
                  predX, predY, lambdaCost, fen, bitDepth, &mv[0], &mv[1], &cost);
  out[0] = mv[0]; out[1] = mv[1]; out[2] = cost; out[25] = mv[0]; out[26] = mv[1];
  /* :4603-4606; bufY[0] is the first sample of the padded buffer = (-80,-80) */
  int notValid = (cost == HOP_MAX_UINT) || (mv[0] == 0 && mv[1] == 0) || (refY00[-80 * refStride - 80] == HOP_NOT_VALID);
  out[3] = notValid;
  if (notValid || stage < 2) return 0;
  int half[2], qter[2];
  cost = hop_o_frac_search(org, orgStride, refPU, refStride, w, h, mv[0], mv[1], predX, predY, lambdaCost, useHad, bitDepth, half, qter);
  out[4] = half[0]; out[5] = half[1]; out[6] = qter[0]; out[7] = qter[1]; out[8] = cost;
  if (stage < 3) return 0;
  int gt[8]; int ssBest[2] = { mv[0], mv[1] };
  int flag = hop_o_gt_search(org, orgStride, refPU, refStride, w, h, mv, half, qter, ssBest, nAmvp, amvpXY,
                             predX, predY, lambdaCost, useHad, bitDepth, &cost, gt);
  out[9] = flag;
  for (int k = 0; k < 8; k++) out[10 + k] = gt[k];
  out[18] = cost; out[19] = mv[0]; out[20] = mv[1]; out[21] = half[0]; out[22] = half[1]; out[23] = qter[0]; out[24] = qter[1];
  return 0;
}
