/* hop_oracle_rdoq.c -- CPU restatement of the reference's rate-distortion optimised quantisation.
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
 * (libhophip.so) never does.
 *
 * Follows TComTrQuant::xRateDistOptQuant (TLibCommon/TComTrQuant.cpp:1489-1999) with xGetCodedLevel (:2123-2173),
 * xGetICRate (:2182-2240), xGetRateSigCoeffGroup / xGetRateLast / xGetRateSigCoef / xGetICost / xGetIEPRate (:2242-2289),
 * calcPatternSigCtx (:2008-2026), getSigCtxInc (:2038-2092), getSigCoeffGroupCtxInc (:2291-2313), the flat-list
 * branch of setErrScaleCoeff (:2360-2379) and the scan tables of TComRom.cpp (initSigLastScan :355-481,
 * g_sigLastScan8x8 :344-349, g_uiGroupIdx :353).  Every cost is a double computed in the reference's operation
 * order (build with -ffp-contract=off).  Pinned against the reference's own function through oracle/ref_harness.cpp
 * (ref_rdoq) with seeded coefficient blocks and seeded bit-estimate tables: tests/golden/rdoq.npz. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include "hop_oracle.h"

#define SCAN_DIAG 0
#define SCAN_HOR 1
#define SCAN_VER 2
#define C1FLAG_NUMBER 8
#define C2FLAG_NUMBER 1
#define COEF_REMAIN_BIN_REDUCTION 3
#define SBH_THRESHOLD 4
#define SCAN_SET_SIZE 16
#define LOG2_SCAN_SET_SIZE 4
#define MLS_GRP_NUM 64
#define MLS_CG_SIZE 4
#define QUANT_SHIFT 14
#define SCALE_BITS 15
#define MAX_TR_DYNAMIC_RANGE 15
#define NUM_QT_CBF_CTX 4

static const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      /* TComRom.cpp:164-167 */
static const int inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                     /* :169-172 */
static const uint32_t sig_last_scan_8x8[3][4] = { {0, 2, 1, 3}, {0, 1, 2, 3}, {0, 2, 1, 3} };   /* :344-349 */
static const uint32_t group_idx[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };   /* :353 */

/* g_auiSigLastScan[scan][i], block side 2 << i (i = 0..4), and g_sigLastScanCG32x32 (the diagonal scan of an 8x8 grid) */
static uint32_t* scan_tab[3][5];
static uint32_t scan_cg32[64];
static int scan_ready = 0;

/* TComRom.cpp:355-481 */
static void init_sig_last_scan(uint32_t* D, uint32_t* Hs, uint32_t* V, int w, int log2_of_quarter)
{
  const uint32_t numPos = (uint32_t)(w * w);
  uint32_t next = 0;
  if (w < 16) {
    uint32_t* T = (w == 8) ? scan_cg32 : D;
    for (uint32_t line = 0; next < numPos; line++) {
      int prim = (int)line, scnd = 0;
      while (prim >= w) { scnd++; prim--; }
      while (prim >= 0 && scnd < w) { T[next++] = (uint32_t)(prim * w + scnd); scnd++; prim--; }
    }
  }
  if (w > 4) {
    const uint32_t nSide = (uint32_t)w >> 2, nBlks = nSide * nSide;
    for (uint32_t blk = 0; blk < nBlks; blk++) {
      next = 0;
      uint32_t init = scan_tab[SCAN_DIAG][log2_of_quarter][blk];        /* diag scan of the (w/4)x(w/4) grid */
      if (w == 32) init = scan_cg32[blk];
      const uint32_t offY = init / nSide, offX = init - offY * nSide;
      const uint32_t offD = 4 * (offX + offY * (uint32_t)w), offScan = 16 * blk;
      for (uint32_t line = 0; next < 16; line++) {
        int prim = (int)line, scnd = 0;
        while (prim >= 4) { scnd++; prim--; }
        while (prim >= 0 && scnd < 4) { D[next + offScan] = (uint32_t)(prim * w + scnd) + offD; next++; scnd++; prim--; }
      }
    }
  }
  uint32_t cnt = 0;
  if (w > 2) {
    const int nSide = w >> 2;
    for (int by = 0; by < nSide; by++) for (int bx = 0; bx < nSide; bx++) {
      const uint32_t off = (uint32_t)(by * 4 * w + bx * 4);
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) Hs[cnt++] = (uint32_t)(y * w + x) + off;
    }
    cnt = 0;
    for (int bx = 0; bx < nSide; bx++) for (int by = 0; by < nSide; by++) {
      const uint32_t off = (uint32_t)(by * 4 * w + bx * 4);
      for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) V[cnt++] = (uint32_t)(y * w + x) + off;
    }
  } else {
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) Hs[cnt++] = (uint32_t)(y * w + x);
    cnt = 0;
    for (int x = 0; x < w; x++) for (int y = 0; y < w; y++) V[cnt++] = (uint32_t)(y * w + x);
  }
}

/* initROM, TComRom.cpp:62-72: sides 2,4,8,16,32 (the reference goes on to 64, which no transform uses) */
void hop_o_scan_init(void)
{
  if (scan_ready) return;
  int c = 2;
  for (int i = 0; i < 5; i++) {
    for (int s = 0; s < 3; s++) scan_tab[s][i] = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(c * c));
    /* "log2Blk = g_aucConvertToBit[nSide] + 1" = index of the table of side w/4: side 2 -> 0, 4 -> 1, 8 -> 2 */
    const int q = c >> 2, lq = (q == 2) ? 0 : (q == 4) ? 1 : (q == 8) ? 2 : 0;
    init_sig_last_scan(scan_tab[0][i], scan_tab[1][i], scan_tab[2][i], c, lq);
    c <<= 1;
  }
  scan_ready = 1;
}

/* scan position -> raster position of an N x N block (N = 4..32); scan_idx 0 diag, 1 hor, 2 ver */
const uint32_t* hop_o_scan(int scan_idx, int log2_size) { hop_o_scan_init(); return scan_tab[scan_idx][log2_size - 1]; }
/* the coefficient-group scan the RDOQ walks (TComTrQuant.cpp:1543-1553) */
const uint32_t* hop_o_scan_cg(int scan_idx, int log2_size)
{
  hop_o_scan_init();
  if (log2_size == 3) return sig_last_scan_8x8[scan_idx];
  if (log2_size == 5) return scan_cg32;
  return scan_tab[scan_idx][log2_size > 3 ? log2_size - 2 - 1 : 0];
}

/* TComDataCU::getCoefScanIdx, TComDataCU.cpp:4001-4056.  dir = luma intra mode (for chroma: the chroma mode with DM
 * already resolved to the luma mode of the CU's first partition) */
int hop_o_coef_scan_idx(int width, int is_luma, int is_intra, int dir)
{
  if (!is_intra) return SCAN_DIAG;
  int ctx;
  switch (width) { case 2: ctx = 6; break; case 4: ctx = 5; break; case 8: ctx = 4; break; case 16: ctx = 3; break;
                   case 32: ctx = 2; break; case 64: ctx = 1; break; default: ctx = 0; break; }
  int scan = SCAN_DIAG;
  const int VER_IDX = 26, HOR_IDX = 10;
  if (is_luma) {
    if (ctx > 3 && ctx < 6) scan = abs(dir - VER_IDX) < 5 ? SCAN_HOR : (abs(dir - HOR_IDX) < 5 ? SCAN_VER : SCAN_DIAG);
  } else {
    if (ctx > 4 && ctx < 7) scan = abs(dir - VER_IDX) < 5 ? SCAN_HOR : (abs(dir - HOR_IDX) < 5 ? SCAN_VER : SCAN_DIAG);
  }
  return scan;
}

/* :2008-2026 */
static int calc_pattern_sig_ctx(const uint32_t* cgFlag, uint32_t px, uint32_t py, int width, int height)
{
  if (width == 4 && height == 4) return -1;
  uint32_t sigRight = 0, sigLower = 0;
  width >>= 2; height >>= 2;
  if (px < (uint32_t)(width - 1)) sigRight = (cgFlag[py * width + px + 1] != 0);
  if (py < (uint32_t)(height - 1)) sigLower = (cgFlag[(py + 1) * width + px] != 0);
  return (int)(sigRight + (sigLower << 1));
}

/* :2038-2092; comp 0 = TEXT_LUMA */
static int get_sig_ctx_inc(int patternSigCtx, uint32_t scanIdx, int posX, int posY, int log2BlockSize, int is_luma)
{
  static const int ctxIndMap[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
  if (posX + posY == 0) return 0;
  if (log2BlockSize == 2) return ctxIndMap[4 * posY + posX];
  const int offset = log2BlockSize == 3 ? (scanIdx == SCAN_DIAG ? 9 : 15) : (is_luma ? 21 : 12);
  const int xs = posX - ((posX >> 2) << 2), ys = posY - ((posY >> 2) << 2);
  int cnt;
  if (patternSigCtx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
  else if (patternSigCtx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
  else if (patternSigCtx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
  else cnt = 2;
  return ((is_luma && ((posX >> 2) + (posY >> 2)) > 0) ? 3 : 0) + offset + cnt;
}

/* :2291-2313 */
static uint32_t get_sig_cg_ctx_inc(const uint32_t* cgFlag, uint32_t px, uint32_t py, int width, int height)
{
  uint32_t r = 0, l = 0;
  width >>= 2; height >>= 2;
  if (px < (uint32_t)(width - 1)) r = (cgFlag[py * width + px + 1] != 0);
  if (py < (uint32_t)(height - 1)) l = (cgFlag[(py + 1) * width + px] != 0);
  return (r || l);
}

/* :2182-2240 */
static int get_ic_rate(const hop_o_estbits* eb, uint32_t absLevel, uint32_t ctxOne, uint32_t ctxAbs, uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx)
{
  int rate = 32768;                                                       /* Int(xGetIEPRate()) */
  const uint32_t baseLevel = (c1Idx < C1FLAG_NUMBER) ? (2 + (c2Idx < C2FLAG_NUMBER)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < ((uint32_t)COEF_REMAIN_BIN_REDUCTION << goRice)) {
      length = symbol >> goRice;
      rate += (int)((length + 1 + goRice) << 15);
    } else {
      length = goRice;
      symbol = symbol - ((uint32_t)COEF_REMAIN_BIN_REDUCTION << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((COEF_REMAIN_BIN_REDUCTION + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < C1FLAG_NUMBER) {
      rate += eb->greaterOneBits[ctxOne][1];
      if (c2Idx < C2FLAG_NUMBER) rate += eb->levelAbsBits[ctxAbs][1];
    }
  } else if (absLevel == 1) {
    rate += eb->greaterOneBits[ctxOne][0];
  } else if (absLevel == 2) {
    rate += eb->greaterOneBits[ctxOne][1];
    rate += eb->levelAbsBits[ctxAbs][0];
  } else {
    rate = 0;
  }
  return rate;
}

/* :2123-2173 */
static uint32_t get_coded_level(const hop_o_estbits* eb, double lambda, double* codedCost, double* codedCost0, double* codedCostSig,
                                int levelDouble, uint32_t maxAbsLevel, uint32_t ctxSig, uint32_t ctxOne, uint32_t ctxAbs, uint32_t goRice,
                                uint32_t c1Idx, uint32_t c2Idx, int qBits, double dTemp, int bLast)
{
  double currCostSig = 0;
  uint32_t bestAbsLevel = 0;
  if (!bLast && maxAbsLevel < 3) {
    *codedCostSig = lambda * (double)eb->significantBits[ctxSig][0];
    *codedCost = *codedCost0 + *codedCostSig;
    if (maxAbsLevel == 0) return bestAbsLevel;
  } else {
    *codedCost = 1.7e+308;                                                /* MAX_DOUBLE, TypeDef.h */
  }
  if (!bLast) currCostSig = lambda * (double)eb->significantBits[ctxSig][1];
  const uint32_t minAbsLevel = (maxAbsLevel > 1 ? maxAbsLevel - 1 : 1);
  for (int absLevel = (int)maxAbsLevel; (uint32_t)absLevel >= minAbsLevel; absLevel--) {
    const double err = (double)(levelDouble - (int)((uint32_t)absLevel << qBits));
    double currCost = err * err * dTemp + lambda * (double)get_ic_rate(eb, (uint32_t)absLevel, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
    currCost += currCostSig;
    if (currCost < *codedCost) { bestAbsLevel = (uint32_t)absLevel; *codedCost = currCost; *codedCostSig = currCostSig; }
  }
  return bestAbsLevel;
}

/* :2253-2268 */
static double get_rate_last(const hop_o_estbits* eb, double lambda, uint32_t posX, uint32_t posY)
{
  const uint32_t ctxX = group_idx[posX], ctxY = group_idx[posY];
  double cost = (double)(eb->lastXBits[ctxX] + eb->lastYBits[ctxY]);
  if (ctxX > 3) cost += 32768.0 * (double)((ctxX - 2) >> 1);
  if (ctxY > 3) cost += 32768.0 * (double)((ctxY - 2) >> 1);
  return lambda * cost;
}

/* src: transform coefficients (Int, raster N x N); dst: levels with sign; *abs_sum is ADDED to like the reference's uiAbsSum.
 * comp 0 luma, 1 Cb, 2 Cr; tr_depth = getTransformIdx; qp_scaled as setQPforQuant hands to setQpParam (:192-214);
 * bit_depth of the component; flat scaling list (m_scalingListEnabledFlag false -> setFlatScalingList, :2416-2434). */
int hop_o_rdoq(const int32_t* src, int32_t* dst, int log2_size, int comp, int is_intra, int scan_idx, int tr_depth,
               int qp_scaled, int bit_depth, int sign_hide, double lambda, const hop_o_estbits* eb, uint32_t* abs_sum)
{
  if (log2_size < 2 || log2_size > 5 || comp < 0 || comp > 2 || scan_idx < 0 || scan_idx > 2 || qp_scaled < 0) return -1;
  hop_o_scan_init();
  const int is_luma = comp == 0;
  const uint32_t width = 1u << log2_size, height = width;
  const int per = qp_scaled / 6, rem = qp_scaled % 6;
  const int transformShift = MAX_TR_DYNAMIC_RANGE - bit_depth - log2_size;
  uint32_t goRice = 0;
  double blockUncodedCost = 0;
  const uint32_t maxNumCoeff = width * height;
  const int qBits = QUANT_SHIFT + per + transformShift;
  /* flat list: every quantiser coefficient is g_quantScales[rem] (:2436-2450); setErrScaleCoeff :2373-2377 */
  const int q = quant_scales[rem];
  double errScale = (double)(1 << SCALE_BITS);
  errScale = errScale * pow(2.0, -2.0 * transformShift);
  const double dTemp = errScale / q / q / (1 << (2 * (bit_depth - 8)));

  static __thread double costCoeff[32 * 32], costSig[32 * 32], costCoeff0[32 * 32];
  static __thread int rateIncUp[32 * 32], rateIncDown[32 * 32], sigRateDelta[32 * 32], deltaU[32 * 32];
  memset(costCoeff, 0, sizeof(double) * maxNumCoeff);
  memset(costSig, 0, sizeof(double) * maxNumCoeff);
  memset(rateIncUp, 0, sizeof(int) * maxNumCoeff);
  memset(rateIncDown, 0, sizeof(int) * maxNumCoeff);
  memset(sigRateDelta, 0, sizeof(int) * maxNumCoeff);
  memset(deltaU, 0, sizeof(int) * maxNumCoeff);

  const uint32_t* scanCG = hop_o_scan_cg(scan_idx, log2_size);
  const uint32_t cgSize = 1u << MLS_CG_SIZE;
  double costCoeffGroupSig[MLS_GRP_NUM];
  uint32_t sigCoeffGroupFlag[MLS_GRP_NUM];
  const uint32_t numBlkSide = width / MLS_CG_SIZE;
  int cgLastScanPos = -1;
  uint32_t ctxSet = 0;
  int c1 = 1, c2 = 0;
  double baseCost = 0;
  int lastScanPos = -1;
  uint32_t c1Idx = 0, c2Idx = 0;
  int baseLevel;
  const uint32_t* scan = hop_o_scan(scan_idx, log2_size);
  memset(costCoeffGroupSig, 0, sizeof(costCoeffGroupSig));
  memset(sigCoeffGroupFlag, 0, sizeof(sigCoeffGroupFlag));
  const uint32_t cgNum = width * height >> MLS_CG_SIZE;
  int scanPos;
  struct { int nnzBeforePos0; double codedLevelandDist, uncodedDist, sigCost, sigCost0; } rd;

  for (int cgScanPos = (int)cgNum - 1; cgScanPos >= 0; cgScanPos--) {
    const uint32_t cgBlkPos = scanCG[cgScanPos];
    const uint32_t cgPosY = cgBlkPos / numBlkSide, cgPosX = cgBlkPos - (cgPosY * numBlkSide);
    memset(&rd, 0, sizeof(rd));
    const int patternSigCtx = calc_pattern_sig_ctx(sigCoeffGroupFlag, cgPosX, cgPosY, (int)width, (int)height);
    for (int scanPosinCG = (int)cgSize - 1; scanPosinCG >= 0; scanPosinCG--) {
      scanPos = cgScanPos * (int)cgSize + scanPosinCG;
      const uint32_t blkPos = scan[scanPos];
      int levelDouble = src[blkPos];
      {
        int64_t v = (int64_t)abs(levelDouble) * q, cap = (int64_t)0x7FFFFFFF - (1 << (qBits - 1));
        levelDouble = (int)(v < cap ? v : cap);
      }
      const uint32_t maxAbsLevel = (uint32_t)((levelDouble + (1 << (qBits - 1))) >> qBits);
      const double err = (double)levelDouble;
      costCoeff0[scanPos] = err * err * dTemp;
      blockUncodedCost += costCoeff0[scanPos];
      dst[blkPos] = (int32_t)maxAbsLevel;
      if (maxAbsLevel > 0 && lastScanPos < 0) {
        lastScanPos = scanPos;
        ctxSet = (scanPos < SCAN_SET_SIZE || !is_luma) ? 0 : 2;
        cgLastScanPos = cgScanPos;
      }
      if (lastScanPos >= 0) {
        uint32_t level;
        const uint32_t oneCtx = 4 * ctxSet + (uint32_t)c1, absCtx = ctxSet + (uint32_t)c2;
        if (scanPos == lastScanPos) {
          level = get_coded_level(eb, lambda, &costCoeff[scanPos], &costCoeff0[scanPos], &costSig[scanPos], levelDouble, maxAbsLevel, 0,
                                  oneCtx, absCtx, goRice, c1Idx, c2Idx, qBits, dTemp, 1);
        } else {
          const uint32_t posY = blkPos >> log2_size, posX = blkPos - (posY << log2_size);
          const uint32_t ctxSig = (uint32_t)get_sig_ctx_inc(patternSigCtx, (uint32_t)scan_idx, (int)posX, (int)posY, log2_size, is_luma);
          level = get_coded_level(eb, lambda, &costCoeff[scanPos], &costCoeff0[scanPos], &costSig[scanPos], levelDouble, maxAbsLevel, ctxSig,
                                  oneCtx, absCtx, goRice, c1Idx, c2Idx, qBits, dTemp, 0);
          sigRateDelta[blkPos] = eb->significantBits[ctxSig][1] - eb->significantBits[ctxSig][0];
        }
        deltaU[blkPos] = (levelDouble - ((int)level << qBits)) >> (qBits - 8);
        if (level > 0) {
          const int rateNow = get_ic_rate(eb, level, oneCtx, absCtx, goRice, c1Idx, c2Idx);
          rateIncUp[blkPos] = get_ic_rate(eb, level + 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
          rateIncDown[blkPos] = get_ic_rate(eb, level - 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
        } else {
          rateIncUp[blkPos] = eb->greaterOneBits[oneCtx][0];
        }
        dst[blkPos] = (int32_t)level;
        baseCost += costCoeff[scanPos];
        baseLevel = (c1Idx < C1FLAG_NUMBER) ? (2 + (c2Idx < C2FLAG_NUMBER)) : 1;
        if (level >= (uint32_t)baseLevel) {
          if (level > 3u * (1u << goRice)) goRice = (goRice + 1 < 4u) ? goRice + 1 : 4u;
        }
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if ((c1 < 3) && (c1 > 0) && level) c1++;
        if ((scanPos % SCAN_SET_SIZE == 0) && (scanPos > 0)) {
          c2 = 0; goRice = 0; c1Idx = 0; c2Idx = 0;
          ctxSet = (scanPos == SCAN_SET_SIZE || !is_luma) ? 0 : 2;
          if (c1 == 0) ctxSet++;
          c1 = 1;
        }
      } else {
        baseCost += costCoeff0[scanPos];
      }
      rd.sigCost += costSig[scanPos];
      if (scanPosinCG == 0) rd.sigCost0 = costSig[scanPos];
      if (dst[blkPos]) {
        sigCoeffGroupFlag[cgBlkPos] = 1;
        rd.codedLevelandDist += costCoeff[scanPos] - costSig[scanPos];
        rd.uncodedDist += costCoeff0[scanPos];
        if (scanPosinCG != 0) rd.nnzBeforePos0++;
      }
    }
    if (cgLastScanPos >= 0) {
      if (cgScanPos) {
        if (sigCoeffGroupFlag[cgBlkPos] == 0) {
          const uint32_t ctxSig = get_sig_cg_ctx_inc(sigCoeffGroupFlag, cgPosX, cgPosY, (int)width, (int)height);
          baseCost += lambda * (double)eb->significantCoeffGroupBits[ctxSig][0] - rd.sigCost;
          costCoeffGroupSig[cgScanPos] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
        } else {
          if (cgScanPos < cgLastScanPos) {
            if (rd.nnzBeforePos0 == 0) { baseCost -= rd.sigCost0; rd.sigCost -= rd.sigCost0; }
            double costZeroCG = baseCost;
            const uint32_t ctxSig = get_sig_cg_ctx_inc(sigCoeffGroupFlag, cgPosX, cgPosY, (int)width, (int)height);
            if (cgScanPos < cgLastScanPos) {
              baseCost += lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
              costZeroCG += lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
              costCoeffGroupSig[cgScanPos] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
            }
            costZeroCG += rd.uncodedDist;
            costZeroCG -= rd.codedLevelandDist;
            costZeroCG -= rd.sigCost;
            if (costZeroCG < baseCost) {
              sigCoeffGroupFlag[cgBlkPos] = 0;
              baseCost = costZeroCG;
              if (cgScanPos < cgLastScanPos) costCoeffGroupSig[cgScanPos] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
              for (int p = (int)cgSize - 1; p >= 0; p--) {
                scanPos = cgScanPos * (int)cgSize + p;
                const uint32_t bp = scan[scanPos];
                if (dst[bp]) { dst[bp] = 0; costCoeff[scanPos] = costCoeff0[scanPos]; costSig[scanPos] = 0; }
              }
            }
          }
        }
      } else {
        sigCoeffGroupFlag[cgBlkPos] = 1;
      }
    }
  }

  /* ===== estimate last position ===== */
  if (lastScanPos < 0) return 0;
  double bestCost = 0;
  int ctxCbf = 0, bestLastIdxP1 = 0;
  if (!is_intra && is_luma && tr_depth == 0) {
    ctxCbf = 0;
    bestCost = blockUncodedCost + lambda * (double)eb->blockRootCbpBits[ctxCbf][0];
    baseCost += lambda * (double)eb->blockRootCbpBits[ctxCbf][1];
  } else {
    ctxCbf = is_luma ? (tr_depth == 0 ? 1 : 0) : tr_depth;             /* getCtxQtCbf, TComDataCU.cpp:1848-1859 */
    ctxCbf = (is_luma ? 0 : 1) * NUM_QT_CBF_CTX + ctxCbf;              /* ( eTType ? TEXT_CHROMA : eTType ) * NUM_QT_CBF_CTX */
    bestCost = blockUncodedCost + lambda * (double)eb->blockCbpBits[ctxCbf][0];
    baseCost += lambda * (double)eb->blockCbpBits[ctxCbf][1];
  }
  int foundLast = 0;
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0; cgScanPos--) {
    const uint32_t cgBlkPos = scanCG[cgScanPos];
    baseCost -= costCoeffGroupSig[cgScanPos];
    if (sigCoeffGroupFlag[cgBlkPos]) {
      for (int p = (int)cgSize - 1; p >= 0; p--) {
        scanPos = cgScanPos * (int)cgSize + p;
        if (scanPos > lastScanPos) continue;
        const uint32_t blkPos = scan[scanPos];
        if (dst[blkPos]) {
          const uint32_t posY = blkPos >> log2_size, posX = blkPos - (posY << log2_size);
          const double costLast = scan_idx == SCAN_VER ? get_rate_last(eb, lambda, posY, posX) : get_rate_last(eb, lambda, posX, posY);
          const double totalCost = baseCost + costLast - costSig[scanPos];
          if (totalCost < bestCost) { bestLastIdxP1 = scanPos + 1; bestCost = totalCost; }
          if (dst[blkPos] > 1) { foundLast = 1; break; }
          baseCost -= costCoeff[scanPos];
          baseCost += costCoeff0[scanPos];
        } else {
          baseCost -= costSig[scanPos];
        }
      }
      if (foundLast) break;
    }
  }
  for (int sp = 0; sp < bestLastIdxP1; sp++) {
    const int bp = (int)scan[sp];
    const int level = dst[bp];
    *abs_sum += (uint32_t)level;
    dst[bp] = (src[bp] < 0) ? -level : level;
  }
  for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) dst[scan[sp]] = 0;

  /* ===== sign bit hiding, :1883-1998 ===== */
  if (sign_hide && *abs_sum >= 2) {
    /* Int arithmetic in the reference: inv*inv*(1<<(2*per)) wraps for per >= 10 (qp_scaled >= 60); reproduced as wrapping int32 */
    const int32_t prod = (int32_t)((uint32_t)(inv_quant_scales[rem] * inv_quant_scales[rem]) * (uint32_t)(1u << ((2 * per) & 31)));
    const int64_t rdFactor = (int64_t)((double)prod / lambda / 16 / (1 << (2 * (bit_depth - 8))) + 0.5);
    int lastCG = -1, absSum = 0, n;
    for (int subSet = (int)((width * height - 1) >> LOG2_SCAN_SET_SIZE); subSet >= 0; subSet--) {
      const int subPos = subSet << LOG2_SCAN_SET_SIZE;
      int firstNZPosInCG = SCAN_SET_SIZE, lastNZPosInCG = -1;
      absSum = 0;
      for (n = SCAN_SET_SIZE - 1; n >= 0; --n) if (dst[scan[n + subPos]]) { lastNZPosInCG = n; break; }
      for (n = 0; n < SCAN_SET_SIZE; n++) if (dst[scan[n + subPos]]) { firstNZPosInCG = n; break; }
      for (n = firstNZPosInCG; n <= lastNZPosInCG; n++) absSum += dst[scan[n + subPos]];
      if (lastNZPosInCG >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZPosInCG - firstNZPosInCG >= SBH_THRESHOLD) {
        const uint32_t signbit = (dst[scan[subPos + firstNZPosInCG]] > 0 ? 0 : 1);
        if (signbit != (uint32_t)(absSum & 0x1)) {
          int64_t minCostInc = INT64_MAX, curCost = INT64_MAX;
          int minPos = -1, finalChange = 0, curChange = 0;
          for (n = (lastCG == 1 ? lastNZPosInCG : SCAN_SET_SIZE - 1); n >= 0; --n) {
            const uint32_t blkPos = scan[n + subPos];
            if (dst[blkPos] != 0) {
              const int64_t costUp = rdFactor * (-deltaU[blkPos]) + rateIncUp[blkPos];
              int64_t costDown = rdFactor * (deltaU[blkPos]) + rateIncDown[blkPos] - ((abs(dst[blkPos]) == 1) ? sigRateDelta[blkPos] : 0);
              if (lastCG == 1 && lastNZPosInCG == n && abs(dst[blkPos]) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else {
                curChange = -1;
                if (n == firstNZPosInCG && abs(dst[blkPos]) == 1) curCost = INT64_MAX; else curCost = costDown;
              }
            } else {
              curCost = rdFactor * (-(abs(deltaU[blkPos]))) + (1 << 15) + rateIncUp[blkPos] + sigRateDelta[blkPos];
              curChange = 1;
              if (n < firstNZPosInCG) {
                const uint32_t thissignbit = (src[blkPos] >= 0 ? 0 : 1);
                if (thissignbit != signbit) curCost = INT64_MAX;
              }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = (int)blkPos; }
          }
          if (dst[minPos] == 32767 || dst[minPos] == -32768) finalChange = -1;
          if (src[minPos] >= 0) dst[minPos] += finalChange; else dst[minPos] -= finalChange;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return 0;
}
