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

#define RQ_SCAN_DIAG 0
#define RQ_SCAN_VER 2
#define RQ_C1FLAG_NUMBER 8
#define RQ_C2FLAG_NUMBER 1
#define RQ_COEF_REMAIN_BIN_REDUCTION 3
#define RQ_SBH_THRESHOLD 4
#define RQ_SCAN_SET_SIZE 16

__constant__ int c_rq_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // TComRom.cpp:164-167
__constant__ int c_rq_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                     // :169-172
__constant__ uint8_t c_rq_group_idx[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };   // :353
__constant__ uint8_t c_rq_ctx_ind_map[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

// scan tables, built on the host (hop_rdoq_build_scans) and kept in device memory: scan[s][log2-2] at offset
// s*1360 + {0,16,80,336}; coefficient-group scans cg[s][log2-2] at 4080 + s*85 + {0,1,5,21}
__device__ static inline const uint16_t* rq_scan(const uint16_t* tabs, int s, int log2) {
  const int off = (log2 == 2) ? 0 : (log2 == 3) ? 16 : (log2 == 4) ? 80 : 336;
  return tabs + s * 1360 + off;
}
__device__ static inline const uint16_t* rq_scan_cg(const uint16_t* tabs, int s, int log2) {
  const int off = (log2 == 2) ? 0 : (log2 == 3) ? 1 : (log2 == 4) ? 5 : 21;
  return tabs + 4080 + s * 85 + off;
}

__device__ static inline int rq_ic_rate(const hop_estbits* eb, uint32_t absLevel, uint32_t ctxOne, uint32_t ctxAbs, uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx) {
  int rate = 32768;                                                       // Int(xGetIEPRate())
  const uint32_t baseLevel = (c1Idx < RQ_C1FLAG_NUMBER) ? (2 + (c2Idx < RQ_C2FLAG_NUMBER)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < ((uint32_t)RQ_COEF_REMAIN_BIN_REDUCTION << goRice)) {
      length = symbol >> goRice;
      rate += (int)((length + 1 + goRice) << 15);
    } else {
      length = goRice;
      symbol = symbol - ((uint32_t)RQ_COEF_REMAIN_BIN_REDUCTION << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((RQ_COEF_REMAIN_BIN_REDUCTION + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < RQ_C1FLAG_NUMBER) {
      rate += eb->greaterOneBits[ctxOne][1];
      if (c2Idx < RQ_C2FLAG_NUMBER) rate += eb->levelAbsBits[ctxAbs][1];
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

__device__ static inline uint32_t rq_coded_level(const hop_estbits* eb, double lambda, double& codedCost, double codedCost0, double& codedCostSig,
                                                 int levelDouble, uint32_t maxAbsLevel, uint32_t ctxSig, uint32_t ctxOne, uint32_t ctxAbs, uint32_t goRice,
                                                 uint32_t c1Idx, uint32_t c2Idx, int qBits, double dTemp, bool bLast) {
  double currCostSig = 0;
  uint32_t bestAbsLevel = 0;
  if (!bLast && maxAbsLevel < 3) {
    codedCostSig = lambda * (double)eb->significantBits[ctxSig][0];
    codedCost = codedCost0 + codedCostSig;
    if (maxAbsLevel == 0) return bestAbsLevel;
  } else {
    codedCost = 1.7e+308;                                                 // MAX_DOUBLE, CommonDef.h:121
  }
  if (!bLast) currCostSig = lambda * (double)eb->significantBits[ctxSig][1];
  const uint32_t minAbsLevel = (maxAbsLevel > 1 ? maxAbsLevel - 1 : 1);
  for (int absLevel = (int)maxAbsLevel; (uint32_t)absLevel >= minAbsLevel; absLevel--) {
    const double err = (double)(levelDouble - (int)((uint32_t)absLevel << qBits));
    double currCost = err * err * dTemp + lambda * (double)rq_ic_rate(eb, (uint32_t)absLevel, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
    currCost += currCostSig;
    if (currCost < codedCost) { bestAbsLevel = (uint32_t)absLevel; codedCost = currCost; codedCostSig = currCostSig; }
  }
  return bestAbsLevel;
}

// the coded_sub_block_flags of a TU are one 64-bit mask per lane (bit = raster index of the group, at most 8 x 8 groups)
__device__ static inline int rq_pattern_sig_ctx(unsigned long long cgFlag, uint32_t px, uint32_t py, int wcg) {
  if (wcg == 1) return -1;                                                // 4x4 block
  uint32_t r = 0, l = 0;
  if (px < (uint32_t)(wcg - 1)) r = (uint32_t)(cgFlag >> (py * wcg + px + 1)) & 1u;
  if (py < (uint32_t)(wcg - 1)) l = (uint32_t)(cgFlag >> ((py + 1) * wcg + px)) & 1u;
  return (int)(r + (l << 1));
}
__device__ static inline uint32_t rq_sig_cg_ctx(unsigned long long cgFlag, uint32_t px, uint32_t py, int wcg) {
  uint32_t r = 0, l = 0;
  if (px < (uint32_t)(wcg - 1)) r = (uint32_t)(cgFlag >> (py * wcg + px + 1)) & 1u;
  if (py < (uint32_t)(wcg - 1)) l = (uint32_t)(cgFlag >> ((py + 1) * wcg + px)) & 1u;
  return (r | l);
}
__device__ static inline int rq_sig_ctx_inc(int patternSigCtx, int scanIdx, int posX, int posY, int log2BlockSize, bool is_luma) {
  if (posX + posY == 0) return 0;
  if (log2BlockSize == 2) return c_rq_ctx_ind_map[4 * posY + posX];
  const int offset = log2BlockSize == 3 ? (scanIdx == RQ_SCAN_DIAG ? 9 : 15) : (is_luma ? 21 : 12);
  const int xs = posX & 3, ys = posY & 3;
  int cnt;
  if (patternSigCtx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
  else if (patternSigCtx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
  else if (patternSigCtx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
  else cnt = 2;
  return ((is_luma && ((posX >> 2) + (posY >> 2)) > 0) ? 3 : 0) + offset + cnt;
}
__device__ static inline double rq_rate_last(const hop_estbits* eb, double lambda, uint32_t posX, uint32_t posY) {
  const uint32_t ctxX = c_rq_group_idx[posX], ctxY = c_rq_group_idx[posY];
  double cost = (double)(eb->lastXBits[ctxX] + eb->lastYBits[ctxY]);
  if (ctxX > 3) cost += 32768.0 * (double)((ctxX - 2) >> 1);
  if (ctxY > 3) cost += 32768.0 * (double)((ctxY - 2) >> 1);
  return lambda * cost;
}

// bytes of per-coefficient state a lane keeps in the work area: cost of the chosen level and of its significance flag (double),
// level, |coeff| * Q, and the four sign-hiding terms (int)
#define RQ_WORK_PER_COEF 40

template <int LOG2>
__global__ __launch_bounds__(64) void k_rdoq(const hop_rdoq_job* __restrict__ jobs, const hop_estbits* __restrict__ tables, const uint16_t* __restrict__ scans,
                                             const int32_t* __restrict__ src_all, int32_t* __restrict__ dst_all, uint32_t* __restrict__ abs_sum_out,
                                             const int* __restrict__ list, const int* __restrict__ count_ptr, char* __restrict__ work) {
  constexpr int N2 = 1 << (2 * LOG2), WCG = (1 << LOG2) >> 2, CGN = N2 >> 4;
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
  double* const wcc = wd + lane; double* const wcs = wd + (size_t)N2 * 64 + lane;
  int* const wi = (int*)(wd + (size_t)2 * N2 * 64);
  int* const wlvl = wi + lane; int* const wlvlD = wi + (size_t)N2 * 64 + lane; int* const wrUp = wi + (size_t)2 * N2 * 64 + lane;
  int* const wrDn = wi + (size_t)3 * N2 * 64 + lane; int* const wsDelta = wi + (size_t)4 * N2 * 64 + lane; int* const wdU = wi + (size_t)5 * N2 * 64 + lane;
#define W(a, sp) a[(size_t)(sp) << 6]

  for (int g = blockIdx.x; g * 64 < count; g += gridDim.x) {
    const int idx = g * 64 + lane;
    if (idx >= count) continue;                                           // no barrier below: a lane only ever reads what it wrote itself
    const int ti = list[idx];
    const hop_rdoq_job jb = jobs[ti];
    const int32_t* src = src_all + jb.coeff_offset;
    int32_t* dst = dst_all + jb.coeff_offset;
    const hop_estbits* eb = tables + jb.estbits_index;
    const bool is_luma = jb.comp == 0;
    const int scan_idx = jb.scan_idx;
    const uint16_t* scan = s_scan[scan_idx];
    const uint16_t* scanCG = s_scanCG[scan_idx];
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6;
    const int transformShift = 15 - jb.bit_depth - LOG2;                  // MAX_TR_DYNAMIC_RANGE - bitDepth - log2
    const int qBits = 14 + per + transformShift;                          // QUANT_SHIFT + per + shift
    const int q = c_rq_quant_scales[rem];
    const double errScale = ldexp((double)(1 << 15), -2 * transformShift);  // (1 << SCALE_BITS) * pow(2.0, -2.0*iTransformShift): exact
    const double dTemp = errScale / q / q / (1 << (2 * (jb.bit_depth - 8)));
    const double lambda = jb.lambda;
    const long long cap = (long long)0x7FFFFFFF - (1 << (qBits - 1));
    uint32_t absSum = 0;

    // ---- the level decision: one walk back along the scan, TComTrQuant.cpp:1574-1792 ----
    unsigned long long cgFlag = 0;
    uint32_t goRice = 0, ctxSet = 0, c1Idx = 0, c2Idx = 0;
    int c1 = 1, c2 = 0, lastScanPos = -1, cgLastScanPos = -1;
    double baseCost = 0, blockUncodedCost = 0;
    for (int cgScanPos = CGN - 1; cgScanPos >= 0; cgScanPos--) {
      const uint32_t cgBlkPos = scanCG[cgScanPos];
      const uint32_t cgPosY = cgBlkPos / WCG, cgPosX = cgBlkPos - cgPosY * WCG;
      int nnzBeforePos0 = 0; double codedLevelandDist = 0, uncodedDist = 0, sigCost = 0, sigCost0 = 0;
      const int patternSigCtx = rq_pattern_sig_ctx(cgFlag, cgPosX, cgPosY, WCG);
      uint32_t nzmask = 0;                                                // positions of this group whose chosen level is not zero
      s_cgSig[cgScanPos][lane] = 0;
      for (int scanPosinCG = 15; scanPosinCG >= 0; scanPosinCG--) {
        const int sp = cgScanPos * 16 + scanPosinCG;
        const uint32_t blkPos = scan[sp];
        // quantisation and the cost of coding nothing, :1545-1570
        int v = src[blkPos]; v = v < 0 ? -v : v;
        const long long t = (long long)v * q;
        const int levelDouble = (int)(t < cap ? t : cap);
        const uint32_t maxAbsLevel = (uint32_t)(levelDouble + (1 << (qBits - 1))) >> qBits;
        const double err0 = (double)levelDouble;
        const double cost0 = err0 * err0 * dTemp;
        W(wlvlD, sp) = levelDouble;
        blockUncodedCost += cost0;
        if (maxAbsLevel > 0 && lastScanPos < 0) {
          lastScanPos = sp;
          ctxSet = (sp < RQ_SCAN_SET_SIZE || !is_luma) ? 0 : 2;
          cgLastScanPos = cgScanPos;
        }
        double cc = 0, cs = 0;
        uint32_t level = 0;
        if (lastScanPos >= 0) {
          const uint32_t oneCtx = 4 * ctxSet + (uint32_t)c1, absCtx = ctxSet + (uint32_t)c2;
          int sDelta = 0;
          if (sp == lastScanPos) {
            level = rq_coded_level(eb, lambda, cc, cost0, cs, levelDouble, maxAbsLevel, 0, oneCtx, absCtx, goRice, c1Idx, c2Idx, qBits, dTemp, true);
          } else {
            const uint32_t posY = blkPos >> LOG2, posX = blkPos - (posY << LOG2);
            const uint32_t ctxSig = (uint32_t)rq_sig_ctx_inc(patternSigCtx, scan_idx, (int)posX, (int)posY, LOG2, is_luma);
            level = rq_coded_level(eb, lambda, cc, cost0, cs, levelDouble, maxAbsLevel, ctxSig, oneCtx, absCtx, goRice, c1Idx, c2Idx, qBits, dTemp, false);
            sDelta = eb->significantBits[ctxSig][1] - eb->significantBits[ctxSig][0];
          }
          W(wsDelta, sp) = sDelta;
          W(wdU, sp) = (levelDouble - (int)(level << qBits)) >> (qBits - 8);
          if (level > 0) {
            const int rateNow = rq_ic_rate(eb, level, oneCtx, absCtx, goRice, c1Idx, c2Idx);
            W(wrUp, sp) = rq_ic_rate(eb, level + 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
            W(wrDn, sp) = rq_ic_rate(eb, level - 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
          } else {
            W(wrUp, sp) = eb->greaterOneBits[oneCtx][0];
            W(wrDn, sp) = 0;
          }
          baseCost += cc;
          const uint32_t baseLevel = (c1Idx < RQ_C1FLAG_NUMBER) ? (2 + (c2Idx < RQ_C2FLAG_NUMBER)) : 1;
          if (level >= baseLevel) {
            if (level > 3u * (1u << goRice)) goRice = (goRice + 1 < 4u) ? goRice + 1 : 4u;
          }
          if (level >= 1) c1Idx++;
          if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
          else if ((c1 < 3) && (c1 > 0) && level) c1++;
          if ((sp % RQ_SCAN_SET_SIZE == 0) && (sp > 0)) {
            c2 = 0; goRice = 0; c1Idx = 0; c2Idx = 0;
            ctxSet = (sp == RQ_SCAN_SET_SIZE || !is_luma) ? 0 : 2;
            if (c1 == 0) ctxSet++;
            c1 = 1;
          }
          W(wcc, sp) = cc; W(wcs, sp) = cs;
        } else {
          baseCost += cost0;
        }
        W(wlvl, sp) = (int)level;
        sigCost += cs;
        if (scanPosinCG == 0) sigCost0 = cs;
        if (level) {
          nzmask |= 1u << scanPosinCG;
          codedLevelandDist += cc - cs;
          uncodedDist += cost0;
          if (scanPosinCG != 0) nnzBeforePos0++;
        }
      }
      if (nzmask) cgFlag |= 1ull << cgBlkPos;
      if (cgLastScanPos >= 0) {
        if (cgScanPos) {
          if (!nzmask) {
            const uint32_t ctxSig = rq_sig_cg_ctx(cgFlag, cgPosX, cgPosY, WCG);
            baseCost += lambda * (double)eb->significantCoeffGroupBits[ctxSig][0] - sigCost;
            s_cgSig[cgScanPos][lane] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
          } else if (cgScanPos < cgLastScanPos) {
            if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
            double costZeroCG = baseCost;
            const uint32_t ctxSig = rq_sig_cg_ctx(cgFlag, cgPosX, cgPosY, WCG);
            baseCost += lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
            costZeroCG += lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
            s_cgSig[cgScanPos][lane] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
            costZeroCG += uncodedDist;
            costZeroCG -= codedLevelandDist;
            costZeroCG -= sigCost;
            if (costZeroCG < baseCost) {
              cgFlag &= ~(1ull << cgBlkPos);
              baseCost = costZeroCG;
              s_cgSig[cgScanPos][lane] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
              for (int p = 15; p >= 0; p--) {
                if (!((nzmask >> p) & 1u)) continue;
                const int sp = cgScanPos * 16 + p;
                const double e = (double)W(wlvlD, sp);
                W(wlvl, sp) = 0; W(wcc, sp) = e * e * dTemp; W(wcs, sp) = 0;
              }
            }
          }
        } else {
          cgFlag |= 1ull << cgBlkPos;
        }
      }
    }
    // ---- the last position, :1794-1866 ----
    int bestLastIdxP1 = 0;
    if (lastScanPos >= 0) {
      double bestCost;
      if (!jb.is_intra && is_luma && jb.tr_depth == 0) {
        bestCost = blockUncodedCost + lambda * (double)eb->blockRootCbpBits[0][0];
        baseCost += lambda * (double)eb->blockRootCbpBits[0][1];
      } else {
        int ctxCbf = is_luma ? (jb.tr_depth == 0 ? 1 : 0) : jb.tr_depth;          // getCtxQtCbf, TComDataCU.cpp:1848-1859
        ctxCbf = (is_luma ? 0 : 1) * 4 + ctxCbf;                                    // NUM_QT_CBF_CTX
        bestCost = blockUncodedCost + lambda * (double)eb->blockCbpBits[ctxCbf][0];
        baseCost += lambda * (double)eb->blockCbpBits[ctxCbf][1];
      }
      bool foundLast = false;
      for (int cgScanPos = cgLastScanPos; cgScanPos >= 0 && !foundLast; cgScanPos--) {
        const uint32_t cgBlkPos = scanCG[cgScanPos];
        baseCost -= s_cgSig[cgScanPos][lane];
        if ((cgFlag >> cgBlkPos) & 1ull) {
          for (int p = 15; p >= 0; p--) {
            const int sp = cgScanPos * 16 + p;
            if (sp > lastScanPos) continue;
            const int lv = W(wlvl, sp);
            if (lv) {
              const uint32_t blkPos = scan[sp];
              const uint32_t posY = blkPos >> LOG2, posX = blkPos - (posY << LOG2);
              const double costLast = scan_idx == RQ_SCAN_VER ? rq_rate_last(eb, lambda, posY, posX) : rq_rate_last(eb, lambda, posX, posY);
              const double totalCost = baseCost + costLast - W(wcs, sp);
              if (totalCost < bestCost) { bestLastIdxP1 = sp + 1; bestCost = totalCost; }
              if (lv > 1) { foundLast = true; break; }
              const double e = (double)W(wlvlD, sp);
              baseCost -= W(wcc, sp);
              baseCost += e * e * dTemp;
            } else {
              baseCost -= W(wcs, sp);
            }
          }
        }
      }
      for (int sp = 0; sp < bestLastIdxP1; sp++) {
        const int level = W(wlvl, sp);
        if (level) {
          absSum += (uint32_t)level;
          if (src[scan[sp]] < 0) W(wlvl, sp) = -level;
        }
      }
      for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) W(wlvl, sp) = 0;
    }
    // ---- sign bit hiding, :1883-1998 ----
    if (lastScanPos >= 0 && jb.sign_hide && absSum >= 2) {
      // Int arithmetic in the reference: inv*inv*(1<<(2*per)) wraps for per >= 10; reproduced as wrapping 32-bit
      const int32_t prod = (int32_t)((uint32_t)(c_rq_inv_quant_scales[rem] * c_rq_inv_quant_scales[rem]) * (uint32_t)(1u << ((2 * per) & 31)));
      const long long rdFactor = (long long)((double)prod / lambda / 16 / (1 << (2 * (jb.bit_depth - 8))) + 0.5);
      int lastCG = -1;
      for (int subSet = lastScanPos >> 4; subSet >= 0; subSet--) {        // the groups above hold no level
        const int subPos = subSet << 4;
        int firstNZ = RQ_SCAN_SET_SIZE, lastNZ = -1, sum = 0, n;
        for (n = RQ_SCAN_SET_SIZE - 1; n >= 0; --n) if (W(wlvl, n + subPos)) { lastNZ = n; break; }
        for (n = 0; n < RQ_SCAN_SET_SIZE; n++) if (W(wlvl, n + subPos)) { firstNZ = n; break; }
        for (n = firstNZ; n <= lastNZ; n++) sum += W(wlvl, n + subPos);
        if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
        if (lastNZ - firstNZ >= RQ_SBH_THRESHOLD) {
          const uint32_t signbit = (W(wlvl, subPos + firstNZ) > 0 ? 0 : 1);
          if (signbit != (uint32_t)(sum & 0x1)) {
            long long minCostInc = 0x7FFFFFFFFFFFFFFFLL, curCost = 0x7FFFFFFFFFFFFFFFLL;
            int minPos = -1, finalChange = 0, curChange = 0;
            for (n = (lastCG == 1 ? lastNZ : RQ_SCAN_SET_SIZE - 1); n >= 0; --n) {
              const int sp = n + subPos;
              const int lv = W(wlvl, sp), alv = lv < 0 ? -lv : lv;
              const int du = W(wdU, sp);
              if (lv != 0) {
                const long long costUp = rdFactor * (-du) + W(wrUp, sp);
                long long costDown = rdFactor * (du) + W(wrDn, sp) - ((alv == 1) ? W(wsDelta, sp) : 0);
                if (lastCG == 1 && lastNZ == n && alv == 1) costDown -= (4 << 15);
                if (costUp < costDown) { curCost = costUp; curChange = 1; }
                else {
                  curChange = -1;
                  if (n == firstNZ && alv == 1) curCost = 0x7FFFFFFFFFFFFFFFLL; else curCost = costDown;
                }
              } else {
                curCost = rdFactor * (-(du < 0 ? -du : du)) + (1 << 15) + W(wrUp, sp) + W(wsDelta, sp);
                curChange = 1;
                if (n < firstNZ) {
                  const uint32_t thissignbit = (src[scan[sp]] >= 0 ? 0 : 1);
                  if (thissignbit != signbit) curCost = 0x7FFFFFFFFFFFFFFFLL;
                }
              }
              if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = sp; }
            }
            const int lm = W(wlvl, minPos);
            if (lm == 32767 || lm == -32768) finalChange = -1;
            W(wlvl, minPos) = (src[scan[minPos]] >= 0) ? lm + finalChange : lm - finalChange;
          }
        }
        if (lastCG == 1) lastCG = 0;
      }
    }
    abs_sum_out[ti] = absSum;
    for (int sp = 0; sp < N2; sp++) dst[scan[sp]] = (sp <= lastScanPos) ? W(wlvl, sp) : 0;
  }
#undef W
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
