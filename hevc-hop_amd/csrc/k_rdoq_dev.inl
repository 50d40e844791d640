// k_rdoq_dev.inl -- the device side of the rate-distortion optimised quantisation of one transform unit (see k_rdoq.hip), shared by k_rdoq (one lane per TU, 64 TUs
// per wave) and the fused leaf kernel of k_cabac.hip (one TU per workgroup).
#define RQ_SCAN_DIAG 0
#define RQ_SCAN_VER 2
#define RQ_C1FLAG_NUMBER 8
#define RQ_C2FLAG_NUMBER 1
#define RQ_COEF_REMAIN_BIN_REDUCTION 3
#define RQ_SBH_THRESHOLD 4
#define RQ_SCAN_SET_SIZE 16

static __constant__ int c_rq_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // TComRom.cpp:164-167
static __constant__ int c_rq_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                     // :169-172
static __constant__ uint8_t c_rq_group_idx[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };   // :353
static __constant__ uint8_t c_rq_ctx_ind_map[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

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

// One TU on the calling lane.  src / dst: its coefficients and levels; scan / scanCG: the TU's coefficient and coefficient-group scans; cgSig: CGN doubles `cgs` apart (the cost of each group's
// coded_sub_block_flag); wd: a work area of N2 * ws * RQ_WORK_PER_COEF bytes laid out [array][scan position][ws lanes], wl: the calling lane's column in it (ws = 64: the lanes of a
// wave side by side, k_rdoq; ws = 1: a TU of its own, k_turd_fused).  A lane only ever reads what it wrote itself: no barrier in here.
template <int LOG2>
__device__ static void rdoq_tu(const hop_rdoq_job& jb, const hop_estbits* __restrict__ eb, const uint16_t* __restrict__ scan, const uint16_t* __restrict__ scanCG,
                               double* __restrict__ cgSig, const int cgs, const int32_t* __restrict__ src, int32_t* __restrict__ dst,
                               uint32_t* __restrict__ abs_sum_out, double* __restrict__ wd, const int ws, const int wl) {
  constexpr int N2 = 1 << (2 * LOG2), WCG = (1 << LOG2) >> 2, CGN = N2 >> 4;
  double* const wcc = wd + wl; double* const wcs = wd + (size_t)N2 * ws + wl;
  int* const wi = (int*)(wd + (size_t)2 * N2 * ws) + wl;
  int* const wlvl = wi; int* const wlvlD = wi + (size_t)N2 * ws; int* const wrUp = wi + (size_t)2 * N2 * ws;
  int* const wrDn = wi + (size_t)3 * N2 * ws; int* const wsDelta = wi + (size_t)4 * N2 * ws; int* const wdU = wi + (size_t)5 * N2 * ws;
#define W(a, sp) a[(size_t)(sp) * ws]
  const bool is_luma = jb.comp == 0;
  const int scan_idx = jb.scan_idx;
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
    cgSig[(size_t)cgScanPos * cgs] = 0;
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
          cgSig[(size_t)cgScanPos * cgs] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
        } else if (cgScanPos < cgLastScanPos) {
          if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
          double costZeroCG = baseCost;
          const uint32_t ctxSig = rq_sig_cg_ctx(cgFlag, cgPosX, cgPosY, WCG);
          baseCost += lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
          costZeroCG += lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
          cgSig[(size_t)cgScanPos * cgs] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][1];
          costZeroCG += uncodedDist;
          costZeroCG -= codedLevelandDist;
          costZeroCG -= sigCost;
          if (costZeroCG < baseCost) {
            cgFlag &= ~(1ull << cgBlkPos);
            baseCost = costZeroCG;
            cgSig[(size_t)cgScanPos * cgs] = lambda * (double)eb->significantCoeffGroupBits[ctxSig][0];
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
      baseCost -= cgSig[(size_t)cgScanPos * cgs];
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
  *abs_sum_out = absSum;
  for (int sp = 0; sp < N2; sp++) dst[scan[sp]] = (sp <= lastScanPos) ? W(wlvl, sp) : 0;
#undef W
}

