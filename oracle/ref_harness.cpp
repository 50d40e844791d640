// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C entry points around the *real* reference functions of the hot path
// (zinsayon/HEVC-HOP, an HM-15.0 fork).  It is compiled by oracle/Makefile.ref
// against the reference sources where they lie under /root/reference and linked
// with the objects built from them (oracle/_ref/libhmref.a); nothing of the
// reference is copied here.  The shared object it yields
// (oracle/_ref/libref_harness.so) is used for two things only:
//   * pinning oracle/hop_oracle.c (our CPU restatement) against the reference,
//   * generating the golden vectors in tests/golden/ (oracle/make_golden.py).
//
// Private/protected members of the reference classes are reached by the usual
// test-harness trick of redefining the access keywords before including the
// reference headers (standard headers are included first so they are unaffected).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cassert>
#include <vector>
#include <list>
#include <map>
#include <string>
#include <sstream>
#include <iostream>
#include <fstream>
#include <algorithm>
#include <stdint.h>

#define private public
#define protected public
#include "TLibCommon/TypeDef.h"
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComMv.h"
#include "TLibCommon/TComPattern.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComPicYuv.h"
#include "TLibCommon/TComYuv.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComSlice.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncSearch.h"
#include "TLibEncoder/TEncSbac.h"
#include "TLibEncoder/TEncBinCoderCABACCounter.h"
#include "TLibCommon/TComBitCounter.h"
#include "TLibCommon/ContextTables.h"
#undef private
#undef protected

// free functions with external linkage in TComTrQuant.cpp (TComTrQuant.cpp:786,829)
void xTrMxN(Int bitDepth, Short *block, Short *coeff, Int iWidth, Int iHeight, UInt uiMode);
void xITrMxN(Int bitDepth, Short *coeff, Short *block, Int iWidth, Int iHeight, UInt uiMode);

namespace {
struct Ctx {
  TEncCfg     cfg;
  TComTrQuant trq;
  TComRdCost  rd;
  TEncSearch  search;
  TComPicYuv* pic;
  TComDataCU  cu;
  bool        inited;
  Ctx() : pic(NULL), inited(false) {}
};
Ctx* g = NULL;
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// initialisation: what TAppEncCfg::xSetGlobal (TAppEncCfg.cpp:1901-1922), TEncTop::create
// (TEncTop.cpp:92), TEncCu::create (TEncCu.cpp:107-112) and TEncTop::init (TEncTop.cpp:299-310)
// do for the members this harness touches; cfg = cfg/3DHencoder_intra_main.cfg
// ---------------------------------------------------------------------------------------------
int ref_init(int bitDepthY, int bitDepthC, int useHadME, int useFastEnc, int searchRange)
{
  if (g) return 0;
  g = new Ctx;
  g_uiMaxCUWidth = 64; g_uiMaxCUHeight = 64;
  g_uiAddCUDepth = 1; g_uiMaxCUDepth = 4;
  g_bitDepthY = bitDepthY; g_bitDepthC = bitDepthC;
  initROM();
  UInt* piTmp = &g_auiZscanToRaster[0];
  initZscanToRaster(5, 1, 0, piTmp);          // m_uhTotalDepth = g_uiMaxCUDepth + 1
  initRasterToZscan(64, 64, 5);
  initRasterToPelXY(64, 64, 5);
  g->cfg.m_uiQuadtreeTULog2MaxSize = 5;
  g->cfg.m_uiQuadtreeTULog2MinSize = 2;
  g->cfg.m_bUseHADME = useHadME != 0;
  g->cfg.m_bUseFastEnc = useFastEnc != 0;
  g->rd.init();
  g->trq.init(1 << 5, true, true, true, true, false);
  g->search.init(&g->cfg, &g->trq, searchRange, 4, 0, 0, NULL, &g->rd, NULL, NULL);
  g->inited = true;
  return 0;
}

void ref_set_bitdepth(int bitDepthY, int bitDepthC) { g_bitDepthY = bitDepthY; g_bitDepthC = bitDepthC; }
void ref_set_lambda(double lambda) { g->rd.setLambda(lambda); }
unsigned ref_lambda_motion_sad() { return g->rd.m_uiLambdaMotionSAD; }

// ---------------------------------------------------------------------------------------------
// distortion kernels (TComRdCost.cpp:513-1016 SAD, :1018 SSE, :1366-1708 HAD, :391 calcHAD)
// ---------------------------------------------------------------------------------------------
unsigned ref_sad(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth, int subShift)
{
  TComPattern pat; pat.initPattern((Pel*)org, NULL, NULL, w, h, so, 0, 0);
  DistParam dp;
  g->rd.setDistParam(&pat, (Pel*)cur, sc, dp);
  dp.iSubShift = subShift; dp.bitDepth = bitDepth; dp.bApplyWeight = false;
  return dp.DistFunc(&dp);
}
unsigned ref_hads(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth)
{
  TComPattern pat; pat.initPattern((Pel*)org, NULL, NULL, w, h, so, 0, 0);
  DistParam dp;
  g->rd.setDistParam(&pat, (Pel*)cur, sc, 1, dp, true);
  dp.bitDepth = bitDepth; dp.bApplyWeight = false;
  return dp.DistFunc(&dp);
}
unsigned ref_sse(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth)
{
  return g->rd.getDistPart(bitDepth, (Pel*)cur, sc, (Pel*)org, so, w, h, TEXT_LUMA, DF_SSE);
}
unsigned ref_calc_had(const int16_t* a, int sa, const int16_t* b, int sb, int w, int h, int bitDepth)
{
  return g->rd.calcHAD(bitDepth, (Pel*)a, sa, (Pel*)b, sb, w, h);
}
unsigned ref_component_bits(int v) { return g->rd.xGetComponentBits(v); }
unsigned ref_bits_gt(const int* v) { return g->rd.getBitsGT(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]); }
double ref_calc_rd_cost(unsigned bits, unsigned dist, int flag) { return g->rd.calcRdCost(bits, dist, flag != 0); }

// ---------------------------------------------------------------------------------------------
// GT / HOP arithmetic (TComPrediction.cpp:807 calcParamProjective, :834 ...C, :904 ProjectiveTransform)
// ---------------------------------------------------------------------------------------------
void ref_calc_param_projective(const int* x, const int* y, double* h, int W, int H)
{
  Int xx[4], yy[4]; for (int i = 0; i < 4; i++) { xx[i] = x[i]; yy[i] = y[i]; }
  g->search.calcParamProjective(xx, yy, h, W, H);
}
void ref_calc_param_projective_c(const double* x, const double* y, double* h, int W, int H)
{
  Double xx[4], yy[4]; for (int i = 0; i < 4; i++) { xx[i] = x[i]; yy[i] = y[i]; }
  g->search.calcParamProjectiveC(xx, yy, h, W, H);
}
void ref_projective_transform(const int16_t* refCentre, int16_t* aux, const double* h, int W, int H, int stride, int nssWindow)
{
  Double hh[9]; memcpy(hh, h, sizeof(hh));
  g->search.ProjectiveTransform((Pel*)refCentre, (Pel*)aux, hh, W, H, stride, nssWindow);
}

// ---------------------------------------------------------------------------------------------
// SS reference picture (TComPicYuv) owned by the harness.  Layout = the reference's own:
// margins 80 luma / 40 chroma (TComPicYuv.cpp:82-85), stride = picW + 160.
// ---------------------------------------------------------------------------------------------
int ref_pic_create(int w, int h)
{
  if (g->pic) { g->pic->destroy(); delete g->pic; }
  g->pic = new TComPicYuv;
  g->pic->create(w, h, 64, 64, 4);
  return g->pic->getStride();
}
// sentinel fill: TComSlice::xGetRefPic path, TComSlice.cpp:252-253 -> setPicPel(NOT_VALID) TComPicYuv.cpp:199
void ref_pic_reset() { g->pic->setPicPel(NOT_VALID); g->pic->setBorderExtension(false); }
// raw access to the padded buffers (comp 0/1/2); returns pointer to sample (0,0), *stride out
int16_t* ref_pic_plane(int comp, int* stride)
{
  *stride = comp ? g->pic->getCStride() : g->pic->getStride();
  return comp == 0 ? g->pic->getLumaAddr() : comp == 1 ? g->pic->getCbAddr() : g->pic->getCrAddr();
}
// what TEncCu::xCopyYuv2SSRef (TEncCu.cpp:1677-1697) does for one finalised CU:
// copy recon into the picture, then whole-picture extendPicBorder (TComPicYuv.cpp:236-275)
void ref_pic_commit_cu(int x, int y, int size, const int16_t* recY, const int16_t* recCb, const int16_t* recCr)
{
  TComPicYuv* p = g->pic;
  for (int r = 0; r < size; r++) memcpy(p->getLumaAddr() + (y + r) * p->getStride() + x, recY + r * size, size * sizeof(Pel));
  int cs = size >> 1;
  for (int r = 0; r < cs; r++) {
    memcpy(p->getCbAddr() + ((y >> 1) + r) * p->getCStride() + (x >> 1), recCb + r * cs, cs * sizeof(Pel));
    memcpy(p->getCrAddr() + ((y >> 1) + r) * p->getCStride() + (x >> 1), recCr + r * cs, cs * sizeof(Pel));
  }
  p->setBorderExtension(false);
  p->extendPicBorder();
}
void ref_pic_extend_border() { g->pic->setBorderExtension(false); g->pic->extendPicBorder(); }

// clipMv for a CU at (cuX,cuY) (TComDataCU.cpp:3492-3504) + both xSetSearchRange overloads
// (TEncSearch.cpp:6204-6259).  The CU-dependent scalars they read are set on a default-constructed
// TComDataCU; the SPS/pic accessors used by the second overload are restated inline below because a
// TComPic/TComSlice graph is not constructible in isolation -- those two lines are marked RESTATED.
static void clip_mv(int picW, int picH, int cuX, int cuY, int& hor, int& ver)
{ // RESTATED from TComDataCU.cpp:3492-3504
  int iMvShift = 2, iOffset = 8;
  int iHorMax = (picW + iOffset - cuX - 1) << iMvShift;
  int iHorMin = (-(int)g_uiMaxCUWidth - iOffset - cuX + 1) << iMvShift;
  int iVerMax = (picH + iOffset - cuY - 1) << iMvShift;
  int iVerMin = (-(int)g_uiMaxCUHeight - iOffset - cuY + 1) << iMvShift;
  hor = std::min(iHorMax, std::max(iHorMin, hor));
  ver = std::min(iVerMax, std::max(iVerMin, ver));
}

// ---------------------------------------------------------------------------------------------
// One PU through the reference's ME chain, exactly the sequence of xMotionEstimation
// (TEncSearch.cpp:4552-4656) from the already-derived search range onwards:
//   getMotionCost(1,0); setPredictor; setCostScale(2); xPatternSearch; validity test :4603-4606;
//   getMotionCost(1,0); setCostScale(1); xPatternSearchFracDIF; setCostScale(0); xPatternSearchGT.
// The three searches are the reference's own member functions.
//   in : PU position (puX,puY) in the harness picture, size, org block, range after xSetSearchRange,
//        offX'/offY' (already transformed, :6239-6240), predictor (quarter-pel), AMVP candidates.
//   out[0..1]  integer MV           out[2] SAD (MV cost removed) or MAX_UINT    out[3] notValid flag
//   out[4..5]  half MV  out[6..7] quarter MV  out[8] cost after fractional search
//   out[9] gtFlag  out[10..17] GT0..GT3 (x,y)  out[18] cost after GT  out[19..20] integer MV after GT
//   out[21..22] half after GT, out[23..24] quarter after GT, out[25..26] ssBestCand[0]
// stage: 1 = integer only, 2 = + fractional, 3 = + GT
// ---------------------------------------------------------------------------------------------
int ref_me_pu(const int16_t* org, int orgStride, int puX, int puY, int w, int h,
              int rngL, int rngR, int rngT, int rngB, int offX, int offY,
              int predX, int predY, int nAmvp, const int* amvpXY, int stage, int64_t* out)
{
  TEncSearch& s = g->search;
  TComPicYuv* p = g->pic;
  Pel* piRefY = p->getLumaAddr() + puY * p->getStride() + puX;
  Int iRefStride = p->getStride();
  TComPattern key; key.initPattern((Pel*)org, NULL, NULL, w, h, orgStride, 0, 0);
  TComMv lt(rngL, rngT), rb(rngR, rngB), mv, pred(predX, predY), half, qter;
  TComMv best[1]; best[0].set(0, 0);
  UInt cost = 0;
  s.m_cDistParam.bApplyWeight = false;
  g->rd.getMotionCost(1, 0);
  g->rd.setPredictor(pred);
  g->rd.setCostScale(2);
  s.xPatternSearch(&key, piRefY, iRefStride, &lt, &rb, mv, cost, offX, offY, best, true);
  out[0] = mv.getHor(); out[1] = mv.getVer(); out[2] = cost; out[25] = best[0].getHor(); out[26] = best[0].getVer();
  bool notValid = (cost == MAX_UINT) || (mv.getHor() == 0 && mv.getVer() == 0) || (p->getBufY()[0] == NOT_VALID);
  out[3] = notValid;
  if (notValid || stage < 2) return 0;
  g->rd.getMotionCost(1, 0);
  g->rd.setCostScale(1);
  s.xPatternSearchFracDIF(&g->cu, &key, piRefY, iRefStride, &mv, half, qter, cost, false);
  g->rd.setCostScale(0);
  out[4] = half.getHor(); out[5] = half.getVer(); out[6] = qter.getHor(); out[7] = qter.getVer(); out[8] = cost;
  if (stage < 3) return 0;
  AMVPInfo* ai = g->cu.getCUMvField(REF_PIC_LIST_0)->getAMVPInfo();
  ai->iN = nAmvp;
  for (int i = 0; i < nAmvp; i++) ai->m_acMvCand[i].set(amvpXY[2 * i], amvpXY[2 * i + 1]);
  TComMv gt0, gt1, gt2, gt3; Bool gtFlag = false;
  s.xPatternSearchGT(&g->cu, &key, piRefY, iRefStride, &mv, &half, &qter, &gt0, &gt1, &gt2, &gt3, gtFlag, cost, false, best);
  out[9] = gtFlag;
  out[10] = gt0.getHor(); out[11] = gt0.getVer(); out[12] = gt1.getHor(); out[13] = gt1.getVer();
  out[14] = gt2.getHor(); out[15] = gt2.getVer(); out[16] = gt3.getHor(); out[17] = gt3.getVer();
  out[18] = cost; out[19] = mv.getHor(); out[20] = mv.getVer();
  out[21] = half.getHor(); out[22] = half.getVer(); out[23] = qter.getHor(); out[24] = qter.getVer();
  return 0;
}

// Search-range derivation for an ISS PU: xSetSearchRange(pred) TEncSearch.cpp:6204-6220 followed by the
// SS overload :6224-6259.  The arithmetic is the reference's via a TComDataCU whose position fields are
// set; the three pic/SPS accessors it needs are RESTATED through clip_mv above.
void ref_set_search_range(int picW, int picH, int cuX, int cuY, int cuSize, int ctuAddr, int frameWidthInCtu,
                          int predX, int predY, int srchRng, int offX, int offY, int firstRow, int firstCol, int* out)
{ // RESTATED control flow of TEncSearch.cpp:6204-6259 (integer arithmetic only)
  int iMvShift = 2;
  int ph = predX, pv = predY; clip_mv(picW, picH, cuX, cuY, ph, pv);
  int lh = ph - (srchRng << iMvShift), lv = pv - (srchRng << iMvShift);
  int rh = ph + (srchRng << iMvShift), rv = pv + (srchRng << iMvShift);
  // TComMv holds Short components
  lh = (Short)lh; lv = (Short)lv; rh = (Short)rh; rv = (Short)rv;
  clip_mv(picW, picH, cuX, cuY, lh, lv); clip_mv(picW, picH, cuX, cuY, rh, rv);
  lh >>= iMvShift; lv >>= iMvShift; rh >>= iMvShift; rv >>= iMvShift;
  int left = lh, right = rh, top = lv, bottom = rv;
  if (firstCol && firstRow) { right = left + 1; top = bottom + 1; }
  else {
    bottom = (bottom > (-offY - 4)) ? (-offY - 4) : bottom;
    offX = -offX - cuSize - 4;
    offY = -offY - cuSize - 4;
    bottom = (firstCol && (bottom > offY)) ? offY : bottom;
    right = (firstRow && (right > offX)) ? offX : right;
    right = (!firstRow && (ctuAddr < frameWidthInCtu) && (right > (offX + (cuSize << 1)))) ? (offX + (cuSize << 1)) : right;
  }
  lh = (Short)(left << iMvShift); lv = (Short)(top << iMvShift); rh = (Short)(right << iMvShift); rv = (Short)(bottom << iMvShift);
  clip_mv(picW, picH, cuX, cuY, lh, lv); clip_mv(picW, picH, cuX, cuY, rh, rv);
  out[0] = lh >> iMvShift; out[1] = rh >> iMvShift; out[2] = lv >> iMvShift; out[3] = rv >> iMvShift;
  out[4] = offX; out[5] = offY;
}

// ---------------------------------------------------------------------------------------------
// Final (normative, decoder-shared) predictor for one PU: the reference's own
// xPredInterLumaBlk / xPredInterChromaBlk (TComPrediction.cpp:639-720, :1235-1347), both branches.
// mv in quarter-pel, gt = GT0..GT3 (x,y).  Destination = 64x64 TComYuv, returned block at (0,0) layout
// predY[h][w], predCb/Cr[h/2][w/2].
// ---------------------------------------------------------------------------------------------
void ref_pred_inter(int puX, int puY, int w, int h, int mvx, int mvy, int useGT, const int* gt,
                    int16_t* predY, int16_t* predCb, int16_t* predCr)
{
  TEncSearch& s = g->search;
  TComPicYuv* p = g->pic;
  int wInCtu = (p->getWidth() + 63) / 64;
  g->cu.m_uiCUAddr = (puY / 64) * wInCtu + (puX / 64);
  g->cu.m_uiAbsIdxInLCU = 0;
  UInt partAddr = g_auiRasterToZscan[((puY & 63) >> 2) * 16 + ((puX & 63) >> 2)];
  TComYuv dst; dst.create(64, 64);
  TComYuv* pd = &dst;
  TComMv mv(mvx, mvy), g0(gt[0], gt[1]), g1(gt[2], gt[3]), g2(gt[4], gt[5]), g3(gt[6], gt[7]);
  s.xPredInterLumaBlk(&g->cu, p, partAddr, &mv, w, h, pd, false, useGT != 0, &g0, &g1, &g2, &g3);
  s.xPredInterChromaBlk(&g->cu, p, partAddr, &mv, w, h, pd, false, useGT != 0, &g0, &g1, &g2, &g3);
  Pel* y = dst.getLumaAddr(partAddr); Pel* cb = dst.getCbAddr(partAddr); Pel* cr = dst.getCrAddr(partAddr);
  for (int r = 0; r < h; r++) memcpy(predY + r * w, y + r * dst.getStride(), w * sizeof(Pel));
  for (int r = 0; r < h / 2; r++) {
    memcpy(predCb + r * (w / 2), cb + r * dst.getCStride(), (w / 2) * sizeof(Pel));
    memcpy(predCr + r * (w / 2), cr + r * dst.getCStride(), (w / 2) * sizeof(Pel));
  }
  dst.destroy();
  g->cu.m_uiCUAddr = 0;
}

// ---------------------------------------------------------------------------------------------
// transforms (TComTrQuant.cpp:786 xTrMxN, :829 xITrMxN): mode = REG_DCT(65535) or DST for 4x4 intra luma
// ---------------------------------------------------------------------------------------------
void ref_fwd_transform(int bitDepth, const int16_t* block, int16_t* coeff, int w, int h, unsigned mode)
{
  std::vector<Short> b(block, block + w * h);
  xTrMxN(bitDepth, &b[0], coeff, w, h, mode);
}
void ref_inv_transform(int bitDepth, const int16_t* coeff, int16_t* block, int w, int h, unsigned mode)
{
  std::vector<Short> c(coeff, coeff + w * h);
  xITrMxN(bitDepth, &c[0], block, w, h, mode);
}

// TComTrQuant::xDeQuant (TComTrQuant.cpp:1124-1183), flat scaling list; qpScaled as setQPforQuant hands to setQpParam
void ref_dequant_flat(int bitDepth, int qpScaled, const int32_t* level, int32_t* coef, int N)
{
  g->trq.m_cQP.setQpParam(qpScaled);
  g->trq.setUseScalingList(false);
  g->trq.xDeQuant(bitDepth, (const TCoeff*)level, (Int*)coef, N, N, 0);
}

// ---------------------------------------------------------------------------------------------
// row a11: TComTrQuant::xRateDistOptQuant (TComTrQuant.cpp:1489-1999) -- the reference's own function.  The TComDataCU
// only has to answer isIntra / getCoefScanIdx / getTransformIdx / getCtxQtCbf / getSlice()->getPPS()->getSignHideFlag():
// its per-partition arrays are pointed at one-element statics.  ttype: 0 TEXT_LUMA, 2 TEXT_CHROMA_U, 3 TEXT_CHROMA_V.
// estBits: an estBitsSbacStruct image (TComTrQuant.h:59-70).  chromaDir must not be DM_CHROMA_IDX (needs a TComPic).
// Returns the scan index the reference chose (getCoefScanIdx).
// ---------------------------------------------------------------------------------------------
int ref_rdoq(const int32_t* src, int32_t* dst, int N, int ttype, int isIntra, int lumaDir, int chromaDir, int trIdx,
             int qpScaled, int bitDepthY, int bitDepthC, int signHide, double lambda, const int32_t* estBits, uint32_t* absSum)
{
  static Char predMode[1]; static UChar trIdxA[1], lumaDirA[1], chromaDirA[1], depthA[1];
  static TComSlice* slice = NULL; static TComPPS* pps = NULL;
  if (!slice) { slice = new TComSlice; pps = new TComPPS; slice->setPPS(pps); }
  g_bitDepthY = bitDepthY; g_bitDepthC = bitDepthC;
  TComTrQuant& t = g->trq;
  t.init(32, true, true, true, false, false);
  t.m_cQP.setQpParam(qpScaled);
  t.setUseScalingList(false);
  t.setFlatScalingList();                                   // quantiser and error-scale tables for the current bit depths
  t.m_dLambda = lambda;
  memcpy(t.m_pcEstBitsSbac, estBits, sizeof(estBitsSbacStruct));
  pps->setSignHideFlag(signHide);
  TComDataCU& cu = g->cu;
  Char* sPred = cu.m_pePredMode; UChar* sTr = cu.m_puhTrIdx; UChar* sL = cu.m_puhLumaIntraDir; UChar* sC = cu.m_puhChromaIntraDir;
  UChar* sD = cu.m_puhDepth; TComSlice* sS = cu.m_pcSlice;
  predMode[0] = isIntra ? MODE_INTRA : MODE_INTER; trIdxA[0] = (UChar)trIdx; lumaDirA[0] = (UChar)lumaDir; chromaDirA[0] = (UChar)chromaDir; depthA[0] = 0;
  cu.m_pePredMode = predMode; cu.m_puhTrIdx = trIdxA; cu.m_puhLumaIntraDir = lumaDirA; cu.m_puhChromaIntraDir = chromaDirA; cu.m_puhDepth = depthA;
  cu.m_pcSlice = slice;
  std::vector<Int> in(src, src + N * N), arl(N * N, 0);
  std::vector<TCoeff> out(N * N, 0);
  Int* parl = &arl[0];
  UInt as = *absSum;
  t.xRateDistOptQuant(&cu, &in[0], &out[0], parl, N, N, as, (TextType)ttype, 0);
  *absSum = as;
  for (int i = 0; i < N * N; i++) dst[i] = out[i];
  const int scanIdx = (int)cu.getCoefScanIdx(0, N, ttype == 0, isIntra != 0);
  cu.m_pePredMode = sPred; cu.m_puhTrIdx = sTr; cu.m_puhLumaIntraDir = sL; cu.m_puhChromaIntraDir = sC; cu.m_puhDepth = sD; cu.m_pcSlice = sS;
  return scanIdx;
}

// row a10: TComTrQuant::xQuant, the non-RDOQ branch (TComTrQuant.cpp:1022-1119) -- the reference's own function with RDOQ switched off.  The branch takes its shift
// from the slice's QP base (ADAPTIVE_QP_SELECTION, :1032-1063) and its scale from m_cQP: both are set from the same qpScaled here (a CU at the slice QP).
// ttype: 0 TEXT_LUMA, 2 TEXT_CHROMA_U, 3 TEXT_CHROMA_V.  Sign-bit hiding off.  Returns uiAcSum.
static unsigned ref_quant_flat_impl(const int32_t* src, int32_t* dst, int N, int ttype, int isIntra, int isISlice, int qpScaled, int bitDepthY, int bitDepthC, int signHide, int lumaDir);
unsigned ref_quant_flat(const int32_t* src, int32_t* dst, int N, int ttype, int isIntra, int isISlice, int qpScaled, int bitDepthY, int bitDepthC)
{ return ref_quant_flat_impl(src, dst, N, ttype, isIntra, isISlice, qpScaled, bitDepthY, bitDepthC, 0, 0); }
// the same with the PPS's sign_data_hiding flag on: xQuant then runs signBitHidingHDQ (TComTrQuant.cpp:868-990, called :1110-1116) along the TU's scan, which
// getCoefScanIdx derives from the intra direction (lumaDir; 4x4 / 8x8 luma: 6..14 -> vertical scan, 22..30 -> horizontal, TComDataCU.cpp:4001-4056)
unsigned ref_quant_flat_sbh(const int32_t* src, int32_t* dst, int N, int ttype, int isIntra, int isISlice, int qpScaled, int bitDepthY, int bitDepthC, int lumaDir)
{ return ref_quant_flat_impl(src, dst, N, ttype, isIntra, isISlice, qpScaled, bitDepthY, bitDepthC, 1, lumaDir); }
static unsigned ref_quant_flat_impl(const int32_t* src, int32_t* dst, int N, int ttype, int isIntra, int isISlice, int qpScaled, int bitDepthY, int bitDepthC, int signHide, int lumaDir)
{
  static Char predMode[1]; static UChar trIdxA[1], lumaDirA[1], chromaDirA[1], depthA[1], tsA[3][1];
  static TComSlice* slice = NULL; static TComPPS* pps = NULL; static TComSPS* sps = NULL;
  if (!slice) { slice = new TComSlice; pps = new TComPPS; sps = new TComSPS; slice->setPPS(pps); slice->setSPS(sps); }
  g_bitDepthY = bitDepthY; g_bitDepthC = bitDepthC;
  TComTrQuant& t = g->trq;
  t.init(32, false, false, true, false, false);
  t.m_cQP.setQpParam(qpScaled);
  t.setUseScalingList(false);
  t.setFlatScalingList();
  pps->setSignHideFlag(signHide); pps->setChromaCbQpOffset(0); pps->setChromaCrQpOffset(0);
  sps->setQpBDOffsetY(0); sps->setQpBDOffsetC(0);
  slice->setSliceType(isISlice ? I_SLICE : P_SLICE);
  // the slice QP base that maps to qpScaled: luma directly; chroma through g_aucChromaScale (identity below 30, so the tests keep chroma QPs there or use the inverse)
  int base = qpScaled;
  if (ttype != 0) { base = -1; for (int q = 0; q <= 57; q++) if ((int)g_aucChromaScale[q] == qpScaled) { base = q; break; } if (base < 0) return 0xFFFFFFFFu; }
  slice->setSliceQpBase(base); slice->setSliceQpDeltaCb(0); slice->setSliceQpDeltaCr(0);
  TComDataCU& cu = g->cu;
  Char* sPred = cu.m_pePredMode; UChar* sTr = cu.m_puhTrIdx; UChar* sL = cu.m_puhLumaIntraDir; UChar* sC = cu.m_puhChromaIntraDir; UChar* sD = cu.m_puhDepth; TComSlice* sS = cu.m_pcSlice;
  UChar* sT[3] = { cu.m_puhTransformSkip[0], cu.m_puhTransformSkip[1], cu.m_puhTransformSkip[2] };
  predMode[0] = isIntra ? MODE_INTRA : MODE_INTER; trIdxA[0] = 0; lumaDirA[0] = (UChar)lumaDir; chromaDirA[0] = 0; depthA[0] = 0; tsA[0][0] = tsA[1][0] = tsA[2][0] = 0;
  cu.m_pePredMode = predMode; cu.m_puhTrIdx = trIdxA; cu.m_puhLumaIntraDir = lumaDirA; cu.m_puhChromaIntraDir = chromaDirA; cu.m_puhDepth = depthA; cu.m_pcSlice = slice;
  for (int k = 0; k < 3; k++) cu.m_puhTransformSkip[k] = tsA[k];
  std::vector<Int> in(src, src + N * N), arl(N * N, 0);
  std::vector<TCoeff> out(N * N, 0);
  Int* parl = &arl[0];
  UInt as = 0;
  t.xQuant(&cu, &in[0], &out[0], parl, N, N, as, (TextType)ttype, 0);
  for (int i = 0; i < N * N; i++) dst[i] = out[i];
  cu.m_pePredMode = sPred; cu.m_puhTrIdx = sTr; cu.m_puhLumaIntraDir = sL; cu.m_puhChromaIntraDir = sC; cu.m_puhDepth = sD; cu.m_pcSlice = sS;
  for (int k = 0; k < 3; k++) cu.m_puhTransformSkip[k] = sT[k];
  return as;
}

// ---------------------------------------------------------------------------------------------
// CABAC bit estimator for residual coding: the reference's own TEncSbac with the counting bin coder
// (TEncBinCABACCounter).  states[150] in the order of hop_o_cabac_ctx (oracle/hop_oracle.h): qt_cbf[8], trans_subdiv[3],
// qt_root_cbf[1], sig_cg[4], sig[42], last_x[30], last_y[30], one[24], abs[6], ts[2].
// ---------------------------------------------------------------------------------------------
namespace {
struct SbacCtx { TEncSbac sbac; TEncBinCABACCounter bin; TComBitCounter bits; TComSlice slice; TComSPS sps; TComPPS pps; bool up; SbacCtx() : up(false) {} };
SbacCtx* gs = NULL;
SbacCtx* sbac_get() {
  if (!gs) { gs = new SbacCtx; gs->sbac.init(&gs->bin); gs->sbac.setBitstream(&gs->bits); gs->slice.setSPS(&gs->sps); gs->slice.setPPS(&gs->pps);
             gs->sps.setMaxTrSize(32); gs->sbac.setSlice(&gs->slice); gs->bin.setBinCountingEnableFlag(false); }
  return gs;
}
struct SetRef { ContextModel* p; int n; };
void sbac_sets(TEncSbac& s, SetRef (&r)[10]) {
  r[0].p = s.m_cCUQtCbfSCModel.get(0); r[0].n = 8; r[1].p = s.m_cCUTransSubdivFlagSCModel.get(0); r[1].n = 3;
  r[2].p = s.m_cCUQtRootCbfSCModel.get(0); r[2].n = 1; r[3].p = s.m_cCUSigCoeffGroupSCModel.get(0); r[3].n = 4;
  r[4].p = s.m_cCUSigSCModel.get(0); r[4].n = 42; r[5].p = s.m_cCuCtxLastX.get(0); r[5].n = 30; r[6].p = s.m_cCuCtxLastY.get(0); r[6].n = 30;
  r[7].p = s.m_cCUOneSCModel.get(0); r[7].n = 24; r[8].p = s.m_cCUAbsSCModel.get(0); r[8].n = 6; r[9].p = s.m_cTransformSkipSCModel.get(0); r[9].n = 2;
}
void sbac_load(TEncSbac& s, const uint8_t* st) { SetRef r[10]; sbac_sets(s, r); int k = 0; for (int i = 0; i < 10; i++) for (int j = 0; j < r[i].n; j++) r[i].p[j].m_ucState = st[k++]; }
void sbac_store(TEncSbac& s, uint8_t* st) { SetRef r[10]; sbac_sets(s, r); int k = 0; for (int i = 0; i < 10; i++) for (int j = 0; j < r[i].n; j++) st[k++] = r[i].p[j].m_ucState; }
}

// ContextModel3DBuffer::initBuffer of the ten sets with the fork's tables (TEncSbac::resetEntropy, TEncSbac.cpp:136-148)
void ref_cabac_init(int sliceType, int qp, uint8_t* states)
{
  SbacCtx* c = sbac_get(); TEncSbac& s = c->sbac; SliceType t = (SliceType)sliceType;
  ContextModel::buildNextStateTable();
  s.m_cCUQtCbfSCModel.initBuffer(t, qp, (UChar*)INIT_QT_CBF); s.m_cCUTransSubdivFlagSCModel.initBuffer(t, qp, (UChar*)INIT_TRANS_SUBDIV_FLAG);
  s.m_cCUQtRootCbfSCModel.initBuffer(t, qp, (UChar*)INIT_QT_ROOT_CBF); s.m_cCUSigCoeffGroupSCModel.initBuffer(t, qp, (UChar*)INIT_SIG_CG_FLAG);
  s.m_cCUSigSCModel.initBuffer(t, qp, (UChar*)INIT_SIG_FLAG); s.m_cCuCtxLastX.initBuffer(t, qp, (UChar*)INIT_LAST); s.m_cCuCtxLastY.initBuffer(t, qp, (UChar*)INIT_LAST);
  s.m_cCUOneSCModel.initBuffer(t, qp, (UChar*)INIT_ONE_FLAG); s.m_cCUAbsSCModel.initBuffer(t, qp, (UChar*)INIT_ABS_FLAG);
  s.m_cTransformSkipSCModel.initBuffer(t, qp, (UChar*)INIT_TRANSFORMSKIP_FLAG);
  sbac_store(s, states);
}
// the CU-level sets in the order of hop_cabac_cu_ctx: skip[3], merge_flag, merge_idx, part_size[4], pred_mode, mvd[2], mvp_idx, gt_flag, gt[2], intra_pred, chroma_pred[2]
void ref_cabac_cu_init(int sliceType, int qp, uint8_t* states19)
{
  SbacCtx* c = sbac_get(); TEncSbac& s = c->sbac; SliceType t = (SliceType)sliceType;
  s.m_cCUSkipFlagSCModel.initBuffer(t, qp, (UChar*)INIT_SKIP_FLAG); s.m_cCUMergeFlagExtSCModel.initBuffer(t, qp, (UChar*)INIT_MERGE_FLAG_EXT);
  s.m_cCUMergeIdxExtSCModel.initBuffer(t, qp, (UChar*)INIT_MERGE_IDX_EXT); s.m_cCUPartSizeSCModel.initBuffer(t, qp, (UChar*)INIT_PART_SIZE);
  s.m_cCUPredModeSCModel.initBuffer(t, qp, (UChar*)INIT_PRED_MODE); s.m_cCUMvdSCModel.initBuffer(t, qp, (UChar*)INIT_MVD); s.m_cMVPIdxSCModel.initBuffer(t, qp, (UChar*)INIT_MVP_IDX);
  s.m_cCUGTFlagExtSCModel.initBuffer(t, qp, (UChar*)INIT_GT_FLAG_EXT); s.m_cCUGTSCModel.initBuffer(t, qp, (UChar*)INIT_GT);
  s.m_cCUIntraPredSCModel.initBuffer(t, qp, (UChar*)INIT_INTRA_PRED_MODE); s.m_cCUChromaPredSCModel.initBuffer(t, qp, (UChar*)INIT_CHROMA_PRED_MODE);
  ContextModel* p[11] = { s.m_cCUSkipFlagSCModel.get(0), s.m_cCUMergeFlagExtSCModel.get(0), s.m_cCUMergeIdxExtSCModel.get(0), s.m_cCUPartSizeSCModel.get(0), s.m_cCUPredModeSCModel.get(0),
                         s.m_cCUMvdSCModel.get(0), s.m_cMVPIdxSCModel.get(0), s.m_cCUGTFlagExtSCModel.get(0), s.m_cCUGTSCModel.get(0), s.m_cCUIntraPredSCModel.get(0),
                         s.m_cCUChromaPredSCModel.get(0) };
  const int n[11] = { 3, 1, 1, 4, 1, 2, 1, 1, 2, 1, 2 };
  int k = 0; for (int i = 0; i < 11; i++) for (int j = 0; j < n[i]; j++) states19[k++] = p[i][j].m_ucState;
}
// TEncSbac::estBit (TEncSbac.cpp:2175-2370); est: an estBitsSbacStruct image, updated in place like the reference's
void ref_cabac_est_bits(const uint8_t* states, int width, int ttype, int32_t* est)
{
  SbacCtx* c = sbac_get();
  ContextModel::buildNextStateTable();
  sbac_load(c->sbac, states);
  c->sbac.estBit((estBitsSbacStruct*)est, width, width, ttype ? TEXT_CHROMA : TEXT_LUMA);   // as TEncEntropy::estimateBit hands it on (TEncEntropy.cpp:669-674)
}
// TEncSbac::codeCoeffNxN (:1829-2092) with the counting coder: fractional bits, contexts updated
uint64_t ref_cabac_coeff_bits(uint8_t* states, const int32_t* coef, int N, int ttype, int isIntra, int lumaDir, int chromaDir,
                              int signHide, int useTS, int tsFlag)
{
  static Char predMode[1]; static UChar trIdxA[1], lumaDirA[1], chromaDirA[1], depthA[1], tsA[3][1]; static Bool bypassA[1];
  SbacCtx* c = sbac_get();
  ContextModel::buildNextStateTable();
  sbac_load(c->sbac, states);
  c->pps.setSignHideFlag(signHide); c->pps.setUseTransformSkip(useTS != 0);
  TComDataCU& cu = g->cu;
  Char* sPred = cu.m_pePredMode; UChar* sTr = cu.m_puhTrIdx; UChar* sL = cu.m_puhLumaIntraDir; UChar* sC = cu.m_puhChromaIntraDir;
  UChar* sD = cu.m_puhDepth; TComSlice* sS = cu.m_pcSlice; Bool* sB = cu.m_CUTransquantBypass;
  UChar* sT[3] = { cu.m_puhTransformSkip[0], cu.m_puhTransformSkip[1], cu.m_puhTransformSkip[2] };
  predMode[0] = isIntra ? MODE_INTRA : MODE_INTER; trIdxA[0] = 0; lumaDirA[0] = (UChar)lumaDir; chromaDirA[0] = (UChar)chromaDir; depthA[0] = 0; bypassA[0] = false;
  for (int k = 0; k < 3; k++) { tsA[k][0] = (UChar)tsFlag; cu.m_puhTransformSkip[k] = tsA[k]; }
  cu.m_pePredMode = predMode; cu.m_puhTrIdx = trIdxA; cu.m_puhLumaIntraDir = lumaDirA; cu.m_puhChromaIntraDir = chromaDirA; cu.m_puhDepth = depthA;
  cu.m_pcSlice = &c->slice; cu.m_CUTransquantBypass = bypassA;
  std::vector<TCoeff> in(coef, coef + N * N);
  c->sbac.resetBits();                                       // keeps the fraction below one bit (TEncBinCABAC::resetBits)
  const uint64_t f0 = c->bin.m_fracBits;
  c->sbac.codeCoeffNxN(&cu, &in[0], 0, N, N, 0, (TextType)ttype);
  const uint64_t frac = ((uint64_t)c->bits.getNumberOfWrittenBits() << 15) + c->bin.m_fracBits - f0;
  sbac_store(c->sbac, states);
  cu.m_pePredMode = sPred; cu.m_puhTrIdx = sTr; cu.m_puhLumaIntraDir = sL; cu.m_puhChromaIntraDir = sC; cu.m_puhDepth = sD; cu.m_pcSlice = sS;
  cu.m_CUTransquantBypass = sB; for (int k = 0; k < 3; k++) cu.m_puhTransformSkip[k] = sT[k];
  return frac;
}

// ---------------------------------------------------------------------------------------------
// row a8b, leaf step of TEncSearch::xEstimateResidualQT (TEncSearch.cpp:6896-7200) for ONE component TU, assembled from
// the reference's own members: TEncSbac::estBit, TComTrQuant::transformNxN (xT + xRateDistOptQuant), TEncSbac::codeQtCbf /
// codeCoeffNxN through the counting coder, TComTrQuant::invtransformNxN, TComRdCost::getDistPart, calcRdCost.  The
// sequencing between them (load snapshot / resetBits / cbf-zero decision) is RESTATED from :6959-7032.
// out[8] and cost as hop_o_tu_rd.
// ---------------------------------------------------------------------------------------------
int ref_tu_rd(const int16_t* resi, int N, int ttype, int qpScaled, int bitDepth, int trDepth, int signHide, int useTS,
              double lambdaRdoq, double lambdaRd, double distWeight, const uint8_t* states, unsigned fracLeft,
              int32_t* levels, uint32_t* out, double* cost)
{
  static Char predMode[1]; static UChar trIdxA[1], lumaDirA[1], chromaDirA[1], depthA[1], tsA[3][1], cbfA[3][1]; static Bool bypassA[1];
  SbacCtx* c = sbac_get();
  ContextModel::buildNextStateTable();
  g_bitDepthY = bitDepth; g_bitDepthC = bitDepth;
  c->pps.setSignHideFlag(signHide); c->pps.setUseTransformSkip(useTS != 0);
  TComTrQuant& t = g->trq;
  t.init(32, true, true, true, false, false);
  t.setUseScalingList(false); t.setFlatScalingList();
  t.m_cQP.setQpParam(qpScaled);
  t.m_dLambda = lambdaRdoq;
  g->rd.setLambda(lambdaRd);
  if (ttype == 2) g->rd.setCbDistortionWeight(distWeight);
  if (ttype == 3) g->rd.setCrDistortionWeight(distWeight);
  TComDataCU& cu = g->cu;
  Char* sPred = cu.m_pePredMode; UChar* sTr = cu.m_puhTrIdx; UChar* sL = cu.m_puhLumaIntraDir; UChar* sC = cu.m_puhChromaIntraDir;
  UChar* sD = cu.m_puhDepth; TComSlice* sS = cu.m_pcSlice; Bool* sB = cu.m_CUTransquantBypass;
  UChar* sT[3] = { cu.m_puhTransformSkip[0], cu.m_puhTransformSkip[1], cu.m_puhTransformSkip[2] };
  UChar* sCbf[3] = { cu.m_puhCbf[0], cu.m_puhCbf[1], cu.m_puhCbf[2] };
  predMode[0] = MODE_INTER; trIdxA[0] = (UChar)trDepth; lumaDirA[0] = 0; chromaDirA[0] = 0; depthA[0] = 0; bypassA[0] = false;
  for (int k = 0; k < 3; k++) { tsA[k][0] = 0; cu.m_puhTransformSkip[k] = tsA[k]; cbfA[k][0] = 0; cu.m_puhCbf[k] = cbfA[k]; }
  cu.m_pePredMode = predMode; cu.m_puhTrIdx = trIdxA; cu.m_puhLumaIntraDir = lumaDirA; cu.m_puhChromaIntraDir = chromaDirA; cu.m_puhDepth = depthA;
  cu.m_pcSlice = &c->slice; cu.m_CUTransquantBypass = bypassA;
  TextType tt = (TextType)ttype;
  // estBit from the snapshot (:6901-6904 / :6922-6925)
  sbac_load(c->sbac, states);
  c->sbac.estBit(t.m_pcEstBitsSbac, N, N, ttype ? TEXT_CHROMA : TEXT_LUMA);
  // transformNxN (:6912)
  std::vector<Pel> r(resi, resi + N * N); std::vector<TCoeff> coef(N * N, 0); std::vector<Int> arl(N * N, 0); Int* parl = &arl[0];
  UInt absSum = 0;
  t.transformNxN(&cu, &r[0], N, &coef[0], parl, N, N, absSum, tt, 0, false);
  const int ci = ttype == 0 ? 0 : ttype == 2 ? 1 : 2;
  cbfA[ci][0] = (UChar)((absSum ? 1 : 0) << trDepth);           // setCbfSubParts( uiAbsSum ? uiSetCbf : 0, ... ), uiSetCbf = 1 << uiTrMode (:6877,6917)
  // bits of cbf + levels from the snapshot (:6957-6962); the coder keeps the fraction below one bit across resetBits
  sbac_load(c->sbac, states);
  c->sbac.resetBits(); c->bin.m_fracBits = fracLeft;
  c->sbac.codeQtCbf(&cu, 0, tt, trDepth);
  c->sbac.codeCoeffNxN(&cu, &coef[0], 0, N, N, 0, tt);
  const UInt singleBits = c->bin.getNumWrittenBits();
  std::vector<Pel> zero(N * N, 0), rec(N * N, 0);
  UInt dist = g->rd.getDistPart(bitDepth, &zero[0], N, &r[0], N, N, N, tt);      // zero-residual distortion (:6984)
  const UInt zeroDist = dist;
  UInt nzDist = 0, nullBits = 0; double chosen = 0;
  if (absSum) {
    t.invtransformNxN(false, ttype ? TEXT_CHROMA : TEXT_LUMA, REG_DCT, &rec[0], N, &coef[0], N, N, 3 + g_eTTable[ttype]);
    nzDist = g->rd.getDistPart(bitDepth, &rec[0], N, &r[0], N, N, N, tt);
    const Double singleCost = g->rd.calcRdCost(singleBits, nzDist);
    sbac_load(c->sbac, states);
    c->sbac.resetBits(); c->bin.m_fracBits = fracLeft;
    c->sbac.codeQtCbfZero(&cu, tt, trDepth);
    nullBits = c->bin.getNumWrittenBits();
    const Double nullCost = g->rd.calcRdCost(nullBits, dist);
    if (nullCost < singleCost) { absSum = 0; std::fill(coef.begin(), coef.end(), 0); chosen = nullCost; }
    else { dist = nzDist; chosen = singleCost; }
  } else {
    chosen = g->rd.calcRdCost(singleBits, dist);
  }
  for (int i = 0; i < N * N; i++) levels[i] = coef[i];
  out[0] = absSum; out[1] = absSum != 0; out[2] = dist; out[3] = zeroDist; out[4] = nzDist; out[5] = singleBits; out[6] = nullBits; out[7] = 0;
  *cost = chosen;
  cu.m_pePredMode = sPred; cu.m_puhTrIdx = sTr; cu.m_puhLumaIntraDir = sL; cu.m_puhChromaIntraDir = sC; cu.m_puhDepth = sD; cu.m_pcSlice = sS;
  cu.m_CUTransquantBypass = sB; for (int k = 0; k < 3; k++) { cu.m_puhTransformSkip[k] = sT[k]; cu.m_puhCbf[k] = sCbf[k]; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// row a8, leaf step of TEncSearch::xIntraCodingLumaBlk / ChromaBlk after the prediction (TEncSearch.cpp:1082-1160): the
// reference's own estBit, transformNxN (intra: DST for 4x4 luma, scan by direction), invtransformNxN, ClipY/ClipC, getDistPart,
// and the bits of cbf flag + levels from the snapshot.  lumaDir / chromaDir select transform and scan as in the reference.
// ---------------------------------------------------------------------------------------------
int ref_tu_intra(const int16_t* org, const int16_t* pred, int N, int ttype, int lumaDir, int chromaDir, int qpScaled, int bitDepth, int trDepth,
                 int signHide, int useTS, double lambdaRdoq, double lambdaRd, double distWeight, const uint8_t* states, unsigned fracLeft,
                 int32_t* levels, int16_t* recon, uint32_t* out, double* cost)
{
  static Char predMode[1]; static UChar trIdxA[1], lumaDirA[1], chromaDirA[1], depthA[1], tsA[3][1], cbfA[3][1]; static Bool bypassA[1];
  SbacCtx* c = sbac_get();
  ContextModel::buildNextStateTable();
  g_bitDepthY = bitDepth; g_bitDepthC = bitDepth;
  c->pps.setSignHideFlag(signHide); c->pps.setUseTransformSkip(useTS != 0);
  TComTrQuant& t = g->trq;
  t.init(32, true, true, true, false, false);
  t.setUseScalingList(false); t.setFlatScalingList();
  t.m_cQP.setQpParam(qpScaled);
  t.m_dLambda = lambdaRdoq;
  g->rd.setLambda(lambdaRd);
  if (ttype == 2) g->rd.setCbDistortionWeight(distWeight);
  if (ttype == 3) g->rd.setCrDistortionWeight(distWeight);
  TComDataCU& cu = g->cu;
  Char* sPred = cu.m_pePredMode; UChar* sTr = cu.m_puhTrIdx; UChar* sL = cu.m_puhLumaIntraDir; UChar* sC = cu.m_puhChromaIntraDir;
  UChar* sD = cu.m_puhDepth; TComSlice* sS = cu.m_pcSlice; Bool* sB = cu.m_CUTransquantBypass;
  UChar* sT[3] = { cu.m_puhTransformSkip[0], cu.m_puhTransformSkip[1], cu.m_puhTransformSkip[2] };
  UChar* sCbf[3] = { cu.m_puhCbf[0], cu.m_puhCbf[1], cu.m_puhCbf[2] };
  predMode[0] = MODE_INTRA; trIdxA[0] = (UChar)trDepth; lumaDirA[0] = (UChar)lumaDir; chromaDirA[0] = (UChar)chromaDir; depthA[0] = 0; bypassA[0] = false;
  for (int k = 0; k < 3; k++) { tsA[k][0] = 0; cu.m_puhTransformSkip[k] = tsA[k]; cbfA[k][0] = 0; cu.m_puhCbf[k] = cbfA[k]; }
  cu.m_pePredMode = predMode; cu.m_puhTrIdx = trIdxA; cu.m_puhLumaIntraDir = lumaDirA; cu.m_puhChromaIntraDir = chromaDirA; cu.m_puhDepth = depthA;
  cu.m_pcSlice = &c->slice; cu.m_CUTransquantBypass = bypassA;
  TextType tt = (TextType)ttype;
  std::vector<Pel> r(N * N);
  for (int i = 0; i < N * N; i++) r[i] = (Pel)(org[i] - pred[i]);
  sbac_load(c->sbac, states);
  c->sbac.estBit(t.m_pcEstBitsSbac, N, N, ttype ? TEXT_CHROMA : TEXT_LUMA);
  std::vector<TCoeff> coef(N * N, 0); std::vector<Int> arl(N * N, 0); Int* parl = &arl[0];
  UInt absSum = 0;
  t.transformNxN(&cu, &r[0], N, &coef[0], parl, N, N, absSum, tt, 0, false);
  const int ci = ttype == 0 ? 0 : ttype == 2 ? 1 : 2;
  cbfA[ci][0] = (UChar)((absSum ? 1 : 0) << trDepth);
  if (absSum) t.invtransformNxN(false, ttype ? TEXT_CHROMA : TEXT_LUMA, ttype ? REG_DCT : (UInt)lumaDir, &r[0], N, &coef[0], N, N, 0 + g_eTTable[ttype]);
  else std::fill(r.begin(), r.end(), 0);
  std::vector<Pel> rec(N * N), o(org, org + N * N);
  for (int i = 0; i < N * N; i++) rec[i] = ttype ? ClipC(pred[i] + r[i]) : ClipY(pred[i] + r[i]);
  const UInt dist = g->rd.getDistPart(bitDepth, &rec[0], N, &o[0], N, N, N, tt);
  sbac_load(c->sbac, states);
  c->sbac.resetBits(); c->bin.m_fracBits = fracLeft;
  c->sbac.codeQtCbf(&cu, 0, tt, trDepth);
  c->sbac.codeCoeffNxN(&cu, &coef[0], 0, N, N, 0, tt);
  const UInt bits = c->bin.getNumWrittenBits();
  for (int i = 0; i < N * N; i++) { levels[i] = coef[i]; recon[i] = rec[i]; }
  out[0] = absSum; out[1] = absSum != 0; out[2] = dist; out[3] = 0; out[4] = 0; out[5] = bits; out[6] = 0; out[7] = 0;
  *cost = g->rd.calcRdCost(bits, dist);
  cu.m_pePredMode = sPred; cu.m_puhTrIdx = sTr; cu.m_puhLumaIntraDir = sL; cu.m_puhChromaIntraDir = sC; cu.m_puhDepth = sD; cu.m_pcSlice = sS;
  cu.m_CUTransquantBypass = sB; for (int k = 0; k < 3; k++) { cu.m_puhTransformSkip[k] = sT[k]; cu.m_puhCbf[k] = sCbf[k]; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// intra rough search pieces (row a7): the reference's own fillReferenceSamples, getPredictorPtr, predIntraLumaAng,
// calcHAD.  The smoothing of TComPattern::initAdiPattern (:237-312) is inline code that needs a TComDataCU/TComPic
// graph; it is RESTATED here (marked) so that the filtered buffer exists for getPredictorPtr.
// rec: reconstruction picture sample (0,0) + stride; flags: bNeighborFlags in the reference's order (4N/4+... units).
// ---------------------------------------------------------------------------------------------
void ref_intra_rough(const int16_t* rec, int recStride, const int16_t* org, int orgStride, int x, int y, int N,
                     const uint8_t* flags, int bitDepth, int strong, uint32_t* satd, int predMode, int16_t* predOut, int* lineOut)
{
  TEncSearch& s = g->search;
  const int U = N / 4, units = 4 * U + 1;
  Bool nb[4 * 16 + 1]; int navail = 0;
  for (int u = 0; u < units; u++) { nb[u] = flags[u] != 0; navail += nb[u]; }
  Int* adi = s.m_piYuvExt;
  const UInt w2 = 2 * N + 1;
  TComPattern pat;
  pat.fillReferenceSamples(bitDepth, (Pel*)(rec + (ptrdiff_t)y * recStride + x), adi, nb, navail, 4, U, units, N, N, w2, w2, recStride, false);
  // ---- RESTATED from TComPattern.cpp:237-312 ----
  {
    const UInt uiCuHeight2 = 2 * N, uiCuWidth2 = 2 * N, uiWH = w2 * w2;
    Int iBufSize = uiCuHeight2 + uiCuWidth2 + 1;
    Int* piFilteredBuf1 = adi + uiWH; Int* piFilteredBuf2 = piFilteredBuf1 + uiWH;
    Int* piFilterBuf = piFilteredBuf2 + uiWH; Int* piFilterBufN = piFilterBuf + iBufSize;
    Int l = 0, i;
    for (i = 0; i < (Int)uiCuHeight2; i++) piFilterBuf[l++] = adi[w2 * (uiCuHeight2 - i)];
    piFilterBuf[l++] = adi[0];
    for (i = 0; i < (Int)uiCuWidth2; i++) piFilterBuf[l++] = adi[1 + i];
    bool done = false;
    if (strong) {
      Int bottomLeft = piFilterBuf[0], topLeft = piFilterBuf[uiCuHeight2], topRight = piFilterBuf[iBufSize - 1];
      Int threshold = 1 << (bitDepth - 5);
      Bool bilinearLeft = abs(bottomLeft + topLeft - 2 * piFilterBuf[N]) < threshold;
      Bool bilinearAbove = abs(topLeft + topRight - 2 * piFilterBuf[uiCuHeight2 + N]) < threshold;
      if (N >= 32 && bilinearLeft && bilinearAbove) {
        Int shift = g_aucConvertToBit[N] + 3;
        piFilterBufN[0] = piFilterBuf[0]; piFilterBufN[uiCuHeight2] = piFilterBuf[uiCuHeight2]; piFilterBufN[iBufSize - 1] = piFilterBuf[iBufSize - 1];
        for (i = 1; i < (Int)uiCuHeight2; i++) piFilterBufN[i] = ((uiCuHeight2 - i) * bottomLeft + i * topLeft + N) >> shift;
        for (i = 1; i < (Int)uiCuWidth2; i++) piFilterBufN[uiCuHeight2 + i] = ((uiCuWidth2 - i) * topLeft + i * topRight + N) >> shift;
        done = true;
      }
    }
    if (!done) {
      piFilterBufN[0] = piFilterBuf[0]; piFilterBufN[iBufSize - 1] = piFilterBuf[iBufSize - 1];
      for (i = 1; i < iBufSize - 1; i++) piFilterBufN[i] = (piFilterBuf[i - 1] + 2 * piFilterBuf[i] + piFilterBuf[i + 1] + 2) >> 2;
    }
    l = 0;
    for (i = 0; i < (Int)uiCuHeight2; i++) piFilteredBuf1[w2 * (uiCuHeight2 - i)] = piFilterBufN[l++];
    piFilteredBuf1[0] = piFilterBufN[l++];
    for (i = 0; i < (Int)uiCuWidth2; i++) piFilteredBuf1[1 + i] = piFilterBufN[l++];
    if (lineOut) for (i = 0; i < iBufSize; i++) { lineOut[i] = piFilterBuf[i]; lineOut[iBufSize + i] = piFilterBufN[i]; }
  }
  // ---- the reference's own prediction + distortion, as estIntraPredQT does (TEncSearch.cpp:2455-2458) ----
  std::vector<Pel> pred(N * N);
  for (UInt m = 0; m < 35; m++) {
    s.predIntraLumaAng(&pat, m, &pred[0], N, N, N, true, true);
    satd[m] = g->rd.calcHAD(bitDepth, (Pel*)(org + (ptrdiff_t)y * orgStride + x), orgStride, &pred[0], N, N, N);
    if ((int)m == predMode && predOut) memcpy(predOut, &pred[0], N * N * sizeof(Pel));
  }
}

} // extern "C"
