// oracle/enc_shim.cpp -- TEST INFRASTRUCTURE.  Encoder-in-the-loop check of the drop-in boundary.
//
// Our own definitions of the three search members of the reference's TEncSearch that the C ABI replaces:
//   TEncSearch::xPatternSearch         TLibEncoder/TEncSearch.cpp:6262-6371   -> hop_o_ss_search    (= hop_ss_search)
//   TEncSearch::xPatternSearchFracDIF  TLibEncoder/TEncSearch.cpp:6564-6610   -> hop_o_frac_search  (= hop_frac_search)
//   TEncSearch::xPatternSearchGT       TLibEncoder/TEncSearch.cpp:4686-6200   -> hop_o_gt_search    (= hop_gt_search)
// oracle/Makefile.ref links them into _ref/TAppEncoderShim in place of the reference's definitions (the reference's object
// keeps everything else; its three symbols are weakened with objcopy, nothing of the reference is edited or copied).  The rest
// of the encoder -- xCompressCU, AMVP, CABAC, RQT, SS-ref upkeep -- is the reference's.  If the shim encoder writes the same
// bitstream and reconstruction as the unmodified one, then for every call the real encoder makes (real predictors, sentinel
// regions, picture borders, every PU shape incl. AMP) the restatement returned what the reference's members return, and the
// argument mapping of INTEGRATION.md carries everything the callee needs (tests/test_encoder_shim.py).
// With HOP_SHIM_TRACE=<file> every call is appended to a binary trace (inputs incl. the reference window + outputs) that
// oracle/make_golden7.py samples into tests/golden/encoder_calls.npz for the GPU replay test.
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
// the reference's members are reached through its own headers; standard headers come first so they are unaffected
#define private public
#define protected public
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComPattern.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibEncoder/TEncSearch.h"
#include "TLibEncoder/TEncSbac.h"
#undef private
#undef protected
extern "C" {
#include "hop_oracle.h"
}

namespace {
FILE* trace_file() {
  static FILE* f = NULL; static bool tried = false;
  if (!tried) { tried = true; const char* p = getenv("HOP_SHIM_TRACE"); if (p && *p) f = fopen(p, "wb"); }
  return f;
}
FILE* rdoq_trace_file() {          // HOP_SHIM_TRACE_RDOQ=<file>: every xRateDistOptQuant call (header, lambda, tables, coefficients in, levels out)
  static FILE* f = NULL; static bool tried = false;
  if (!tried) { tried = true; const char* p = getenv("HOP_SHIM_TRACE_RDOQ"); if (p && *p) f = fopen(p, "wb"); }
  return f;
}
unsigned long g_calls[3] = { 0, 0, 0 };
struct Report { ~Report() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: ss %lu frac %lu gt %lu\n", g_calls[0], g_calls[1], g_calls[2]); } } g_report;

// one trace record: header of int32 + the original block + the reference rows the call can read
//   kind(0 ss,1 frac,2 gt) w h nIn nOut winX0 winY0 winW winH, in[nIn], out[nOut], org[w*h], window[winW*winH]
void trace(int kind, int w, int h, const int* in, int nIn, const int64_t* out, int nOut, const Pel* org, int orgStride,
           const Pel* refPU, int refStride, int x0, int y0, int x1, int y1) {
  FILE* f = trace_file(); if (!f) return;
  const int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
  int32_t hd[9] = { kind, w, h, nIn, nOut, x0, y0, ww, wh };
  fwrite(hd, 4, 9, f); fwrite(in, 4, nIn, f); fwrite(out, 8, nOut, f);
  for (int y = 0; y < h; y++) fwrite(org + y * orgStride, 2, w, f);
  for (int y = y0; y <= y1; y++) fwrite(refPU + (ptrdiff_t)y * refStride + x0, 2, ww, f);
}
}

Void TEncSearch::xPatternSearch(TComPattern* pcPatternKey, Pel* piRefY, Int iRefStride, TComMv* pcMvSrchRngLT, TComMv* pcMvSrchRngRB,
                                TComMv& rcMv, UInt& ruiSAD, Int riOffsetX, Int riOffsetY, TComMv* ssBestCand, Bool isSSE)
{
  if (!isSSE) { fprintf(stderr, "hop shim: xPatternSearch outside an SS slice is not on the replaced path\n"); abort(); }
  g_calls[0]++;
  const int w = pcPatternKey->getROIYWidth(), h = pcPatternKey->getROIYHeight();
  int bx = 0, by = 0; uint32_t sad = 0;
  const int L = pcMvSrchRngLT->getHor(), R = pcMvSrchRngRB->getHor(), T = pcMvSrchRngLT->getVer(), B = pcMvSrchRngRB->getVer();
  hop_o_ss_search(pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride, w, h, L, R, T, B, riOffsetX, riOffsetY,
                  m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(), m_pcRdCost->m_uiCost,
                  m_pcEncCfg->getUseFastEnc() ? 1 : 0, g_bitDepthY, &bx, &by, &sad);
  if (trace_file()) {
    const int in[10] = { L, R, T, B, riOffsetX, riOffsetY, m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(), (int)m_pcRdCost->m_uiCost,
                         m_pcEncCfg->getUseFastEnc() ? 1 : 0 };
    const int64_t out[3] = { bx, by, (int64_t)sad };
    if (L <= R && T <= B) trace(0, w, h, in, 10, out, 3, pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride, L, T, R + w - 1, B + h - 1);
  }
  if (sad == 0xFFFFFFFFu) { ruiSAD = MAX_UINT; return; }         // no valid candidate: rcMv and ssBestCand stay as they were (:6356-6360)
  rcMv.set(bx, by);
  ssBestCand[0].set(bx, by);
  ruiSAD = sad;
}

Void TEncSearch::xPatternSearchFracDIF(TComDataCU* pcCU, TComPattern* pcPatternKey, Pel* piRefY, Int iRefStride, TComMv* pcMvInt,
                                       TComMv& rcMvHalf, TComMv& rcMvQter, UInt& ruiCost, Bool biPred)
{
  if (biPred) { fprintf(stderr, "hop shim: bi-prediction is not on the replaced path\n"); abort(); }
  g_calls[1]++;
  const int w = pcPatternKey->getROIYWidth(), h = pcPatternKey->getROIYHeight();
  int half[2], qter[2];
  const uint32_t cost = hop_o_frac_search(pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride, w, h,
                                          pcMvInt->getHor(), pcMvInt->getVer(), m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(),
                                          m_pcRdCost->m_uiCost, m_pcEncCfg->getUseHADME() ? 1 : 0, g_bitDepthY, half, qter);
  if (trace_file()) {
    const int in[6] = { pcMvInt->getHor(), pcMvInt->getVer(), m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(), (int)m_pcRdCost->m_uiCost,
                        m_pcEncCfg->getUseHADME() ? 1 : 0 };
    const int64_t out[5] = { half[0], half[1], qter[0], qter[1], (int64_t)cost };
    trace(1, w, h, in, 6, out, 5, pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride,
          pcMvInt->getHor() - 4, pcMvInt->getVer() - 4, pcMvInt->getHor() + w + 4, pcMvInt->getVer() + h + 4);   // the members' own reach (:7818-8011)
  }
  rcMvHalf.set(half[0], half[1]);
  rcMvQter.set(qter[0], qter[1]);
  ruiCost = cost;
  m_pcRdCost->setCostScale(0);                                      // what the reference's member leaves behind (:6596)
}

Void TEncSearch::xPatternSearchGT(TComDataCU* pcCU, TComPattern* pcPatternKey, Pel* piRefY, Int iRefStride, TComMv* pcMvInt, TComMv* rcMvHalf,
                                  TComMv* rcMvQter, TComMv* rcGT0, TComMv* rcGT1, TComMv* rcGT2, TComMv* rcGT3, Bool& gtFlag, UInt& ruiCost,
                                  Bool biPred, TComMv* bestSSCand)
{
  if (biPred) { fprintf(stderr, "hop shim: bi-prediction is not on the replaced path\n"); abort(); }
  g_calls[2]++;
  const int w = pcPatternKey->getROIYWidth(), h = pcPatternKey->getROIYHeight();
  int mv[2] = { pcMvInt->getHor(), pcMvInt->getVer() }, half[2] = { rcMvHalf->getHor(), rcMvHalf->getVer() }, qter[2] = { rcMvQter->getHor(), rcMvQter->getVer() };
  const int ssBest[2] = { bestSSCand[0].getHor(), bestSSCand[0].getVer() };
  AMVPInfo* ai = pcCU->getCUMvField(REF_PIC_LIST_0)->getAMVPInfo();
  int amvp[2 * AMVP_MAX_NUM_CANDS_MEM];
  for (int i = 0; i < ai->iN; i++) { amvp[2 * i] = ai->m_acMvCand[i].getHor(); amvp[2 * i + 1] = ai->m_acMvCand[i].getVer(); }
  uint32_t cost = ruiCost; int gt[8];
  const int in0[12] = { mv[0], mv[1], half[0], half[1], qter[0], qter[1], ssBest[0], ssBest[1], m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(),
                        (int)m_pcRdCost->m_uiCost, (int)ruiCost };
  const int flag = hop_o_gt_search(pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride, w, h, mv, half, qter, ssBest, ai->iN, amvp,
                                   m_pcRdCost->m_mvPredictor.getHor(), m_pcRdCost->m_mvPredictor.getVer(), m_pcRdCost->m_uiCost,
                                   m_pcEncCfg->getUseHADME() ? 1 : 0, g_bitDepthY, &cost, gt);
  if (trace_file()) {
    int in[12 + 4 + 2 * AMVP_MAX_NUM_CANDS_MEM]; memcpy(in, in0, sizeof(in0));
    in[12] = m_pcEncCfg->getUseHADME() ? 1 : 0; in[13] = ai->iN;
    // where the PU lies in the picture (the replay puts the recorded windows back there): from the SS reference's own origin
    TComPicYuv* ssref = pcCU->getSlice()->getRefPic(REF_PIC_LIST_0, 0)->getPicYuvRec();
    const ptrdiff_t delta = piRefY - ssref->getLumaAddr();
    in[14] = (int)(delta % iRefStride); in[15] = (int)(delta / iRefStride);
    for (int i = 0; i < 2 * ai->iN; i++) in[16 + i] = amvp[i];
    const int64_t out[16] = { flag, gt[0], gt[1], gt[2], gt[3], gt[4], gt[5], gt[6], gt[7], (int64_t)cost, mv[0], mv[1], half[0], half[1], qter[0], qter[1] };
    // every start vector's 2W x 2H patch (+ the 8-tap margins): the starts are the SS best and the AMVP candidates at integer precision
    int x0 = in0[6], x1 = in0[6], y0 = in0[7], y1 = in0[7];
    for (int i = 0; i < ai->iN; i++) {
      const int ax = amvp[2 * i] >> 2, ay = amvp[2 * i + 1] >> 2;
      if (ax < x0) x0 = ax; if (ax > x1) x1 = ax; if (ay < y0) y0 = ay; if (ay > y1) y1 = ay;
    }
    if (in0[0] < x0) x0 = in0[0]; if (in0[0] > x1) x1 = in0[0]; if (in0[1] < y0) y0 = in0[1]; if (in0[1] > y1) y1 = in0[1];
    x0 -= w / 2 + 6; y0 -= h / 2 + 6; x1 += w + w / 2 + 6; y1 += h + h / 2 + 6;
    const int mX = ssref->getLumaMargin(), mY = ssref->getLumaMargin();          // keep the dump inside the padded plane
    if (x0 < -mX - in[14]) x0 = -mX - in[14];
    if (y0 < -mY - in[15]) y0 = -mY - in[15];
    if (x1 > ssref->getWidth() + mX - 1 - in[14]) x1 = ssref->getWidth() + mX - 1 - in[14];
    if (y1 > ssref->getHeight() + mY - 1 - in[15]) y1 = ssref->getHeight() + mY - 1 - in[15];
    trace(2, w, h, in, 16 + 2 * ai->iN, out, 16, pcPatternKey->getROIY(), pcPatternKey->getPatternLStride(), piRefY, iRefStride, x0, y0, x1, y1);
  }
  gtFlag = flag != 0;
  rcGT0->set(gt[0], gt[1]); rcGT1->set(gt[2], gt[3]); rcGT2->set(gt[4], gt[5]); rcGT3->set(gt[6], gt[7]);
  pcMvInt->set(mv[0], mv[1]); rcMvHalf->set(half[0], half[1]); rcMvQter->set(qter[0], qter[1]);
  ruiCost = cost;
}

// ---- rows a6 / a9 / a10 / a11: the predictor and transform/quantisation members, same method ----
//   TComPrediction::xPredInterLumaBlk / xPredInterChromaBlk  TLibCommon/TComPrediction.cpp:639-720, 1235-1347  -> hop_o_pred_inter (= hop_pred_inter)
//   TComTrQuant::xT / xIT                                    TLibCommon/TComTrQuant.cpp:1341-1400              -> hop_o_fwd_transform / hop_o_inv_transform
//   TComTrQuant::xDeQuant                                    TLibCommon/TComTrQuant.cpp:1124-1183              -> hop_o_dequant_flat
//   TComTrQuant::xRateDistOptQuant                           TLibCommon/TComTrQuant.cpp:1489-1999              -> hop_o_rdoq (= hop_rdoq)
namespace { unsigned long g_calls2[6] = { 0, 0, 0, 0, 0, 0 };
struct Report2 { ~Report2() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: predY %lu predC %lu xT %lu xIT %lu dequant %lu rdoq %lu\n",
                                                                  g_calls2[0], g_calls2[1], g_calls2[2], g_calls2[3], g_calls2[4], g_calls2[5]); } } g_report2; }

Void TComPrediction::xPredInterLumaBlk(TComDataCU* cu, TComPicYuv* refPic, UInt partAddr, TComMv* mv, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                       Bool bUseGT, TComMv* mGT0, TComMv* mGT1, TComMv* mGT2, TComMv* mGT3)
{
  if (bi) { fprintf(stderr, "hop shim: bi-prediction is not on the replaced path\n"); abort(); }
  g_calls2[0]++;
  const int gt[8] = { mGT0->getHor(), mGT0->getVer(), mGT1->getHor(), mGT1->getVer(), mGT2->getHor(), mGT2->getVer(), mGT3->getHor(), mGT3->getVer() };
  const UInt z = cu->getZorderIdxInCU() + partAddr;
  std::vector<int16_t> y(width * height), cb(width * height / 4), cr(width * height / 4);
  hop_o_pred_inter(refPic->getLumaAddr(cu->getAddr(), z), refPic->getStride(), refPic->getCbAddr(cu->getAddr(), z), refPic->getCrAddr(cu->getAddr(), z), refPic->getCStride(),
                   0, 0, width, height, mv->getHor(), mv->getVer(), bUseGT ? 1 : 0, gt, g_bitDepthY, g_bitDepthC, &y[0], &cb[0], &cr[0]);
  Pel* dst = dstPic->getLumaAddr(partAddr); const int ds = dstPic->getStride();
  for (int r = 0; r < height; r++) memcpy(dst + r * ds, &y[r * width], width * sizeof(Pel));
}

Void TComPrediction::xPredInterChromaBlk(TComDataCU* cu, TComPicYuv* refPic, UInt partAddr, TComMv* mv, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                         Bool bUseGT, TComMv* mGT0, TComMv* mGT1, TComMv* mGT2, TComMv* mGT3)
{
  if (bi) { fprintf(stderr, "hop shim: bi-prediction is not on the replaced path\n"); abort(); }
  g_calls2[1]++;
  const int gt[8] = { mGT0->getHor(), mGT0->getVer(), mGT1->getHor(), mGT1->getVer(), mGT2->getHor(), mGT2->getVer(), mGT3->getHor(), mGT3->getVer() };
  const UInt z = cu->getZorderIdxInCU() + partAddr;
  std::vector<int16_t> y(width * height), cb(width * height / 4), cr(width * height / 4);
  hop_o_pred_inter(refPic->getLumaAddr(cu->getAddr(), z), refPic->getStride(), refPic->getCbAddr(cu->getAddr(), z), refPic->getCrAddr(cu->getAddr(), z), refPic->getCStride(),
                   0, 0, width, height, mv->getHor(), mv->getVer(), bUseGT ? 1 : 0, gt, g_bitDepthY, g_bitDepthC, &y[0], &cb[0], &cr[0]);
  Pel* dcb = dstPic->getCbAddr(partAddr); Pel* dcr = dstPic->getCrAddr(partAddr); const int ds = dstPic->getCStride();
  const int cw = width >> 1, ch = height >> 1;
  for (int r = 0; r < ch; r++) { memcpy(dcb + r * ds, &cb[r * cw], cw * sizeof(Pel)); memcpy(dcr + r * ds, &cr[r * cw], cw * sizeof(Pel)); }
}

Void TComTrQuant::xT(Int bitDepth, UInt uiMode, Pel* piBlkResi, UInt uiStride, Int* psCoeff, Int iWidth, Int iHeight)
{
  g_calls2[2]++;
  int16_t block[32 * 32], coeff[32 * 32];
  for (int j = 0; j < iHeight; j++) memcpy(block + j * iWidth, piBlkResi + j * uiStride, iWidth * sizeof(int16_t));
  hop_o_fwd_transform(bitDepth, block, coeff, iWidth, uiMode != REG_DCT && iWidth == 4);   // the mode only selects the DST for 4x4 (xTrMxN :793-799)
  for (int j = 0; j < iWidth * iHeight; j++) psCoeff[j] = coeff[j];
}

Void TComTrQuant::xIT(Int bitDepth, UInt uiMode, Int* plCoef, Pel* pResidual, UInt uiStride, Int iWidth, Int iHeight)
{
  g_calls2[3]++;
  int16_t block[32 * 32], coeff[32 * 32];
  for (int j = 0; j < iWidth * iHeight; j++) coeff[j] = (int16_t)plCoef[j];
  hop_o_inv_transform(bitDepth, coeff, block, iWidth, uiMode != REG_DCT && iWidth == 4);
  for (int j = 0; j < iHeight; j++) memcpy(pResidual + j * uiStride, block + j * iWidth, iWidth * sizeof(int16_t));
}

Void TComTrQuant::xDeQuant(Int bitDepth, const TCoeff* pSrc, Int* pDes, Int iWidth, Int iHeight, Int scalingListType)
{
  if (getUseScalingList()) { fprintf(stderr, "hop shim: scaling lists are not on the replaced path\n"); abort(); }
  g_calls2[4]++;
  hop_o_dequant_flat(bitDepth, m_cQP.m_iQP, pSrc, pDes, iWidth);
}

Void TComTrQuant::xRateDistOptQuant(TComDataCU* pcCU, Int* plSrcCoeff, TCoeff* piDstCoeff, Int*& piArlDstCoeff, UInt uiWidth, UInt uiHeight, UInt& uiAbsSum,
                                    TextType eTType, UInt uiAbsPartIdx)
{
  g_calls2[5]++;
  const int log2 = g_aucConvertToBit[uiWidth] + 2, comp = eTType == TEXT_LUMA ? 0 : (eTType == TEXT_CHROMA_U ? 1 : 2);
  const bool intra = pcCU->isIntra(uiAbsPartIdx);
  uint32_t as = uiAbsSum;
  hop_o_rdoq(plSrcCoeff, piDstCoeff, log2, comp, intra ? 1 : 0, (int)pcCU->getCoefScanIdx(uiAbsPartIdx, uiWidth, eTType == TEXT_LUMA, intra),
             pcCU->getTransformIdx(uiAbsPartIdx), m_cQP.m_iQP, eTType == TEXT_LUMA ? g_bitDepthY : g_bitDepthC,
             pcCU->getSlice()->getPPS()->getSignHideFlag() ? 1 : 0, m_dLambda, (const hop_o_estbits*)m_pcEstBitsSbac, &as);
  if (FILE* f = rdoq_trace_file()) {
    const int32_t hd[10] = { log2, comp, intra ? 1 : 0, (int)pcCU->getCoefScanIdx(uiAbsPartIdx, uiWidth, eTType == TEXT_LUMA, intra), pcCU->getTransformIdx(uiAbsPartIdx), m_cQP.m_iQP,
                             eTType == TEXT_LUMA ? g_bitDepthY : g_bitDepthC, pcCU->getSlice()->getPPS()->getSignHideFlag() ? 1 : 0, (int32_t)uiAbsSum, (int32_t)as };
    fwrite(hd, 4, 10, f); fwrite(&m_dLambda, 8, 1, f); fwrite(m_pcEstBitsSbac, sizeof(estBitsSbacStruct), 1, f);
    fwrite(plSrcCoeff, 4, uiWidth * uiHeight, f); fwrite(piDstCoeff, 4, uiWidth * uiHeight, f);
  }
  uiAbsSum = as;
}

// ---- the RDOQ bit-estimate tables: TEncSbac::estBit (TLibEncoder/TEncSbac.cpp:2175-2370) -> hop_o_cabac_est_bits (= hop_cabac_est_bits) ----
namespace { unsigned long g_calls3[1] = { 0 };
struct Report3 { ~Report3() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: estBit %lu\n", g_calls3[0]); } } g_report3; }
Void TEncSbac::estBit(estBitsSbacStruct* pcEstBitsSbac, Int width, Int height, TextType eTType)
{
  g_calls3[0]++;
  hop_o_cabac_ctx c;
  struct { ContextModel* p; int n; uint8_t* d; } sets[10] = {
    { m_cCUQtCbfSCModel.get(0), 8, c.qt_cbf }, { m_cCUTransSubdivFlagSCModel.get(0), 3, c.trans_subdiv }, { m_cCUQtRootCbfSCModel.get(0), 1, c.qt_root_cbf },
    { m_cCUSigCoeffGroupSCModel.get(0), 4, c.sig_cg }, { m_cCUSigSCModel.get(0), 42, c.sig }, { m_cCuCtxLastX.get(0), 30, c.last_x }, { m_cCuCtxLastY.get(0), 30, c.last_y },
    { m_cCUOneSCModel.get(0), 24, c.one }, { m_cCUAbsSCModel.get(0), 6, c.abs }, { m_cTransformSkipSCModel.get(0), 2, c.ts } };
  for (int i = 0; i < 10; i++) for (int j = 0; j < sets[i].n; j++) sets[i].d[j] = sets[i].p[j].m_ucState;
  hop_o_cabac_est_bits(&c, width, eTType == TEXT_LUMA ? 0 : 1, (hop_o_estbits*)pcEstBitsSbac);
}

// ---- rows a7 / a12: intra reference samples and prediction, HAD and SSE ----
//   TComPattern::fillReferenceSamples   TLibCommon/TComPattern.cpp:374-558     -> hop_o_intra_fill_refs   (luma: 4-sample units; the chroma calls keep the reference's)
//   TComPrediction::predIntraLumaAng    TLibCommon/TComPrediction.cpp:340-372  -> hop_o_intra_pred
//   TComRdCost::calcHAD                 TLibCommon/TComRdCost.cpp:391-442      -> hop_o_calc_had
//   TComRdCost::getDistPart             TLibCommon/TComRdCost.cpp:477-503      -> hop_o_sse / hop_o_sad (+ the chroma weight as the reference applies it)
extern "C" void hop_ref_orig_fill_refs(TComPattern*, Int, Pel*, Int*, Bool*, Int, Int, Int, Int, UInt, UInt, UInt, UInt, Int, Bool);   // the reference's own definition (Makefile.ref)
namespace { unsigned long g_calls4[4] = { 0, 0, 0, 0 };
struct Report4 { ~Report4() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: fillRefs %lu intraPred %lu calcHAD %lu distPart %lu\n",
                                                                  g_calls4[0], g_calls4[1], g_calls4[2], g_calls4[3]); } } g_report4; }

Void TComPattern::fillReferenceSamples(Int bitDepth, Pel* piRoiOrigin, Int* piAdiTemp, Bool* bNeighborFlags, Int iNumIntraNeighbor, Int iUnitSize, Int iNumUnitsInCu,
                                       Int iTotalUnits, UInt uiCuWidth, UInt uiCuHeight, UInt uiWidth, UInt uiHeight, Int iPicStride, Bool bLMmode)
{
  if ((iUnitSize != 4 && iUnitSize != 2) || bLMmode || uiCuWidth != uiCuHeight) {
    hop_ref_orig_fill_refs(this, bitDepth, piRoiOrigin, piAdiTemp, bNeighborFlags, iNumIntraNeighbor, iUnitSize, iNumUnitsInCu, iTotalUnits, uiCuWidth, uiCuHeight,
                           uiWidth, uiHeight, iPicStride, bLMmode);
    return;
  }
  g_calls4[0]++;
  const int N = (int)uiCuWidth;
  uint8_t flags[68]; int L[4 * 64 + 1];
  for (int i = 0; i < iTotalUnits; i++) flags[i] = bNeighborFlags[i] ? 1 : 0;
  hop_o_intra_fill_refs_u(piRoiOrigin, iPicStride, 0, 0, N, iUnitSize, flags, bitDepth, L);          // 4-sample units for luma, 2-sample units for the chroma planes
  piAdiTemp[0] = L[2 * N];                                                  // corner, above row, left column of the (2N+1)^2 buffer
  for (int i = 0; i < 2 * N; i++) piAdiTemp[1 + i] = L[2 * N + 1 + i];
  for (int i = 0; i < 2 * N; i++) piAdiTemp[(1 + i) * uiWidth] = L[2 * N - 1 - i];
}

Void TComPrediction::predIntraLumaAng(TComPattern* pcTComPattern, UInt uiDirMode, Pel* piPred, UInt uiStride, Int iWidth, Int iHeight, Bool bAbove, Bool bLeft)
{
  if (!bAbove || !bLeft || iWidth != iHeight) { fprintf(stderr, "hop shim: predIntraLumaAng without both neighbours is not on the replaced path\n"); abort(); }
  g_calls4[1]++;
  const int N = iWidth, sw = 2 * N + 1;
  int Lu[4 * 64 + 1], Lf[4 * 64 + 1];
  const Int* u = pcTComPattern->getAdiOrgBuf(N, N, m_piYuvExt); const Int* f = u + sw * sw;
  Lu[2 * N] = u[0]; Lf[2 * N] = f[0];
  for (int i = 0; i < 2 * N; i++) { Lu[2 * N + 1 + i] = u[1 + i]; Lf[2 * N + 1 + i] = f[1 + i]; Lu[2 * N - 1 - i] = u[(1 + i) * sw]; Lf[2 * N - 1 - i] = f[(1 + i) * sw]; }
  int16_t pred[64 * 64];
  hop_o_intra_pred(Lu, Lf, N, (int)uiDirMode, g_bitDepthY, pred);
  for (int r = 0; r < N; r++) memcpy(piPred + r * uiStride, pred + r * N, N * sizeof(Pel));
}

UInt TComRdCost::calcHAD(Int bitDepth, Pel* pi0, Int iStride0, Pel* pi1, Int iStride1, Int iWidth, Int iHeight)
{
  g_calls4[2]++;
  return hop_o_calc_had(pi0, iStride0, pi1, iStride1, iWidth, iHeight, bitDepth);
}

namespace { int g_fin_pending = 0; }   // set by the residual-quadtree shim: the next three getDistPart calls are the final Y / Cb / Cr distortions of that CU (:6807-6810)
UInt TComRdCost::getDistPart(Int bitDepth, Pel* piCur, Int iCurStride, Pel* piOrg, Int iOrgStride, UInt uiBlkWidth, UInt uiBlkHeight, TextType eText, DFunc eDFunc)
{
  g_calls4[3]++;
  UInt d;
  if (eDFunc == DF_SSE) d = hop_o_sse(piOrg, iOrgStride, piCur, iCurStride, uiBlkWidth, uiBlkHeight, bitDepth);
  else if (eDFunc == DF_SAD) d = hop_o_sad(piOrg, iOrgStride, piCur, iCurStride, uiBlkWidth, uiBlkHeight, bitDepth, 0);
  else { fprintf(stderr, "hop shim: getDistPart with distortion function %d is not on the replaced path\n", (int)eDFunc); abort(); }
  UInt out = d;
  if (eText == TEXT_CHROMA_U) out = (UInt)((Int)(m_cbDistortionWeight * d));
  if (eText == TEXT_CHROMA_V) out = (UInt)((Int)(m_crDistortionWeight * d));
  if (g_fin_pending > 0) {                                          // HOP_SHIM_TRACE_FIN=<file>: width, distortion, reconstruction, original of the three final calls
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_FIN"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      const int32_t hd[2] = { (int32_t)uiBlkWidth, (int32_t)out }; fwrite(hd, 4, 2, f);
      for (UInt y = 0; y < uiBlkHeight; y++) fwrite(piCur + y * iCurStride, 2, uiBlkWidth, f);
      for (UInt y = 0; y < uiBlkHeight; y++) fwrite(piOrg + y * iOrgStride, 2, uiBlkWidth, f);
    }
    g_fin_pending--;
  }
  return out;
}

// ---- a9 transform skip, a13 SS-reference upkeep ----
//   TComTrQuant::xTransformSkip / xITransformSkip  TLibCommon/TComTrQuant.cpp:1402-1460  -> hop_o_transform_skip / hop_o_inv_transform_skip
//   TEncCu::xCopyYuv2SSRef                         TLibEncoder/TEncCu.cpp:1677-1715      -> hop_o_ssref_commit_cu (= hop_ssref_commit_cus) at the leaf
#define private public
#define protected public
#include "TLibEncoder/TEncCu.h"
#undef private
#undef protected
namespace { unsigned long g_calls5[3] = { 0, 0, 0 };
struct Report5 { ~Report5() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: tskip %lu itskip %lu commit %lu\n", g_calls5[0], g_calls5[1], g_calls5[2]); } } g_report5; }

Void TComTrQuant::xTransformSkip(Int bitDepth, Pel* piBlkResi, UInt uiStride, Int* psCoeff, Int width, Int height)
{
  g_calls5[0]++;
  int16_t blk[32 * 32];
  for (int j = 0; j < height; j++) memcpy(blk + j * width, piBlkResi + j * uiStride, width * sizeof(int16_t));
  hop_o_transform_skip(bitDepth, blk, psCoeff, width);
}

Void TComTrQuant::xITransformSkip(Int bitDepth, Int* plCoef, Pel* pResidual, UInt uiStride, Int width, Int height)
{
  g_calls5[1]++;
  int16_t blk[32 * 32];
  hop_o_inv_transform_skip(bitDepth, plCoef, blk, width);
  for (int j = 0; j < height; j++) memcpy(pResidual + j * uiStride, blk + j * width, width * sizeof(int16_t));
}

Void TEncCu::xCopyYuv2SSRef(TComPic* rpcPic, UInt uiCUAddr, UInt uiAbsPartIdx, UInt uiDepth, UInt uiSrcDepth, TComDataCU* pcCU, UInt uiLPelX, UInt uiTPelY)
{
  // the quadtree walk down to blocks that lie inside the picture and the slice segment is the caller's logic, kept as it is;
  // the leaf -- block copy + re-extension of every border -- is the ABI call
  const UInt size = g_uiMaxCUWidth >> uiDepth, nParts = pcCU->getPic()->getNumPartInCU(), span = nParts >> (uiDepth << 1);
  TComSlice* slice = pcCU->getPic()->getSlice(pcCU->getPic()->getCurrSliceIdx());
  const UInt first = rpcPic->getPicSym()->getInverseCUOrderMap(pcCU->getAddr()) * nParts + uiAbsPartIdx;
  const UInt segStart = slice->getSliceSegmentCurStartCUAddr(), segEnd = slice->getSliceSegmentCurEndCUAddr();
  const bool cutByStart = segStart > first && segStart < first + span, cutByEnd = segEnd > first && segEnd < first + span;
  const UInt picW = slice->getSPS()->getPicWidthInLumaSamples(), picH = slice->getSPS()->getPicHeightInLumaSamples();
  if (!cutByStart && !cutByEnd && uiLPelX + size - 1 < picW && uiTPelY + size - 1 < picH) {
    g_calls5[2]++;
    TComYuv* src = m_ppcRecoYuvBest[uiSrcDepth];
    const UInt srcSize = g_uiMaxCUWidth >> uiSrcDepth;
    const UInt raster = g_auiZscanToRaster[uiAbsPartIdx], perRow = rpcPic->getNumPartInWidth();
    const UInt bx = ((raster % perRow) * 4) % srcSize, by = ((raster / perRow) * 4) % srcSize;       // the block inside the source CU, in samples (4-sample partitions)
    std::vector<int16_t> y(size * size), cb(size * size / 4), cr(size * size / 4);
    for (UInt r = 0; r < size; r++) memcpy(&y[r * size], src->getLumaAddr() + (by + r) * src->getStride() + bx, size * sizeof(int16_t));
    for (UInt r = 0; r < size / 2; r++) {
      memcpy(&cb[r * (size / 2)], src->getCbAddr() + (by / 2 + r) * src->getCStride() + bx / 2, (size / 2) * sizeof(int16_t));
      memcpy(&cr[r * (size / 2)], src->getCrAddr() + (by / 2 + r) * src->getCStride() + bx / 2, (size / 2) * sizeof(int16_t));
    }
    TComPicYuv* rec = rpcPic->getPicYuvRec();
    hop_o_ssref_commit_cu(rec->getLumaAddr(), rec->getCbAddr(), rec->getCrAddr(), rec->getWidth(), rec->getHeight(), (int)uiLPelX, (int)uiTPelY, (int)size, &y[0], &cb[0], &cr[0]);
    rec->setBorderExtension(true);                                     // the state the reference's extendPicBorder leaves
    return;
  }
  const UInt quarter = span >> 2;
  for (UInt q = 0; q < 4; q++) {
    const UInt idx = uiAbsPartIdx + q * quarter, x = uiLPelX + (size >> 1) * (q & 1), y = uiTPelY + (size >> 1) * (q >> 1);
    const UInt at = rpcPic->getPicSym()->getInverseCUOrderMap(pcCU->getAddr()) * nParts + idx;
    if (at + quarter > segStart && at < segEnd && x < picW && y < picH) xCopyYuv2SSRef(rpcPic, uiCUAddr, idx, uiDepth + 1, uiSrcDepth, pcCU, x, y);
  }
}

// ---- row a8b: the residual quadtree search of an SS/GT CU ----
//   TEncSearch::xEstimateResidualQT  TLibEncoder/TEncSearch.cpp:6824-7560 (+ xEncodeResidualQT :7562-7655)  -> hop_o_rqt
// The top-level call (from encodeResAndCalcRdInterCU, :6700) hands over the CU's residual, the quantiser / lambda / weight set and
// the coder state; the restatement walks the whole tree and leaves what the reference's function leaves: transform depth, cbf and
// transform-skip arrays of the CU, the per-layer coefficient and residual buffers xSetResidualQTData reads, the coder state.
namespace { unsigned long g_calls6[1] = { 0 };
struct Report6 { ~Report6() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: rqt %lu\n", g_calls6[0]); } } g_report6;
struct SbacSets { ContextModel* p[10]; int n[10]; };
SbacSets sbac_sets(TEncSbac* s) {
  SbacSets r = { { s->m_cCUQtCbfSCModel.get(0), s->m_cCUTransSubdivFlagSCModel.get(0), s->m_cCUQtRootCbfSCModel.get(0), s->m_cCUSigCoeffGroupSCModel.get(0), s->m_cCUSigSCModel.get(0),
                   s->m_cCuCtxLastX.get(0), s->m_cCuCtxLastY.get(0), s->m_cCUOneSCModel.get(0), s->m_cCUAbsSCModel.get(0), s->m_cTransformSkipSCModel.get(0) },
                 { 8, 3, 1, 4, 42, 30, 30, 24, 6, 2 } };
  return r;
}
void coder_get(TEncSbac* s, hop_o_coder* c) {
  SbacSets r = sbac_sets(s); uint8_t* d = (uint8_t*)&c->ctx;
  for (int i = 0; i < 10; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState;
  c->frac = s->m_pcBinIf->getTEncBinCABAC()->m_fracBits;
}
void coder_put(TEncSbac* s, const hop_o_coder* c) {
  SbacSets r = sbac_sets(s); const uint8_t* d = (const uint8_t*)&c->ctx;
  for (int i = 0; i < 10; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++;
  s->m_pcBinIf->getTEncBinCABAC()->m_fracBits = c->frac;
}
}

Void TEncSearch::xEstimateResidualQT(TComDataCU* pcCU, UInt uiQuadrant, UInt uiAbsPartIdx, UInt absTUPartIdx, TComYuv* pcResi, const UInt uiDepth,
                                     Double& rdCost, UInt& ruiBits, UInt& ruiDist, UInt* puiZeroDist)
{
  if (uiAbsPartIdx != 0 || uiDepth != pcCU->getDepth(0)) { fprintf(stderr, "hop shim: xEstimateResidualQT is replaced at the CU level only\n"); abort(); }
  g_calls6[0]++;
  TComSlice* sl = pcCU->getSlice();
  hop_o_rqt_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.log2_cu = g_aucConvertToBit[sl->getSPS()->getMaxCUWidth() >> uiDepth] + 2;
  m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_LUMA, sl->getSPS()->getQpBDOffsetY(), 0); cfg.qp[0] = m_pcTrQuant->m_cQP.m_iQP;
  m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_CHROMA, sl->getSPS()->getQpBDOffsetC(), sl->getPPS()->getChromaCbQpOffset() + sl->getSliceQpDeltaCb()); cfg.qp[1] = m_pcTrQuant->m_cQP.m_iQP;
  m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_CHROMA, sl->getSPS()->getQpBDOffsetC(), sl->getPPS()->getChromaCrQpOffset() + sl->getSliceQpDeltaCr()); cfg.qp[2] = m_pcTrQuant->m_cQP.m_iQP;
  cfg.bit_depth_y = g_bitDepthY; cfg.bit_depth_c = g_bitDepthC;
  cfg.sign_hide = sl->getPPS()->getSignHideFlag() ? 1 : 0;
  cfg.use_ts = (sl->getPPS()->getUseTransformSkip() && !pcCU->isLosslessCoded(0)) ? 1 : 0;
  if (!m_pcEncCfg->getUseRDOQ() || !m_pcEncCfg->getUseRDOQTS() || pcCU->isLosslessCoded(0)) { fprintf(stderr, "hop shim: the residual quadtree is replaced for RDOQ + RDOQTS, lossy coding\n"); abort(); }
  cfg.log2_max_tu = sl->getSPS()->getQuadtreeTULog2MaxSize(); cfg.log2_min_tu_in_cu = pcCU->getQuadtreeTULog2MinSizeInCU(0);
  cfg.inter_split_flag = (sl->getSPS()->getQuadtreeTUMaxDepthInter() == 1 && pcCU->getPredictionMode(0) == MODE_INTER && pcCU->getPartitionSize(0) != SIZE_2Nx2N) ? 1 : 0;
  cfg.lambda_rd = m_pcRdCost->m_dLambda;
  for (int c = 0; c < 3; c++) cfg.lambda_rdoq[c] = m_pcTrQuant->m_lambdas[c];
  cfg.dist_weight[0] = 1.0; cfg.dist_weight[1] = m_pcRdCost->m_cbDistortionWeight; cfg.dist_weight[2] = m_pcRdCost->m_crDistortionWeight;
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  std::vector<int16_t> planes[4][3];
  for (int l = 0; l < 4; l++) {
    st.coef[l][0] = m_ppcQTTempCoeffY[l]; st.coef[l][1] = m_ppcQTTempCoeffCb[l]; st.coef[l][2] = m_ppcQTTempCoeffCr[l];   // the reference's own layer buffers, same layout
    for (int c = 0; c < 3; c++) { planes[l][c].assign(c ? cu * cu / 4 : cu * cu, 0); st.resi[l][c] = &planes[l][c][0]; }
  }
  // what the arrays hold on entry is part of the function's input where it is not overwritten (a 64x64 CU never writes depth 0)
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  for (int l = 0; l < 4; l++) {                                      // and so do the residual layers
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int y = 0; y < cu; y++) memcpy(st.resi[l][0] + y * cu, t.getLumaAddr() + y * t.getStride(), cu * sizeof(Pel));
    for (int y = 0; y < cu / 2; y++) { memcpy(st.resi[l][1] + y * (cu / 2), t.getCbAddr() + y * t.getCStride(), (cu / 2) * sizeof(Pel));
                                       memcpy(st.resi[l][2] + y * (cu / 2), t.getCrAddr() + y * t.getCStride(), (cu / 2) * sizeof(Pel)); }
  }
  hop_o_coder coder; coder_get(m_pcRDGoOnSbacCoder, &coder);
  const hop_o_coder coder_in = coder;
  m_pcRDGoOnSbacCoder->store(m_pppcRDSbacCoder[uiDepth][CI_QT_TRAFO_ROOT]);
  double cost = 0; uint32_t bits = 0, dist = 0, zd = 0;
  hop_o_rqt(&cfg, pcResi->getLumaAddr(), pcResi->getStride(), pcResi->getCbAddr(), pcResi->getCrAddr(), pcResi->getCStride(), &coder, &st, &cost, &bits, &dist, &zd);
  rdCost += cost; ruiBits += bits; ruiDist += dist; if (puiZeroDist) *puiZeroDist += zd;
  {                                                                 // HOP_SHIM_TRACE_RQT=<file>: cfg, coder in / out, residual planes, results, arrays, chosen levels
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_RQT"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&coder_in, sizeof(coder_in), 1, f); fwrite(&coder, sizeof(coder), 1, f);
      for (int y = 0; y < cu; y++) fwrite(pcResi->getLumaAddr() + y * pcResi->getStride(), 2, cu, f);
      for (int y = 0; y < cu / 2; y++) fwrite(pcResi->getCbAddr() + y * pcResi->getCStride(), 2, cu / 2, f);
      for (int y = 0; y < cu / 2; y++) fwrite(pcResi->getCrAddr() + y * pcResi->getCStride(), 2, cu / 2, f);
      const uint32_t o4[4] = { bits, dist, zd, 0 }; fwrite(&cost, 8, 1, f); fwrite(o4, 4, 4, f);
      fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
      std::vector<int32_t> fin(cu * cu * 3 / 2); hop_o_rqt_final_coeffs(&cfg, &st, &fin[0]); fwrite(&fin[0], 4, fin.size(), f);
    }
  }
  memcpy(pcCU->m_puhTrIdx, st.tr_idx, parts);
  for (int c = 0; c < 3; c++) { memcpy(pcCU->m_puhCbf[c], st.cbf[c], parts); memcpy(pcCU->m_puhTransformSkip[c], st.tskip[c], parts); }
  for (int l = 0; l < 4; l++) {
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int y = 0; y < cu; y++) memcpy(t.getLumaAddr() + y * t.getStride(), st.resi[l][0] + y * cu, cu * sizeof(Pel));
    for (int y = 0; y < cu / 2; y++) { memcpy(t.getCbAddr() + y * t.getCStride(), st.resi[l][1] + y * (cu / 2), (cu / 2) * sizeof(Pel));
                                       memcpy(t.getCrAddr() + y * t.getCStride(), st.resi[l][2] + y * (cu / 2), (cu / 2) * sizeof(Pel)); }
  }
  coder_put(m_pcRDGoOnSbacCoder, &coder);
  g_fin_pending = 3;
}

// ---- the CU-level syntax bits of an SS/GT CU: TEncSearch::xAddSymbolBitsInter (TLibEncoder/TEncSearch.cpp:7779-7810) -> hop_o_inter_cu_bits ----
namespace { unsigned long g_calls7[1] = { 0 };
struct Report7 { ~Report7() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: cuBits %lu\n", g_calls7[0]); } } g_report7;
struct CuSets { ContextModel* p[9]; int n[9]; };
CuSets cu_sets(TEncSbac* s) {
  CuSets r = { { s->m_cCUSkipFlagSCModel.get(0), s->m_cCUMergeFlagExtSCModel.get(0), s->m_cCUMergeIdxExtSCModel.get(0), s->m_cCUPartSizeSCModel.get(0), s->m_cCUPredModeSCModel.get(0),
                 s->m_cCUMvdSCModel.get(0), s->m_cMVPIdxSCModel.get(0), s->m_cCUGTFlagExtSCModel.get(0), s->m_cCUGTSCModel.get(0) }, { 3, 1, 1, 4, 1, 2, 1, 1, 2 } };
  return r;
}
}

Void TEncSearch::xAddSymbolBitsInter(TComDataCU* pcCU, UInt uiQp, UInt uiTrMode, UInt& ruiBits, TComYuv*& rpcYuvRec, TComYuv* pcYuvPred, TComYuv*& rpcYuvResi)
{
  g_calls7[0]++;
  TComSlice* sl = pcCU->getSlice();
  const UInt depth = pcCU->getDepth(0);
  if (sl->getPPS()->getTransquantBypassEnableFlag() || sl->getPPS()->getUseDQP() || sl->getNumRefIdx(REF_PIC_LIST_0) != 1 || sl->getNumRefIdx(REF_PIC_LIST_1) > 0 || sl->isIntra()) {
    fprintf(stderr, "hop shim: xAddSymbolBitsInter is replaced for one SS reference, no transquant bypass, no delta QP\n"); abort();
  }
  hop_o_rqt_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.log2_cu = g_aucConvertToBit[sl->getSPS()->getMaxCUWidth() >> depth] + 2;
  cfg.sign_hide = sl->getPPS()->getSignHideFlag() ? 1 : 0; cfg.use_ts = sl->getPPS()->getUseTransformSkip() ? 1 : 0;
  cfg.log2_max_tu = sl->getSPS()->getQuadtreeTULog2MaxSize(); cfg.log2_min_tu_in_cu = pcCU->getQuadtreeTULog2MinSizeInCU(0);
  cfg.inter_split_flag = (sl->getSPS()->getQuadtreeTUMaxDepthInter() == 1 && pcCU->getPartitionSize(0) != SIZE_2Nx2N) ? 1 : 0;
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  std::vector<int32_t> coef(cu * cu * 3 / 2);
  memcpy(&coef[0], pcCU->getCoeffY(), cu * cu * 4); memcpy(&coef[cu * cu], pcCU->getCoeffCb(), cu * cu); memcpy(&coef[cu * cu + cu * cu / 4], pcCU->getCoeffCr(), cu * cu);
  hop_o_cu_syntax y; memset(&y, 0, sizeof(y));
  const PartSize ps = pcCU->getPartitionSize(0);
  y.part_size = (int)ps; y.n_pu = ps == SIZE_2Nx2N ? 1 : (ps == SIZE_NxN ? 4 : 2);
  y.skip_flag = pcCU->isSkipped(0) ? 1 : 0; y.skip_ctx = (int)pcCU->getCtxSkipFlag(0);
  y.amp_acc = sl->getSPS()->getAMPAcc(depth) ? 1 : 0; y.is_min_cu = depth == g_uiMaxCUDepth - g_uiAddCUDepth; y.max_merge_cand = (int)sl->getMaxNumMergeCand();
  const UInt puOffset = (g_auiPUOffset[UInt(ps)] << ((sl->getSPS()->getMaxCUDepth() - depth) << 1)) >> 4;
  for (int p = 0; p < y.n_pu; p++) {
    const UInt idx = p * puOffset;
    y.pu[p].merge_flag = pcCU->getMergeFlag(idx) ? 1 : 0; y.pu[p].merge_idx = (int)pcCU->getMergeIndex(idx);
    if (!y.pu[p].merge_flag && pcCU->getInterDir(idx) != 1) { fprintf(stderr, "hop shim: a PU that is not predicted from list 0\n"); abort(); }
    const TComMv d = pcCU->getCUMvField(REF_PIC_LIST_0)->getMvd(idx);
    y.pu[p].mvd[0] = d.getHor(); y.pu[p].mvd[1] = d.getVer(); y.pu[p].mvp_idx = pcCU->getMVPIdx(REF_PIC_LIST_0, idx);
    y.pu[p].gt_flag = pcCU->getGTFlag(idx) ? 1 : 0;
    const TComMv g0 = pcCU->getCUGT0Field(REF_PIC_LIST_0)->getMv(idx), g1 = pcCU->getCUGT1Field(REF_PIC_LIST_0)->getMv(idx), g2 = pcCU->getCUGT2Field(REF_PIC_LIST_0)->getMv(idx),
                g3 = pcCU->getCUGT3Field(REF_PIC_LIST_0)->getMv(idx);
    const int gt[8] = { g0.getHor(), g0.getVer(), g1.getHor(), g1.getVer(), g2.getHor(), g2.getVer(), g3.getHor(), g3.getVer() };
    memcpy(y.pu[p].gt, gt, sizeof(gt));
  }
  TEncSbac* sb = m_pcRDGoOnSbacCoder;
  hop_o_coder coder; coder_get(sb, &coder);
  uint8_t cuctx[16]; { CuSets r = cu_sets(sb); uint8_t* d = cuctx; for (int i = 0; i < 9; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  int skipped = 0;
  const hop_o_coder coder_in = coder; uint8_t cu_in[16]; memcpy(cu_in, cuctx, 16);
  const uint32_t bits = hop_o_inter_cu_bits(&cfg, &y, &st, &coef[0], &coder, cuctx, &skipped);
  {                                                                 // HOP_SHIM_TRACE_CUBITS=<file>: cfg, syntax, arrays, levels, coder / CU contexts in and out, bits, skipped
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_CUBITS"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
      fwrite(&coef[0], 4, coef.size(), f); fwrite(&coder_in, sizeof(coder_in), 1, f); fwrite(cu_in, 1, 16, f); fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 16, f);
      const int32_t o2[2] = { (int32_t)bits, skipped }; fwrite(o2, 4, 2, f);
    }
  }
  if (skipped && !y.skip_flag) pcCU->setSkipFlagSubParts(true, 0, depth);
  coder_put(sb, &coder);
  { CuSets r = cu_sets(sb); const uint8_t* d = cuctx; for (int i = 0; i < 9; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++; }
  ruiBits += bits;
}

// ---- the mode-decision half of the intra rough search: TEncSearch::xModeBitsIntra (:7734-7745) -> hop_o_intra_mode_bits, xUpdateCandList (:7747-7767) -> hop_o_cand_update ----
namespace { unsigned long g_calls8[2] = { 0, 0 };
struct Report8 { ~Report8() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: modeBits %lu candList %lu\n", g_calls8[0], g_calls8[1]); } } g_report8; }

UInt TEncSearch::xModeBitsIntra(TComDataCU* pcCU, UInt uiMode, UInt uiPU, UInt uiPartOffset, UInt uiDepth, UInt uiInitTrDepth)
{
  g_calls8[0]++;
  m_pcRDGoOnSbacCoder->loadIntraDirModeLuma(m_pppcRDSbacCoder[uiDepth][CI_CURR_BEST]);
  pcCU->setLumaIntraDirSubParts(uiMode, uiPartOffset, uiDepth + uiInitTrDepth);
  Int preds[3] = { -1, -1, -1 };
  const Int predNum = pcCU->getIntraDirLumaPredictor(uiPartOffset, preds);
  ContextModel* cm = m_pcRDGoOnSbacCoder->m_cCUIntraPredSCModel.get(0);
  uint8_t st = cm->m_ucState; uint64_t frac = m_pcRDGoOnSbacCoder->m_pcBinIf->getTEncBinCABAC()->m_fracBits;
  const uint32_t bits = hop_o_intra_mode_bits(&st, &frac, (int)uiMode, preds, predNum);
  {                                                                 // HOP_SHIM_TRACE_MODEBITS=<file>: state, fraction, mode, predictors -> bits
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_MODEBITS"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) { const int32_t rec[8] = { cm->m_ucState, (int32_t)(m_pcRDGoOnSbacCoder->m_pcBinIf->getTEncBinCABAC()->m_fracBits & 32767), (int32_t)uiMode, preds[0], preds[1], preds[2], predNum, (int32_t)bits };
             fwrite(rec, 4, 8, f); }
  }
  cm->m_ucState = st; m_pcRDGoOnSbacCoder->m_pcBinIf->getTEncBinCABAC()->m_fracBits = frac;
  return bits;
}

UInt TEncSearch::xUpdateCandList(UInt uiMode, Double uiCost, UInt uiFastCandNum, UInt* CandModeList, Double* CandCostList)
{
  g_calls8[1]++;
  return (UInt)hop_o_cand_update((int)uiMode, uiCost, (int)uiFastCandNum, CandModeList, CandCostList);
}

// ---- the bits of an intra CU's quadtree: TEncSearch::xGetIntraBitsQT (TLibEncoder/TEncSearch.cpp:957-980) -> hop_o_intra_cu_bits ----
namespace { unsigned long g_calls9[1] = { 0 };
struct Report9 { ~Report9() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: intraBits %lu\n", g_calls9[0]); } } g_report9;
struct CuSets2 { ContextModel* p[11]; int n[11]; };
CuSets2 cu_sets2(TEncSbac* s) {
  CuSets2 r = { { s->m_cCUSkipFlagSCModel.get(0), s->m_cCUMergeFlagExtSCModel.get(0), s->m_cCUMergeIdxExtSCModel.get(0), s->m_cCUPartSizeSCModel.get(0), s->m_cCUPredModeSCModel.get(0),
                  s->m_cCUMvdSCModel.get(0), s->m_cMVPIdxSCModel.get(0), s->m_cCUGTFlagExtSCModel.get(0), s->m_cCUGTSCModel.get(0), s->m_cCUIntraPredSCModel.get(0),
                  s->m_cCUChromaPredSCModel.get(0) }, { 3, 1, 1, 4, 1, 2, 1, 1, 2, 1, 2 } };
  return r;
}
}

UInt TEncSearch::xGetIntraBitsQT(TComDataCU* pcCU, UInt uiTrDepth, UInt uiAbsPartIdx, Bool bLuma, Bool bChroma, Bool bRealCoeff)
{
  g_calls9[0]++;
  TComSlice* sl = pcCU->getSlice();
  if (bRealCoeff || sl->getSPS()->getUsePCM() || sl->getPPS()->getTransquantBypassEnableFlag()) {
    fprintf(stderr, "hop shim: xGetIntraBitsQT is replaced for the layer coefficients, no PCM, no transquant bypass\n"); abort();
  }
  const UInt depth = pcCU->getDepth(0);
  hop_o_rqt_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.log2_cu = g_aucConvertToBit[sl->getSPS()->getMaxCUWidth() >> depth] + 2;
  cfg.sign_hide = sl->getPPS()->getSignHideFlag() ? 1 : 0; cfg.use_ts = sl->getPPS()->getUseTransformSkip() ? 1 : 0;
  cfg.log2_max_tu = sl->getSPS()->getQuadtreeTULog2MaxSize(); cfg.log2_min_tu_in_cu = pcCU->getQuadtreeTULog2MinSizeInCU(0);
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  for (int l = 0; l < 4; l++) { st.coef[l][0] = m_ppcQTTempCoeffY[l]; st.coef[l][1] = m_ppcQTTempCoeffCb[l]; st.coef[l][2] = m_ppcQTTempCoeffCr[l]; }
  hop_o_intra_syntax y; memset(&y, 0, sizeof(y));
  y.part_nxn = pcCU->getPartitionSize(0) == SIZE_NxN ? 1 : 0;
  y.skip_flag = pcCU->isSkipped(0) ? 1 : 0; y.skip_ctx = sl->isIntra() ? -1 : (int)pcCU->getCtxSkipFlag(0); y.is_min_cu = depth == g_uiMaxCUDepth - g_uiAddCUDepth;   // -1: an I slice codes neither skip flag nor prediction mode
  for (int p = 0; p < (y.part_nxn ? 4 : 1); p++) {
    const UInt idx = p * (parts >> 2);
    y.luma_dir[p] = pcCU->getLumaIntraDir(idx);
    Int pr[3] = { -1, -1, -1 }; y.pred_num[p] = pcCU->getIntraDirLumaPredictor(idx, pr);
    for (int k = 0; k < 3; k++) y.preds[p][k] = pr[k];
  }
  y.chroma_is_dm = pcCU->getChromaIntraDir(0) == DM_CHROMA_IDX; y.chroma_dir = pcCU->getChromaIntraDir(0);
  TEncSbac* sb = m_pcRDGoOnSbacCoder;
  hop_o_coder coder; coder_get(sb, &coder);
  uint8_t cuctx[20] = { 0 }; { CuSets2 r = cu_sets2(sb); uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  const hop_o_coder coder_in = coder; uint8_t cu_in[20]; memcpy(cu_in, cuctx, 20);
  const uint32_t bits = hop_o_intra_cu_bits(&cfg, &y, &st, (int)uiTrDepth, (int)uiAbsPartIdx, bLuma ? 1 : 0, bChroma ? 1 : 0, &coder, cuctx);
  {                                                                 // HOP_SHIM_TRACE_INTRABITS=<file>
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_INTRABITS"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      const int32_t nd[4] = { (int32_t)uiTrDepth, (int32_t)uiAbsPartIdx, bLuma ? 1 : 0, bChroma ? 1 : 0 };
      fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(nd, 4, 4, f); fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
      for (int l = 0; l < 4; l++) { fwrite(st.coef[l][0], 4, cu * cu, f); fwrite(st.coef[l][1], 4, cu * cu / 4, f); fwrite(st.coef[l][2], 4, cu * cu / 4, f); }
      fwrite(&coder_in, sizeof(coder_in), 1, f); fwrite(cu_in, 1, 20, f); fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
      const int32_t b1 = (int32_t)bits; fwrite(&b1, 4, 1, f);
    }
  }
  coder_put(sb, &coder);
  { CuSets2 r = cu_sets2(sb); const uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++; }
  return bits;
}

// what both intra shims hand to the restatement: quantiser / lambda / tree limits, the CU's syntax elements, the neighbour flags of every node from (uiTrDepth, uiAbsPartIdx) downwards
// (TComPattern::initAdiPattern, TComPattern.cpp:199-211: position-only)
namespace {
void intra_env(TEncSearch* self, TComDataCU* pcCU, UInt uiTrDepth, UInt uiAbsPartIdx, hop_o_rqt_cfg& cfg_out, hop_o_intra_syntax& y_out, std::vector<uint8_t>& avail_out)
{
  TComSlice* sl = pcCU->getSlice();
  const UInt depth = pcCU->getDepth(0);
  hop_o_rqt_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.log2_cu = g_aucConvertToBit[sl->getSPS()->getMaxCUWidth() >> depth] + 2;
  self->m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_LUMA, sl->getSPS()->getQpBDOffsetY(), 0); cfg.qp[0] = self->m_pcTrQuant->m_cQP.m_iQP;
  cfg.bit_depth_y = g_bitDepthY; cfg.bit_depth_c = g_bitDepthC;
  cfg.sign_hide = sl->getPPS()->getSignHideFlag() ? 1 : 0; cfg.use_ts = sl->getPPS()->getUseTransformSkip() ? 1 : 0;
  cfg.log2_max_tu = sl->getSPS()->getQuadtreeTULog2MaxSize(); cfg.log2_min_tu_in_cu = pcCU->getQuadtreeTULog2MinSizeInCU(uiAbsPartIdx);
  cfg.lambda_rd = self->m_pcRdCost->m_dLambda;
  for (int c = 0; c < 3; c++) cfg.lambda_rdoq[c] = self->m_pcTrQuant->m_lambdas[c];
  cfg.dist_weight[0] = 1.0; cfg.dist_weight[1] = self->m_pcRdCost->m_cbDistortionWeight; cfg.dist_weight[2] = self->m_pcRdCost->m_crDistortionWeight;
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_intra_syntax y; memset(&y, 0, sizeof(y));
  y.part_nxn = pcCU->getPartitionSize(0) == SIZE_NxN ? 1 : 0;
  y.skip_flag = pcCU->isSkipped(0) ? 1 : 0; y.skip_ctx = sl->isIntra() ? -1 : (int)pcCU->getCtxSkipFlag(0); y.is_min_cu = depth == g_uiMaxCUDepth - g_uiAddCUDepth;   // -1: an I slice codes neither skip flag nor prediction mode
  for (int p = 0; p < (y.part_nxn ? 4 : 1); p++) {
    const UInt idx = p * (parts >> 2);
    y.luma_dir[p] = pcCU->getLumaIntraDir(idx);
    Int pr[3] = { -1, -1, -1 }; y.pred_num[p] = pcCU->getIntraDirLumaPredictor(idx, pr);
    for (int k = 0; k < 3; k++) y.preds[p][k] = pr[k];
  }
  y.chroma_is_dm = pcCU->getChromaIntraDir(0) == DM_CHROMA_IDX; y.chroma_dir = pcCU->getChromaIntraDir(0);
  // neighbour flags of every node below this one (TComPattern::initAdiPattern, TComPattern.cpp:199-211)
  std::vector<uint8_t> avail((size_t)341 * HOP_O_AVAIL_PITCH, 0);
  for (UInt d = uiTrDepth; cfg.log2_cu - (int)d >= 2 && cfg.log2_cu - (int)d >= cfg.log2_min_tu_in_cu; d++) {
    const int log2 = cfg.log2_cu - (int)d; if (log2 > 5) continue;
    const int np = 1 << (2 * (log2 - 2)), units = (1 << log2) / 4, first = (int)uiAbsPartIdx, count = parts >> (2 * uiTrDepth);
    for (int p = first; p < first + count; p += np) {
      UInt lt, rt, lb; Bool fl[4 * MAX_NUM_SPU_W + 1]; memset(fl, 0, sizeof(fl));
      pcCU->deriveLeftRightTopIdxAdi(lt, rt, (UInt)p, d); pcCU->deriveLeftBottomIdxAdi(lb, (UInt)p, d);
      TComPattern* pt = pcCU->getPattern();
      fl[units * 2] = pt->isAboveLeftAvailable(pcCU, lt);
      pt->isAboveAvailable(pcCU, lt, rt, fl + units * 2 + 1); pt->isAboveRightAvailable(pcCU, lt, rt, fl + units * 3 + 1);
      pt->isLeftAvailable(pcCU, lt, lb, fl + units * 2 - 1); pt->isBelowLeftAvailable(pcCU, lt, lb, fl + units - 1);
      uint8_t* a = &avail[(size_t)hop_o_intra_node_index((int)d, log2, p) * HOP_O_AVAIL_PITCH];
      for (int i = 0; i < 4 * units + 1; i++) a[i] = fl[i] ? 1 : 0;
    }
  }
  cfg_out = cfg; y_out = y; avail_out.swap(avail);
}
}

// ---- the luma transform tree of an intra PU: TEncSearch::xRecurIntraCodingQT, bLumaOnly (TLibEncoder/TEncSearch.cpp:1361-1710) -> hop_o_intra_rqt ----
// The call from estIntraPredQT (:2524 per candidate mode with bCheckFirst, :2587 for the best one without) hands over the PU; the restatement
// walks the tree and leaves what the reference leaves: transform depth / cbf / transform-skip arrays, the level and reconstruction layers
// xSetIntraResultQT reads, the reconstruction picture, the coder state.  Neighbour availability of every node comes from the reference's
// own TComPattern helpers (it depends on positions only).
extern "C" void hop_ref_orig_recur_intra(TEncSearch*, TComDataCU*, UInt, UInt, Bool, TComYuv*, TComYuv*, TComYuv*, UInt&, UInt&, Bool, Double&);   // the reference's own definitions (Makefile.ref)
extern "C" void hop_ref_orig_est_intra(TEncSearch*, TComDataCU*, TComYuv*, TComYuv*, TComYuv*, TComYuv*, UInt&, Bool);
namespace { bool hand_back(const char* name) { const char* e = getenv("HOP_SHIM_ORIG"); return e && strstr(e, name); } }
namespace { unsigned long g_calls11[1] = { 0 };
struct Report11 { ~Report11() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: intraRqt %lu\n", g_calls11[0]); } } g_report11; }

Void TEncSearch::xRecurIntraCodingQT(TComDataCU* pcCU, UInt uiTrDepth, UInt uiAbsPartIdx, Bool bLumaOnly, TComYuv* pcOrgYuv, TComYuv* pcPredYuv, TComYuv* pcResiYuv,
                                     UInt& ruiDistY, UInt& ruiDistC, Bool bCheckFirst, Double& dRDCost)
{
  static const bool orig = hand_back("xRecurIntraCodingQT");
  if (orig) { hop_ref_orig_recur_intra(this, pcCU, uiTrDepth, uiAbsPartIdx, bLumaOnly, pcOrgYuv, pcPredYuv, pcResiYuv, ruiDistY, ruiDistC, bCheckFirst, dRDCost); return; }
  TComSlice* sl = pcCU->getSlice();
  if (!bLumaOnly || m_pcEncCfg->getRDpenalty() || !m_pcEncCfg->getUseRDOQ() || !m_pcEncCfg->getUseRDOQTS() || pcCU->getCUTransquantBypass(0) ||
      sl->getSPS()->getUsePCM()) {
    fprintf(stderr, "hop shim: xRecurIntraCodingQT is replaced for luma-only trees, RDOQ + RDOQTS, no RD penalty, lossy\n"); abort();
  }
  g_calls11[0]++;
  hop_o_rqt_cfg cfg; hop_o_intra_syntax y; std::vector<uint8_t> avail;
  intra_env(this, pcCU, uiTrDepth, uiAbsPartIdx, cfg, y, avail);
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  std::vector<int16_t> planes[4];
  for (int l = 0; l < 4; l++) {
    st.coef[l][0] = m_ppcQTTempCoeffY[l]; st.coef[l][1] = m_ppcQTTempCoeffCb[l]; st.coef[l][2] = m_ppcQTTempCoeffCr[l];
    planes[l].assign(cu * cu, 0); st.resi[l][0] = &planes[l][0];
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int r = 0; r < cu; r++) memcpy(st.resi[l][0] + r * cu, t.getLumaAddr() + r * t.getStride(), cu * sizeof(Pel));
  }
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  TComPicYuv* recPic = pcCU->getPic()->getPicYuvRec();
  hop_o_intra_rqt_in in; memset(&in, 0, sizeof(in));
  in.org = pcOrgYuv->getLumaAddr(); in.org_stride = pcOrgYuv->getStride();
  in.rec = recPic->getLumaAddr(pcCU->getAddr(), pcCU->getZorderIdxInCU()); in.rec_stride = recPic->getStride();
  in.avail = &avail[0]; in.strong = sl->getSPS()->getUseStrongIntraSmoothing() ? 1 : 0; in.check_first = bCheckFirst ? 1 : 0;
  in.ts_fast = m_pcEncCfg->getUseTransformSkipFast() ? 1 : 0;
  TEncSbac* sb = m_pcRDGoOnSbacCoder;
  hop_o_coder coder; coder_get(sb, &coder);
  uint8_t cuctx[20] = { 0 }; { CuSets2 r = cu_sets2(sb); uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  // trace input: HOP_SHIM_TRACE_IRQT=<file>
  static FILE* f = NULL; static bool tried = false;
  if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_IRQT"); if (pth && *pth) f = fopen(pth, "wb"); }
  const int W = 2 * cu + 1;
  const int px = (int)pcCU->getCUPelX(), py = (int)pcCU->getCUPelY(), pw = recPic->getWidth(), ph = recPic->getHeight();
  if (f) {
    const int32_t nd[8] = { (int32_t)uiTrDepth, (int32_t)uiAbsPartIdx, in.check_first, in.ts_fast, in.strong, 0, 0, 0 };
    fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(nd, 4, 8, f); fwrite(&avail[0], 1, avail.size(), f);
    for (int r = 0; r < cu; r++) fwrite(in.org + r * in.org_stride, 2, cu, f);
    std::vector<int16_t> win((size_t)W * W, 0);                       // the picture around the CU, from (-1, -1); outside the picture: 0 (never read)
    for (int r = 0; r < W; r++) for (int c = 0; c < W; c++) {
      const int X = px - 1 + c, Y = py - 1 + r;
      if (X >= 0 && Y >= 0 && X < pw && Y < ph) win[(size_t)r * W + c] = in.rec[(ptrdiff_t)(r - 1) * in.rec_stride + (c - 1)];
    }
    fwrite(&win[0], 2, win.size(), f);
    fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
    fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
  }
  double cost = 0; uint32_t dist = 0;
  hop_o_intra_rqt(&cfg, &y, &in, (int)uiTrDepth, (int)uiAbsPartIdx, &coder, cuctx, &st, &cost, &dist);
  dRDCost += cost; ruiDistY += dist;
  if (f) {
    const uint32_t o2[2] = { dist, 0 }; fwrite(&cost, 8, 1, f); fwrite(o2, 4, 2, f);
    fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
    fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
    for (int r = 0; r < cu; r++) fwrite(in.rec + (ptrdiff_t)r * in.rec_stride, 2, cu, f);
    std::vector<int32_t> fin(cu * cu * 3 / 2); hop_o_rqt_final_coeffs(&cfg, &st, &fin[0]); fwrite(&fin[0], 4, cu * cu, f);
  }
  memcpy(pcCU->m_puhTrIdx, st.tr_idx, parts);
  memcpy(pcCU->m_puhCbf[0], st.cbf[0], parts); memcpy(pcCU->m_puhTransformSkip[0], st.tskip[0], parts);
  for (int l = 0; l < 4; l++) {
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int r = 0; r < cu; r++) memcpy(t.getLumaAddr() + r * t.getStride(), st.resi[l][0] + r * cu, cu * sizeof(Pel));
  }
  coder_put(sb, &coder);
  { CuSets2 r = cu_sets2(sb); const uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++; }
}

// ---- the luma intra search of a CU: TEncSearch::estIntraPredQT, bLumaOnly (TLibEncoder/TEncSearch.cpp:2386-2710) -> hop_o_intra_luma_search ----
// xCheckRDCostIntra hands over the CU; the restatement derives the most probable modes, runs the rough search, the candidate and the final transform trees per PU
// and leaves what the reference leaves: luma directions, transform depth / cbf / transform-skip arrays, the luma levels in the CU's coefficient buffer, the
// reconstruction in pcRecoYuv and (all but the last PU: the decided block; the last PU: whatever the final pass wrote) in the picture, the total distortion.
namespace { unsigned long g_calls12[1] = { 0 };
struct Report12 { ~Report12() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: intraSearch %lu\n", g_calls12[0]); } } g_report12; }

Void TEncSearch::estIntraPredQT(TComDataCU* pcCU, TComYuv* pcOrgYuv, TComYuv* pcPredYuv, TComYuv* pcResiYuv, TComYuv* pcRecoYuv, UInt& ruiDistC, Bool bLumaOnly)
{
  static const bool orig = hand_back("estIntraPredQT");
  if (orig) { hop_ref_orig_est_intra(this, pcCU, pcOrgYuv, pcPredYuv, pcResiYuv, pcRecoYuv, ruiDistC, bLumaOnly); return; }
  TComSlice* sl = pcCU->getSlice();
  if (!bLumaOnly || m_pcEncCfg->getRDpenalty() || !m_pcEncCfg->getUseRDOQ() || !m_pcEncCfg->getUseRDOQTS() || pcCU->getCUTransquantBypass(0) ||
      sl->getSPS()->getUsePCM()) {
    fprintf(stderr, "hop shim: estIntraPredQT is replaced for luma-only searches, RDOQ + RDOQTS, no RD penalty, lossy\n"); abort();
  }
  g_calls12[0]++;
  const UInt uiDepth = pcCU->getDepth(0);
  const UInt d0 = pcCU->getPartitionSize(0) == SIZE_2Nx2N ? 0 : 1, npu = pcCU->getNumPartitions(), q = pcCU->getTotalNumPart() >> 2;
  pcCU->setQPSubParts(sl->getPPS()->getUseDQP() ? pcCU->getQP(0) : sl->getSliceQp(), 0, uiDepth);
  hop_o_rqt_cfg cfg; hop_o_intra_syntax y; std::vector<uint8_t> avail((size_t)341 * HOP_O_AVAIL_PITCH, 0);
  for (UInt pu = 0; pu < npu; pu++) {
    std::vector<uint8_t> a1; intra_env(this, pcCU, d0, pu * q, cfg, y, a1);
    for (size_t i = 0; i < avail.size(); i++) avail[i] |= a1[i];
  }
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4), N = cu >> d0;
  hop_o_intra_search_in sin; memset(&sin, 0, sizeof(sin));
  std::vector<uint8_t> rough(4 * 68, 0);
  for (UInt pu = 0; pu < npu; pu++) {
    const UInt part = pu * q; UInt tp;
    TComDataCU* l = pcCU->getPULeft(tp, pcCU->getZorderIdxInCU() + part);
    sin.left_dir[pu] = l ? (l->isIntra(tp) ? l->getLumaIntraDir(tp) : DC_IDX) : DC_IDX;
    TComDataCU* a = pcCU->getPUAbove(tp, pcCU->getZorderIdxInCU() + part, true, true);
    sin.above_dir[pu] = a ? (a->isIntra(tp) ? a->getLumaIntraDir(tp) : DC_IDX) : DC_IDX;
    UInt lt, rt, lb; Bool fl[4 * MAX_NUM_SPU_W + 1]; memset(fl, 0, sizeof(fl));
    const int units = N / 4;
    pcCU->deriveLeftRightTopIdxAdi(lt, rt, part, d0); pcCU->deriveLeftBottomIdxAdi(lb, part, d0);
    TComPattern* pt = pcCU->getPattern();
    fl[units * 2] = pt->isAboveLeftAvailable(pcCU, lt);
    pt->isAboveAvailable(pcCU, lt, rt, fl + units * 2 + 1); pt->isAboveRightAvailable(pcCU, lt, rt, fl + units * 3 + 1);
    pt->isLeftAvailable(pcCU, lt, lb, fl + units * 2 - 1); pt->isBelowLeftAvailable(pcCU, lt, lb, fl + units - 1);
    for (int i = 0; i < 4 * units + 1; i++) rough[68 * pu + i] = fl[i] ? 1 : 0;
  }
  sin.rough_flags = &rough[0]; sin.sqrt_lambda = m_pcRdCost->getSqrtLambda(); sin.num_full_rd = g_aucIntraModeNumFast[pcCU->getIntraSizeIdx(0)];
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  std::vector<int16_t> planes[4];
  for (int l = 0; l < 4; l++) {
    st.coef[l][0] = m_ppcQTTempCoeffY[l]; st.coef[l][1] = m_ppcQTTempCoeffCb[l]; st.coef[l][2] = m_ppcQTTempCoeffCr[l];
    planes[l].assign(cu * cu, 0); st.resi[l][0] = &planes[l][0];
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int r = 0; r < cu; r++) memcpy(st.resi[l][0] + r * cu, t.getLumaAddr() + r * t.getStride(), cu * sizeof(Pel));
  }
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  TComPicYuv* recPic = pcCU->getPic()->getPicYuvRec();
  hop_o_intra_rqt_in in; memset(&in, 0, sizeof(in));
  in.org = pcOrgYuv->getLumaAddr(); in.org_stride = pcOrgYuv->getStride();
  in.rec = recPic->getLumaAddr(pcCU->getAddr(), pcCU->getZorderIdxInCU()); in.rec_stride = recPic->getStride();
  in.avail = &avail[0]; in.strong = sl->getSPS()->getUseStrongIntraSmoothing() ? 1 : 0; in.ts_fast = m_pcEncCfg->getUseTransformSkipFast() ? 1 : 0;
  TEncSbac* best = m_pppcRDSbacCoder[uiDepth][CI_CURR_BEST];
  hop_o_coder coder; coder_get(best, &coder);
  uint8_t cuctx[20] = { 0 }; { CuSets2 r = cu_sets2(best); uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  static FILE* f = NULL; static bool tried = false;                 // HOP_SHIM_TRACE_ISEARCH=<file>
  if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_ISEARCH"); if (pth && *pth) f = fopen(pth, "wb"); }
  const int W = 2 * cu + 1;
  const int px = (int)pcCU->getCUPelX(), py = (int)pcCU->getCUPelY(), pw = recPic->getWidth(), ph = recPic->getHeight();
  if (f) {
    const int32_t nd[4] = { in.ts_fast, in.strong, sin.num_full_rd, 0 };
    fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(nd, 4, 4, f); fwrite(sin.left_dir, 4, 4, f); fwrite(sin.above_dir, 4, 4, f); fwrite(&sin.sqrt_lambda, 8, 1, f);
    fwrite(&rough[0], 1, rough.size(), f); fwrite(&avail[0], 1, avail.size(), f);
    for (int r = 0; r < cu; r++) fwrite(in.org + r * in.org_stride, 2, cu, f);
    std::vector<int16_t> win((size_t)W * W, 0);
    for (int r = 0; r < W; r++) for (int c = 0; c < W; c++) {
      const int X = px - 1 + c, Y = py - 1 + r;
      if (X >= 0 && Y >= 0 && X < pw && Y < ph) win[(size_t)r * W + c] = in.rec[(ptrdiff_t)(r - 1) * in.rec_stride + (c - 1)];
    }
    fwrite(&win[0], 2, win.size(), f);
    fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
  }
  int bestDir[4] = { 0, 0, 0, 0 }, ncand[4] = { 0, 0, 0, 0 }; uint32_t distY = 0;
  std::vector<int16_t> reco((size_t)cu * cu, 0);
  hop_o_intra_luma_search(&cfg, &y, &in, &sin, &coder, cuctx, &st, bestDir, pcCU->getCoeffY(), &reco[0], &distY, ncand);
  if (f) {
    fwrite(bestDir, 4, 4, f); fwrite(ncand, 4, 4, f); fwrite(&distY, 4, 1, f);
    fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
    fwrite(pcCU->getCoeffY(), 4, cu * cu, f); fwrite(&reco[0], 2, cu * cu, f);
    for (int r = 0; r < cu; r++) fwrite(in.rec + (ptrdiff_t)r * in.rec_stride, 2, cu, f);
  }
  for (int l = 0; l < 4; l++) {
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int r = 0; r < cu; r++) memcpy(t.getLumaAddr() + r * t.getStride(), st.resi[l][0] + r * cu, cu * sizeof(Pel));
  }
  for (int r = 0; r < cu; r++) memcpy(pcRecoYuv->getLumaAddr() + r * pcRecoYuv->getStride(), &reco[(size_t)r * cu], cu * sizeof(Pel));
  memcpy(pcCU->m_puhTrIdx, st.tr_idx, parts); memcpy(pcCU->m_puhTransformSkip[0], st.tskip[0], parts);
  for (int p = 0; p < parts; p++) pcCU->m_puhCbf[0][p] = (UChar)(npu > 1 ? (st.cbf[0][p] & ~1) : st.cbf[0][p]);   // as the PU loop leaves them (:2662-2663) ...
  for (UInt pu = 0; pu < npu; pu++) { pcCU->setLumaIntraDirSubParts(bestDir[pu], pu * q, uiDepth + d0); pcCU->copyToPic(uiDepth, pu, d0); }
  memcpy(pcCU->m_puhCbf[0], st.cbf[0], parts);                                                                     // ... then the combined cbf of an NxN CU (:2667-2685)
  m_pcRDGoOnSbacCoder->load(best);
  ruiDistC = 0;
  pcCU->getTotalDistortion() = distY;
}

// ---- the chroma intra search of a CU: TEncSearch::estIntraPredChromaQT (TLibEncoder/TEncSearch.cpp:2720-2785) -> hop_o_intra_chroma_search ----
extern "C" void hop_ref_orig_est_chroma(TEncSearch*, TComDataCU*, TComYuv*, TComYuv*, TComYuv*, TComYuv*, UInt);
namespace { unsigned long g_calls13[1] = { 0 };
struct Report13 { ~Report13() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: chromaSearch %lu\n", g_calls13[0]); } } g_report13; }

Void TEncSearch::estIntraPredChromaQT(TComDataCU* pcCU, TComYuv* pcOrgYuv, TComYuv* pcPredYuv, TComYuv* pcResiYuv, TComYuv* pcRecoYuv, UInt uiPreCalcDistC)
{
  static const bool orig = hand_back("estIntraPredChromaQT");
  if (orig) { hop_ref_orig_est_chroma(this, pcCU, pcOrgYuv, pcPredYuv, pcResiYuv, pcRecoYuv, uiPreCalcDistC); return; }
  TComSlice* sl = pcCU->getSlice();
  if (m_pcEncCfg->getRDpenalty() || !m_pcEncCfg->getUseRDOQ() || !m_pcEncCfg->getUseRDOQTS() || pcCU->getCUTransquantBypass(0) || sl->getSPS()->getUsePCM()) {
    fprintf(stderr, "hop shim: estIntraPredChromaQT is replaced for RDOQ + RDOQTS, no RD penalty, lossy\n"); abort();
  }
  g_calls13[0]++;
  const UInt uiDepth = pcCU->getDepth(0);
  hop_o_rqt_cfg cfg; hop_o_intra_syntax y; std::vector<uint8_t> avail;
  intra_env(this, pcCU, 0, 0, cfg, y, avail);
  m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_CHROMA, sl->getSPS()->getQpBDOffsetC(), sl->getPPS()->getChromaCbQpOffset() + sl->getSliceQpDeltaCb()); cfg.qp[1] = m_pcTrQuant->m_cQP.m_iQP;
  m_pcTrQuant->setQPforQuant(pcCU->getQP(0), TEXT_CHROMA, sl->getSPS()->getQpBDOffsetC(), sl->getPPS()->getChromaCrQpOffset() + sl->getSliceQpDeltaCr()); cfg.qp[2] = m_pcTrQuant->m_cQP.m_iQP;
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4), half = cu / 2;
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  std::vector<int16_t> planes[4][3];
  for (int l = 0; l < 4; l++) {
    st.coef[l][0] = m_ppcQTTempCoeffY[l]; st.coef[l][1] = m_ppcQTTempCoeffCb[l]; st.coef[l][2] = m_ppcQTTempCoeffCr[l];
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int c = 1; c < 3; c++) {
      planes[l][c].assign(half * half, 0); st.resi[l][c] = &planes[l][c][0];
      const Pel* src = c == 1 ? t.getCbAddr() : t.getCrAddr();
      for (int r = 0; r < half; r++) memcpy(st.resi[l][c] + r * half, src + r * t.getCStride(), half * sizeof(Pel));
    }
  }
  memcpy(st.tr_idx, pcCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], pcCU->m_puhCbf[c], parts); memcpy(st.tskip[c], pcCU->m_puhTransformSkip[c], parts); }
  TComPicYuv* recPic = pcCU->getPic()->getPicYuvRec();
  hop_o_intra_chroma_in in; memset(&in, 0, sizeof(in));
  in.org_cb = pcOrgYuv->getCbAddr(); in.org_cr = pcOrgYuv->getCrAddr(); in.org_stride = pcOrgYuv->getCStride();
  in.rec_cb = recPic->getCbAddr(pcCU->getAddr(), pcCU->getZorderIdxInCU()); in.rec_cr = recPic->getCrAddr(pcCU->getAddr(), pcCU->getZorderIdxInCU()); in.rec_stride = recPic->getCStride();
  in.avail = &avail[0]; in.ts_fast = m_pcEncCfg->getUseTransformSkipFast() ? 1 : 0;
  TEncSbac* best = m_pppcRDSbacCoder[uiDepth][CI_CURR_BEST];
  hop_o_coder coder; coder_get(best, &coder);
  uint8_t cuctx[20] = { 0 }; { CuSets2 r = cu_sets2(best); uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  static FILE* f = NULL; static bool tried = false;                 // HOP_SHIM_TRACE_CSEARCH=<file>
  if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_CSEARCH"); if (pth && *pth) f = fopen(pth, "wb"); }
  const int W = cu + 1;                                             // chroma window: (2 * half + 1)^2 from (-1, -1)
  const int px = (int)pcCU->getCUPelX() / 2, py = (int)pcCU->getCUPelY() / 2, pw = recPic->getWidth() / 2, ph = recPic->getHeight() / 2;
  if (f) {
    const int32_t nd[4] = { in.ts_fast, 0, 0, 0 };
    fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(nd, 4, 4, f); fwrite(&avail[0], 1, avail.size(), f);
    for (int c = 0; c < 2; c++) for (int r = 0; r < half; r++) fwrite((c ? in.org_cr : in.org_cb) + r * in.org_stride, 2, half, f);
    for (int c = 0; c < 2; c++) {
      std::vector<int16_t> win((size_t)W * W, 0);
      const int16_t* rp = c ? in.rec_cr : in.rec_cb;
      for (int r = 0; r < W; r++) for (int cc = 0; cc < W; cc++) {
        const int X = px - 1 + cc, Y = py - 1 + r;
        if (X >= 0 && Y >= 0 && X < pw && Y < ph) win[(size_t)r * W + cc] = rp[(ptrdiff_t)(r - 1) * in.rec_stride + (cc - 1)];
      }
      fwrite(&win[0], 2, win.size(), f);
    }
    fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
    fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
  }
  int bestMode = 0; uint32_t bestDist = 0;
  std::vector<int16_t> rcb((size_t)half * half, 0), rcr((size_t)half * half, 0);
  hop_o_intra_chroma_search(&cfg, &y, &in, &coder, cuctx, &st, &bestMode, &bestDist, pcCU->getCoeffCb(), pcCU->getCoeffCr(), &rcb[0], &rcr[0]);
  if (f) {
    const int32_t o2[2] = { bestMode, (int32_t)bestDist }; fwrite(o2, 4, 2, f);
    fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
    fwrite(pcCU->getCoeffCb(), 4, half * half, f); fwrite(pcCU->getCoeffCr(), 4, half * half, f); fwrite(&rcb[0], 2, rcb.size(), f); fwrite(&rcr[0], 2, rcr.size(), f);
    for (int c = 0; c < 2; c++) for (int r = 0; r < half; r++) fwrite((c ? in.rec_cr : in.rec_cb) + (ptrdiff_t)r * in.rec_stride, 2, half, f);
  }
  for (int l = 0; l < 4; l++) {
    TComYuv& t = m_pcQTTempTComYuv[l];
    for (int c = 1; c < 3; c++) { Pel* dst = c == 1 ? t.getCbAddr() : t.getCrAddr(); for (int r = 0; r < half; r++) memcpy(dst + r * t.getCStride(), st.resi[l][c] + r * half, half * sizeof(Pel)); }
  }
  for (int r = 0; r < half; r++) { memcpy(pcRecoYuv->getCbAddr() + r * pcRecoYuv->getCStride(), &rcb[(size_t)r * half], half * sizeof(Pel));
                                   memcpy(pcRecoYuv->getCrAddr() + r * pcRecoYuv->getCStride(), &rcr[(size_t)r * half], half * sizeof(Pel)); }
  for (int c = 1; c < 3; c++) { memcpy(pcCU->m_puhCbf[c], st.cbf[c], parts); memcpy(pcCU->m_puhTransformSkip[c], st.tskip[c], parts); }
  pcCU->setChromIntraDirSubParts(bestMode, 0, uiDepth);
  pcCU->getTotalDistortion() += bestDist - uiPreCalcDistC;
  m_pcRDGoOnSbacCoder->load(best);
}

// ---- an intra CU candidate: TEncCu::xCheckRDCostIntra (TLibEncoder/TEncCu.cpp:1455-1507) - the searches as the reference calls them, the CU's bits -> hop_o_intra_cu_total_bits ----
extern "C" void hop_ref_orig_check_intra(TEncCu*, TComDataCU*&, TComDataCU*&, PartSize);
namespace { unsigned long g_calls14[1] = { 0 };
struct Report14 { ~Report14() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: intraCu %lu\n", g_calls14[0]); } } g_report14; }

Void TEncCu::xCheckRDCostIntra(TComDataCU*& rpcBestCU, TComDataCU*& rpcTempCU, PartSize eSize)
{
  static const bool orig = hand_back("xCheckRDCostIntra");
  if (orig) { hop_ref_orig_check_intra(this, rpcBestCU, rpcTempCU, eSize); return; }
  TComSlice* sl = rpcTempCU->getSlice();
  if (sl->getPPS()->getTransquantBypassEnableFlag() || sl->getSPS()->getUsePCM() || getdQPFlag()) {
    fprintf(stderr, "hop shim: xCheckRDCostIntra is replaced without transquant bypass, PCM, cu_qp_delta\n"); abort();
  }
  g_calls14[0]++;
  const UInt uiDepth = rpcTempCU->getDepth(0);
  rpcTempCU->setSkipFlagSubParts(false, 0, uiDepth);
  rpcTempCU->setPartSizeSubParts(eSize, 0, uiDepth);
  rpcTempCU->setPredModeSubParts(MODE_INTRA, 0, uiDepth);
  UInt uiPreCalcDistC = 0;
  m_pcPredSearch->estIntraPredQT(rpcTempCU, m_ppcOrigYuv[uiDepth], m_ppcPredYuvTemp[uiDepth], m_ppcResiYuvTemp[uiDepth], m_ppcRecoYuvTemp[uiDepth], uiPreCalcDistC, true);
  m_ppcRecoYuvTemp[uiDepth]->copyToPicLuma(rpcTempCU->getPic()->getPicYuvRec(), rpcTempCU->getAddr(), rpcTempCU->getZorderIdxInCU());
  m_pcPredSearch->estIntraPredChromaQT(rpcTempCU, m_ppcOrigYuv[uiDepth], m_ppcPredYuvTemp[uiDepth], m_ppcResiYuvTemp[uiDepth], m_ppcRecoYuvTemp[uiDepth], uiPreCalcDistC);
  // the bits of the finished CU
  hop_o_rqt_cfg cfg; hop_o_intra_syntax y; std::vector<uint8_t> avail;
  intra_env(m_pcPredSearch, rpcTempCU, 0, 0, cfg, y, avail);
  const int cu = 1 << cfg.log2_cu, parts = (cu / 4) * (cu / 4);
  hop_o_rqt_state st; memset(&st, 0, sizeof(st));
  memcpy(st.tr_idx, rpcTempCU->m_puhTrIdx, parts);
  for (int c = 0; c < 3; c++) { memcpy(st.cbf[c], rpcTempCU->m_puhCbf[c], parts); memcpy(st.tskip[c], rpcTempCU->m_puhTransformSkip[c], parts); }
  std::vector<int32_t> coef((size_t)cu * cu * 3 / 2);
  memcpy(&coef[0], rpcTempCU->getCoeffY(), (size_t)cu * cu * 4);
  memcpy(&coef[(size_t)cu * cu], rpcTempCU->getCoeffCb(), (size_t)cu * cu); memcpy(&coef[(size_t)cu * cu * 5 / 4], rpcTempCU->getCoeffCr(), (size_t)cu * cu);
  TEncSbac* sb = m_pcRDGoOnSbacCoder;
  hop_o_coder coder; coder_get(sb, &coder);
  uint8_t cuctx[20] = { 0 }; { CuSets2 r = cu_sets2(sb); uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  const hop_o_coder coder_in = coder; uint8_t cu_in[20]; memcpy(cu_in, cuctx, 20);
  const uint32_t bits = hop_o_intra_cu_total_bits(&cfg, &y, &st, &coef[0], &coder, cuctx);
  {                                                                 // HOP_SHIM_TRACE_INTRACU=<file>
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_INTRACU"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      fwrite(&cfg, sizeof(cfg), 1, f); fwrite(&y, sizeof(y), 1, f); fwrite(st.tr_idx, 1, 256, f); fwrite(st.cbf, 1, 768, f); fwrite(st.tskip, 1, 768, f);
      fwrite(&coef[0], 4, coef.size(), f); fwrite(&coder_in, sizeof(coder_in), 1, f); fwrite(cu_in, 1, 20, f); fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 20, f);
      const uint32_t o2[2] = { bits, (uint32_t)rpcTempCU->getTotalDistortion() }; fwrite(o2, 4, 2, f);
    }
  }
  coder_put(sb, &coder);
  { CuSets2 r = cu_sets2(sb); const uint8_t* d = cuctx; for (int i = 0; i < 11; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++; }
  m_pcRDGoOnSbacCoder->store(m_pppcRDSbacCoder[uiDepth][CI_TEMP_BEST]);
  rpcTempCU->getTotalBits() = bits;
  rpcTempCU->getTotalBins() = ((TEncBinCABAC*)((TEncSbac*)m_pcEntropyCoder->m_pcEntropyCoderIf)->getEncBinIf())->getBinsCoded();
  rpcTempCU->getTotalCost() = hop_o_calc_rd_cost(bits, rpcTempCU->getTotalDistortion(), m_pcRdCost->m_dLambda);
  xCheckDQP(rpcTempCU);
  xCheckBestMode(rpcBestCU, rpcTempCU, uiDepth);
}

// ---- an SS/GT CU without residual: the bSkipRes branch of TEncSearch::encodeResAndCalcRdInterCU (TLibEncoder/TEncSearch.cpp:6635-6668) -> hop_o_inter_cu_skip ----
// (every other call goes back to the reference's definition, whose quadtree and syntax-bit members are replaced above)
extern "C" void hop_ref_orig_encode_res(TEncSearch*, TComDataCU*, TComYuv*, TComYuv*, TComYuv*&, TComYuv*&, TComYuv*&, Bool);
namespace { unsigned long g_calls15[1] = { 0 };
struct Report15 { ~Report15() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: cuSkip %lu\n", g_calls15[0]); } } g_report15; }

Void TEncSearch::encodeResAndCalcRdInterCU(TComDataCU* pcCU, TComYuv* pcYuvOrg, TComYuv* pcYuvPred, TComYuv*& rpcYuvResi, TComYuv*& rpcYuvResiBest, TComYuv*& rpcYuvRec, Bool bSkipRes)
{
  if (!bSkipRes || pcCU->isIntra(0)) { hop_ref_orig_encode_res(this, pcCU, pcYuvOrg, pcYuvPred, rpcYuvResi, rpcYuvResiBest, rpcYuvRec, bSkipRes); return; }
  TComSlice* sl = pcCU->getSlice();
  if (sl->getPPS()->getTransquantBypassEnableFlag()) { fprintf(stderr, "hop shim: the skip variant is replaced without transquant bypass\n"); abort(); }
  g_calls15[0]++;
  const UInt depth = pcCU->getDepth(0);
  pcCU->setSkipFlagSubParts(true, 0, depth);
  rpcYuvResi->clear();
  pcYuvPred->copyToPartYuv(rpcYuvRec, 0);
  hop_o_rqt_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.log2_cu = g_aucConvertToBit[sl->getSPS()->getMaxCUWidth() >> depth] + 2; cfg.bit_depth_y = g_bitDepthY; cfg.bit_depth_c = g_bitDepthC;
  cfg.lambda_rd = m_pcRdCost->m_dLambda; cfg.dist_weight[0] = 1.0; cfg.dist_weight[1] = m_pcRdCost->m_cbDistortionWeight; cfg.dist_weight[2] = m_pcRdCost->m_crDistortionWeight;
  const int cu = 1 << cfg.log2_cu;
  std::vector<int16_t> pr[3], og[3];
  for (int c = 0; c < 3; c++) {
    const int w = c ? cu / 2 : cu; pr[c].resize((size_t)w * w); og[c].resize((size_t)w * w);
    const Pel* p = c == 0 ? rpcYuvRec->getLumaAddr() : c == 1 ? rpcYuvRec->getCbAddr() : rpcYuvRec->getCrAddr(); const int ps = c ? rpcYuvRec->getCStride() : rpcYuvRec->getStride();
    const Pel* o = c == 0 ? pcYuvOrg->getLumaAddr() : c == 1 ? pcYuvOrg->getCbAddr() : pcYuvOrg->getCrAddr(); const int os = c ? pcYuvOrg->getCStride() : pcYuvOrg->getStride();
    for (int r = 0; r < w; r++) { memcpy(&pr[c][(size_t)r * w], p + r * ps, w * sizeof(Pel)); memcpy(&og[c][(size_t)r * w], o + r * os, w * sizeof(Pel)); }
  }
  m_pcRDGoOnSbacCoder->load(m_pppcRDSbacCoder[depth][CI_CURR_BEST]);
  TEncSbac* sb = m_pcRDGoOnSbacCoder;
  hop_o_coder coder; coder_get(sb, &coder);
  uint8_t cuctx[16] = { 0 }; { CuSets r = cu_sets(sb); uint8_t* d = cuctx; for (int i = 0; i < 9; i++) for (int j = 0; j < r.n[i]; j++) *d++ = r.p[i][j].m_ucState; }
  const hop_o_coder coder_in = coder; uint8_t cu_in[16]; memcpy(cu_in, cuctx, 16);
  const int16_t* pp[3] = { &pr[0][0], &pr[1][0], &pr[2][0] }; const int16_t* oo[3] = { &og[0][0], &og[1][0], &og[2][0] };
  uint32_t d3[3]; double cost = 0;
  const int skipCtx = (int)pcCU->getCtxSkipFlag(0), mergeIdx = (int)pcCU->getMergeIndex(0), maxCand = (int)sl->getMaxNumMergeCand();
  const uint32_t bits = hop_o_inter_cu_skip(&cfg, skipCtx, mergeIdx, maxCand, pp, oo, &coder, cuctx, d3, &cost);
  {                                                                 // HOP_SHIM_TRACE_CUSKIP=<file>
    static FILE* f = NULL; static bool tried = false;
    if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_CUSKIP"); if (pth && *pth) f = fopen(pth, "wb"); }
    if (f) {
      const int32_t nd[4] = { skipCtx, mergeIdx, maxCand, 0 };
      fwrite(&cfg, sizeof(cfg), 1, f); fwrite(nd, 4, 4, f);
      for (int c = 0; c < 3; c++) fwrite(&pr[c][0], 2, pr[c].size(), f);
      for (int c = 0; c < 3; c++) fwrite(&og[c][0], 2, og[c].size(), f);
      fwrite(&coder_in, sizeof(coder_in), 1, f); fwrite(cu_in, 1, 16, f); fwrite(&coder, sizeof(coder), 1, f); fwrite(cuctx, 1, 16, f);
      const uint32_t o4[4] = { bits, d3[0], d3[1], d3[2] }; fwrite(o4, 4, 4, f); fwrite(&cost, 8, 1, f);
    }
  }
  coder_put(sb, &coder);
  { CuSets r = cu_sets(sb); const uint8_t* d = cuctx; for (int i = 0; i < 9; i++) for (int j = 0; j < r.n[i]; j++) r.p[i][j].m_ucState = *d++; }
  pcCU->getTotalBits() = bits;
  pcCU->getTotalDistortion() = d3[0] + d3[1] + d3[2];
  pcCU->getTotalCost() = cost;
  m_pcRDGoOnSbacCoder->store(m_pppcRDSbacCoder[depth][CI_TEMP_BEST]);
  pcCU->setCbfSubParts(0, 0, 0, 0, depth);
  pcCU->setTrIdxSubParts(0, 0, depth);
}

// ---- chroma intra prediction: TComPrediction::predIntraChromaAng (TLibCommon/TComPrediction.cpp:375-390) -> hop_o_intra_pred_chroma ----
namespace { unsigned long g_calls10[1] = { 0 };
struct Report10 { ~Report10() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop shim calls: chromaPred %lu\n", g_calls10[0]); } } g_report10; }
Void TComPrediction::predIntraChromaAng(Int* piSrc, UInt uiDirMode, Pel* piPred, UInt uiStride, Int iWidth, Int iHeight, Bool bAbove, Bool bLeft)
{
  if (iWidth != iHeight) { fprintf(stderr, "hop shim: predIntraChromaAng on a non-square block\n"); abort(); }
  g_calls10[0]++;
  const int N = iWidth, sw = 2 * N + 1;
  int L[4 * 64 + 1];
  L[2 * N] = piSrc[0];
  for (int i = 0; i < 2 * N; i++) { L[2 * N + 1 + i] = piSrc[1 + i]; L[2 * N - 1 - i] = piSrc[(1 + i) * sw]; }
  int16_t pred[64 * 64];
  hop_o_intra_pred_chroma(L, N, (int)uiDirMode, g_bitDepthC, pred);
  for (int r = 0; r < N; r++) memcpy(piPred + r * uiStride, pred + r * N, N * sizeof(Pel));
}

// ---- observers for the RD spine (row a0): nothing is replaced here, the calls are logged and handed to the reference's own definitions ----
// HOP_SHIM_TRACE_BEST=<file>: one text line per candidate that reaches TEncCu::xCheckBestMode (TLibEncoder/TEncCu.cpp:1557-1590), in the format of
// hopspine::CtuWorker::trace_candidate: depth x y size pred_mode part_size skip merge bits dist cost.
// HOP_SHIM_TRACE_CTU=<file>: after every TEncCu::compressCU (:246-264) the CTU's finished per-partition data, binary: address, cost, bits, distortion, then 256 records of
// depth, pred_mode, part_size, skip, merge_flag, merge_idx, gt_flag, luma_dir, chroma_dir, tr_idx, cbf[3], mv[2], gt[8] (int16 each).
extern "C" void hop_ref_orig_check_best(TEncCu*, TComDataCU*&, TComDataCU*&, UInt);
extern "C" void hop_ref_orig_compress_cu(TEncCu*, TComDataCU*&);
Void TEncCu::xCheckBestMode(TComDataCU*& rpcBestCU, TComDataCU*& rpcTempCU, UInt uiDepth)
{
  static FILE* f = NULL; static bool tried = false;
  if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_BEST"); if (pth && *pth) f = fopen(pth, "w"); }
  if (f) {
    TComDataCU* c = rpcTempCU;
    fprintf(f, "%d %d %d %d %d %d %d %d %u %u %.17g\n", (int)uiDepth, (int)c->getCUPelX(), (int)c->getCUPelY(), (int)c->getWidth(0), (int)c->getPredictionMode(0), (int)c->getPartitionSize(0),
            (int)c->getSkipFlag(0), (int)c->getMergeFlag(0), c->getTotalBits(), c->getTotalDistortion(), c->getTotalCost());
    fflush(f);
  }
  hop_ref_orig_check_best(this, rpcBestCU, rpcTempCU, uiDepth);
}
Void TEncCu::compressCU(TComDataCU*& rpcCU)
{
  hop_ref_orig_compress_cu(this, rpcCU);
  static FILE* f = NULL; static bool tried = false;
  if (!tried) { tried = true; const char* pth = getenv("HOP_SHIM_TRACE_CTU"); if (pth && *pth) f = fopen(pth, "wb"); }
  if (f) {
    TComDataCU* c = rpcCU;
    const int32_t addr = (int32_t)c->getAddr(); const double cost = c->getTotalCost(); const uint32_t bd[2] = { c->getTotalBits(), c->getTotalDistortion() };
    fwrite(&addr, 4, 1, f); fwrite(&cost, 8, 1, f); fwrite(bd, 4, 2, f);
    for (UInt i = 0; i < c->getTotalNumPart(); i++) {
      int16_t r[23];
      r[0] = c->getDepth(i); r[1] = c->getPredictionMode(i); r[2] = c->getPartitionSize(i); r[3] = c->getSkipFlag(i); r[4] = c->getMergeFlag(i); r[5] = c->getMergeIndex(i);
      r[6] = c->getGTFlag(i); r[7] = c->getLumaIntraDir(i); r[8] = c->getChromaIntraDir(i); r[9] = c->getTransformIdx(i);
      r[10] = c->getCbf(i, TEXT_LUMA); r[11] = c->getCbf(i, TEXT_CHROMA_U); r[12] = c->getCbf(i, TEXT_CHROMA_V);
      const TComMv m = c->getCUMvField(REF_PIC_LIST_0)->getMv(i); r[13] = m.getHor(); r[14] = m.getVer();
      const TComMv g0 = c->getCUGT0Field(REF_PIC_LIST_0)->getMv(i), g1 = c->getCUGT1Field(REF_PIC_LIST_0)->getMv(i), g2 = c->getCUGT2Field(REF_PIC_LIST_0)->getMv(i),
                   g3 = c->getCUGT3Field(REF_PIC_LIST_0)->getMv(i);
      r[15] = g0.getHor(); r[16] = g0.getVer(); r[17] = g1.getHor(); r[18] = g1.getVer(); r[19] = g2.getHor(); r[20] = g2.getVer(); r[21] = g3.getHor(); r[22] = g3.getVer();
      fwrite(r, 2, 23, f);
    }
    fflush(f);
  }
}
