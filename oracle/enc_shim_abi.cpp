// oracle/enc_shim_abi.cpp -- the reference-side binding of include/hophip.h, compiled against the reference's own headers.
//
// Members of the reference encoder re-defined as calls into the C ABI of libhophip.so, one job per call (the synchronous integration):
//   TEncSearch::xMotionEstimation       TLibEncoder/TEncSearch.cpp:4479-4683  -> hop_set_search_range + hop_me_search (SS, fractional, GT) + hop_me_finish
//   TComPrediction::xPredInterLumaBlk   TLibCommon/TComPrediction.cpp:639-720  \  hop_pred_inter (one PU, both called for the same PU by xPredInterUni :528-552;
//   TComPrediction::xPredInterChromaBlk TLibCommon/TComPrediction.cpp:1235-1347 /  the luma call computes, the chroma call hands the kept planes over)
//   TEncCu::xCopyYuv2SSRef              TLibEncoder/TEncCu.cpp:1677-1715       -> the reference's own definition (the host copy other members still read) + hop_ssref_commit_cus
// plus the residency the ABI asks for: one hop_ctx per encoder, the original uploaded when a new picture is met (TEncTop.cpp:363-368), the SS reference reset with it
// (TComSlice.cpp:241-255).
// oracle/Makefile.ref links this file with the reference's objects (the three symbols weakened with objcopy, nothing of the reference edited or copied) and with
// libhophip.so into oracle/_ref/TAppEncoderAbi: the LINK is the check -- every callee name, argument struct and ownership rule of include/hophip.h is met from the
// reference's types.  It cannot run in the build container (libhophip refuses to create a context without a gfx950 device) and the reference does not travel to the GPU box.
// With -DHOP_ABI_DRYRUN the hop_* calls are replaced by stand-ins inside this file that write every marshalled struct to HOP_ABI_DUMP and answer "no valid candidate":
// oracle/_ref/TAppEncoderAbiDry runs here, and tests/test_abi.py checks that no field of any marshalled struct is left at the poison value it was filled with before.
// This file never touches the CPU restatement (oracle/hop_oracle*.c).
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
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncSearch.h"
#include "TLibEncoder/TEncCu.h"
#undef private
#undef protected
#include "../include/hophip.h"

#ifdef HOP_ABI_DRYRUN
// stand-ins for the library (dry run): the marshalled inputs go to a dump, the answers are "nothing valid" / zero predictions
namespace {
FILE* dump() { static FILE* f = NULL; static bool tried = false; if (!tried) { tried = true; const char* p = getenv("HOP_ABI_DUMP"); if (p && *p) f = fopen(p, "wb"); } return f; }
void put(int kind, const void* p, size_t n) { FILE* f = dump(); if (!f) return; const int32_t h[2] = { kind, (int32_t)n }; fwrite(h, 4, 2, f); fwrite(p, 1, n, f); fflush(f); }
}
extern "C" {
int hop_ctx_create(hop_ctx** out, int w, int h, int bdy, int bdc, int dev) { const int32_t a[5] = { w, h, bdy, bdc, dev }; put(1, a, sizeof(a)); *out = (hop_ctx*)malloc(8); return HOP_OK; }
int hop_upload_orig(hop_ctx*, const int16_t* y, int sy, const int16_t*, const int16_t*, int sc) { const int32_t a[3] = { sy, sc, y ? y[0] : -1 }; put(2, a, sizeof(a)); return HOP_OK; }
int hop_ssref_reset(hop_ctx*) { return HOP_OK; }
int hop_ssref_commit_cus(hop_ctx*, int n, const int32_t* r, const int16_t*, const int16_t*, const int16_t*) { put(3, r, (size_t)n * 16); return HOP_OK; }
void hop_set_search_range(int, int, int, int, int, int, int, int, int, int, int, int, int, int, int out[6]) { out[0] = 0; out[1] = -1; out[2] = 0; out[3] = -1; out[4] = 0; out[5] = 0; }
int hop_me_search(hop_ctx*, int n, const hop_pu_job* j, hop_pu_result* r, int) { put(4, j, (size_t)n * sizeof(hop_pu_job)); memset(r, 0, (size_t)n * sizeof(hop_pu_result)); for (int i = 0; i < n; i++) { r[i].not_valid = 1; r[i].sad = 0xFFFFFFFFu; } return HOP_OK; }
void hop_me_finish(const hop_pu_job*, const hop_pu_result*, int, uint32_t b, int mv[2], uint32_t* bo, uint32_t* co) { mv[0] = mv[1] = 0; *bo = b; *co = 0xFFFFFFFFu; }
int hop_pred_inter(hop_ctx*, int n, const hop_pred_job* j, int16_t* y, int16_t* cb, int16_t* cr) { put(5, j, (size_t)n * sizeof(hop_pred_job)); memset(y, 0, (size_t)j[0].w * j[0].h * 2); memset(cb, 0, (size_t)j[0].w * j[0].h / 2); memset(cr, 0, (size_t)j[0].w * j[0].h / 2); return HOP_OK; }
const char* hop_last_error(const hop_ctx*) { return "dry run"; }
}
#endif

namespace {
const unsigned char POISON = 0xA5;                   // every marshalled struct starts out filled with it (the dry-run test looks for survivors)
hop_ctx* g_ctx = NULL; const TComPic* g_pic = NULL;

void fail(const char* what) { fprintf(stderr, "hop abi shim: %s failed: %s\n", what, hop_last_error(g_ctx)); exit(1); }

// residency: the context, and per picture the original + a sentinel-filled SS reference
void resident(TComDataCU* pcCU) {
  TComSlice* sl = pcCU->getSlice();
  if (!g_ctx && hop_ctx_create(&g_ctx, sl->getSPS()->getPicWidthInLumaSamples(), sl->getSPS()->getPicHeightInLumaSamples(), g_bitDepthY, g_bitDepthC, 0) != HOP_OK) fail("hop_ctx_create");
  if (g_pic != pcCU->getPic()) {
    g_pic = pcCU->getPic();
    TComPicYuv* org = pcCU->getPic()->getPicYuvOrg();
    if (hop_upload_orig(g_ctx, org->getLumaAddr(), org->getStride(), org->getCbAddr(), org->getCrAddr(), org->getCStride()) != HOP_OK) fail("hop_upload_orig");
    if (hop_ssref_reset(g_ctx) != HOP_OK) fail("hop_ssref_reset");
  }
}
int16_t g_cb[64 * 64 / 4], g_cr[64 * 64 / 4]; int g_kept_w = 0, g_kept_h = 0;    // the chroma planes of the PU the luma call predicted
}

Void TEncSearch::xMotionEstimation(TComDataCU* pcCU, TComYuv* pcYuvOrg, Int iPartIdx, RefPicList eRefPicList, TComMv* pcMvPred, Int iRefIdxPred, TComMv& rcMv, UInt& ruiBits, UInt& ruiCost,
                                   Bool& bNotValCU, Bool& bUseGT, TComMv& rcGT0, TComMv& rcGT1, TComMv& rcGT2, TComMv& rcGT3, Bool& gtFlag, Bool bBi)
{
  if (bBi || eRefPicList != REF_PIC_LIST_0 || !pcCU->getSlice()->isIntraSS()) { fprintf(stderr, "hop abi shim: xMotionEstimation is bound for the SS reference of an ISS slice\n"); exit(1); }
  resident(pcCU);
  UInt partAddr; Int w, h, offX, offY; Bool firstRow, firstCol;
  pcCU->getPartIndexAndSize(iPartIdx, partAddr, w, h);
  pcCU->getPartOffset(iPartIdx, partAddr, offX, offY, firstRow, firstCol);
  hop_pu_job j; memset(&j, POISON, sizeof(j));
  const UInt z = pcCU->getZorderIdxInCU() + partAddr;
  j.pu_x = pcCU->getPic()->getCU(pcCU->getAddr())->getCUPelX() + g_auiRasterToPelX[g_auiZscanToRaster[z]];
  j.pu_y = pcCU->getPic()->getCU(pcCU->getAddr())->getCUPelY() + g_auiRasterToPelY[g_auiZscanToRaster[z]];
  j.w = w; j.h = h;
  int rg[6];
  hop_set_search_range(pcCU->getSlice()->getSPS()->getPicWidthInLumaSamples(), pcCU->getSlice()->getSPS()->getPicHeightInLumaSamples(), pcCU->getCUPelX(), pcCU->getCUPelY(),
                       pcCU->getWidth(0), pcCU->getAddr(), pcCU->getPic()->getFrameWidthInCU(), pcMvPred->getHor(), pcMvPred->getVer(), m_aaiAdaptSR[eRefPicList][iRefIdxPred],
                       offX, offY, firstRow, firstCol, rg);
  j.rng_left = rg[0]; j.rng_right = rg[1]; j.rng_top = rg[2]; j.rng_bottom = rg[3]; j.off_x = rg[4]; j.off_y = rg[5];
  j.pred_x = pcMvPred->getHor(); j.pred_y = pcMvPred->getVer();
  m_pcRdCost->getMotionCost(1, 0);
  j.lambda_cost = m_pcRdCost->m_uiCost;
  const AMVPInfo* am = pcCU->getCUMvField(eRefPicList)->getAMVPInfo();
  j.n_amvp = am->iN > 2 ? 2 : am->iN;
  for (int k = 0; k < 2; k++) { j.amvp[2 * k] = k < j.n_amvp ? am->m_acMvCand[k].getHor() : 0; j.amvp[2 * k + 1] = k < j.n_amvp ? am->m_acMvCand[k].getVer() : 0; }
  j.flags = (m_pcEncCfg->getUseFastEnc() ? HOP_FLAG_FEN : 0) | (m_pcEncCfg->getUseHADME() ? HOP_FLAG_HADME : 0);
  hop_pu_result r;
  const int stage = bUseGT ? HOP_STAGE_GT : HOP_STAGE_FRAC;
  if (hop_me_search(g_ctx, 1, &j, &r, stage) != HOP_OK) fail("hop_me_search");
  if (r.not_valid) { bNotValCU = true; rcMv.set((Short)r.mv_int[0], (Short)r.mv_int[1]); ruiCost = r.sad; return; }     // :4603-4611
  int mv[2]; uint32_t bits = 0, cost = 0;
  hop_me_finish(&j, &r, stage, ruiBits, mv, &bits, &cost);
  rcMv.set((Short)mv[0], (Short)mv[1]); ruiBits = bits; ruiCost = cost;
  gtFlag = bUseGT && r.gt_flag;
  rcGT0.set((Short)(bUseGT ? r.gt[0] : 0), (Short)(bUseGT ? r.gt[1] : 0)); rcGT1.set((Short)(bUseGT ? r.gt[2] : 0), (Short)(bUseGT ? r.gt[3] : 0));
  rcGT2.set((Short)(bUseGT ? r.gt[4] : 0), (Short)(bUseGT ? r.gt[5] : 0)); rcGT3.set((Short)(bUseGT ? r.gt[6] : 0), (Short)(bUseGT ? r.gt[7] : 0));
}

Void TComPrediction::xPredInterLumaBlk(TComDataCU* cu, TComPicYuv* refPic, UInt partAddr, TComMv* mv, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                       Bool bUseGT, TComMv* mGT0, TComMv* mGT1, TComMv* mGT2, TComMv* mGT3)
{
  if (bi) { fprintf(stderr, "hop abi shim: bi-prediction is not on the bound path\n"); exit(1); }
  resident(cu);
  hop_pred_job j; memset(&j, POISON, sizeof(j));
  const UInt z = cu->getZorderIdxInCU() + partAddr;
  j.pu_x = cu->getPic()->getCU(cu->getAddr())->getCUPelX() + g_auiRasterToPelX[g_auiZscanToRaster[z]];
  j.pu_y = cu->getPic()->getCU(cu->getAddr())->getCUPelY() + g_auiRasterToPelY[g_auiZscanToRaster[z]];
  j.w = width; j.h = height; j.mv_x = mv->getHor(); j.mv_y = mv->getVer(); j.use_gt = bUseGT ? 1 : 0; j.dst_row_off = 0;
  const TComMv* g[4] = { mGT0, mGT1, mGT2, mGT3 };
  for (int k = 0; k < 4; k++) { j.gt[2 * k] = g[k]->getHor(); j.gt[2 * k + 1] = g[k]->getVer(); }
  std::vector<int16_t> y((size_t)width * height);
  if (hop_pred_inter(g_ctx, 1, &j, &y[0], g_cb, g_cr) != HOP_OK) fail("hop_pred_inter");
  g_kept_w = width; g_kept_h = height;
  Pel* dst = dstPic->getLumaAddr(partAddr); const int ds = dstPic->getStride();
  for (int r = 0; r < height; r++) memcpy(dst + r * ds, &y[(size_t)r * width], width * sizeof(Pel));
  (void)refPic;                                       // the SS reference is resident in the context
}

Void TComPrediction::xPredInterChromaBlk(TComDataCU* cu, TComPicYuv*, UInt partAddr, TComMv*, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                         Bool, TComMv*, TComMv*, TComMv*, TComMv*)
{
  if (bi || width != g_kept_w || height != g_kept_h) { fprintf(stderr, "hop abi shim: chroma planes are handed over after the luma call of the same PU\n"); exit(1); }
  Pel* dcb = dstPic->getCbAddr(partAddr); Pel* dcr = dstPic->getCrAddr(partAddr); const int ds = dstPic->getCStride();
  const int cw = width >> 1, ch = height >> 1;
  for (int r = 0; r < ch; r++) { memcpy(dcb + r * ds, &g_cb[r * cw], cw * sizeof(Pel)); memcpy(dcr + r * ds, &g_cr[r * cw], cw * sizeof(Pel)); }
  (void)cu;
}

extern "C" void hop_ref_orig_copy_ssref(TEncCu*, TComPic*, UInt, UInt, UInt, UInt, TComDataCU*, UInt, UInt);   // the reference's own definition (Makefile.ref)
Void TEncCu::xCopyYuv2SSRef(TComPic* rpcPic, UInt uiCUAddr, UInt uiAbsPartIdx, UInt uiDepth, UInt uiSrcDepth, TComDataCU* pcCU, UInt uiLPelX, UInt uiTPelY)
{
  hop_ref_orig_copy_ssref(this, rpcPic, uiCUAddr, uiAbsPartIdx, uiDepth, uiSrcDepth, pcCU, uiLPelX, uiTPelY);   // the host copy (other members of the reference still read it)
  resident(pcCU);
  // the blocks of this CU that lie inside the picture, from the host copy the reference has just updated
  TComPicYuv* rec = rpcPic->getPicYuvRec();
  const int picW = rec->getWidth(), picH = rec->getHeight();
  struct Walk { static void go(TComPicYuv* rec, int picW, int picH, int x, int y, int size) {
    if (x >= picW || y >= picH) return;
    if (x + size > picW || y + size > picH) { const int h = size >> 1; for (int q = 0; q < 4; q++) go(rec, picW, picH, x + (q & 1) * h, y + (q >> 1) * h, h); return; }
    int32_t r4[4]; memset(r4, POISON, sizeof(r4)); r4[0] = x; r4[1] = y; r4[2] = size; r4[3] = 0;
    std::vector<int16_t> py((size_t)size * size), pb((size_t)size * size / 4), pr((size_t)size * size / 4);
    for (int r = 0; r < size; r++) memcpy(&py[(size_t)r * size], rec->getLumaAddr() + (size_t)(y + r) * rec->getStride() + x, size * sizeof(Pel));
    for (int r = 0; r < size / 2; r++) { memcpy(&pb[(size_t)r * (size / 2)], rec->getCbAddr() + (size_t)(y / 2 + r) * rec->getCStride() + x / 2, (size / 2) * sizeof(Pel));
                                         memcpy(&pr[(size_t)r * (size / 2)], rec->getCrAddr() + (size_t)(y / 2 + r) * rec->getCStride() + x / 2, (size / 2) * sizeof(Pel)); }
    if (hop_ssref_commit_cus(g_ctx, 1, r4, &py[0], &pb[0], &pr[0]) != HOP_OK) fail("hop_ssref_commit_cus");
  } };
  Walk::go(rec, picW, picH, (int)uiLPelX, (int)uiTPelY, (int)(g_uiMaxCUWidth >> uiDepth));
}
