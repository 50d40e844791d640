// oracle/dec_shim_abi.cpp -- the reference-side binding of include/hophip.h for the DECODER, compiled against the reference's own headers.
//
// The GT / SS predictor is shared between encoder and decoder (TComPrediction::xPredInterUni under TDecCu::xReconInter, TLibDecoder/TDecCu.cpp:383-476).  Members of the
// reference decoder re-defined as calls into the C ABI of libhophip.so, one PU per call:
//   TComPrediction::xPredInterLumaBlk   TLibCommon/TComPrediction.cpp:639-720   \  hop_pred_inter (the luma call computes all three planes, the chroma call hands the kept
//   TComPrediction::xPredInterChromaBlk TLibCommon/TComPrediction.cpp:1235-1347 /  chroma planes over): SS prediction with quarter-sample vectors and the GT warp
//   TDecCu::xFindSSRef2Copy             TLibDecoder/TDecCu.cpp:459-471            -> the reference's own definition (the host copy the decoder keeps) + hop_ssref_commit_cus
// plus the residency: one hop_ctx per decoder, the SS reference back to the sentinel when a new picture is met (TComSlice.cpp:241-255).  No original picture exists on
// this side and none is needed.  oracle/Makefile.ref links this file with the reference's decoder objects (the three symbols weakened with objcopy) and libhophip.so into
// oracle/_ref/TAppDecoderAbi.  It runs on the GPU box (tests/test_gpu_encoder_pic.py): a stream is decoded with every SS / GT prediction coming from the device-resident
// reference picture, and the decoded pictures must be the encoder's reconstruction, byte for byte.  This file never touches the CPU restatement.
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
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibDecoder/TDecCu.h"
#undef private
#undef protected
#include "../include/hophip.h"

namespace {
struct DecBinding {
  hop_ctx* ctx; const TComPic* pic; int poc; unsigned long pictures, predictions, gt_predictions, commits;
  DecBinding() : ctx(NULL), pic(NULL), poc(-1), pictures(0), predictions(0), gt_predictions(0), commits(0) {}
  ~DecBinding() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop dec binding: pictures %lu predictions %lu gt %lu commits %lu\n", pictures, predictions, gt_predictions, commits); if (ctx) hop_ctx_destroy(ctx); }
  void fail(const char* what) { fprintf(stderr, "hop dec binding: %s failed: %s\n", what, hop_last_error(ctx)); exit(1); }
  void resident(TComDataCU* cu) {
    TComSlice* sl = cu->getSlice();
    if (!sl->isIntraSS() || g_bitDepthY != 8 || g_bitDepthC != 8) { fprintf(stderr, "hop dec binding: bound for the SS reference of 8-bit ISS pictures\n"); exit(1); }
    if (!ctx && hop_ctx_create(&ctx, sl->getSPS()->getPicWidthInLumaSamples(), sl->getSPS()->getPicHeightInLumaSamples(), g_bitDepthY, g_bitDepthC, 0) != HOP_OK) fail("hop_ctx_create");
    if (pic != cu->getPic() || poc != sl->getPOC()) {
      pic = cu->getPic(); poc = sl->getPOC(); pictures++;
      if (hop_ssref_reset(ctx) != HOP_OK) fail("hop_ssref_reset");
    }
  }
} g_d;
int16_t g_cb[64 * 64 / 4], g_cr[64 * 64 / 4]; int g_kept_w = 0, g_kept_h = 0;    // the chroma planes of the PU the luma call predicted
}

Void TComPrediction::xPredInterLumaBlk(TComDataCU* cu, TComPicYuv* refPic, UInt partAddr, TComMv* mv, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                       Bool bUseGT, TComMv* mGT0, TComMv* mGT1, TComMv* mGT2, TComMv* mGT3)
{
  if (bi) { fprintf(stderr, "hop dec binding: bi-prediction is not on the bound path\n"); exit(1); }
  g_d.resident(cu);
  hop_pred_job j; memset(&j, 0, sizeof(j));
  const UInt z = cu->getZorderIdxInCU() + partAddr;
  j.pu_x = cu->getPic()->getCU(cu->getAddr())->getCUPelX() + g_auiRasterToPelX[g_auiZscanToRaster[z]];
  j.pu_y = cu->getPic()->getCU(cu->getAddr())->getCUPelY() + g_auiRasterToPelY[g_auiZscanToRaster[z]];
  j.w = width; j.h = height; j.mv_x = mv->getHor(); j.mv_y = mv->getVer(); j.use_gt = bUseGT ? 1 : 0; j.dst_row_off = 0;
  const TComMv* g[4] = { mGT0, mGT1, mGT2, mGT3 };
  for (int k = 0; k < 4; k++) { j.gt[2 * k] = g[k]->getHor(); j.gt[2 * k + 1] = g[k]->getVer(); }
  std::vector<int16_t> y((size_t)width * height);
  if (hop_pred_inter(g_d.ctx, 1, &j, &y[0], g_cb, g_cr) != HOP_OK) g_d.fail("hop_pred_inter");
  g_kept_w = width; g_kept_h = height; g_d.predictions++; g_d.gt_predictions += bUseGT ? 1 : 0;
  Pel* dst = dstPic->getLumaAddr(partAddr); const int ds = dstPic->getStride();
  for (int r = 0; r < height; r++) memcpy(dst + r * ds, &y[(size_t)r * width], width * sizeof(Pel));
  (void)refPic;                                       // the SS reference is resident in the context
}

Void TComPrediction::xPredInterChromaBlk(TComDataCU* cu, TComPicYuv*, UInt partAddr, TComMv*, Int width, Int height, TComYuv*& dstPic, Bool bi,
                                         Bool, TComMv*, TComMv*, TComMv*, TComMv*)
{
  if (bi || width != g_kept_w || height != g_kept_h) { fprintf(stderr, "hop dec binding: chroma planes are handed over after the luma call of the same PU\n"); exit(1); }
  Pel* dcb = dstPic->getCbAddr(partAddr); Pel* dcr = dstPic->getCrAddr(partAddr); const int ds = dstPic->getCStride();
  const int cw = width >> 1, ch = height >> 1;
  for (int r = 0; r < ch; r++) { memcpy(dcb + r * ds, &g_cb[r * cw], cw * sizeof(Pel)); memcpy(dcr + r * ds, &g_cr[r * cw], cw * sizeof(Pel)); }
  (void)cu;
}

extern "C" void hop_ref_orig_find_ssref(TDecCu*, TComDataCU*&, UInt, UInt);      // the reference's own definition (Makefile.ref)
Void TDecCu::xFindSSRef2Copy(TComDataCU*& pcCU, UInt uiZorderIdx, UInt uiDepth)
{
  hop_ref_orig_find_ssref(this, pcCU, uiZorderIdx, uiDepth);                      // the decoder's own SS picture (other members still read it)
  g_d.resident(pcCU);
  // the CU's reconstruction (TDecCu::xCopyToPic has just put it into the picture) into the resident SS reference
  TComPicYuv* rec = pcCU->getPic()->getPicYuvRec();
  const int size = (int)(g_uiMaxCUWidth >> uiDepth), x = (int)pcCU->getCUPelX(), y = (int)pcCU->getCUPelY();
  int32_t r4[4] = { x, y, size, 0 };
  std::vector<int16_t> py((size_t)size * size), pb((size_t)size * size / 4), pr((size_t)size * size / 4);
  for (int r = 0; r < size; r++) memcpy(&py[(size_t)r * size], rec->getLumaAddr() + (size_t)(y + r) * rec->getStride() + x, size * sizeof(Pel));
  for (int r = 0; r < size / 2; r++) { memcpy(&pb[(size_t)r * (size / 2)], rec->getCbAddr() + (size_t)(y / 2 + r) * rec->getCStride() + x / 2, (size / 2) * sizeof(Pel));
                                       memcpy(&pr[(size_t)r * (size / 2)], rec->getCrAddr() + (size_t)(y / 2 + r) * rec->getCStride() + x / 2, (size / 2) * sizeof(Pel)); }
  if (hop_ssref_commit_cus(g_d.ctx, 1, r4, &py[0], &pb[0], &pr[0]) != HOP_OK) g_d.fail("hop_ssref_commit_cus");
  g_d.commits++;
}
