// hop_hostlogic.cpp -- host-only logic of libhophip's C ABI (plain C++, no HIP): search-range derivation, MV / GT bit costs, the
// bookkeeping tail of xMotionEstimation, and the single-bin helpers of the counting coder the RD spine (hop_spine.cpp) needs for
// split flags.  Built into libhophip.so; also compiled into the CPU instantiation of the spine that the tests use (oracle/Makefile).
#include <stdint.h>
#include <string.h>
#include "../../include/hophip.h"

// TComRdCost::xGetComponentBits, TLibCommon/TComRdCost.cpp:270-284
static inline uint32_t hl_component_bits(int v) {
  uint32_t t = (v <= 0) ? (uint32_t)((-v << 1) + 1) : (uint32_t)(v << 1);
  uint32_t len = 1; while (t != 1) { t >>= 1; len += 2; } return len;
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// host logic
// ---------------------------------------------------------------------------------------------
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
// TComDataCU::clipMv, TLibCommon/TComDataCU.cpp:3492-3504 (g_uiMaxCUWidth/Height = 64)
static void clip_mv(int pic_w, int pic_h, int cu_x, int cu_y, int& hor, int& ver) {
  const int sh = 2, off = 8;
  int hmax = (pic_w + off - cu_x - 1) * 4, hmin = (-64 - off - cu_x + 1) * 4;
  int vmax = (pic_h + off - cu_y - 1) * 4, vmin = (-64 - off - cu_y + 1) * 4;
  (void)sh;
  hor = imin(hmax, imax(hmin, hor));
  ver = imin(vmax, imax(vmin, ver));
}

void hop_set_search_range(int pic_w, int pic_h, int cu_x, int cu_y, int cu_size, int ctu_addr, int frame_width_in_ctu,
                          int pred_x, int pred_y, int search_range, int off_x, int off_y, int first_row, int first_col, int out[6]) {
  // TEncSearch::xSetSearchRange(pcCU, cMvPred, iSrchRng, LT, RB), TEncSearch.cpp:6204-6220; TComMv stores Short
  int ph = pred_x, pv = pred_y;
  clip_mv(pic_w, pic_h, cu_x, cu_y, ph, pv);
  int lh = (int16_t)(ph - search_range * 4), lv = (int16_t)(pv - search_range * 4);
  int rh = (int16_t)(ph + search_range * 4), rv = (int16_t)(pv + search_range * 4);
  clip_mv(pic_w, pic_h, cu_x, cu_y, lh, lv);
  clip_mv(pic_w, pic_h, cu_x, cu_y, rh, rv);
  int left = lh >> 2, top = lv >> 2, right = rh >> 2, bottom = rv >> 2;
  // SS overload, TEncSearch.cpp:6224-6259
  if (first_col && first_row) {
    right = left + 1;
    top = bottom + 1;
  } else {
    bottom = (bottom > (-off_y - 4)) ? (-off_y - 4) : bottom;
    off_x = -off_x - cu_size - 4;
    off_y = -off_y - cu_size - 4;
    bottom = (first_col && (bottom > off_y)) ? off_y : bottom;
    right = (first_row && (right > off_x)) ? off_x : right;
    right = (!first_row && (ctu_addr < frame_width_in_ctu) && (right > (off_x + (cu_size << 1)))) ? (off_x + (cu_size << 1)) : right;
  }
  lh = (int16_t)(left * 4); lv = (int16_t)(top * 4); rh = (int16_t)(right * 4); rv = (int16_t)(bottom * 4);
  clip_mv(pic_w, pic_h, cu_x, cu_y, lh, lv);
  clip_mv(pic_w, pic_h, cu_x, cu_y, rh, rv);
  out[0] = lh >> 2; out[1] = rh >> 2; out[2] = lv >> 2; out[3] = rv >> 2; out[4] = off_x; out[5] = off_y;
}

uint32_t hop_component_bits(int v) { return hl_component_bits(v); }
uint32_t hop_bits_gt(const int v[8]) {   // IT_GT_AFFINE: corners 0..2 only, TComRdCost.h:205-215
  uint32_t b = 0; for (int i = 0; i < 6; i++) b += hl_component_bits(v[i]); return b;
}

void hop_me_finish(const hop_pu_job* job, const hop_pu_result* res, int stage, uint32_t bits_in,
                   int mv_qpel[2], uint32_t* bits_out, uint32_t* cost_out) {
  // TEncSearch.cpp:4654-4682 (fWeight = 1: uni-prediction; cost scale is 0 at this point)
  int mvx, mvy;
  if (stage >= HOP_STAGE_GT) { mvx = (res->mv_final[0] << 2) + (res->half_final[0] << 1) + res->qter_final[0]; mvy = (res->mv_final[1] << 2) + (res->half_final[1] << 1) + res->qter_final[1]; }
  else if (stage == HOP_STAGE_FRAC) { mvx = (res->mv_int[0] << 2) + (res->half[0] << 1) + res->qter[0]; mvy = (res->mv_int[1] << 2) + (res->half[1] << 1) + res->qter[1]; }
  else { mvx = res->mv_int[0] << 2; mvy = res->mv_int[1] << 2; }
  mv_qpel[0] = mvx; mv_qpel[1] = mvy;
  uint32_t mv_bits = hl_component_bits(mvx - job->pred_x) + hl_component_bits(mvy - job->pred_y);
  uint32_t bits = bits_in + mv_bits + 1;                       // + GT flag (:4669)
  if (stage >= HOP_STAGE_GT) {
    // :4673-4678 -- the chained '==' evaluates left to right on ints/bools; restated literally
    const int32_t* g = res->gt;
    int chain = (g[0] == g[1]);
    chain = (chain == g[2]); chain = (chain == g[3]); chain = (chain == g[4]);
    chain = (chain == g[5]); chain = (chain == g[6]); chain = (chain == g[7]);
    if (!chain) { int v[8]; for (int i = 0; i < 8; i++) v[i] = g[i]; bits += hop_bits_gt(v); }
  }
  uint32_t cost_mv = (job->lambda_cost * mv_bits) >> 16, cost_bits = (job->lambda_cost * bits) >> 16;
  uint32_t cost = (stage == HOP_STAGE_INT) ? res->sad + ((job->lambda_cost * mv_bits) >> 16) : res->cost;
  *bits_out = bits;
  *cost_out = (uint32_t)((double)cost - (double)cost_mv) + cost_bits;   // floor(fWeight*(cost - mvcost)) + cost(bits)
}

} // extern "C"

// ---- counting-coder helpers for single context-coded bins (split_cu_flag) ----
// state transition of ContextModel::update (TLibCommon/ContextModel.cpp:67-106) on m_ucState = state << 1 | MPS, and the fractional bit
// table ContextModel::m_entropyBits (:108-128, FAST_BIT_EST) indexed by m_ucState ^ bin
static const uint8_t hl_next_mps[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
static const uint8_t hl_next_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
static const int32_t hl_entropy_bits[128] = {
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };


// offsets of the context sets inside hop_cabac_ctx.state (the reference's set order, TEncSbac.cpp:76-88)
#define CX_QT_CBF 0
#define CX_TRANS_SUBDIV 8
#define CX_ROOT_CBF 11
#define CX_SIG_CG 12
#define CX_SIG 16
#define CX_LAST_X 58
#define CX_LAST_Y 88
#define CX_ONE 118
#define CX_ABS 142
#define CX_TS 148
#define CX_COUNT 150
#define h_entropy_bits hl_entropy_bits

// ---- host: initialisation values, rows = slice types B, P, I, ISS, PSS (TypeDef.h:418-427) ----
#define CNU 154
static const uint8_t h_init[5][CX_COUNT] = {
  /* B   */ { 153, 111, CNU, CNU, 149, 92, 167, 154,   224, 167, 122,   79,   121, 140, 61, 154,
              170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140,
              125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
              107, 167, 91, 107, 107, 167,   139, 139 },
  /* P   */ { 153, 111, CNU, CNU, 149, 107, 167, 154,   124, 138, 94,   79,   121, 140, 61, 154,
              155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
              107, 167, 91, 122, 107, 167,   139, 139 },
  /* I   */ { 111, 141, CNU, CNU, 94, 138, 182, 154,   153, 138, 138,   CNU,   91, 171, 134, 141,
              111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
              110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
              138, 153, 136, 167, 152, 152,   139, 139 },
  /* ISS */ { 153, 111, CNU, CNU, 149, 107, 167, 154,   124, 138, 94,   79,   121, 140, 61, 154,
              155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
              107, 167, 91, 122, 107, 167,   139, 139 },
  /* PSS */ { 153, 111, CNU, CNU, 149, 107, 167, 154,   124, 138, 94,   79,   121, 140, 61, 154,
              155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU,
              154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
              107, 167, 91, 122, 107, 167,   139, 139 },
};

static uint8_t h_ctx_init(int qp, int initValue) {          // ContextModel::init, ContextModel.cpp:56-65
  qp = qp < 0 ? 0 : qp > 51 ? 51 : qp;
  const int slope = (initValue >> 4) * 5 - 45, offset = ((initValue & 15) << 3) - 16;
  int initState = ((slope * qp) >> 4) + offset;
  initState = initState < 1 ? 1 : initState > 126 ? 126 : initState;
  const unsigned mp = (initState >= 64);
  return (uint8_t)(((mp ? (initState - 64) : (63 - initState)) << 1) + mp);
}
static int h_conv_to_bit(int w) { return w == 4 ? 0 : w == 8 ? 1 : w == 16 ? 2 : 3; }

extern "C" {

int hop_cabac_init(hop_cabac_ctx* ctx, int slice_type, int qp) {
  if (!ctx || slice_type < 0 || slice_type > 4) return HOP_ERR_ARG;
  for (int i = 0; i < CX_COUNT; i++) ctx->state[i] = h_ctx_init(qp, h_init[slice_type][i]);
  ctx->state[150] = ctx->state[151] = 0;
  return HOP_OK;
}

// CU-level sets of hop_cabac_cu_ctx: skip[3], merge_flag, merge_idx, part_size[4], pred_mode, mvd[2], mvp_idx, gt_flag, gt[2], intra_pred, chroma_pred[2];
// rows B, P, I, ISS, PSS (TLibCommon/ContextTables.h:140-310, 472-482)
static const uint8_t h_cu_init[5][20] = {
  { 197, 185, 201, 154, 137, 154, 139, 154, 154, 134, 169, 198, 168, 154, 169, 198, 183, 152, 139, CNU },
  { 197, 185, 201, 110, 122, 154, 139, 154, 154, 149, 140, 198, 168, 110, 140, 198, 154, 152, 139, CNU },
  { CNU, CNU, CNU, CNU, CNU, 184, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, 184,  63, 139, CNU },
  { 197, 185, 201, 110, 122, 154, 139, 154, 154, 149, 140, 198, 168, 110, 140, 198, 154, 152, 139, CNU },
  { 197, 185, 201, 110, 122, 154, 139, 154, 154, 149, 140, 198, 168, 110, 140, 198, 154, 152, 139, CNU } };
int hop_cabac_cu_init(hop_cabac_cu_ctx* ctx, int slice_type, int qp) {
  if (!ctx || slice_type < 0 || slice_type > 4) return HOP_ERR_ARG;
  for (int i = 0; i < 19; i++) ctx->state[i] = h_ctx_init(qp, h_cu_init[slice_type][i]);
  ctx->state[19] = 0;
  return HOP_OK;
}

int hop_cabac_est_bits(const hop_cabac_ctx* ctx, int width, int comp, hop_estbits* eb) {
  if (!ctx || !eb || (width != 4 && width != 8 && width != 16 && width != 32) || comp < 0 || comp > 2 || (comp && width == 32)) return HOP_ERR_ARG;
  static const uint8_t grp[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };
  const uint8_t* s = ctx->state;
  const int chroma = comp != 0;
  // estCBFBit reads 12 models from a set of 8 and 4 from the root set of 1: the sets behind them are read too (reproduced)
  for (int i = 0; i < 12; i++) { eb->blockCbpBits[i][0] = h_entropy_bits[s[CX_QT_CBF + i] ^ 0]; eb->blockCbpBits[i][1] = h_entropy_bits[s[CX_QT_CBF + i] ^ 1]; }
  for (int i = 0; i < 4; i++) { eb->blockRootCbpBits[i][0] = h_entropy_bits[s[CX_ROOT_CBF + i] ^ 0]; eb->blockRootCbpBits[i][1] = h_entropy_bits[s[CX_ROOT_CBF + i] ^ 1]; }
  for (int i = 0; i < 2; i++) for (int b = 0; b < 2; b++) eb->significantCoeffGroupBits[i][b] = h_entropy_bits[s[CX_SIG_CG + 2 * chroma + i] ^ b];
  int firstCtx = 1, numCtx = 8;
  if (width >= 16) { firstCtx = chroma ? 12 : 21; numCtx = chroma ? 3 : 6; }
  else if (width == 8) { firstCtx = 9; numCtx = chroma ? 3 : 12; }
  const int base = CX_SIG + (chroma ? 27 : 0);
  for (int b = 0; b < 2; b++) eb->significantBits[0][b] = h_entropy_bits[s[base] ^ b];
  for (int i = firstCtx; i < firstCtx + numCtx; i++) for (int b = 0; b < 2; b++) eb->significantBits[i][b] = h_entropy_bits[s[base + i] ^ b];
  const int cb = h_conv_to_bit(width);
  const int off = chroma ? 0 : (cb * 3 + ((cb + 1) >> 2)), sh = chroma ? cb : ((cb + 3) >> 2);
  const uint8_t* px = s + CX_LAST_X + 15 * chroma; const uint8_t* py = s + CX_LAST_Y + 15 * chroma;
  int bitsX = 0, bitsY = 0, c;
  for (c = 0; c < grp[width - 1]; c++) { const int o = off + (c >> sh); eb->lastXBits[c] = bitsX + h_entropy_bits[px[o] ^ 0]; bitsX += h_entropy_bits[px[o] ^ 1]; }
  eb->lastXBits[c] = bitsX;
  for (c = 0; c < grp[width - 1]; c++) { const int o = off + (c >> sh); eb->lastYBits[c] = bitsY + h_entropy_bits[py[o] ^ 0]; bitsY += h_entropy_bits[py[o] ^ 1]; }
  eb->lastYBits[c] = bitsY;
  const int no = chroma ? 8 : 16, na = chroma ? 2 : 4, oo = CX_ONE + (chroma ? 16 : 0), oa = CX_ABS + (chroma ? 4 : 0);
  for (int i = 0; i < no; i++) { eb->greaterOneBits[i][0] = h_entropy_bits[s[oo + i] ^ 0]; eb->greaterOneBits[i][1] = h_entropy_bits[s[oo + i] ^ 1]; }
  for (int i = 0; i < na; i++) { eb->levelAbsBits[i][0] = h_entropy_bits[s[oa + i] ^ 0]; eb->levelAbsBits[i][1] = h_entropy_bits[s[oa + i] ^ 1]; }
  return HOP_OK;
}

} // extern "C"


extern "C" {

uint32_t hop_cabac_bin_bits(uint8_t* state, int bin) {
  const uint8_t s = *state;
  const uint32_t bits = (uint32_t)hl_entropy_bits[s ^ (bin & 1)];
  *state = ((s & 1) == (bin & 1)) ? hl_next_mps[s] : hl_next_lps[s];
  return bits;
}
uint32_t hop_cabac_trm_bits(int bin) { return (uint32_t)hl_entropy_bits[126 ^ (bin & 1)]; }   // ContextModel::getEntropyBitsTrm, ContextModel.h:86

int hop_cabac_split_init(uint8_t split_ctx[3], int slice_type, int qp) {   // INIT_SPLIT_FLAG, TLibCommon/ContextTables.h:126-136 (rows B, P, I, ISS, PSS)
  static const uint8_t init[5][3] = { { 107, 139, 126 }, { 107, 139, 126 }, { 139, 141, 157 }, { 107, 139, 126 }, { 107, 139, 126 } };
  if (!split_ctx || slice_type < 0 || slice_type > 4) return HOP_ERR_ARG;
  qp = qp < 0 ? 0 : qp > 51 ? 51 : qp;
  for (int i = 0; i < 3; i++) {                                            // ContextModel::init, ContextModel.cpp:56-65
    const int v = init[slice_type][i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset;
    st = st < 1 ? 1 : st > 126 ? 126 : st;
    const unsigned mp = st >= 64;
    split_ctx[i] = (uint8_t)(((mp ? (st - 64) : (63 - st)) << 1) + mp);
  }
  return HOP_OK;
}

// sizeof of every struct of hophip.h as this library was compiled: a binding checks its own mirror against it before the first call (hophip.py does, at load)
int hop_sizeof(const char* name) {
  if (!name) return -1;
#define S(T) if (!strcmp(name, #T)) return (int)sizeof(T)
  S(hop_pu_job);
  S(hop_pu_result);
  S(hop_pred_job);
  S(hop_dist_job);
  S(hop_tu_job);
  S(hop_tu_result);
  S(hop_intra_job);
  S(hop_estbits);
  S(hop_rdoq_job);
  S(hop_cabac_ctx);
  S(hop_coeff_bits_job);
  S(hop_tu_rd_job);
  S(hop_tu_rd_result);
  S(hop_intra_modes_job);
  S(hop_intra_modes_result);
  S(hop_rqt_job);
  S(hop_rqt_result);
  S(hop_cu_final);
  S(hop_cabac_cu_ctx);
  S(hop_cu_syntax);
  S(hop_intra_cu_syntax);
  S(hop_intra_rqt_opt);
  S(hop_intra_search_job);
  S(hop_intra_search_result);
  S(hop_intra_chroma_result);
  S(hop_inter_class);
  S(hop_intra_class);
  S(hop_cu_part);
  S(hop_enc_params);
  S(hop_deblock_params);
  S(hop_sao_param);
  S(hop_sao_params);
#undef S
  return -1;
}

} // extern "C"
