/* hop_oracle_cabac.c -- CPU restatement of the CABAC bit estimator for residual coding (first building block of
 * SURVEY 8(a) row a0 / 8(f)-1; it feeds rows a8, a8b and a11).  TEST INFRASTRUCTURE ONLY.
 *
 * Follows: ContextModel::init (TLibCommon/ContextModel.cpp:56-65), the state transition and fractional-bit tables
 * (:67-128, FAST_BIT_EST = 1, TypeDef.h:107), ContextModel3DBuffer::initBuffer (ContextModel3DBuffer.cpp:68-77),
 * the initialisation values of the residual-coding context sets (TLibCommon/ContextTables.h:340-546, the fork's five
 * slice types B, P, I, ISS, PSS), the counting bin coder (TLibEncoder/TEncBinCoderCABACCounter.cpp:72-108),
 * TEncSbac::estBit / estCBFBit / estSignificantCoeffGroupMapBit / estSignificantMapBit / estSignificantCoefficientsBit
 * (TLibEncoder/TEncSbac.cpp:2175-2370), codeCoeffNxN (:1829-2092), codeLastSignificantXY (:1772-1827),
 * codeTransformSkipFlags (:1608-1628), xWriteCoefRemainExGolomb (:381-402), codeQtCbf / codeQtRootCbf bit of a flag.
 * Pinned against the reference's own TEncSbac + TEncBinCABACCounter through oracle/ref_harness.cpp. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include "hop_oracle.h"

static const uint8_t next_state_mps[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
static const uint8_t next_state_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
static const int32_t entropy_bits[128] = {   /* ContextModel.cpp:104-113 (FAST_BIT_EST) */
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };

#define CNU 154
/* initialisation values, rows = slice types B, P, I, ISS, PSS (TypeDef.h:418-427); ContextTables.h:340-546 */
static const uint8_t init_qt_cbf[5][8] = {
  { 153, 111, CNU, CNU, 149, 92, 167, 154 }, { 153, 111, CNU, CNU, 149, 107, 167, 154 }, { 111, 141, CNU, CNU, 94, 138, 182, 154 },
  { 153, 111, CNU, CNU, 149, 107, 167, 154 }, { 153, 111, CNU, CNU, 149, 107, 167, 154 } };
static const uint8_t init_qt_root_cbf[5][1] = { { 79 }, { 79 }, { CNU }, { 79 }, { 79 } };
static const uint8_t init_last[5][30] = {
  { 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU },
  { 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU },
  { 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU },
  { 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU },
  { 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU, CNU } };
static const uint8_t init_sig_cg[5][4] = { { 121, 140, 61, 154 }, { 121, 140, 61, 154 }, { 91, 171, 134, 141 }, { 121, 140, 61, 154 }, { 121, 140, 61, 154 } };
static const uint8_t init_sig[5][42] = {
  { 170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140 },
  { 155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140 },
  { 111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111 },
  { 155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140 },
  { 155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140 } };
static const uint8_t init_one[5][24] = {
  { 154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182 },
  { 154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182 },
  { 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197 },
  { 154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182 },
  { 154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182 } };
static const uint8_t init_abs[5][6] = { { 107, 167, 91, 107, 107, 167 }, { 107, 167, 91, 122, 107, 167 }, { 138, 153, 136, 167, 152, 152 },
                                        { 107, 167, 91, 122, 107, 167 }, { 107, 167, 91, 122, 107, 167 } };
static const uint8_t init_trans_subdiv[5][3] = { { 224, 167, 122 }, { 124, 138, 94 }, { 153, 138, 138 }, { 124, 138, 94 }, { 124, 138, 94 } };
static const uint8_t init_ts[5][2] = { { 139, 139 }, { 139, 139 }, { 139, 139 }, { 139, 139 }, { 139, 139 } };

/* ContextModel::init, ContextModel.cpp:56-65 */
uint8_t hop_o_ctx_init(int qp, int initValue)
{
  qp = qp < 0 ? 0 : qp > 51 ? 51 : qp;
  const int slope = (initValue >> 4) * 5 - 45, offset = ((initValue & 15) << 3) - 16;
  int initState = ((slope * qp) >> 4) + offset;
  initState = initState < 1 ? 1 : initState > 126 ? 126 : initState;
  const unsigned mp = (initState >= 64);
  return (uint8_t)(((mp ? (initState - 64) : (63 - initState)) << 1) + mp);
}
int32_t hop_o_ctx_bits(uint8_t state, int bin) { return entropy_bits[state ^ bin]; }
uint8_t hop_o_ctx_next(uint8_t state, int bin) { return ((state & 1) == bin) ? next_state_mps[state] : next_state_lps[state]; }

/* TEncSbac::resetEntropy for the residual-coding sets (TEncSbac.cpp:136-148); slice_type 0 B, 1 P, 2 I, 3 ISS, 4 PSS */
int hop_o_cabac_init(hop_o_cabac_ctx* c, int slice_type, int qp)
{
  if (slice_type < 0 || slice_type > 4) return -1;
  for (int i = 0; i < 8; i++) c->qt_cbf[i] = hop_o_ctx_init(qp, init_qt_cbf[slice_type][i]);
  c->qt_root_cbf[0] = hop_o_ctx_init(qp, init_qt_root_cbf[slice_type][0]);
  for (int i = 0; i < 3; i++) c->trans_subdiv[i] = hop_o_ctx_init(qp, init_trans_subdiv[slice_type][i]);
  for (int i = 0; i < 4; i++) c->sig_cg[i] = hop_o_ctx_init(qp, init_sig_cg[slice_type][i]);
  for (int i = 0; i < 42; i++) c->sig[i] = hop_o_ctx_init(qp, init_sig[slice_type][i]);
  for (int i = 0; i < 30; i++) { c->last_x[i] = hop_o_ctx_init(qp, init_last[slice_type][i]); c->last_y[i] = c->last_x[i]; }
  for (int i = 0; i < 24; i++) c->one[i] = hop_o_ctx_init(qp, init_one[slice_type][i]);
  for (int i = 0; i < 6; i++) c->abs[i] = hop_o_ctx_init(qp, init_abs[slice_type][i]);
  for (int i = 0; i < 2; i++) c->ts[i] = hop_o_ctx_init(qp, init_ts[slice_type][i]);
  return 0;
}

static const uint8_t grp_idx[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };
static const uint8_t min_in_grp[10] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 };
static int conv_to_bit(int w) { return w == 4 ? 0 : w == 8 ? 1 : w == 16 ? 2 : w == 32 ? 3 : 4; }   /* g_aucConvertToBit */

/* TEncSbac::estBit, TEncSbac.cpp:2175-2370; comp 0 = TEXT_LUMA, else chroma.  Only the entries the reference writes for
 * this (width, component) are written; the others keep their value (the reference's struct is persistent too). */
void hop_o_cabac_est_bits(const hop_o_cabac_ctx* c, int width, int comp, hop_o_estbits* eb)
{
  const int height = width, chroma = comp != 0;
  /* estCBFBit reads 3*NUM_QT_CBF_CTX = 12 models from a set of 8 and 4 from the root set of 1: it runs on into the sets
   * laid out behind them (TEncSbac.cpp:76-79: qt_cbf[8], trans_subdiv[3], root_cbf[1], sig_cg[4]) -- reproduced */
  {
    const uint8_t run[16] = { c->qt_cbf[0], c->qt_cbf[1], c->qt_cbf[2], c->qt_cbf[3], c->qt_cbf[4], c->qt_cbf[5], c->qt_cbf[6], c->qt_cbf[7],
                              c->trans_subdiv[0], c->trans_subdiv[1], c->trans_subdiv[2], c->qt_root_cbf[0], c->sig_cg[0], c->sig_cg[1], c->sig_cg[2], c->sig_cg[3] };
    for (int i = 0; i < 12; i++) { eb->blockCbpBits[i][0] = entropy_bits[run[i] ^ 0]; eb->blockCbpBits[i][1] = entropy_bits[run[i] ^ 1]; }
    for (int i = 0; i < 4; i++) { eb->blockRootCbpBits[i][0] = entropy_bits[run[11 + i] ^ 0]; eb->blockRootCbpBits[i][1] = entropy_bits[run[11 + i] ^ 1]; }
  }
  for (int i = 0; i < 2; i++) for (int b = 0; b < 2; b++) eb->significantCoeffGroupBits[i][b] = entropy_bits[c->sig_cg[2 * chroma + i] ^ b];
  int firstCtx = 1, numCtx = 8;
  if (width >= 16) { firstCtx = chroma ? 12 : 21; numCtx = chroma ? 3 : 6; }
  else if (width == 8) { firstCtx = 9; numCtx = chroma ? 3 : 12; }
  const int base = chroma ? 27 : 0;                              /* NUM_SIG_FLAG_CTX_LUMA */
  for (int b = 0; b < 2; b++) eb->significantBits[0][b] = entropy_bits[c->sig[base] ^ b];
  for (int i = firstCtx; i < firstCtx + numCtx; i++) for (int b = 0; b < 2; b++) eb->significantBits[i][b] = entropy_bits[c->sig[base + i] ^ b];
  int bitsX = 0, bitsY = 0;
  const int cb = conv_to_bit(width);
  const int offX = chroma ? 0 : (cb * 3 + ((cb + 1) >> 2)), offY = offX;
  const int shX = chroma ? cb : ((cb + 3) >> 2), shY = shX;
  const uint8_t* px = c->last_x + 15 * chroma; const uint8_t* py = c->last_y + 15 * chroma;
  int ctx;
  for (ctx = 0; ctx < grp_idx[width - 1]; ctx++) {
    const int o = offX + (ctx >> shX);
    eb->lastXBits[ctx] = bitsX + entropy_bits[px[o] ^ 0];
    bitsX += entropy_bits[px[o] ^ 1];
  }
  eb->lastXBits[ctx] = bitsX;
  for (ctx = 0; ctx < grp_idx[height - 1]; ctx++) {
    const int o = offY + (ctx >> shY);
    eb->lastYBits[ctx] = bitsY + entropy_bits[py[o] ^ 0];
    bitsY += entropy_bits[py[o] ^ 1];
  }
  eb->lastYBits[ctx] = bitsY;
  /* estSignificantCoefficientsBit, :2340-2370 */
  if (!chroma) {
    for (int i = 0; i < 16; i++) { eb->greaterOneBits[i][0] = entropy_bits[c->one[i] ^ 0]; eb->greaterOneBits[i][1] = entropy_bits[c->one[i] ^ 1]; }
    for (int i = 0; i < 4; i++) { eb->levelAbsBits[i][0] = entropy_bits[c->abs[i] ^ 0]; eb->levelAbsBits[i][1] = entropy_bits[c->abs[i] ^ 1]; }
  } else {
    for (int i = 0; i < 8; i++) { eb->greaterOneBits[i][0] = entropy_bits[c->one[16 + i] ^ 0]; eb->greaterOneBits[i][1] = entropy_bits[c->one[16 + i] ^ 1]; }
    for (int i = 0; i < 2; i++) { eb->levelAbsBits[i][0] = entropy_bits[c->abs[4 + i] ^ 0]; eb->levelAbsBits[i][1] = entropy_bits[c->abs[4 + i] ^ 1]; }
  }
}

#define BIN(ctxp, b) do { frac += (uint64_t)entropy_bits[*(ctxp) ^ (b)]; *(ctxp) = hop_o_ctx_next(*(ctxp), (b)); } while (0)

/* TEncSbac::codeCoeffNxN with the counting bin coder: fractional bits (15 fractional binary places) of coding the levels of
 * one TU, contexts updated in place.  coef raster N x N; use_ts: PPS transform_skip_enabled; ts_flag: the TU's flag. */
uint64_t hop_o_cabac_coeff_bits(hop_o_cabac_ctx* c, const int32_t* coef, int log2_size, int comp, int scan_idx, int sign_hide,
                                int use_ts, int ts_flag)
{
  const int width = 1 << log2_size, n = width * width, chroma = comp != 0;
  uint64_t frac = 0;
  int numSig = 0;
  for (int i = 0; i < n; i++) numSig += coef[i] != 0;
  if (numSig == 0) return 0;
  if (use_ts && width == 4) BIN(&c->ts[chroma], ts_flag ? 1 : 0);
  const uint32_t* scan = hop_o_scan(scan_idx, log2_size);
  const uint32_t* scanCG = hop_o_scan_cg(scan_idx, log2_size);
  uint32_t cgFlag[64];
  const int numBlkSide = width >> 2;
  memset(cgFlag, 0, sizeof(cgFlag));
  int scanPosLast = -1, posLast;
  do {
    posLast = (int)scan[++scanPosLast];
    const int py = posLast >> log2_size, px = posLast - (py << log2_size);
    if (coef[posLast]) cgFlag[numBlkSide * (py >> 2) + (px >> 2)] = 1;
    numSig -= (coef[posLast] != 0);
  } while (numSig > 0);
  /* codeLastSignificantXY */
  {
    int posY = posLast >> log2_size, posX = posLast - (posY << log2_size);
    if (scan_idx == 2) { int t = posX; posX = posY; posY = t; }
    uint8_t* pX = c->last_x + 15 * chroma; uint8_t* pY = c->last_y + 15 * chroma;
    const int gX = grp_idx[posX], gY = grp_idx[posY];
    const int cb = conv_to_bit(width);
    const int offX = chroma ? 0 : (cb * 3 + ((cb + 1) >> 2)), shX = chroma ? cb : ((cb + 3) >> 2);
    int k;
    for (k = 0; k < gX; k++) BIN(pX + offX + (k >> shX), 1);
    if (gX < grp_idx[width - 1]) BIN(pX + offX + (k >> shX), 0);
    for (k = 0; k < gY; k++) BIN(pY + offX + (k >> shX), 1);
    if (gY < grp_idx[width - 1]) BIN(pY + offX + (k >> shX), 0);
    if (gX > 3) frac += 32768ull * (uint64_t)((gX - 2) >> 1);
    if (gY > 3) frac += 32768ull * (uint64_t)((gY - 2) >> 1);
    (void)min_in_grp;
  }
  uint8_t* baseCG = c->sig_cg + 2 * chroma;
  uint8_t* baseSig = c->sig + (chroma ? 27 : 0);
  const int lastScanSet = scanPosLast >> 4;
  unsigned c1 = 1, goRice = 0;
  int scanPosSig = scanPosLast;
  for (int subSet = lastScanSet; subSet >= 0; subSet--) {
    int numNonZero = 0;
    const int subPos = subSet << 4;
    goRice = 0;
    int absCoeff[16];
    unsigned coeffSigns = 0;
    int lastNZ = -1, firstNZ = 16;
    if (scanPosSig == scanPosLast) {
      absCoeff[0] = abs(coef[posLast]); coeffSigns = (coef[posLast] < 0); numNonZero = 1;
      lastNZ = scanPosSig; firstNZ = scanPosSig; scanPosSig--;
    }
    const int cgBlkPos = (int)scanCG[subSet], cgPosY = cgBlkPos / numBlkSide, cgPosX = cgBlkPos - cgPosY * numBlkSide;
    if (subSet == lastScanSet || subSet == 0) cgFlag[cgBlkPos] = 1;
    else {
      unsigned r = 0, l = 0;
      if (cgPosX < numBlkSide - 1) r = (cgFlag[cgPosY * numBlkSide + cgPosX + 1] != 0);
      if (cgPosY < numBlkSide - 1) l = (cgFlag[(cgPosY + 1) * numBlkSide + cgPosX] != 0);
      BIN(baseCG + (r || l), cgFlag[cgBlkPos] != 0);
    }
    if (cgFlag[cgBlkPos]) {
      int patternSigCtx;
      if (width == 4) patternSigCtx = -1;
      else {
        unsigned r = 0, l = 0;
        if (cgPosX < numBlkSide - 1) r = (cgFlag[cgPosY * numBlkSide + cgPosX + 1] != 0);
        if (cgPosY < numBlkSide - 1) l = (cgFlag[(cgPosY + 1) * numBlkSide + cgPosX] != 0);
        patternSigCtx = (int)(r + (l << 1));
      }
      for (; scanPosSig >= subPos; scanPosSig--) {
        const int blkPos = (int)scan[scanPosSig], posY = blkPos >> log2_size, posX = blkPos - (posY << log2_size);
        const int sig = (coef[blkPos] != 0);
        if (scanPosSig > subPos || subSet == 0 || numNonZero) {
          /* getSigCtxInc, TComTrQuant.cpp:2038-2092 */
          static const int ctxIndMap[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
          int ctxSig;
          if (posX + posY == 0) ctxSig = 0;
          else if (log2_size == 2) ctxSig = ctxIndMap[4 * posY + posX];
          else {
            const int offset = log2_size == 3 ? (scan_idx == 0 ? 9 : 15) : (!chroma ? 21 : 12);
            const int xs = posX & 3, ys = posY & 3;
            int cnt;
            if (patternSigCtx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
            else if (patternSigCtx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
            else if (patternSigCtx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
            else cnt = 2;
            ctxSig = ((!chroma && ((posX >> 2) + (posY >> 2)) > 0) ? 3 : 0) + offset + cnt;
          }
          BIN(baseSig + ctxSig, sig);
        }
        if (sig) {
          absCoeff[numNonZero] = abs(coef[blkPos]);
          coeffSigns = 2 * coeffSigns + (coef[blkPos] < 0);
          numNonZero++;
          if (lastNZ == -1) lastNZ = scanPosSig;
          firstNZ = scanPosSig;
        }
      }
    } else {
      scanPosSig = subPos - 1;
    }
    if (numNonZero > 0) {
      const int signHidden = (lastNZ - firstNZ >= 4);
      unsigned ctxSet = (subSet > 0 && !chroma) ? 2 : 0;
      if (c1 == 0) ctxSet++;
      c1 = 1;
      uint8_t* baseOne = c->one + (chroma ? 16 : 0) + 4 * ctxSet;
      const int numC1 = numNonZero < 8 ? numNonZero : 8;
      int firstC2 = -1;
      for (int idx = 0; idx < numC1; idx++) {
        const int sym = absCoeff[idx] > 1;
        BIN(baseOne + c1, sym);
        if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = idx; }
        else if ((c1 < 3) && (c1 > 0)) c1++;
      }
      if (c1 == 0) {
        uint8_t* baseAbs = c->abs + (chroma ? 4 : 0) + ctxSet;
        if (firstC2 != -1) BIN(baseAbs, absCoeff[firstC2] > 2);
      }
      if (sign_hide && signHidden) frac += 32768ull * (uint64_t)(numNonZero - 1);
      else frac += 32768ull * (uint64_t)numNonZero;
      int firstCoeff2 = 1;
      if (c1 == 0 || numNonZero > 8) {
        for (int idx = 0; idx < numNonZero; idx++) {
          const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (absCoeff[idx] >= baseLevel) {
            /* xWriteCoefRemainExGolomb: only the number of equiprobable bins matters for the counter */
            int codeNumber = absCoeff[idx] - baseLevel; unsigned length;
            if (codeNumber < (3 << goRice)) { length = (unsigned)codeNumber >> goRice; frac += 32768ull * (uint64_t)(length + 1 + goRice); }
            else {
              length = goRice; codeNumber -= (3 << goRice);
              while (codeNumber >= (1 << length)) codeNumber -= (1 << (length++));
              frac += 32768ull * (uint64_t)(3 + length + 1 - goRice + length);
            }
            if (absCoeff[idx] > 3 * (1 << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
          }
          if (absCoeff[idx] >= 2) firstCoeff2 = 0;
        }
      }
    }
  }
  return frac;
}

/* bits of one coded_block_flag (TEncSbac::codeQtCbf :1590-1603: context set by component class, index getCtxQtCbf) and of
 * the root flag (:1723-1727); contexts updated */
uint64_t hop_o_cabac_cbf_bits(hop_o_cabac_ctx* c, int comp, int tr_depth, int cbf)
{
  uint64_t frac = 0;
  const int chroma = comp != 0, ctx = chroma ? tr_depth : (tr_depth == 0 ? 1 : 0);
  BIN(&c->qt_cbf[4 * chroma + ctx], cbf ? 1 : 0);
  return frac;
}
/* TEncSbac::codeTransformSubdivFlag (TEncSbac.cpp:756-759): context = 5 - log2 of the transform size */
uint64_t hop_o_cabac_subdiv_bits(hop_o_cabac_ctx* c, int ctx, int flag)
{
  uint64_t frac = 0;
  BIN(&c->trans_subdiv[ctx], flag ? 1 : 0);
  return frac;
}
uint64_t hop_o_cabac_root_cbf_bits(hop_o_cabac_ctx* c, int cbf)
{
  uint64_t frac = 0;
  BIN(&c->qt_root_cbf[0], cbf ? 1 : 0);
  return frac;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Row a8b, leaf step: the evaluation of ONE component TU inside TEncSearch::xEstimateResidualQT
 * (TLibEncoder/TEncSearch.cpp:6896-7200, the default transform; the 4x4 transform-skip retry :7210-7440 and the split
 * recursion are the caller's): estBit from the snapshot (:6901-6904), transformNxN = xT + xRateDistOptQuant (:6912), the bits
 * of cbf flag + levels from the snapshot (:6959-6962, integer bits = (fraction left in the coder + counted) >> 15,
 * TEncBinCoderCABACCounter.cpp:60-63 after TEncBinCABAC::resetBits), zero-residual distortion (:6984), inverse path and
 * its distortion (:6998-7001), the cbf-zero decision on calcRdCost values (:7008-7032; TComRdCost.cpp:59-111).
 * resi: residual block (Pel, stride N).  frac_left: m_fracBits & 32767 of the snapshot.  dist_weight: 1.0 for luma, the
 * chroma weight of getDistPart (TComRdCost.cpp:493-497) otherwise.  out[8]: abs_sum, cbf, dist, zero_dist, nonzero_dist,
 * single_bits, null_bits, -; *cost = the cost of the choice. levels: final levels. */
static double calc_rd_cost(uint32_t bits, uint32_t dist, double lambda)
{
  double c = ((double)dist + (double)((int)(bits * lambda + .5)));
  return (double)(uint32_t)floor(c);
}
double hop_o_calc_rd_cost(uint32_t bits, uint32_t dist, double lambda) { return calc_rd_cost(bits, dist, lambda); }
static uint32_t weighted(uint32_t sse, int comp, double w) { return comp ? (uint32_t)(int)(w * sse) : sse; }

int hop_o_tu_rd(const int16_t* resi, int log2_size, int comp, int qp_scaled, int bit_depth, int tr_depth, int sign_hide, int use_ts,
                double lambda_rdoq, double lambda_rd, double dist_weight, const hop_o_cabac_ctx* snap, uint32_t frac_left,
                int32_t* levels, uint32_t* out, double* cost)
{
  const int N = 1 << log2_size, n2 = N * N;
  if (log2_size < 2 || log2_size > 5 || (comp && log2_size == 5)) return -1;
  int16_t c16[32 * 32], r2[32 * 32], zero[32 * 32];
  int32_t c32[32 * 32], dq[32 * 32];
  hop_o_estbits eb;
  memset(&eb, 0, sizeof(eb));
  hop_o_cabac_est_bits(snap, N, comp, &eb);
  hop_o_fwd_transform(bit_depth, resi, c16, N, 0);
  for (int i = 0; i < n2; i++) c32[i] = c16[i];
  uint32_t absSum = 0;
  memset(levels, 0, sizeof(int32_t) * (size_t)n2);
  hop_o_rdoq(c32, levels, log2_size, comp, 0, 0, tr_depth, qp_scaled, bit_depth, sign_hide, lambda_rdoq, &eb, &absSum);
  hop_o_cabac_ctx s = *snap;
  uint64_t f = hop_o_cabac_cbf_bits(&s, comp, tr_depth, absSum != 0);
  f += hop_o_cabac_coeff_bits(&s, levels, log2_size, comp, 0, sign_hide, use_ts, 0);
  const uint32_t singleBits = (uint32_t)((frac_left + f) >> 15);
  memset(zero, 0, sizeof(zero));
  const uint32_t zeroDist = weighted(hop_o_sse(zero, N, resi, N, N, N, bit_depth), comp, dist_weight);
  uint32_t dist = zeroDist, nzDist = 0, nullBits = 0;
  double chosen = 0;
  if (absSum) {
    hop_o_dequant_flat(bit_depth, qp_scaled, levels, dq, N);
    for (int i = 0; i < n2; i++) c16[i] = (int16_t)dq[i];
    hop_o_inv_transform(bit_depth, c16, r2, N, 0);
    nzDist = weighted(hop_o_sse(r2, N, resi, N, N, N, bit_depth), comp, dist_weight);
    const double singleCost = calc_rd_cost(singleBits, nzDist, lambda_rd);
    hop_o_cabac_ctx z = *snap;
    nullBits = (uint32_t)((frac_left + hop_o_cabac_cbf_bits(&z, comp, tr_depth, 0)) >> 15);
    const double nullCost = calc_rd_cost(nullBits, zeroDist, lambda_rd);
    if (nullCost < singleCost) { absSum = 0; memset(levels, 0, sizeof(int32_t) * (size_t)n2); chosen = nullCost; }
    else { dist = nzDist; chosen = singleCost; }
  } else {
    chosen = calc_rd_cost(singleBits, zeroDist, lambda_rd);      /* not computed by the reference at this point; reported for convenience */
  }
  out[0] = absSum; out[1] = absSum != 0; out[2] = dist; out[3] = zeroDist; out[4] = nzDist; out[5] = singleBits; out[6] = nullBits; out[7] = 0;
  *cost = chosen;
  return 0;
}

/* Row a8, leaf step: TEncSearch::xIntraCodingLumaBlk after the prediction (TLibEncoder/TEncSearch.cpp:1082-1160; the chroma
 * twin xIntraCodingChromaBlk :1164-1330 has the same arithmetic): residual, estBit, transformNxN (DST-VII for 4x4 luma,
 * scan by intra direction, RDOQ with the intra tables), inverse path, reconstruction clip, distortion against the original.
 * org / pred / recon: N x N, stride N.  out[8]: abs_sum, cbf, dist, -, -, bits of cbf flag + levels from the snapshot (what
 * xGetIntraBitsQT will count for this block, :1340-1359 via xEncCoeffQT), -, -. */
/* ts_flag: the 4x4 transform-skip variant (pcCU->getTransformSkip of the block: xTransformSkip / xITransformSkip instead of the transform, the flag coded as 1) */
int hop_o_tu_intra_ts(const int16_t* org, const int16_t* pred, int log2_size, int comp, int scan_idx, int use_dst, int qp_scaled, int bit_depth,
                      int tr_depth, int sign_hide, int use_ts, int ts_flag, double lambda_rdoq, double lambda_rd, double dist_weight,
                      const hop_o_cabac_ctx* snap, uint32_t frac_left, int32_t* levels, int16_t* recon, uint32_t* out, double* cost)
{
  const int N = 1 << log2_size, n2 = N * N;
  if (log2_size < 2 || log2_size > 5 || (comp && log2_size == 5)) return -1;
  int16_t resi[32 * 32], c16[32 * 32], r2[32 * 32];
  int32_t c32[32 * 32], dq[32 * 32];
  const int dst = use_dst && N == 4 && comp == 0;
  for (int i = 0; i < n2; i++) resi[i] = (int16_t)(org[i] - pred[i]);
  hop_o_estbits eb;
  memset(&eb, 0, sizeof(eb));
  hop_o_cabac_est_bits(snap, N, comp, &eb);
  if (ts_flag) hop_o_transform_skip(bit_depth, resi, c32, N);
  else { hop_o_fwd_transform(bit_depth, resi, c16, N, dst); for (int i = 0; i < n2; i++) c32[i] = c16[i]; }
  uint32_t absSum = 0;
  memset(levels, 0, sizeof(int32_t) * (size_t)n2);
  hop_o_rdoq(c32, levels, log2_size, comp, 1, scan_idx, tr_depth, qp_scaled, bit_depth, sign_hide, lambda_rdoq, &eb, &absSum);
  hop_o_cabac_ctx s = *snap;
  uint64_t f = hop_o_cabac_cbf_bits(&s, comp, tr_depth, absSum != 0);
  f += hop_o_cabac_coeff_bits(&s, levels, log2_size, comp, scan_idx, sign_hide, use_ts, ts_flag);
  const uint32_t bits = (uint32_t)((frac_left + f) >> 15);
  if (absSum) {
    hop_o_dequant_flat(bit_depth, qp_scaled, levels, dq, N);
    if (ts_flag) hop_o_inv_transform_skip(bit_depth, dq, r2, N);
    else { for (int i = 0; i < n2; i++) c16[i] = (int16_t)dq[i]; hop_o_inv_transform(bit_depth, c16, r2, N, dst); }
  } else memset(r2, 0, sizeof(int16_t) * (size_t)n2);
  const int maxVal = (1 << bit_depth) - 1;
  for (int i = 0; i < n2; i++) { int v = pred[i] + r2[i]; recon[i] = (int16_t)(v < 0 ? 0 : v > maxVal ? maxVal : v); }
  const uint32_t dist = weighted(hop_o_sse(recon, N, org, N, N, N, bit_depth), comp, dist_weight);
  out[0] = absSum; out[1] = absSum != 0; out[2] = dist; out[3] = 0; out[4] = 0; out[5] = bits; out[6] = 0; out[7] = 0;
  *cost = calc_rd_cost(bits, dist, lambda_rd);
  return 0;
}
int hop_o_tu_intra(const int16_t* org, const int16_t* pred, int log2_size, int comp, int scan_idx, int use_dst, int qp_scaled, int bit_depth,
                   int tr_depth, int sign_hide, int use_ts, double lambda_rdoq, double lambda_rd, double dist_weight,
                   const hop_o_cabac_ctx* snap, uint32_t frac_left, int32_t* levels, int16_t* recon, uint32_t* out, double* cost)
{
  return hop_o_tu_intra_ts(org, pred, log2_size, comp, scan_idx, use_dst, qp_scaled, bit_depth, tr_depth, sign_hide, use_ts, 0, lambda_rdoq, lambda_rd, dist_weight,
                           snap, frac_left, levels, recon, out, cost);
}
