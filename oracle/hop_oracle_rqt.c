/* hop_oracle_rqt.c -- CPU restatement of the residual quadtree search of one SS/GT ("inter") CU.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows TEncSearch::xEstimateResidualQT (TLibEncoder/TEncSearch.cpp:6824-7560) with xEncodeResidualQT (:7562-7655):
 * per node the full-block evaluation of Y, Cb, Cr (estBit, transformNxN = xT + xRateDistOptQuant, bits of cbf flag + levels
 * counted from the node's entry state, inverse path, cbf-zero decision :6959-7200), the 4x4 transform-skip retry
 * (:7210-7440), the node's own bits (:7443-7466), the four children evaluated one after the other on the coder state the
 * previous one left (:7480-7485), the recount of the whole subtree in syntax order (:7503-7511) and the split decision
 * (:7514-7522).  Costs are TComRdCost::calcRdCost values; bit counts are the counting coder's integer bits, fraction carried
 * as the reference's TEncBinCABAC::resetBits does.
 *
 * Pinned in the loop: oracle/enc_shim.cpp puts this function in place of the reference's xEstimateResidualQT inside the
 * reference encoder, which must then write the unmodified encoder's bitstream (tests/test_encoder_shim.py). */
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include "hop_oracle.h"

typedef struct {
  const hop_o_rqt_cfg* cfg;
  const int16_t* resi[3]; int stride[3];
  hop_o_coder* cur;                 /* m_pcRDGoOnSbacCoder */
  hop_o_coder root[4], test[4];     /* m_pppcRDSbacCoder[depth][CI_QT_TRAFO_ROOT / _TEST], by transform depth */
  hop_o_rqt_state* st;
  int parts;                        /* 4x4 partitions in the CU */
} Rqt;

static int zx(int p) { int x = 0; for (int b = 0; b < 4; b++) x |= ((p >> (2 * b)) & 1) << b; return 4 * x; }       /* partition -> luma position in the CU */
static int zy(int p) { int y = 0; for (int b = 0; b < 4; b++) y |= ((p >> (2 * b + 1)) & 1) << b; return 4 * y; }
static void reset_bits(hop_o_coder* c) { c->frac &= 32767; }                       /* TEncBinCABAC::resetBits keeps the fraction */
static uint32_t written(const hop_o_coder* c) { return (uint32_t)(c->frac >> 15); }
static void set_parts(uint8_t* a, int first, int count, int v) { memset(a + first, v, (size_t)count); }
static int bit_depth(const Rqt* r, int comp) { return comp ? r->cfg->bit_depth_c : r->cfg->bit_depth_y; }
static uint32_t weighted(const Rqt* r, int comp, uint32_t sse) { return comp ? (uint32_t)(int)(r->cfg->dist_weight[comp] * sse) : sse; }

/* encodeQtCbf + encodeCoeffNxN of one component TU on the current coder (:6959-6962) */
static void code_cbf_and_coeff(Rqt* r, int comp, int trMode, int part, const int32_t* coef, int log2)
{
  hop_o_coder* c = r->cur;
  c->frac += hop_o_cabac_cbf_bits(&c->ctx, comp, trMode, (r->st->cbf[comp][part] >> trMode) & 1);
  c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef, log2, comp, 0, r->cfg->sign_hide, r->cfg->use_ts, r->st->tskip[comp][part]);
}

/* transformNxN (TComTrQuant.cpp:1204-1258) of one component with the tables of the current coder state */
static uint32_t transform_quant(Rqt* r, int comp, int log2, const int16_t* res, int stride, int trMode, int skip, int32_t* coef)
{
  const int N = 1 << log2, bd = bit_depth(r, comp);
  int16_t blk[32 * 32], c16[32 * 32]; int32_t c32[32 * 32];
  hop_o_estbits eb; memset(&eb, 0, sizeof(eb));
  hop_o_cabac_est_bits(&r->cur->ctx, N, comp ? 1 : 0, &eb);
  for (int y = 0; y < N; y++) memcpy(blk + y * N, res + y * stride, (size_t)N * sizeof(int16_t));
  if (skip) hop_o_transform_skip(bd, blk, c32, N);
  else { hop_o_fwd_transform(bd, blk, c16, N, 0); for (int i = 0; i < N * N; i++) c32[i] = c16[i]; }
  uint32_t absSum = 0;
  memset(coef, 0, sizeof(int32_t) * (size_t)(N * N));
  hop_o_rdoq(c32, coef, log2, comp, 0, 0, trMode, r->cfg->qp[comp], bd, r->cfg->sign_hide, r->cfg->lambda_rdoq[comp], &eb, &absSum);
  return absSum;
}

/* invtransformNxN (:1260-1310) into the layer's residual plane; returns the distortion against the residual */
static uint32_t inverse_and_dist(Rqt* r, int comp, int log2, const int32_t* coef, int skip, int16_t* out, int ostride, const int16_t* res, int stride)
{
  const int N = 1 << log2, bd = bit_depth(r, comp);
  int32_t dq[32 * 32]; int16_t c16[32 * 32], blk[32 * 32];
  hop_o_dequant_flat(bd, r->cfg->qp[comp], coef, dq, N);
  if (skip) hop_o_inv_transform_skip(bd, dq, blk, N);
  else { for (int i = 0; i < N * N; i++) c16[i] = (int16_t)dq[i]; hop_o_inv_transform(bd, c16, blk, N, 0); }
  for (int y = 0; y < N; y++) memcpy(out + y * ostride, blk + y * N, (size_t)N * sizeof(int16_t));
  return weighted(r, comp, hop_o_sse(out, ostride, res, stride, N, N, bd));
}

static void encode_tree(Rqt* r, int part, int trMode, int log2, int subdivAndCbf, int comp);

static void node(Rqt* r, int part, int trMode, int log2, double* rdCost, uint32_t* ruiBits, uint32_t* ruiDist, uint32_t* zeroDist)
{
  const hop_o_rqt_cfg* g = r->cfg; hop_o_rqt_state* st = r->st;
  const int nparts = r->parts >> (2 * trMode);                       /* partitions of this node */
  int checkFull = (g->inter_split_flag && trMode == 0 && log2 > g->log2_min_tu_in_cu) ? 0 : (log2 <= g->log2_max_tu);
  const int checkSplit = log2 > g->log2_min_tu_in_cu;
  int codeChroma = 1, trModeC = trMode, log2C = log2 - 1;
  if (log2 == 2) { log2C++; trModeC--; codeChroma = (part % (r->parts >> (2 * trModeC))) == 0; }
  const int npartsC = r->parts >> (2 * trModeC);
  const int setCbf = 1 << trMode;
  double singleCost = 1.7e+308; uint32_t singleBits = 0, singleDist = 0;
  uint32_t absSum[3] = { 0, 0, 0 }; int bestSkip[3] = { 0, 0, 0 };
  r->root[trMode] = *r->cur;
  const int px = zx(part), py = zy(part);
  const int layer = g->log2_max_tu - log2, cu = 1 << g->log2_cu;

  if (checkFull) {
    int32_t* coef[3] = { st->coef[layer][0] + 16 * part, st->coef[layer][1] + ((16 * part) >> 2), st->coef[layer][2] + ((16 * part) >> 2) };
    const int lg[3] = { log2, log2C, log2C };
    const int16_t* res[3] = { r->resi[0] + py * r->stride[0] + px, r->resi[1] + (py >> 1) * r->stride[1] + (px >> 1), r->resi[2] + (py >> 1) * r->stride[2] + (px >> 1) };
    int16_t* rec[3] = { st->resi[layer][0] + py * cu + px, st->resi[layer][1] + (py >> 1) * (cu >> 1) + (px >> 1), st->resi[layer][2] + (py >> 1) * (cu >> 1) + (px >> 1) };
    const int rs[3] = { cu, cu >> 1, cu >> 1 };
    const int ncomp = codeChroma ? 3 : 1;
    set_parts(st->tr_idx, part, nparts, trMode);
    double minCost[3] = { 1.7e+308, 1.7e+308, 1.7e+308 };
    const int checkSkip[3] = { g->use_ts && log2 == 2, g->use_ts && log2C == 2, g->use_ts && log2C == 2 };
    set_parts(st->tskip[0], part, nparts, 0);
    if (codeChroma) { set_parts(st->tskip[1], part, npartsC, 0); set_parts(st->tskip[2], part, npartsC, 0); }
    /* transforms + quantisation: the coder stands at the entry state for all three (:6901-6952) */
    for (int c = 0; c < ncomp; c++) absSum[c] = transform_quant(r, c, lg[c], res[c], r->stride[c], trMode, 0, coef[c]);
    set_parts(st->cbf[0], part, nparts, absSum[0] ? setCbf : 0);
    if (codeChroma) { set_parts(st->cbf[1], part, npartsC, absSum[1] ? setCbf : 0); set_parts(st->cbf[2], part, npartsC, absSum[2] ? setCbf : 0); }
    /* bits of each component from the entry state (:6957-6977) */
    uint32_t sBits[3] = { 0, 0, 0 };
    for (int c = 0; c < ncomp; c++) {
      if (c) *r->cur = r->root[trMode];
      reset_bits(r->cur);
      code_cbf_and_coeff(r, c, trMode, part, coef[c], lg[c]);
      sBits[c] = written(r->cur);
    }
    uint32_t distC[3] = { 0, 0, 0 };
    int16_t zero[32 * 32]; memset(zero, 0, sizeof(zero));
    for (int c = 0; c < ncomp; c++) {
      const int N = 1 << lg[c];
      distC[c] = weighted(r, c, hop_o_sse(zero, N, res[c], r->stride[c], N, N, bit_depth(r, c)));
      if (zeroDist) *zeroDist += distC[c];
      if (absSum[c]) {
        const uint32_t nz = inverse_and_dist(r, c, lg[c], coef[c], 0, rec[c], rs[c], res[c], r->stride[c]);
        const double sc = hop_o_calc_rd_cost(sBits[c], nz, g->lambda_rd);
        *r->cur = r->root[trMode]; reset_bits(r->cur);
        r->cur->frac += hop_o_cabac_cbf_bits(&r->cur->ctx, c, trMode, 0);
        const double nc = hop_o_calc_rd_cost(written(r->cur), distC[c], g->lambda_rd);
        if (nc < sc) { absSum[c] = 0; memset(coef[c], 0, sizeof(int32_t) * (size_t)(N * N)); if (checkSkip[c]) minCost[c] = nc; }
        else { distC[c] = nz; if (checkSkip[c]) minCost[c] = sc; }
      } else if (checkSkip[c]) {
        *r->cur = r->root[trMode]; reset_bits(r->cur);
        r->cur->frac += hop_o_cabac_cbf_bits(&r->cur->ctx, c, trMode, 0);
        minCost[c] = hop_o_calc_rd_cost(written(r->cur), distC[c], g->lambda_rd);
      }
      if (!absSum[c]) for (int y = 0; y < N; y++) memset(rec[c] + y * rs[c], 0, (size_t)N * sizeof(int16_t));
    }
    set_parts(st->cbf[0], part, nparts, absSum[0] ? setCbf : 0);
    if (codeChroma) { set_parts(st->cbf[1], part, npartsC, absSum[1] ? setCbf : 0); set_parts(st->cbf[2], part, npartsC, absSum[2] ? setCbf : 0); }

    /* transform-skip retry, luma (:7210-7292) */
    if (checkSkip[0]) {
      int32_t best[16]; int16_t bestRes[16];
      memcpy(best, coef[0], sizeof(best));
      for (int y = 0; y < 4; y++) memcpy(bestRes + 4 * y, rec[0] + y * rs[0], 8);
      *r->cur = r->root[trMode];
      set_parts(st->tskip[0], part, nparts, 1);
      const uint32_t as = transform_quant(r, 0, 2, res[0], r->stride[0], trMode, 1, coef[0]);
      set_parts(st->cbf[0], part, nparts, as ? setCbf : 0);
      uint32_t nz = 0; double sc = 0;
      if (as) {
        reset_bits(r->cur);
        code_cbf_and_coeff(r, 0, trMode, part, coef[0], 2);
        const uint32_t b = written(r->cur);
        nz = inverse_and_dist(r, 0, 2, coef[0], 1, rec[0], rs[0], res[0], r->stride[0]);
        sc = hop_o_calc_rd_cost(b, nz, g->lambda_rd);
      }
      if (!as || minCost[0] < sc) {
        set_parts(st->tskip[0], part, nparts, 0);
        memcpy(coef[0], best, sizeof(best));
        for (int y = 0; y < 4; y++) memcpy(rec[0] + y * rs[0], bestRes + 4 * y, 8);
      } else { distC[0] = nz; absSum[0] = as; bestSkip[0] = 1; }
      set_parts(st->cbf[0], part, nparts, absSum[0] ? setCbf : 0);
    }
    /* transform-skip retry, chroma (:7294-7437): both planes are transformed first, then each is counted from the entry state */
    if (codeChroma && checkSkip[1]) {
      int32_t best[3][16]; int16_t bestRes[3][16]; uint32_t as[3] = { 0, 0, 0 };
      for (int c = 1; c < 3; c++) { memcpy(best[c], coef[c], sizeof(best[c])); for (int y = 0; y < 4; y++) memcpy(bestRes[c] + 4 * y, rec[c] + y * rs[c], 8); }
      *r->cur = r->root[trMode];
      set_parts(st->tskip[1], part, npartsC, 1); set_parts(st->tskip[2], part, npartsC, 1);
      for (int c = 1; c < 3; c++) as[c] = transform_quant(r, c, 2, res[c], r->stride[c], trMode, 1, coef[c]);
      set_parts(st->cbf[1], part, npartsC, as[1] ? setCbf : 0); set_parts(st->cbf[2], part, npartsC, as[2] ? setCbf : 0);
      for (int c = 1; c < 3; c++) {
        uint32_t nz = 0; double sc = 0;
        if (as[c]) {
          if (c == 2) *r->cur = r->root[trMode];
          reset_bits(r->cur);
          code_cbf_and_coeff(r, c, trMode, part, coef[c], 2);
          const uint32_t b = written(r->cur);
          nz = inverse_and_dist(r, c, 2, coef[c], 1, rec[c], rs[c], res[c], r->stride[c]);
          sc = hop_o_calc_rd_cost(b, nz, g->lambda_rd);
        }
        if (!as[c] || minCost[c] < sc) {
          set_parts(st->tskip[c], part, npartsC, 0);
          memcpy(coef[c], best[c], sizeof(best[c]));
          for (int y = 0; y < 4; y++) memcpy(rec[c] + y * rs[c], bestRes[c] + 4 * y, 8);
        } else { distC[c] = nz; absSum[c] = as[c]; bestSkip[c] = 1; }
      }
      set_parts(st->cbf[1], part, npartsC, absSum[1] ? setCbf : 0); set_parts(st->cbf[2], part, npartsC, absSum[2] ? setCbf : 0);
    }

    /* the node coded as one TU, from the entry state (:7439-7466) */
    *r->cur = r->root[trMode]; reset_bits(r->cur);
    hop_o_coder* c = r->cur;
    if (log2 > g->log2_min_tu_in_cu) c->frac += hop_o_cabac_subdiv_bits(&c->ctx, 5 - log2, 0);
    if (codeChroma) {
      c->frac += hop_o_cabac_cbf_bits(&c->ctx, 1, trMode, (st->cbf[1][part] >> trMode) & 1);
      c->frac += hop_o_cabac_cbf_bits(&c->ctx, 2, trMode, (st->cbf[2][part] >> trMode) & 1);
    }
    c->frac += hop_o_cabac_cbf_bits(&c->ctx, 0, trMode, (st->cbf[0][part] >> trMode) & 1);
    for (int k = 0; k < ncomp; k++)
      c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef[k], lg[k], k, 0, g->sign_hide, g->use_ts, st->tskip[k][part]);
    singleBits = written(c);
    singleDist = distC[0] + distC[1] + distC[2];
    singleCost = hop_o_calc_rd_cost(singleBits, singleDist, g->lambda_rd);
  }

  if (checkSplit) {
    if (checkFull) { r->test[trMode] = *r->cur; *r->cur = r->root[trMode]; }
    uint32_t subDist = 0, subBits = 0; double subCost = 0.0;
    const int q = nparts >> 2;
    for (int k = 0; k < 4; k++) node(r, part + k * q, trMode + 1, log2 - 1, &subCost, &subBits, &subDist, checkFull ? NULL : zeroDist);
    int any[3] = { 0, 0, 0 };
    for (int c = 0; c < 3; c++) for (int k = 0; k < 4; k++) any[c] |= (st->cbf[c][part + k * q] >> (trMode + 1)) & 1;
    for (int c = 0; c < 3; c++) for (int k = 0; k < nparts; k++) st->cbf[c][part + k] |= (uint8_t)(any[c] << trMode);
    *r->cur = r->root[trMode]; reset_bits(r->cur);
    encode_tree(r, part, trMode, log2, 1, 0);
    encode_tree(r, part, trMode, log2, 0, 0);
    encode_tree(r, part, trMode, log2, 0, 1);
    encode_tree(r, part, trMode, log2, 0, 2);
    subBits = written(r->cur);
    subCost = hop_o_calc_rd_cost(subBits, subDist, g->lambda_rd);
    if (any[0] || any[1] || any[2] || !checkFull) {
      if (subCost < singleCost) { *rdCost += subCost; *ruiBits += subBits; *ruiDist += subDist; return; }
    }
    set_parts(st->tskip[0], part, nparts, bestSkip[0]);
    if (codeChroma) { set_parts(st->tskip[1], part, npartsC, bestSkip[1]); set_parts(st->tskip[2], part, npartsC, bestSkip[2]); }
    *r->cur = r->test[trMode];
  }
  *rdCost += singleCost; *ruiBits += singleBits; *ruiDist += singleDist;
  set_parts(st->tr_idx, part, nparts, trMode);
  set_parts(st->cbf[0], part, nparts, absSum[0] ? setCbf : 0);
  if (codeChroma) { set_parts(st->cbf[1], part, npartsC, absSum[1] ? setCbf : 0); set_parts(st->cbf[2], part, npartsC, absSum[2] ? setCbf : 0); }
}

/* xEncodeResidualQT (:7562-7655): the subtree as the arrays describe it, flags first (subdivAndCbf) then one component's levels */
static void encode_tree(Rqt* r, int part, int curTrMode, int log2, int subdivAndCbf, int comp)
{
  const hop_o_rqt_cfg* g = r->cfg; hop_o_rqt_state* st = r->st; hop_o_coder* c = r->cur;
  const int trMode = st->tr_idx[part], subdiv = curTrMode != trMode;
  if (subdivAndCbf && log2 <= g->log2_max_tu && log2 > g->log2_min_tu_in_cu) c->frac += hop_o_cabac_subdiv_bits(&c->ctx, 5 - log2, subdiv);
  if (subdivAndCbf) {
    const int first = curTrMode == 0;
    if (first || log2 > 2) {
      if (first || ((st->cbf[1][part] >> (curTrMode - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 1, curTrMode, (st->cbf[1][part] >> curTrMode) & 1);
      if (first || ((st->cbf[2][part] >> (curTrMode - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 2, curTrMode, (st->cbf[2][part] >> curTrMode) & 1);
    }
  }
  if (!subdiv) {
    const int layer = g->log2_max_tu - log2;
    int codeChroma = 1, trModeC = trMode, log2C = log2 - 1;
    if (log2 == 2) { log2C++; trModeC--; codeChroma = (part % (r->parts >> (2 * trModeC))) == 0; }
    if (subdivAndCbf) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 0, trMode, (st->cbf[0][part] >> trMode) & 1);
    else {
      if (comp == 0 && ((st->cbf[0][part] >> trMode) & 1))
        c->frac += hop_o_cabac_coeff_bits(&c->ctx, st->coef[layer][0] + 16 * part, log2, 0, 0, g->sign_hide, g->use_ts, st->tskip[0][part]);
      if (codeChroma && comp && ((st->cbf[comp][part] >> trMode) & 1))
        c->frac += hop_o_cabac_coeff_bits(&c->ctx, st->coef[layer][comp] + ((16 * part) >> 2), log2C, comp, 0, g->sign_hide, g->use_ts, st->tskip[comp][part]);
    }
  } else if (subdivAndCbf || ((st->cbf[comp][part] >> curTrMode) & 1)) {
    const int q = (r->parts >> (2 * curTrMode)) >> 2;
    for (int k = 0; k < 4; k++) encode_tree(r, part + k * q, curTrMode + 1, log2 - 1, subdivAndCbf, comp);
  }
}

void hop_o_rqt(const hop_o_rqt_cfg* cfg, const int16_t* resiY, int strideY, const int16_t* resiCb, const int16_t* resiCr, int strideC,
               hop_o_coder* coder, hop_o_rqt_state* st, double* cost, uint32_t* bits, uint32_t* dist, uint32_t* zero_dist)
{
  Rqt r; memset(&r, 0, sizeof(r));
  r.cfg = cfg; r.resi[0] = resiY; r.resi[1] = resiCb; r.resi[2] = resiCr; r.stride[0] = strideY; r.stride[1] = r.stride[2] = strideC;
  r.cur = coder; r.st = st; r.parts = 1 << (2 * (cfg->log2_cu - 2));
  *cost = 0; *bits = 0; *dist = 0; if (zero_dist) *zero_dist = 0;
  node(&r, 0, 0, cfg->log2_cu, cost, bits, dist, zero_dist);
}

/* the chosen transform units' levels in the CU's coefficient layout -- what xSetResidualQTData (:7658-7777) copies into
 * getCoeffY/Cb/Cr: a TU's block is contiguous from its first partition on (16 luma, 4 + 4 chroma levels per partition), so
 * taking every partition from the layer of its transform depth takes every chosen block whole.  out: 1.5 * size^2 ints. */
void hop_o_rqt_final_coeffs(const hop_o_rqt_cfg* cfg, const hop_o_rqt_state* st, int32_t* out)
{
  const int cu2 = 1 << (2 * cfg->log2_cu), parts = cu2 >> 4;
  for (int p = 0; p < parts; p++) {
    const int layer = cfg->log2_max_tu - (cfg->log2_cu - st->tr_idx[p]);
    memcpy(out + 16 * p, st->coef[layer][0] + 16 * p, 16 * sizeof(int32_t));
    memcpy(out + cu2 + 4 * p, st->coef[layer][1] + 4 * p, 4 * sizeof(int32_t));
    memcpy(out + cu2 + (cu2 >> 2) + 4 * p, st->coef[layer][2] + 4 * p, 4 * sizeof(int32_t));
  }
}

/* The tail of TEncSearch::encodeResAndCalcRdInterCU around the quadtree (TLibEncoder/TEncSearch.cpp:6700-6723, :6804-6812), without the
 * CU-level syntax bits in between (xAddSymbolBitsInter, the caller's): the root-cbf-zero test -- bits of a zero rqt_root_cbf counted on the
 * coder as xEstimateResidualQT left it, against the quadtree's cost --, the arrays cleared if it wins, the reconstruction
 * Clip(prediction + residual of the chosen transform units) and its distortion against the original (chroma weighted per plane).
 * coder: the state after hop_o_rqt; cost / zero_dist: its results.  pred / org / rec: CU planes, pitch = CU size (chroma half).
 * Returns the root cbf (0: the zero residual won); dist3: Y, Cb, Cr distortion of the reconstruction. */
int hop_o_inter_cu_finish(const hop_o_rqt_cfg* cfg, hop_o_rqt_state* st, const hop_o_coder* coder, double cost, uint32_t zero_dist,
                          const int16_t* const pred[3], const int16_t* const org[3], int16_t* const rec[3], uint32_t dist3[3], int32_t* final_coef)
{
  const int cu = 1 << cfg->log2_cu, cu2 = cu * cu, parts = cu2 >> 4;
  hop_o_coder c = *coder;
  c.frac &= 32767;
  c.frac += hop_o_cabac_root_cbf_bits(&c.ctx, 0);
  const double zeroCost = hop_o_calc_rd_cost((uint32_t)(c.frac >> 15), zero_dist, cfg->lambda_rd);
  const int root = !(zeroCost < cost);
  if (!root) {
    memset(st->tr_idx, 0, (size_t)parts);
    for (int k = 0; k < 3; k++) { memset(st->cbf[k], 0, (size_t)parts); memset(st->tskip[k], 0, (size_t)parts); }
    if (final_coef) memset(final_coef, 0, sizeof(int32_t) * (size_t)(cu2 + cu2 / 2));
  } else if (final_coef) hop_o_rqt_final_coeffs(cfg, st, final_coef);
  for (int k = 0; k < 3; k++) {
    const int w = k ? cu >> 1 : cu, bd = k ? cfg->bit_depth_c : cfg->bit_depth_y, maxv = (1 << bd) - 1;
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) {
      int r = 0;
      if (root) {                                                    /* xSetResidualQTData, spatial: the layer of the partition's transform depth */
        const int px = k ? 2 * x : x, py = k ? 2 * y : y;
        int p = 0; for (int b = 0; b < 4; b++) p |= (((px >> 2) >> b) & 1) << (2 * b) | (((py >> 2) >> b) & 1) << (2 * b + 1);
        const int layer = cfg->log2_max_tu - (cfg->log2_cu - st->tr_idx[p]);
        r = st->resi[layer][k][y * w + x];
      }
      int v = pred[k][y * w + x] + r;
      rec[k][y * w + x] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
    }
    const uint32_t sse = hop_o_sse(rec[k], w, org[k], w, w, w, bd);
    dist3[k] = k ? (uint32_t)(int)(cfg->dist_weight[k] * sse) : sse;
  }
  return root;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The CU-level syntax of an SS/GT ("inter") CU, counted: TEncSearch::xAddSymbolBitsInter (TLibEncoder/TEncSearch.cpp:7779-7810) =
 * skip flag, and either the merge index (skipped CU) or prediction mode, partition size (TEncSbac.cpp:469-566), per PU the merge
 * flag / index or MVD (:944-1048), MVP index (:434-467), GT flag (:654-677) and the GT corner vectors (codeGT :1051-1330: corners 0..2,
 * coded like MVDs with their own two contexts), the root cbf and the transform tree in bitstream order (TEncEntropy::encodeCoeff
 * :633-660, xEncodeTransform :219-394).  One reference list with one picture, no transquant bypass, no delta QP.
 * cu_ctx: the CU-level context states used here -- skip[3], merge_flag, merge_idx, part_size[4], pred_mode, mvd[2], mvp_idx, gt_flag,
 * gt[2] (16 bytes, this order); coder: the residual sets + the fraction, both updated.  coef: the CU's levels in the layout of
 * hop_o_rqt_final_coeffs.  Returns the integer bits (what getNumberOfWrittenBits adds to ruiBits). */
typedef struct { hop_o_coder* c; uint8_t* cu; } Syn;
#define CU_SKIP 0
#define CU_MERGE_FLAG 3
#define CU_MERGE_IDX 4
#define CU_PART 5
#define CU_PRED 9
#define CU_MVD 10
#define CU_MVP 12
#define CU_GTF 13
#define CU_GT 14
static void bin(Syn* s, int idx, int b) { s->c->frac += (uint64_t)hop_o_ctx_bits(s->cu[idx], b); s->cu[idx] = hop_o_ctx_next(s->cu[idx], b); }
static void ep(Syn* s, int n) { s->c->frac += (uint64_t)32768 * (uint64_t)n; }
static int eg_bins(uint32_t sym, int k) { int n = 0; while (sym >= (1u << k)) { n++; sym -= 1u << k; k++; } return n + 1 + k; }   /* xWriteEpExGolomb :354-374 */
static void vec_like_mvd(Syn* s, int base, const int* v, int ncomp)
{                                                                       /* codeMvd / codeGT: != 0 flags, > 1 flags, then remainder + sign in bypass */
  for (int i = 0; i < ncomp; i++) bin(s, base, v[i] != 0);
  for (int i = 0; i < ncomp; i++) if (v[i]) bin(s, base + 1, abs(v[i]) > 1);
  for (int i = 0; i < ncomp; i++) if (v[i]) { if (abs(v[i]) > 1) ep(s, eg_bins((uint32_t)abs(v[i]) - 2, 1)); ep(s, 1); }
}
static void merge_index(Syn* s, int idx, int num)
{
  if (num <= 1) return;
  for (int ui = 0; ui < num - 1; ui++) { const int sym = ui == idx ? 0 : 1; if (ui == 0) bin(s, CU_MERGE_IDX, sym); else ep(s, 1); if (!sym) break; }
}
static void part_size(Syn* s, const hop_o_cu_syntax* y, int log2_cu)
{
  const int e = y->part_size;
  if (e == 0) { bin(s, CU_PART, 1); return; }
  if (e == 1 || e == 4 || e == 5) {                                   /* 2NxN, 2NxnU, 2NxnD */
    bin(s, CU_PART, 0); bin(s, CU_PART + 1, 1);
    if (y->amp_acc) { if (e == 1) bin(s, CU_PART + 3, 1); else { bin(s, CU_PART + 3, 0); ep(s, 1); } }
    return;
  }
  if (e == 2 || e == 6 || e == 7) {                                   /* Nx2N, nLx2N, nRx2N */
    bin(s, CU_PART, 0); bin(s, CU_PART + 1, 0);
    if (y->is_min_cu && log2_cu != 3) bin(s, CU_PART + 2, 1);
    if (y->amp_acc) { if (e == 2) bin(s, CU_PART + 3, 1); else { bin(s, CU_PART + 3, 0); ep(s, 1); } }
    return;
  }
  if (y->is_min_cu && log2_cu != 3) { bin(s, CU_PART, 0); bin(s, CU_PART + 1, 0); bin(s, CU_PART + 2, 0); }      /* NxN */
}
/* xEncodeTransform */
static int tu_scan(const hop_o_intra_syntax* y, int parts, int part, int width, int comp);
static void transform_tree(const hop_o_rqt_cfg* g, const hop_o_intra_syntax* iy /* NULL: an SS/GT ("inter") CU */, const hop_o_rqt_state* st, const int32_t* coef, hop_o_coder* c,
                           int part, int trIdx, int log2, int* bakPart)
{
  const int parts = 1 << (2 * (g->log2_cu - 2)), cu2 = 1 << (2 * g->log2_cu);
  const int subdiv = st->tr_idx[part] > trIdx;
  int cbfY = (st->cbf[0][part] >> trIdx) & 1, cbfU = (st->cbf[1][part] >> trIdx) & 1, cbfV = (st->cbf[2][part] >> trIdx) & 1;
  if (log2 == 2) {
    const int pn = parts >> (2 * (trIdx - 1));
    if (part % pn == 0) *bakPart = part;
    else if (part % pn == pn - 1) { cbfU = (st->cbf[1][*bakPart] >> trIdx) & 1; cbfV = (st->cbf[2][*bakPart] >> trIdx) & 1; }
  }
  if (iy && iy->part_nxn && trIdx == 0) { /* intra NxN: the split is inferred (TEncEntropy.cpp:246-249) */ }
  else if (!iy && g->inter_split_flag && trIdx == 0) { /* QuadtreeTUMaxDepthInter == 1, partition != 2Nx2N: the split is inferred */ }
  else if (log2 > g->log2_max_tu) { }
  else if (log2 == 2) { }
  else if (log2 == g->log2_min_tu_in_cu) { }
  else c->frac += hop_o_cabac_subdiv_bits(&c->ctx, 5 - log2, subdiv);
  const int first = trIdx == 0;
  if (first || log2 > 2) {
    if (first || ((st->cbf[1][part] >> (trIdx - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 1, trIdx, (st->cbf[1][part] >> trIdx) & 1);
    if (first || ((st->cbf[2][part] >> (trIdx - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 2, trIdx, (st->cbf[2][part] >> trIdx) & 1);
  }
  if (subdiv) {
    const int q = (parts >> (2 * trIdx)) >> 2;
    for (int k = 0; k < 4; k++) transform_tree(g, iy, st, coef, c, part + k * q, trIdx + 1, log2 - 1, bakPart);
    return;
  }
  if (iy || !(trIdx == 0 && !((st->cbf[1][part]) & 1) && !((st->cbf[2][part]) & 1)))                  /* inferred for a non-intra CU without chroma cbf (:334-338) */
    c->frac += hop_o_cabac_cbf_bits(&c->ctx, 0, st->tr_idx[part], (st->cbf[0][part] >> st->tr_idx[part]) & 1);
  if (cbfY) c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef + 16 * part, log2, 0, iy ? tu_scan(iy, parts, part, 1 << log2, 0) : 0, g->sign_hide, g->use_ts, st->tskip[0][part]);
  if (log2 > 2) {
    if (cbfU) c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef + cu2 + 4 * part, log2 - 1, 1, iy ? tu_scan(iy, parts, part, 1 << (log2 - 1), 1) : 0, g->sign_hide, g->use_ts, st->tskip[1][part]);
    if (cbfV) c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef + cu2 + (cu2 >> 2) + 4 * part, log2 - 1, 2, iy ? tu_scan(iy, parts, part, 1 << (log2 - 1), 2) : 0, g->sign_hide, g->use_ts, st->tskip[2][part]);
  } else {
    const int pn = parts >> (2 * (trIdx - 1));
    if (part % pn == pn - 1) {
      if (cbfU) c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef + cu2 + 4 * *bakPart, 2, 1, iy ? tu_scan(iy, parts, *bakPart, 4, 1) : 0, g->sign_hide, g->use_ts, st->tskip[1][*bakPart]);
      if (cbfV) c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef + cu2 + (cu2 >> 2) + 4 * *bakPart, 2, 2, iy ? tu_scan(iy, parts, *bakPart, 4, 2) : 0, g->sign_hide, g->use_ts, st->tskip[2][*bakPart]);
    }
  }
}

uint32_t hop_o_inter_cu_bits(const hop_o_rqt_cfg* cfg, const hop_o_cu_syntax* y, const hop_o_rqt_state* st, const int32_t* coef, hop_o_coder* coder, uint8_t cu_ctx[16],
                             int* skipped)
{
  Syn s = { coder, cu_ctx };
  const int root = (st->cbf[0][0] & 1) || (st->cbf[1][0] & 1) || (st->cbf[2][0] & 1);
  coder->frac &= 32767;                                                 /* resetBits */
  if (y->pu[0].merge_flag && y->part_size == 0 && !root) {
    *skipped = 1;
    bin(&s, CU_SKIP + y->skip_ctx, 1);
    merge_index(&s, y->pu[0].merge_idx, y->max_merge_cand);
    return (uint32_t)(coder->frac >> 15);
  }
  *skipped = y->skip_flag;
  bin(&s, CU_SKIP + y->skip_ctx, y->skip_flag ? 1 : 0);
  bin(&s, CU_PRED, 0);                                                  /* MODE_INTER */
  part_size(&s, y, cfg->log2_cu);
  for (int p = 0; p < y->n_pu; p++) {
    bin(&s, CU_MERGE_FLAG, y->pu[p].merge_flag ? 1 : 0);
    if (y->pu[p].merge_flag) { merge_index(&s, y->pu[p].merge_idx, y->max_merge_cand); continue; }
    vec_like_mvd(&s, CU_MVD, y->pu[p].mvd, 2);
    bin(&s, CU_MVP, y->pu[p].mvp_idx ? 1 : 0);                          /* xWriteUnaryMaxSymbol with one candidate bit */
    bin(&s, CU_GTF, y->pu[p].gt_flag ? 1 : 0);
    if (y->pu[p].gt_flag) vec_like_mvd(&s, CU_GT, y->pu[p].gt, 6);      /* corners 0..2 (IT_GT_AFFINE) */
  }
  if (!(y->pu[0].merge_flag && y->part_size == 0)) coder->frac += hop_o_cabac_root_cbf_bits(&coder->ctx, root);
  if (root) { int bak = 0; transform_tree(cfg, NULL, st, coef, coder, 0, 0, cfg->log2_cu, &bak); }
  return (uint32_t)(coder->frac >> 15);
}

/* ------------------------------------------------------------------------------------------------------------------
 * The bits of an intra CU's quadtree as the intra search counts them: TEncSearch::xGetIntraBitsQT (TLibEncoder/TEncSearch.cpp:957-980) =
 * xEncIntraHeader (:887-954: skip flag, prediction mode, partition size, the luma directions of the CU or of the PU that starts at this
 * node, the chroma direction), xEncSubdivCbfQT (:764-830: split flags if luma, chroma cbfs if chroma, luma cbf at the leaves) and
 * xEncCoeffQT (:833-884) of the asked components, from the node (tr_depth, part) downwards, with the levels of the layer buffers.
 * cu_ctx: hop_cabac_cu_ctx (20 bytes: ... [16] prev_intra_luma_pred_flag, [17..18] chroma prediction).  No PCM, no transquant bypass. */
#define CU_IPRED 16
#define CU_CPRED 17
static void intra_dir(Syn* s, int dir, const int* preds, int pred_num)
{                                                                       /* codeIntraDirLumaAng for one PU, TEncSbac.cpp:770-831 */
  int idx = -1;
  for (int i = 0; i < pred_num; i++) if (dir == preds[i]) idx = i;
  bin(s, CU_IPRED, idx != -1);
  ep(s, idx == -1 ? 5 : (idx ? 2 : 1));
}
static int tu_scan(const hop_o_intra_syntax* y, int parts, int part, int width, int comp)
{
  const int dir = comp ? (y->chroma_is_dm ? y->luma_dir[0] : y->chroma_dir) : y->luma_dir[y->part_nxn ? part / (parts >> 2) : 0];
  return hop_o_coef_scan_idx(width, comp == 0, 1, dir);
}
static void intra_subdiv_cbf(const hop_o_rqt_cfg* g, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, hop_o_coder* c, int parts, int trDepth, int part, int bLuma, int bChroma)
{
  const int trMode = st->tr_idx[part], subdiv = trMode > trDepth, log2 = g->log2_cu - trDepth;
  if (y->part_nxn && trDepth == 0) { }
  else if (log2 > g->log2_max_tu) { }
  else if (log2 == 2) { }
  else if (log2 == g->log2_min_tu_in_cu) { }
  else if (bLuma) c->frac += hop_o_cabac_subdiv_bits(&c->ctx, 5 - log2, subdiv);
  if (bChroma && log2 > 2) {
    if (trDepth == 0 || ((st->cbf[1][part] >> (trDepth - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 1, trDepth, (st->cbf[1][part] >> trDepth) & 1);
    if (trDepth == 0 || ((st->cbf[2][part] >> (trDepth - 1)) & 1)) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 2, trDepth, (st->cbf[2][part] >> trDepth) & 1);
  }
  if (subdiv) {
    const int q = (parts >> (2 * trDepth)) >> 2;
    for (int k = 0; k < 4; k++) intra_subdiv_cbf(g, y, st, c, parts, trDepth + 1, part + k * q, bLuma, bChroma);
    return;
  }
  if (bLuma) c->frac += hop_o_cabac_cbf_bits(&c->ctx, 0, trMode, (st->cbf[0][part] >> trMode) & 1);
}
static void intra_coeff(const hop_o_rqt_cfg* g, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, hop_o_coder* c, int parts, int trDepth, int part, int comp)
{
  const int trMode = st->tr_idx[part], log2 = g->log2_cu - trDepth;
  if (trMode > trDepth) {
    const int q = (parts >> (2 * trDepth)) >> 2;
    for (int k = 0; k < 4; k++) intra_coeff(g, y, st, c, parts, trDepth + 1, part + k * q, comp);
    return;
  }
  int d = trDepth;
  if (comp && log2 == 2) { d--; if (part % (parts >> (2 * d)) != 0) return; }
  const int lg = g->log2_cu - d - (comp ? 1 : 0), layer = g->log2_max_tu - log2;
  const int32_t* coef = st->coef[layer][comp] + (comp ? (16 * part) >> 2 : 16 * part);
  c->frac += hop_o_cabac_coeff_bits(&c->ctx, coef, lg, comp, tu_scan(y, parts, part, 1 << lg, comp), g->sign_hide, g->use_ts, st->tskip[comp][part]);
}

uint32_t hop_o_intra_cu_bits(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, int tr_depth, int part, int b_luma, int b_chroma,
                             hop_o_coder* coder, uint8_t cu_ctx[20])
{
  Syn s = { coder, cu_ctx };
  const int parts = 1 << (2 * (cfg->log2_cu - 2));
  coder->frac &= 32767;                                                 /* resetBits */
  if (b_luma) {
    if (part == 0) {
      if (y->skip_ctx >= 0) {                                           /* skip_ctx < 0: an I slice, which codes neither (TEncEntropy::encodeSkipFlag / encodePredMode) */
        bin(&s, CU_SKIP + y->skip_ctx, y->skip_flag ? 1 : 0);
        bin(&s, CU_PRED, 1);
      }
      if (y->is_min_cu) bin(&s, CU_PART, y->part_nxn ? 0 : 1);          /* codePartSize, intra */
    }
    if (!y->part_nxn) { if (part == 0) intra_dir(&s, y->luma_dir[0], y->preds[0], y->pred_num[0]); }
    else {
      const int q = parts >> 2;
      if (tr_depth == 0) for (int p = 0; p < 4; p++) intra_dir(&s, y->luma_dir[p], y->preds[p], y->pred_num[p]);
      else if (part % q == 0) intra_dir(&s, y->luma_dir[part / q], y->preds[part / q], y->pred_num[part / q]);
    }
  }
  if (b_chroma && part == 0) {                                          /* codeIntraDirChroma, TEncSbac.cpp:833-862 */
    if (y->chroma_is_dm) bin(&s, CU_CPRED, 0); else { bin(&s, CU_CPRED, 1); ep(&s, 2); }
  }
  intra_subdiv_cbf(cfg, y, st, coder, parts, tr_depth, part, b_luma, b_chroma);
  if (b_luma) intra_coeff(cfg, y, st, coder, parts, tr_depth, part, 0);
  if (b_chroma) { intra_coeff(cfg, y, st, coder, parts, tr_depth, part, 1); intra_coeff(cfg, y, st, coder, parts, tr_depth, part, 2); }
  return (uint32_t)(coder->frac >> 15);
}

/* ---- row a8: the luma transform tree of one intra PU, TEncSearch::xRecurIntraCodingQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:1361-1710) ----
 * Per node: the block coded as one TU (xIntraCodingLumaBlk :1003-1161 = reference samples from the reconstruction picture, prediction,
 * residual, estBit on the current coder state, transform / RDOQ, inverse path, reconstruction into the layer plane and into the
 * picture, SSE), for 4x4 blocks also as a transform-skip block (:1424-1523; the better one kept, the first restored from the
 * stash of xStoreIntraResultQT / xLoadIntraResultQT :1786-1955), its bits through xGetIntraBitsQT (= hop_o_intra_cu_bits); then the
 * four children on the state the previous one left, their bits recounted from the node's entry state (:1576-1640), and the cheaper
 * alternative kept - the picture gets the single block's reconstruction back when it wins (:1642-1700).  RDpenalty = 0. */
typedef struct { hop_o_coder c; uint8_t cu[20]; } ISnap;
typedef struct {
  const hop_o_rqt_cfg* cfg; const hop_o_intra_syntax* syn; const hop_o_intra_rqt_in* in; hop_o_rqt_state* st;
  hop_o_coder* coder; uint8_t* cu;
} Irq;
static ISnap isnap(const Irq* r) { ISnap s; s.c = *r->coder; memcpy(s.cu, r->cu, 20); return s; }
static void irestore(Irq* r, const ISnap* s) { *r->coder = s->c; memcpy(r->cu, s->cu, 20); }
int hop_o_intra_node_index(int tr_depth, int log2_size, int part)
{
  static const int first[5] = { 0, 1, 5, 21, 85 };
  return first[tr_depth] + (part >> (2 * (log2_size - 2)));
}

static void intra_luma_leaf(Irq* r, int trDepth, int part, int tsFlag, uint32_t* dist)
{
  const hop_o_rqt_cfg* g = r->cfg;
  const int log2 = g->log2_cu - trDepth, N = 1 << log2, cu = 1 << g->log2_cu, layer = g->log2_max_tu - log2, nparts = 1 << (2 * (log2 - 2));
  const int parts = 1 << (2 * (g->log2_cu - 2)), x = zx(part), y = zy(part);
  const int dir = r->syn->luma_dir[r->syn->part_nxn ? part / (parts >> 2) : 0];
  int L[4 * 32 + 1], F[4 * 32 + 1];
  int16_t org[32 * 32], pred[32 * 32], rec[32 * 32];
  hop_o_intra_fill_refs_u(r->in->rec, r->in->rec_stride, x, y, N, 4, r->in->avail + (size_t)hop_o_intra_node_index(trDepth, log2, part) * HOP_O_AVAIL_PITCH, g->bit_depth_y, L);
  hop_o_intra_smooth(L, N, g->bit_depth_y, r->in->strong, F);
  hop_o_intra_pred(L, F, N, dir, g->bit_depth_y, pred);
  for (int j = 0; j < N; j++) memcpy(org + j * N, r->in->org + (size_t)(y + j) * r->in->org_stride + x, sizeof(int16_t) * (size_t)N);
  set_parts(r->st->tr_idx, part, nparts, trDepth);
  uint32_t out[8]; double c;
  int32_t* coef = r->st->coef[layer][0] + 16 * part;
  hop_o_tu_intra_ts(org, pred, log2, 0, hop_o_coef_scan_idx(N, 1, 1, dir), 1, g->qp[0], g->bit_depth_y, trDepth, g->sign_hide, g->use_ts, tsFlag,
                    g->lambda_rdoq[0], g->lambda_rd, 1.0, &r->coder->ctx, 0, coef, rec, out, &c);
  set_parts(r->st->cbf[0], part, nparts, (out[0] ? 1 : 0) << trDepth);
  for (int j = 0; j < N; j++) {
    memcpy(r->st->resi[layer][0] + (size_t)(y + j) * cu + x, rec + j * N, sizeof(int16_t) * (size_t)N);
    memcpy(r->in->rec + (ptrdiff_t)(y + j) * r->in->rec_stride + x, rec + j * N, sizeof(int16_t) * (size_t)N);
  }
  *dist += out[2];
}

static void intra_node(Irq* r, int trDepth, int part, double* rdCost, uint32_t* distY)
{
  const hop_o_rqt_cfg* g = r->cfg;
  hop_o_rqt_state* st = r->st;
  const int log2 = g->log2_cu - trDepth, N = 1 << log2, cu = 1 << g->log2_cu, layer = g->log2_max_tu - log2, nparts = 1 << (2 * (log2 - 2));
  const int full = log2 <= g->log2_max_tu;
  int split = log2 > g->log2_min_tu_in_cu;
  if (r->in->check_first && full) split = 0;                           /* HHI_RQT_INTRA_SPEEDUP, :1387-1399 */
  const double MAXD = 1.7e+308;                                        /* MAX_DOUBLE */
  double singleCost = MAXD;
  uint32_t singleDist = 0, singleCbf = 0;
  int best = 0;
  const int ts = g->use_ts && N == 4 && (!r->in->ts_fast || r->syn->part_nxn);
  ISnap root, test, tbest;
  memset(&root, 0, sizeof(root)); memset(&test, 0, sizeof(test)); memset(&tbest, 0, sizeof(tbest));
  if (full) {
    if (ts) {
      root = isnap(r);
      int32_t keepCoef[16]; int16_t keepRec[16];
      for (int modeId = 0; modeId < 2; modeId++) {
        uint32_t d = 0; double c;
        set_parts(st->tskip[0], part, nparts, modeId);
        intra_luma_leaf(r, trDepth, part, modeId, &d);
        const uint32_t cbf = (st->cbf[0][part] >> trDepth) & 1;
        if (modeId == 1 && cbf == 0) c = MAXD;
        else c = hop_o_calc_rd_cost(hop_o_intra_cu_bits(g, r->syn, st, trDepth, part, 1, 0, r->coder, r->cu), d, g->lambda_rd);
        if (c < singleCost) {
          singleCost = c; singleDist = d; singleCbf = cbf; best = modeId;
          if (best == 0) {
            memcpy(keepCoef, st->coef[layer][0] + 16 * part, sizeof(keepCoef));
            for (int j = 0; j < 4; j++) memcpy(keepRec + 4 * j, st->resi[layer][0] + (size_t)(zy(part) + j) * cu + zx(part), 8);
            tbest = isnap(r);
          }
        }
        if (modeId == 0) irestore(r, &root);
      }
      set_parts(st->tskip[0], part, nparts, best);
      if (best == 0) {
        memcpy(st->coef[layer][0] + 16 * part, keepCoef, sizeof(keepCoef));
        for (int j = 0; j < 4; j++) {
          memcpy(st->resi[layer][0] + (size_t)(zy(part) + j) * cu + zx(part), keepRec + 4 * j, 8);
          memcpy(r->in->rec + (ptrdiff_t)(zy(part) + j) * r->in->rec_stride + zx(part), keepRec + 4 * j, 8);
        }
        set_parts(st->cbf[0], part, nparts, (int)(singleCbf << trDepth));
        irestore(r, &tbest);
      }
    } else {
      set_parts(st->tskip[0], part, nparts, 0);
      if (split) root = isnap(r);
      intra_luma_leaf(r, trDepth, part, 0, &singleDist);
      if (split) singleCbf = (st->cbf[0][part] >> trDepth) & 1;
      singleCost = hop_o_calc_rd_cost(hop_o_intra_cu_bits(g, r->syn, st, trDepth, part, 1, 0, r->coder, r->cu), singleDist, g->lambda_rd);
    }
  }
  if (split) {
    if (full) { test = isnap(r); irestore(r, &root); } else root = isnap(r);
    double splitCost = 0.0; uint32_t splitDist = 0, splitCbf = 0;
    const int q = nparts >> 2;
    for (int i = 0; i < 4; i++) {
      intra_node(r, trDepth + 1, part + i * q, &splitCost, &splitDist);
      splitCbf |= (st->cbf[0][part + i * q] >> (trDepth + 1)) & 1;
    }
    for (int o = 0; o < nparts; o++) st->cbf[0][part + o] |= (uint8_t)(splitCbf << trDepth);
    irestore(r, &root);
    splitCost = hop_o_calc_rd_cost(hop_o_intra_cu_bits(g, r->syn, st, trDepth, part, 1, 0, r->coder, r->cu), splitDist, g->lambda_rd);
    if (splitCost < singleCost) { *distY += splitDist; *rdCost += splitCost; return; }
    irestore(r, &test);
    set_parts(st->tr_idx, part, nparts, trDepth);
    set_parts(st->cbf[0], part, nparts, (int)(singleCbf << trDepth));
    set_parts(st->tskip[0], part, nparts, best);
    for (int j = 0; j < N; j++)
      memcpy(r->in->rec + (ptrdiff_t)(zy(part) + j) * r->in->rec_stride + zx(part), st->resi[layer][0] + (size_t)(zy(part) + j) * cu + zx(part), sizeof(int16_t) * (size_t)N);
  }
  *distY += singleDist; *rdCost += singleCost;
}

void hop_o_intra_rqt(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn, const hop_o_intra_rqt_in* in, int tr_depth, int part,
                     hop_o_coder* coder, uint8_t cu_ctx[20], hop_o_rqt_state* st, double* cost, uint32_t* dist)
{
  Irq r = { cfg, syn, in, st, coder, cu_ctx };
  intra_node(&r, tr_depth, part, cost, dist);
}

/* ---- row a8: the luma intra search of one CU, TEncSearch::estIntraPredQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:2386-2710) ----
 * Per PU (1 or 4, in order): the most probable modes (TComDataCU::getIntraDirLumaPredictor, TComDataCU.cpp:1772-1830, from the directions
 * left and above: a PU of this CU once it is decided, the caller's value otherwise), the 35-mode rough search and candidate list
 * (:2430-2493), every candidate through the transform tree with bCheckFirst from the CI_CURR_BEST state (:2507-2553), the best one
 * again with the full tree (:2555-2590), the better of the two kept the way xSetIntraResultQT does (levels into the CU layout,
 * reconstruction into the CU's plane, arrays aside); the decided PU's reconstruction goes into the picture unless it is the last
 * (:2603-2660), the cbf of an NxN CU is combined at depth 0 (:2667-2685).  The picture block of the last PU is left as the final
 * pass wrote it, as in the reference. */
static void luma_mpm(int left, int above, int preds[3], int* mode)
{
  if (left == above) {
    *mode = 1;
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = 0; preds[1] = 1; preds[2] = 26; }
  } else {
    *mode = 2;
    preds[0] = left; preds[1] = above;
    preds[2] = (left && above) ? 0 : ((left + above) < 2 ? 26 : 1);
  }
}

void hop_o_intra_luma_search(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn_in, const hop_o_intra_rqt_in* in, const hop_o_intra_search_in* sin,
                             const hop_o_coder* coder_in, const uint8_t cu_ctx_in[20], hop_o_rqt_state* st, int best_dir[4], int32_t* coef_y, int16_t* reco_y,
                             uint32_t* dist_y, int* n_cand_out)
{
  const int cu = 1 << cfg->log2_cu, parts = 1 << (2 * (cfg->log2_cu - 2)), nxn = syn_in->part_nxn ? 1 : 0, npu = nxn ? 4 : 1, d0 = nxn;
  const int q = parts >> (2 * d0), N = cu >> d0;
  hop_o_intra_syntax syn = *syn_in;
  uint32_t overall = 0;
  uint8_t keep_tr[256], keep_cbf[256], keep_ts[256];
  for (int pu = 0; pu < npu; pu++) {
    const int part = pu * q, x = zx(part), y = zy(part);
    /* most probable modes */
    const int left = (nxn && (pu & 1)) ? best_dir[pu - 1] : sin->left_dir[pu], above = (nxn && (pu & 2)) ? best_dir[pu - 2] : sin->above_dir[pu];
    int preds[3], mpm_mode;
    luma_mpm(left, above, preds, &mpm_mode);
    for (int k = 0; k < 3; k++) syn.preds[pu][k] = preds[k];
    syn.pred_num[pu] = 3;
    /* rough search, candidate list */
    uint32_t satd[35], modes[16]; double costs[16];
    hop_o_intra_rough(in->rec, in->rec_stride, in->org, in->org_stride, x, y, N, sin->rough_flags + 68 * pu, cfg->bit_depth_y, in->strong, satd);
    int n = hop_o_intra_cand_list(satd, cu_ctx_in[16], (uint32_t)(coder_in->frac & 32767), sin->sqrt_lambda, preds, 3, mpm_mode, sin->num_full_rd, modes, costs);
    if (n_cand_out) n_cand_out[pu] = n;
    /* candidates with bCheckFirst, then the best one with the full tree */
    double bestCost = 1.7e+308; uint32_t bestDist = 0; int bestMode = 0;
    hop_o_intra_rqt_in rin = *in;
    for (int pass = 0; pass <= n; pass++) {
      const int mode = pass < n ? (int)modes[pass] : bestMode;
      syn.luma_dir[pu] = mode;
      rin.check_first = pass < n ? 1 : 0;
      hop_o_coder coder = *coder_in; uint8_t cuc[20]; memcpy(cuc, cu_ctx_in, 20);
      double cost = 0.0; uint32_t dist = 0;
      hop_o_intra_rqt(cfg, &syn, &rin, d0, part, &coder, cuc, st, &cost, &dist);
      if (cost < bestCost) {
        bestMode = mode; bestDist = dist; bestCost = cost;
        for (int p = part; p < part + q; p++) {                          /* xSetIntraResultQT */
          const int layer = cfg->log2_max_tu - (cfg->log2_cu - st->tr_idx[p]);
          memcpy(coef_y + 16 * p, st->coef[layer][0] + 16 * p, 16 * sizeof(int32_t));
          for (int j = 0; j < 4; j++) memcpy(reco_y + (size_t)(zy(p) + j) * cu + zx(p), st->resi[layer][0] + (size_t)(zy(p) + j) * cu + zx(p), 4 * sizeof(int16_t));
        }
        memcpy(keep_tr + part, st->tr_idx + part, (size_t)q); memcpy(keep_cbf + part, st->cbf[0] + part, (size_t)q); memcpy(keep_ts + part, st->tskip[0] + part, (size_t)q);
      }
    }
    overall += bestDist;
    memcpy(st->tr_idx + part, keep_tr + part, (size_t)q); memcpy(st->cbf[0] + part, keep_cbf + part, (size_t)q); memcpy(st->tskip[0] + part, keep_ts + part, (size_t)q);
    if (pu != npu - 1)
      for (int j = 0; j < N; j++) memcpy(in->rec + (ptrdiff_t)(y + j) * in->rec_stride + x, reco_y + (size_t)(y + j) * cu + x, sizeof(int16_t) * (size_t)N);
    best_dir[pu] = bestMode;
    syn.luma_dir[pu] = bestMode;
  }
  if (npu > 1) {
    uint32_t comb = 0;
    for (int pu = 0; pu < 4; pu++) comb |= (st->cbf[0][pu * q] >> 1) & 1;
    for (int p = 0; p < parts; p++) st->cbf[0][p] |= (uint8_t)comb;
  }
  *dist_y = overall;
}

/* ---- row a8: the chroma intra search of one CU, TEncSearch::estIntraPredChromaQT (TLibEncoder/TEncSearch.cpp:2720-2785) ----
 * The five allowed chroma directions (TComDataCU::getAllowedChromaDir, TComDataCU.cpp:1746-1764), each through xRecurIntraChromaCodingQT (:2130-2277: the chroma
 * blocks of the luma tree's transform units in order - xIntraCodingChromaBlk :1164-1330 = prediction from the chroma reconstruction picture, residual, estBit on
 * the current coder state, transform / RDOQ with the chroma quantiser and lambda, inverse path, reconstruction into the layer plane and the picture, weighted SSE
 * - with the transform-skip retry per component for 4x4 blocks, :2176-2247, whose bit counts (xGetIntraBitsQTChroma :982-1001) move the coder on), then the
 * chroma bits of the CU from the CI_CURR_BEST state (xGetIntraBitsQT, chroma only) and the cost; the best one kept as xSetIntraResultChromaQT keeps it. */
typedef struct {
  const hop_o_rqt_cfg* cfg; const hop_o_intra_syntax* syn; const hop_o_intra_chroma_in* in; hop_o_rqt_state* st;
  hop_o_coder* coder; uint8_t* cu;
} Icq;

static void chroma_blk(Icq* r, int orgDepth, int part, int comp, int tsFlag, uint32_t* dist)
{
  const hop_o_rqt_cfg* g = r->cfg;
  const int parts = 1 << (2 * (g->log2_cu - 2)), log2Luma = g->log2_cu - orgDepth, layer = g->log2_max_tu - log2Luma;
  int d = orgDepth;
  if (log2Luma == 2) d--;
  const int lg = g->log2_cu - d - 1, N = 1 << lg, half = 1 << (g->log2_cu - 1), nparts = parts >> (2 * d);
  const int x = zx(part) >> 1, y = zy(part) >> 1;
  int dir = r->syn->chroma_is_dm ? r->syn->luma_dir[0] : r->syn->chroma_dir;
  int16_t* rec = comp == 1 ? r->in->rec_cb : r->in->rec_cr;
  const int16_t* orgp = comp == 1 ? r->in->org_cb : r->in->org_cr;
  int L[4 * 32 + 1];
  int16_t org[32 * 32], pred[32 * 32], out_rec[32 * 32];
  hop_o_intra_fill_refs_u(rec, r->in->rec_stride, x, y, N, 2, r->in->avail + (size_t)hop_o_intra_node_index(d, g->log2_cu - d, part) * HOP_O_AVAIL_PITCH, g->bit_depth_c, L);
  hop_o_intra_pred_chroma(L, N, dir, g->bit_depth_c, pred);
  for (int j = 0; j < N; j++) memcpy(org + j * N, orgp + (size_t)(y + j) * r->in->org_stride + x, sizeof(int16_t) * (size_t)N);
  uint32_t out[8]; double c;
  int32_t* coef = r->st->coef[layer][comp] + ((16 * part) >> 2);
  hop_o_tu_intra_ts(org, pred, lg, comp, tu_scan(r->syn, parts, part, N, comp), 0, g->qp[comp], g->bit_depth_c, orgDepth, g->sign_hide, g->use_ts, tsFlag,
                    g->lambda_rdoq[comp], g->lambda_rd, g->dist_weight[comp], &r->coder->ctx, 0, coef, out_rec, out, &c);
  set_parts(r->st->cbf[comp], part, nparts, (out[0] ? 1 : 0) << orgDepth);
  for (int j = 0; j < N; j++) {
    memcpy(r->st->resi[layer][comp] + (size_t)(y + j) * half + x, out_rec + j * N, sizeof(int16_t) * (size_t)N);
    memcpy(rec + (ptrdiff_t)(y + j) * r->in->rec_stride + x, out_rec + j * N, sizeof(int16_t) * (size_t)N);
  }
  *dist += out[2];
}

static void chroma_node(Icq* r, int trDepth, int part, uint32_t* dist)
{
  const hop_o_rqt_cfg* g = r->cfg;
  hop_o_rqt_state* st = r->st;
  const int parts = 1 << (2 * (g->log2_cu - 2));
  if (st->tr_idx[part] != trDepth) {
    const int q = (parts >> (2 * trDepth)) >> 2;
    uint32_t su = 0, sv = 0;
    for (int k = 0; k < 4; k++) {
      chroma_node(r, trDepth + 1, part + k * q, dist);
      su |= (st->cbf[1][part + k * q] >> (trDepth + 1)) & 1; sv |= (st->cbf[2][part + k * q] >> (trDepth + 1)) & 1;
    }
    for (int o = 0; o < 4 * q; o++) { st->cbf[1][part + o] |= (uint8_t)(su << trDepth); st->cbf[2][part + o] |= (uint8_t)(sv << trDepth); }
    return;
  }
  const int log2 = g->log2_cu - trDepth;
  int actual = trDepth;
  if (log2 == 2) { actual--; if (part % (parts >> (2 * actual)) != 0) return; }
  const int nparts = parts >> (2 * actual), layer = g->log2_max_tu - log2, half = 1 << (g->log2_cu - 1);
  int ts = g->use_ts && log2 <= 3;
  if (r->in->ts_fast) {
    ts = ts && log2 < 3;
    if (ts) { int nb = 0; for (int p = part; p < part + 4; p++) nb += st->tskip[0][p]; ts = ts && nb > 0; }
  }
  if (!ts) {
    set_parts(st->tskip[1], part, nparts, 0); set_parts(st->tskip[2], part, nparts, 0);
    chroma_blk(r, trDepth, part, 1, 0, dist); chroma_blk(r, trDepth, part, 2, 0, dist);
    return;
  }
  const double MAXD = 1.7e+308;
  hop_o_coder root = *r->coder;
  const int x = zx(part) >> 1, y = zy(part) >> 1;
  for (int comp = 1; comp <= 2; comp++) {
    double single = MAXD; int best = 0; uint32_t sdist = 0, scbf = 0;
    hop_o_coder tbest = root;
    int32_t keepCoef[16]; int16_t keepRec[16];
    int16_t* rec = comp == 1 ? r->in->rec_cb : r->in->rec_cr;
    int32_t* coef = st->coef[layer][comp] + ((16 * part) >> 2);
    for (int mode = 0; mode < 2; mode++) {
      set_parts(st->tskip[comp], part, nparts, mode);
      uint32_t d = 0; double c;
      chroma_blk(r, trDepth, part, comp, mode, &d);
      const uint32_t cbf = (st->cbf[comp][part] >> trDepth) & 1;
      if (mode == 1 && cbf == 0) c = MAXD;
      else {
        reset_bits(r->coder);
        intra_coeff(g, r->syn, st, r->coder, parts, trDepth, part, comp);
        c = hop_o_calc_rd_cost(written(r->coder), d, g->lambda_rd);
      }
      if (c < single) {
        single = c; sdist = d; best = mode; scbf = cbf;
        if (best == 0) {
          memcpy(keepCoef, coef, sizeof(keepCoef));
          for (int j = 0; j < 4; j++) memcpy(keepRec + 4 * j, st->resi[layer][comp] + (size_t)(y + j) * half + x, 8);
          tbest = *r->coder;
        }
      }
      if (mode == 0) *r->coder = root;
    }
    if (best == 0) {
      memcpy(coef, keepCoef, sizeof(keepCoef));
      for (int j = 0; j < 4; j++) {
        memcpy(st->resi[layer][comp] + (size_t)(y + j) * half + x, keepRec + 4 * j, 8);
        memcpy(rec + (ptrdiff_t)(y + j) * r->in->rec_stride + x, keepRec + 4 * j, 8);
      }
      set_parts(st->cbf[comp], part, nparts, (int)(scbf << trDepth));
      *r->coder = tbest;
    }
    set_parts(st->tskip[comp], part, nparts, best);
    *dist += sdist;
    if (comp == 1) root = *r->coder;
  }
}

void hop_o_intra_chroma_search(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn_in, const hop_o_intra_chroma_in* in, const hop_o_coder* coder_in, const uint8_t cu_ctx_in[20],
                               hop_o_rqt_state* st, int* best_mode, uint32_t* best_dist, int32_t* coef_cb, int32_t* coef_cr, int16_t* reco_cb, int16_t* reco_cr)
{
  const int parts = 1 << (2 * (cfg->log2_cu - 2)), half = 1 << (cfg->log2_cu - 1);
  hop_o_intra_syntax syn = *syn_in;
  int list[5] = { 0, 26, 10, 1, 36 };
  for (int i = 0; i < 4; i++) if (syn.luma_dir[0] == list[i]) { list[i] = 34; break; }
  double bestCost = 1.7e+308;
  uint8_t keep[4][256];
  *best_mode = 0; *best_dist = 0;
  for (int m = 0; m < 5; m++) {
    hop_o_coder coder = *coder_in; uint8_t cuc[20]; memcpy(cuc, cu_ctx_in, 20);
    syn.chroma_is_dm = list[m] == 36; syn.chroma_dir = list[m];
    Icq r = { cfg, &syn, in, st, &coder, cuc };
    uint32_t dist = 0;
    chroma_node(&r, 0, 0, &dist);
    if (cfg->use_ts) { coder = *coder_in; memcpy(cuc, cu_ctx_in, 20); }
    const uint32_t bits = hop_o_intra_cu_bits(cfg, &syn, st, 0, 0, 0, 1, &coder, cuc);
    const double cost = hop_o_calc_rd_cost(bits, dist, cfg->lambda_rd);
    if (cost < bestCost) {
      bestCost = cost; *best_dist = dist; *best_mode = list[m];
      for (int p = 0; p < parts; p++) {                                 /* xSetIntraResultChromaQT, :2280-2345 */
        const int d = st->tr_idx[p], log2 = cfg->log2_cu - d, layer = cfg->log2_max_tu - log2;
        int dd = d; if (log2 == 2) dd--;
        const int np = parts >> (2 * dd), first = p - p % np, lgc = cfg->log2_cu - dd - 1, N = 1 << lgc;
        if (p != first) continue;
        for (int comp = 1; comp <= 2; comp++) {
          memcpy((comp == 1 ? coef_cb : coef_cr) + 4 * p, st->coef[layer][comp] + 4 * p, sizeof(int32_t) * (size_t)N * N);
          for (int j = 0; j < N; j++)
            memcpy((comp == 1 ? reco_cb : reco_cr) + (size_t)((zy(p) >> 1) + j) * half + (zx(p) >> 1), st->resi[layer][comp] + (size_t)((zy(p) >> 1) + j) * half + (zx(p) >> 1), sizeof(int16_t) * (size_t)N);
        }
      }
      memcpy(keep[0], st->cbf[1], (size_t)parts); memcpy(keep[1], st->cbf[2], (size_t)parts); memcpy(keep[2], st->tskip[1], (size_t)parts); memcpy(keep[3], st->tskip[2], (size_t)parts);
    }
  }
  memcpy(st->cbf[1], keep[0], (size_t)parts); memcpy(st->cbf[2], keep[1], (size_t)parts); memcpy(st->tskip[1], keep[2], (size_t)parts); memcpy(st->tskip[2], keep[3], (size_t)parts);
}


/* ---- rows a0 / a8: the bits of a finished intra CU as TEncCu::xCheckRDCostIntra counts them (TLibEncoder/TEncCu.cpp:1483-1503) ----
 * resetBits, skip flag, prediction mode, partition size (encodeSkipFlag / encodePredMode / encodePartSize), the luma directions of all PUs and the chroma direction
 * (encodePredInfo: TEncEntropy::encodeIntraDirModeLuma with the grouped flags of codeIntraDirLumaAng - the context-coded bins come in the same order as one PU
 * after the other, so the sum is the same -, encodeIntraDirModeChroma), then encodeCoeff = xEncodeTransform (TEncEntropy.cpp:219-420) on the CU's final levels:
 * split flags, chroma cbfs, luma cbf and the levels of a transform unit interleaved, scans by direction.  No PCM, no transquant bypass, no cu_qp_delta, not an I slice.
 * coef: Y | Cb | Cr in the CU layout.  Returns the bits; the coder afterwards is what the caller stores as CI_TEMP_BEST. */
uint32_t hop_o_intra_cu_total_bits(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, const int32_t* coef, hop_o_coder* coder, uint8_t cu_ctx[20])
{
  Syn s = { coder, cu_ctx };
  coder->frac &= 32767;
  if (y->skip_ctx >= 0) {                                               /* skip_ctx < 0: an I slice */
    bin(&s, CU_SKIP + y->skip_ctx, y->skip_flag ? 1 : 0);
    bin(&s, CU_PRED, 1);
  }
  if (y->is_min_cu) bin(&s, CU_PART, y->part_nxn ? 0 : 1);
  for (int p = 0; p < (y->part_nxn ? 4 : 1); p++) intra_dir(&s, y->luma_dir[p], y->preds[p], y->pred_num[p]);
  if (y->chroma_is_dm) bin(&s, CU_CPRED, 0); else { bin(&s, CU_CPRED, 1); ep(&s, 2); }
  int bak = 0;
  transform_tree(cfg, y, st, coef, coder, 0, 0, cfg->log2_cu, &bak);
  return (uint32_t)(coder->frac >> 15);
}


/* ---- row a8b, the variant without residual: TEncSearch::encodeResAndCalcRdInterCU with bSkipRes (TLibEncoder/TEncSearch.cpp:6635-6668) ----
 * The reconstruction is the prediction; its distortion against the original per plane (getDistPart: the chroma planes weighted), the bits of a set skip flag and
 * the merge index from the CI_CURR_BEST state, calcRdCost.  pred / org: the CU's planes (pitch = CU size, chroma half).  The coder afterwards is CI_TEMP_BEST. */
uint32_t hop_o_inter_cu_skip(const hop_o_rqt_cfg* cfg, int skip_ctx, int merge_idx, int max_merge_cand, const int16_t* const pred[3], const int16_t* const org[3],
                             hop_o_coder* coder, uint8_t cu_ctx[16], uint32_t dist3[3], double* cost)
{
  const int cu = 1 << cfg->log2_cu;
  for (int c = 0; c < 3; c++) {
    const int w = c ? cu >> 1 : cu;
    const uint32_t sse = hop_o_sse(org[c], w, pred[c], w, w, w, c ? cfg->bit_depth_c : cfg->bit_depth_y);
    dist3[c] = c ? (uint32_t)(int)(cfg->dist_weight[c] * sse) : sse;
  }
  Syn s = { coder, cu_ctx };
  coder->frac &= 32767;
  bin(&s, CU_SKIP + skip_ctx, 1);
  merge_index(&s, merge_idx, max_merge_cand);
  const uint32_t bits = (uint32_t)(coder->frac >> 15);
  *cost = hop_o_calc_rd_cost(bits, dist3[0] + dist3[1] + dist3[2], cfg->lambda_rd);
  return bits;
}
