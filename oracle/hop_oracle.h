/* hop_oracle.h -- declarations of the CPU restatement (TEST INFRASTRUCTURE ONLY, see hop_oracle.c). */
#ifndef HOP_ORACLE_H
#define HOP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
uint32_t hop_o_sad(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth, int subShift);
uint32_t hop_o_sse(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth);
uint32_t hop_o_hads(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bitDepth);
uint32_t hop_o_calc_had(const int16_t* a, int sa, const int16_t* b, int sb, int w, int h, int bitDepth);
uint32_t hop_o_component_bits(int v);
uint32_t hop_o_bits_gt(const int v[8]);
void hop_o_ssref_reset(int16_t* bufY, int16_t* bufCb, int16_t* bufCr, int picW, int picH);
void hop_o_ssref_commit_cu(int16_t* y, int16_t* cb, int16_t* cr, int picW, int picH,
                           int x0, int y0, int size, const int16_t* recY, const int16_t* recCb, const int16_t* recCr);
void hop_o_set_search_range(int picW, int picH, int cuX, int cuY, int cuSize, int ctuAddr, int frameWidthInCtu,
                            int predX, int predY, int srchRng, int offX, int offY, int firstRow, int firstCol, int out[6]);
int hop_o_ss_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                    int left, int right, int top, int bottom, int offX, int offY,
                    int predX, int predY, uint32_t lambdaCost, int fen, int bitDepth,
                    int* bestX, int* bestY, uint32_t* sadOut);
uint32_t hop_o_frac_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                           int mvX, int mvY, int predX, int predY, uint32_t lambdaCost, int useHad, int bitDepth,
                           int half[2], int qter[2]);
void hop_o_calc_param_projective(const int x[4], const int y[4], double h[9], int Width, int Height);
void hop_o_calc_param_projective_c(const double x[4], const double y[4], double h[9], int Width, int Height);
void hop_o_projective_transform(const int16_t* refCentre, int16_t* aux, const double h[9], int W, int H, int stride, int nssWindow);
int hop_o_gt_search(const int16_t* org, int orgStride, const int16_t* refPU, int refStride, int w, int h,
                    int mvInt[2], int half[2], int qter[2], const int ssBest[2], int nAmvp, const int* amvpXY,
                    int predX, int predY, uint32_t lambdaCost, int useHad, int bitDepth,
                    uint32_t* cost, int gt[8]);
void hop_o_pred_inter(const int16_t* refY, int strideY, const int16_t* refCb, const int16_t* refCr, int strideC,
                      int puX, int puY, int w, int h, int mvx, int mvy, int useGT, const int gt[8],
                      int bitDepthY, int bitDepthC, int16_t* predY, int16_t* predCb, int16_t* predCr);
long hop_o_warp_counter(int reset);   /* warps evaluated by hop_o_gt_search / hop_o_projective_transform since the last reset */
int hop_o_me_pu(const int16_t* org, int orgStride, const int16_t* refY00, int refStride, int puX, int puY, int w, int h,
                int rngL, int rngR, int rngT, int rngB, int offX, int offY,
                int predX, int predY, int nAmvp, const int* amvpXY, uint32_t lambdaCost,
                int fen, int useHad, int bitDepth, int stage, int64_t* out);
/* ---- a7 (hop_oracle_intra.c): reference samples L[4N+1] = left column bottom-up, corner, above row left-to-right ---- */
void hop_o_intra_fill_refs(const int16_t* rec, int stride, int x, int y, int N, const uint8_t* flags, int bitDepth, int* L);
void hop_o_intra_fill_refs_u(const int16_t* rec, int stride, int x, int y, int N, int unit, const uint8_t* flags, int bitDepth, int* L);
void hop_o_intra_pred_chroma(const int* L, int N, int mode, int bitDepth, int16_t* pred);
void hop_o_intra_smooth(const int* L, int N, int bitDepth, int strong, int* F);
void hop_o_intra_pred(const int* Lunf, const int* Lfil, int N, int mode, int bitDepth, int16_t* pred);
void hop_o_intra_rough(const int16_t* rec, int recStride, const int16_t* org, int orgStride, int x, int y, int N,
                       const uint8_t* flags, int bitDepth, int strong, uint32_t satd[35]);
uint32_t hop_o_intra_mode_bits(uint8_t* ctx_state, uint64_t* frac, int mode, const int preds[3], int pred_num);
int hop_o_cand_update(int mode, double cost, int n, uint32_t* modes, double* costs);
int hop_o_intra_cand_list(const uint32_t satd[35], uint8_t ctx_state, uint32_t frac_left, double sqrt_lambda, const int preds[3], int pred_num, int mpm_cand,
                          int num_full_rd, uint32_t* out_modes, double* out_costs);
/* ---- a9 / a10 (hop_oracle_tq.c) ---- */
void hop_o_fwd_transform(int bitDepth, const int16_t* block, int16_t* coeff, int N, int useDst);
void hop_o_inv_transform(int bitDepth, const int16_t* coeff, int16_t* block, int N, int useDst);
void hop_o_transform_skip(int bitDepth, const int16_t* resi, int32_t* coef, int N);
void hop_o_inv_transform_skip(int bitDepth, const int32_t* coef, int16_t* resi, int N);
uint32_t hop_o_quant_flat(int bitDepth, int qpScaled, int isISlice, const int32_t* coef, int32_t* level, int N);
uint32_t hop_o_tu_roundtrip_sbh(int bitDepth, int qpScaled, int isISlice, int useDst, int transformSkip, int N, int signHide, int scanIdx,
                                const int16_t* org, const int16_t* pred, int32_t* level, int16_t* recon, uint32_t* sse);
uint32_t hop_o_quant_flat_sbh(int bitDepth, int qpScaled, int isISlice, const int32_t* coef, int32_t* level, int N, int scan_idx);   /* + signBitHidingHDQ */
void hop_o_dequant_flat(int bitDepth, int qpScaled, const int32_t* level, int32_t* coef, int N);
/* ---- a11: rate-distortion optimised quantisation (hop_oracle_rdoq.c) ---- */
/* the reference's estBitsSbacStruct (TLibCommon/TComTrQuant.h:59-70), same member order and sizes */
typedef struct {
  int significantCoeffGroupBits[2][2];
  int significantBits[42][2];
  int lastXBits[32];
  int lastYBits[32];
  int greaterOneBits[24][2];
  int levelAbsBits[6][2];
  int blockCbpBits[12][2];
  int blockRootCbpBits[4][2];
} hop_o_estbits;
/* context states (ContextModel::m_ucState = state << 1 | MPS) of the sets residual coding uses, in the reference's set order */
typedef struct {
  uint8_t qt_cbf[8];        /* [luma, chroma][NUM_QT_CBF_CTX] */
  uint8_t trans_subdiv[3];
  uint8_t qt_root_cbf[1];
  uint8_t sig_cg[4];        /* [luma, chroma][2] */
  uint8_t sig[42];          /* 27 luma + 15 chroma */
  uint8_t last_x[30];       /* [luma, chroma][15] */
  uint8_t last_y[30];
  uint8_t one[24];          /* 16 luma + 8 chroma */
  uint8_t abs[6];           /* 4 luma + 2 chroma */
  uint8_t ts[2];            /* transform_skip_flag [luma, chroma] */
} hop_o_cabac_ctx;          /* 150 bytes */
uint8_t hop_o_ctx_init(int qp, int initValue);
int32_t hop_o_ctx_bits(uint8_t state, int bin);
uint8_t hop_o_ctx_next(uint8_t state, int bin);
int hop_o_cabac_init(hop_o_cabac_ctx* c, int slice_type, int qp);
void hop_o_cabac_est_bits(const hop_o_cabac_ctx* c, int width, int comp, hop_o_estbits* eb);
uint64_t hop_o_cabac_coeff_bits(hop_o_cabac_ctx* c, const int32_t* coef, int log2_size, int comp, int scan_idx, int sign_hide, int use_ts, int ts_flag);
uint64_t hop_o_cabac_cbf_bits(hop_o_cabac_ctx* c, int comp, int tr_depth, int cbf);
uint64_t hop_o_cabac_root_cbf_bits(hop_o_cabac_ctx* c, int cbf);
uint64_t hop_o_cabac_subdiv_bits(hop_o_cabac_ctx* c, int ctx, int flag);
double hop_o_calc_rd_cost(uint32_t bits, uint32_t dist, double lambda);
/* ---- a8b: the residual quadtree of one SS/GT ("inter") CU (hop_oracle_rqt.c) ---- */
typedef struct { hop_o_cabac_ctx ctx; uint64_t frac; } hop_o_coder;   /* TEncSbac contexts + TEncBinCABACCounter::m_fracBits */
typedef struct {
  int log2_cu;                      /* 3..6 */
  int qp[3];                        /* what setQPforQuant hands to setQpParam for Y, Cb, Cr */
  int bit_depth_y, bit_depth_c;
  int sign_hide, use_ts;            /* PPS sign_data_hiding, transform_skip_enabled (RDOQTS on) */
  int log2_max_tu, log2_min_tu_in_cu;
  int inter_split_flag;             /* QuadtreeTUMaxDepthInter == 1 && partition != 2Nx2N (TEncSearch.cpp:6831) */
  double lambda_rd, lambda_rdoq[3], dist_weight[3];
} hop_o_rqt_cfg;
typedef struct {                    /* everything the caller reads afterwards; partitions = 4x4 units of the CU in z-order */
  uint8_t tr_idx[256], cbf[3][256], tskip[3][256];
  int32_t* coef[4][3];              /* [layer = log2_max_tu - log2 size][comp]: TU of partition p at 16*p (chroma (16*p)>>2), N x N raster */
  int16_t* resi[4][3];              /* [layer][comp]: reconstructed residual planes of the CU (pitch = CU size, chroma half) */
} hop_o_rqt_state;
void hop_o_rqt(const hop_o_rqt_cfg* cfg, const int16_t* resiY, int strideY, const int16_t* resiCb, const int16_t* resiCr, int strideC,
               hop_o_coder* coder, hop_o_rqt_state* st, double* cost, uint32_t* bits, uint32_t* dist, uint32_t* zero_dist);
void hop_o_rqt_final_coeffs(const hop_o_rqt_cfg* cfg, const hop_o_rqt_state* st, int32_t* out);
typedef struct {
  int part_size;                  /* PartSize: 0 2Nx2N, 1 2NxN, 2 Nx2N, 3 NxN, 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N */
  int n_pu, skip_flag, skip_ctx;  /* isSkipped, getCtxSkipFlag */
  int amp_acc, is_min_cu, max_merge_cand;
  struct { int merge_flag, merge_idx, mvd[2], mvp_idx, gt_flag, gt[8]; } pu[4];
} hop_o_cu_syntax;
uint32_t hop_o_inter_cu_bits(const hop_o_rqt_cfg* cfg, const hop_o_cu_syntax* y, const hop_o_rqt_state* st, const int32_t* coef, hop_o_coder* coder, uint8_t cu_ctx[16],
                             int* skipped);
typedef struct {
  int part_nxn;                    /* 0: 2Nx2N, 1: NxN */
  int skip_flag, skip_ctx, is_min_cu;
  int luma_dir[4], preds[4][3], pred_num[4];   /* per PU: getLumaIntraDir, getIntraDirLumaPredictor */
  int chroma_is_dm, chroma_dir;    /* chroma direction == DM_CHROMA_IDX; otherwise the direction itself (selects the scan) */
} hop_o_intra_syntax;
uint32_t hop_o_intra_cu_bits(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, int tr_depth, int part, int b_luma, int b_chroma,
                             hop_o_coder* coder, uint8_t cu_ctx[20]);
/* row a8: the luma transform tree of one intra PU (xRecurIntraCodingQT, luma only).  st->resi[layer][0] are the reconstruction layer planes
 * (m_pcQTTempTComYuv, pitch = CU size), st->coef[layer][0] the level layers; avail holds the neighbour flags of every node of the CU's quadtree. */
#define HOP_O_AVAIL_PITCH 36
typedef struct {
  const int16_t* org; int org_stride;   /* luma original, at the CU origin */
  int16_t* rec; int rec_stride;         /* reconstruction picture at the CU origin: neighbours read at negative offsets, blocks written as the search goes */
  const uint8_t* avail;                 /* [hop_o_intra_node_index(tr_depth, log2_size, part)][HOP_O_AVAIL_PITCH]: 4 * (size / 4) + 1 flags, bottom-left first */
  int strong;                           /* SPS strong_intra_smoothing */
  int check_first;                      /* bCheckFirst */
  int ts_fast;                          /* TransformSkipFast: transform skip tried in NxN CUs only */
} hop_o_intra_rqt_in;
int hop_o_intra_node_index(int tr_depth, int log2_size, int part);
void hop_o_intra_rqt(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn, const hop_o_intra_rqt_in* in, int tr_depth, int part,
                     hop_o_coder* coder, uint8_t cu_ctx[20], hop_o_rqt_state* st, double* cost, uint32_t* dist);
/* row a8: the luma intra search of one CU (estIntraPredQT, luma only): rough search, candidate list, the candidates through hop_o_intra_rqt, the final pass.
 * coder_in / cu_ctx_in: the CI_CURR_BEST state (every tree starts from it; the coder is unchanged afterwards).  syn: partition, skip flag / context, is_min_cu
 * (directions and predictors are derived here).  coef_y: the CU's luma levels (getCoeffY layout), reco_y: the CU's luma reconstruction (pitch = CU size),
 * st arrays: tr_idx / cbf[0] / tskip[0] of the CU afterwards; best_dir[pu]; n_cand_out[pu] (may be NULL): candidates tested. */
typedef struct {
  int left_dir[4], above_dir[4];  /* per PU: the luma direction getIntraDirLumaPredictor sees left / above when that neighbour lies outside the CU (DC = 1 when it is
                                     unavailable, not intra, or above the CTU row); ignored where the neighbour is a PU of this CU */
  const uint8_t* rough_flags;     /* [PU][68]: neighbour flags of the PU's block (the rough search; 64x64 needs 65) */
  double sqrt_lambda;             /* TComRdCost::getSqrtLambda */
  int num_full_rd;                /* g_aucIntraModeNumFast of the PU size: 8 for 4x4 and 8x8, 3 above */
} hop_o_intra_search_in;
void hop_o_intra_luma_search(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn, const hop_o_intra_rqt_in* in, const hop_o_intra_search_in* sin,
                             const hop_o_coder* coder_in, const uint8_t cu_ctx_in[20], hop_o_rqt_state* st, int best_dir[4], int32_t* coef_y, int16_t* reco_y,
                             uint32_t* dist_y, int* n_cand_out);
/* row a8: the chroma intra search of one CU (estIntraPredChromaQT).  st: tr_idx and tskip[0] as the luma search left them (input), cbf[1..2] / tskip[1..2] afterwards;
 * coef[layer][1..2] / resi[layer][1..2]: level layers and reconstruction layer planes (pitch = half the CU size).  best_mode: 0 / 26 / 10 / 1 / 34 or 36 (DM_CHROMA_IDX);
 * coef_cb / coef_cr: the CU's chroma levels (getCoeffCb/Cr layout: TU of partition p at 4 p), reco_cb / reco_cr: its chroma reconstruction (pitch = half the CU size). */
typedef struct {
  const int16_t* org_cb; const int16_t* org_cr; int org_stride;   /* chroma originals at the CU origin */
  int16_t* rec_cb; int16_t* rec_cr; int rec_stride;               /* chroma reconstruction pictures at the CU origin (read and written as the search goes) */
  const uint8_t* avail;                                           /* the node table of hop_o_intra_rqt_in (the same flags, per 2-sample unit here) */
  int ts_fast;
} hop_o_intra_chroma_in;
void hop_o_intra_chroma_search(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* syn, const hop_o_intra_chroma_in* in, const hop_o_coder* coder_in, const uint8_t cu_ctx_in[20],
                               hop_o_rqt_state* st, int* best_mode, uint32_t* best_dist, int32_t* coef_cb, int32_t* coef_cr, int16_t* reco_cb, int16_t* reco_cr);
/* rows a0 / a8: the bits of a finished intra CU (the counting part of TEncCu::xCheckRDCostIntra): header, directions, xEncodeTransform on the final levels (Y | Cb | Cr) */
uint32_t hop_o_intra_cu_total_bits(const hop_o_rqt_cfg* cfg, const hop_o_intra_syntax* y, const hop_o_rqt_state* st, const int32_t* coef, hop_o_coder* coder, uint8_t cu_ctx[20]);
/* row a8b without residual (encodeResAndCalcRdInterCU, bSkipRes): distortion of the prediction per plane, bits of skip flag + merge index, cost */
uint32_t hop_o_inter_cu_skip(const hop_o_rqt_cfg* cfg, int skip_ctx, int merge_idx, int max_merge_cand, const int16_t* const pred[3], const int16_t* const org[3],
                             hop_o_coder* coder, uint8_t cu_ctx[16], uint32_t dist3[3], double* cost);
int hop_o_inter_cu_finish(const hop_o_rqt_cfg* cfg, hop_o_rqt_state* st, const hop_o_coder* coder, double cost, uint32_t zero_dist,
                          const int16_t* const pred[3], const int16_t* const org[3], int16_t* const rec[3], uint32_t dist3[3], int32_t* final_coef);
int hop_o_tu_rd(const int16_t* resi, int log2_size, int comp, int qp_scaled, int bit_depth, int tr_depth, int sign_hide, int use_ts,
                double lambda_rdoq, double lambda_rd, double dist_weight, const hop_o_cabac_ctx* snap, uint32_t frac_left,
                int32_t* levels, uint32_t* out, double* cost);
int hop_o_tu_intra_ts(const int16_t* org, const int16_t* pred, int log2_size, int comp, int scan_idx, int use_dst, int qp_scaled, int bit_depth,
                      int tr_depth, int sign_hide, int use_ts, int ts_flag, double lambda_rdoq, double lambda_rd, double dist_weight,
                      const hop_o_cabac_ctx* snap, uint32_t frac_left, int32_t* levels, int16_t* recon, uint32_t* out, double* cost);
int hop_o_tu_intra(const int16_t* org, const int16_t* pred, int log2_size, int comp, int scan_idx, int use_dst, int qp_scaled, int bit_depth,
                   int tr_depth, int sign_hide, int use_ts, double lambda_rdoq, double lambda_rd, double dist_weight,
                   const hop_o_cabac_ctx* snap, uint32_t frac_left, int32_t* levels, int16_t* recon, uint32_t* out, double* cost);
void hop_o_scan_init(void);
const uint32_t* hop_o_scan(int scan_idx, int log2_size);
const uint32_t* hop_o_scan_cg(int scan_idx, int log2_size);
int hop_o_coef_scan_idx(int width, int is_luma, int is_intra, int dir);
int hop_o_rdoq(const int32_t* src, int32_t* dst, int log2_size, int comp, int is_intra, int scan_idx, int tr_depth,
               int qp_scaled, int bit_depth, int sign_hide, double lambda, const hop_o_estbits* eb, uint32_t* abs_sum);

/* ---- after the search: the deblocking filter, hop_oracle_lf.c ---- */
/* one 4x4 unit of a finished picture: the layout of hop_cu_part (include/hophip.h) */
typedef struct {
  uint8_t depth, pred_mode, part_size, skip, merge_flag, merge_idx, gt_flag, inter_dir;
  int8_t  ref_idx, mvp_idx, mvp_num; uint8_t luma_dir, chroma_dir, tr_idx;
  uint8_t cbf[3], tskip[3];
  int16_t mv[2], mvd[2], gt[8];
} hop_o_cu_part;
int hop_o_deblock_frame(int w, int h, int bit_depth, int qp, int beta_offset_div2, int tc_offset_div2, int cb_qp_offset, int cr_qp_offset, int disable,
                        const hop_o_cu_part* parts, int16_t* y, int16_t* cb, int16_t* cr);

/* ---- sample adaptive offset, hop_oracle_sao.c: the layout of hop_sao_param (include/hophip.h) ---- */
typedef struct { int8_t mode, type, aux, pad; int8_t offset[32]; } hop_o_sao_param;
int hop_o_sao_stats(int w, int h, int bit_depth, const int16_t* const src[3], const int16_t* const org[3], int32_t* stats);
int hop_o_sao_apply(int w, int h, int bit_depth, const int16_t* const src[3], const hop_o_sao_param* params, int16_t* const dst[3]);

#ifdef __cplusplus
}
#endif
#endif
