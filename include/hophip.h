/*
 * hophip.h -- C ABI of libhophip.so: the MI355X (gfx950) implementation of the HEVC-HOP hot path.
 *
 * The reference (zinsayon/HEVC-HOP, an HM-15.0 fork) has no plugin/FFI layer; its "boundary" is the
 * C++ surface of TEncSearch / TComPrediction / TComRdCost / TEncCu.
 * Each entry point below names the reference interface it replaces (paths relative to
 * source/Lib in the reference tree); INTEGRATION.md shows the C++ shim a maintainer adds.
 *
 * Conventions
 *   - Pel = int16_t, strides in elements, coordinates in luma samples of the picture.
 *   - All pictures live in HBM inside the context: the original (no margins) and the
 *     self-similarity ("SS") reference in the reference's own TComPicYuv layout (margins 80 luma /
 *     40 chroma, TLibCommon/TComPicYuv.cpp:82-85; sentinel NOT_VALID = -1, CommonDef.h:126).
 *   - Host-array entry points are synchronous: the caller owns every pointer it passes and the
 *     library keeps none of them after return.  *_device entry points take device pointers and
 *     are asynchronous on the context stream (hop_sync to wait); they exist so that a caller
 *     that already lives on the GPU (and bench.py) does not pay PCIe per batch.
 *   - Every function returns HOP_OK (0) or a negative hop_status; hop_last_error() has the text.
 *     "No valid SS candidate" is not an error: it is reported per PU through
 *     hop_pu_result.not_valid and sad = 0xFFFFFFFF, the sentinels the reference uses
 *     (TLibEncoder/TEncSearch.cpp:6356-6360, :4603-4611).
 */
#ifndef HOPHIP_H
#define HOPHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hop_ctx hop_ctx;

typedef enum {
  HOP_OK = 0,
  HOP_ERR_ARG = -1,       /* bad argument (shape not on the reference's PU list, NULL, out of picture) */
  HOP_ERR_DEVICE = -2,    /* HIP runtime error (no gfx950 device, allocation, launch) */
  HOP_ERR_STATE = -3      /* call order (e.g. search before hop_upload_orig) */
} hop_status;

/* ---- stage selectors for hop_me_search ---- */
#define HOP_STAGE_INT  1  /* SS integer full search only              (xPatternSearch)            */
#define HOP_STAGE_FRAC 2  /* + half/quarter-pel refinement            (xPatternSearchFracDIF)     */
#define HOP_STAGE_GT   3  /* + GT/HOP 4-corner diamond search         (xPatternSearchGT)          */

/* ---- job flags ---- */
#define HOP_FLAG_FEN     1  /* FEN: row-subsampled SAD for PUs taller than 8 (TEncSearch.cpp:6303-6309) */
#define HOP_FLAG_HADME   2  /* HadamardME: HAD cost in fractional + GT search (TEncSearch.cpp:719,4772) */

/* One prediction unit through the ME chain of TEncSearch::xMotionEstimation
 * (TLibEncoder/TEncSearch.cpp:4479-4683).  The search range and offX'/offY' are the values AFTER
 * both xSetSearchRange overloads (:6204-6259); hop_set_search_range() computes them. */
typedef struct {
  int32_t  pu_x, pu_y, w, h;                       /* PU rectangle in the picture */
  int32_t  rng_left, rng_right, rng_top, rng_bottom; /* integer-pel search range, inclusive */
  int32_t  off_x, off_y;                           /* causality offsets after :6239-6240 */
  int32_t  pred_x, pred_y;                         /* MV predictor, quarter-pel (setPredictor :4559) */
  uint32_t lambda_cost;                            /* m_uiLambdaMotionSAD = floor(65536*sqrt(lambda)), TComRdCost.cpp:171 */
  int32_t  n_amvp;                                 /* AMVP candidates used as extra GT start vectors (:5100-5105), 0..2 */
  int32_t  amvp[4];                                /* (x,y) quarter-pel each */
  int32_t  flags;                                  /* HOP_FLAG_* */
} hop_pu_job;

typedef struct {
  int32_t  mv_int[2];      /* integer MV from the SS search (rcMv after xPatternSearch) */
  uint32_t sad;            /* its SAD with the MV cost removed (:6365); 0xFFFFFFFF if none valid */
  int32_t  not_valid;      /* bNotValCU condition (:4603-4611) */
  int32_t  half[2];        /* after xPatternSearchFracDIF */
  int32_t  qter[2];
  uint32_t frac_cost;
  int32_t  gt_flag;        /* after xPatternSearchGT */
  int32_t  gt[8];          /* GT0..GT3 (x,y) */
  uint32_t cost;           /* ruiCost after the last executed stage */
  int32_t  mv_final[2];    /* integer MV after GT (rewritten when GT wins, :5455-5457) */
  int32_t  half_final[2];
  int32_t  qter_final[2];
} hop_pu_result;

/* One PU for the final (normative, decoder-shared) predictor:
 * TComPrediction::xPredInterLumaBlk / xPredInterChromaBlk incl. the GT branch
 * (TLibCommon/TComPrediction.cpp:639-720, :723-805, :1235-1420). */
typedef struct {
  int32_t pu_x, pu_y, w, h;
  int32_t mv_x, mv_y;      /* quarter-pel */
  int32_t use_gt;
  int32_t gt[8];
  int32_t dst_row_off;     /* the prediction is written dst_row_off luma rows below the PU (candidate slots, hop_ctx_set_slots: k * picture height); 0: at the PU */
} hop_pred_job;

/* distortion kinds for hop_distortion (TLibCommon/TComRdCost.cpp) */
#define HOP_DIST_SAD  0    /* xGetSAD*            :513-1011  */
#define HOP_DIST_SSE  1    /* xGetSSE*            :1018-1360 */
#define HOP_DIST_HADS 2    /* xGetHADs            :1641-1708 */

typedef struct {
  int32_t x, y, w, h;      /* luma rectangle; comp 1/2 use (x/2,y/2,w/2,h/2) */
  int32_t comp;            /* 0 Y, 1 Cb, 2 Cr */
  int32_t kind;            /* HOP_DIST_* */
} hop_dist_job;

/* ---- context ---- */
/* replaces: TEncTop::create/init wiring of m_cSearch/m_cRdCost (TLibEncoder/TEncTop.cpp:89-101,299-310)
 * and TComPicYuv::create for the SS reference (TLibCommon/TComPicYuv.cpp:69-120). */
int hop_ctx_create(hop_ctx** out, int pic_w, int pic_h, int bit_depth_y, int bit_depth_c, int device);
void hop_ctx_destroy(hop_ctx* ctx);
/* A second handle on the same resident pictures with its own stream and work areas (no counterpart in the single-threaded reference): requests issued through different views
 * run concurrently on the device; the caller keeps their rectangles apart.  Views are destroyed before their parent. */
int hop_ctx_create_view(hop_ctx* parent, hop_ctx** out);
const char* hop_last_error(const hop_ctx* ctx);     /* ctx may be NULL: error of the failed create */
int hop_sync(hop_ctx* ctx);
void* hop_stream(hop_ctx* ctx);                     /* hipStream_t the *_device calls are ordered on */

/* Candidate slots (no counterpart in the reference, whose candidates run one after the other through m_ppcPredYuvTemp / m_ppcRecoYuvTemp): the original, prediction and
 * reconstruction pictures exist slots + 1 times, copy k at rows k * pic_h.  A request whose y is y + k * pic_h works on copy k, so candidates of one CU evaluated side by
 * side do not overwrite each other; the searches and the predictor read the one SS reference at the true position (hop_pred_job.dst_row_off names the copy the
 * prediction goes to).  Call before hop_upload_orig; hop_encode_frame then evaluates the candidates of a CU side by side: from 16 slots the merge / SS / GT / intra
 * candidates of TEncCu::xCompressCU's first pass (TEncCu.cpp:451-637), from 24 also the AMP shapes in both forms deriveTestModeAMP (:292-356) can ask for, from 48 in levels
 * of 24 also the candidates of the CU's first sub-CU (72, 96: of the chain of first sub-CUs below it) -- what the reference would not have tested is dropped, its decisions
 * are replayed in its order, the results do not depend on the number.  0 <= slots <= 128. */
int hop_ctx_set_slots(hop_ctx* ctx, int slots);

/* The transform-unit leaf step (hop_tu_rd and everything built on it) has two forms with identical results: a pipeline of 13 kernels that lays the serial stages out one
 * lane per TU (batches of thousands of TUs) and one kernel with a workgroup per TU (the launch-bound batches of the RD search).  Batches of up to max_tus TUs take the
 * one-kernel form (default 8192, or the environment's HOP_FUSED_LEAF); 0 = always the pipeline. */
int hop_set_fused_leaf(hop_ctx* ctx, int max_tus);

/* Several independent pictures of equal size in ONE context (no counterpart in the reference, which codes one picture at a time: the pictures of an all-intra sequence do
 * not depend on each other, TAppEncTop.cpp:432-520 codes them in a loop).  The context is created with the height of the stack: picture k occupies rows
 * k * sub_pitch .. k * sub_pitch + sub_h - 1, sub_pitch = hop_stack_pitch(sub_h) or more (a multiple of 64), pic_h = (n - 1) * sub_pitch + sub_h; the rows between the
 * pictures are never read.  Every entry point keeps taking coordinates in the stack; hop_encode_frame codes the pictures side by side, one batch of requests serving CTUs
 * of all of them, each exactly as in a context of its own.  sub_h = sub_pitch = 0 returns to one picture. */
int hop_stack_pitch(int sub_h);
int hop_ctx_set_stack(hop_ctx* ctx, int sub_h, int sub_pitch);

/* ---- picture residency ---- */
/* replaces: the host copy of the original into TComPic (TLibEncoder/TEncTop.cpp:363-368) */
int hop_upload_orig(hop_ctx* ctx, const int16_t* y, int stride_y, const int16_t* cb, const int16_t* cr, int stride_c);
/* replaces: TComSlice::xGetRefPic sentinel fill (TLibCommon/TComSlice.cpp:241-255 -> TComPicYuv::setPicPel :199-207) */
int hop_ssref_reset(hop_ctx* ctx);
/* replaces: TEncCu::xCopyYuv2SSRef leaf (TLibEncoder/TEncCu.cpp:1677-1697) incl. the border re-extension
 * (TLibCommon/TComPicYuv.cpp:236-275), done incrementally for the margins the CU can change.
 * n CUs; rect[i] = {x, y, size, 0}; rec_y/cb/cr: n contiguous blocks of size^2 / (size/2)^2 samples. */
int hop_ssref_commit_cus(hop_ctx* ctx, int n, const int32_t* rect4, const int16_t* rec_y, const int16_t* rec_cb, const int16_t* rec_cr);
/* commit straight from a device-resident reconstruction picture (planes without margins, pitch = pic_w / pic_w/2) */
int hop_ssref_commit_cus_device(hop_ctx* ctx, int n, const int32_t* d_rect4, const int16_t* d_rec_y, const int16_t* d_rec_cb, const int16_t* d_rec_cr);
/* whole padded plane (comp 0: (pic_h+160) x (pic_w+160); 1/2: (pic_h/2+80) x (pic_w/2+80)) -- tests, debugging, checkpoints */
int hop_ssref_download(hop_ctx* ctx, int comp, int16_t* dst);
int hop_ssref_upload(hop_ctx* ctx, int comp, const int16_t* src);

/* ---- host logic ---- */
/* replaces: TEncSearch::xSetSearchRange (both overloads, TLibEncoder/TEncSearch.cpp:6204-6259) with
 * TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3492-3504).  out = {left,right,top,bottom,offX',offY'} */
void hop_set_search_range(int pic_w, int pic_h, int cu_x, int cu_y, int cu_size, int ctu_addr, int frame_width_in_ctu,
                          int pred_x, int pred_y, int search_range, int off_x, int off_y, int first_row, int first_col, int out[6]);
/* replaces: TComRdCost::xGetComponentBits / getBitsGT (TLibCommon/TComRdCost.cpp:270-284, TComRdCost.h:205-215) */
uint32_t hop_component_bits(int v);
uint32_t hop_bits_gt(const int v[8]);
/* replaces: the bits/cost bookkeeping at the tail of xMotionEstimation (TEncSearch.cpp:4654-4682):
 * folds one hop_pu_result into (mv quarter-pel, bits, cost); bits_in = ruiBits on entry. */
void hop_me_finish(const hop_pu_job* job, const hop_pu_result* res, int stage, uint32_t bits_in,
                   int mv_qpel[2], uint32_t* bits_out, uint32_t* cost_out);

/* single bins of the counting coder, for the split_cu_flag the RD spine counts itself (host only):
 * replaces: TEncBinCABACCounter::encodeBin (TLibEncoder/TEncBinCoderCABACCounter.cpp:72-78) = ContextModel::getEntropyBits + update (TLibCommon/ContextModel.h:78,
 * ContextModel.cpp:67-128): returns the fractional bits (15 binary places) of coding `bin` in *state and moves the state on; encodeBinTrm (:104-108) for the
 * end_of_slice_segment_flag; ContextModel3DBuffer::initBuffer with INIT_SPLIT_FLAG (TLibCommon/ContextTables.h:126-136), slice_type as hop_cabac_init. */
uint32_t hop_cabac_bin_bits(uint8_t* state, int bin);
uint32_t hop_cabac_trm_bits(int bin);
int hop_cabac_split_init(uint8_t split_ctx[3], int slice_type, int qp);

/* ---- the hot path ---- */
/* replaces: TEncSearch::xPatternSearch (:6262-6371) + xPatternSearchFracDIF (:6564-6610) +
 * xPatternSearchGT (:4686-5467) for a batch of PUs; the original block is read from the resident
 * original picture at (pu_x, pu_y). */
int hop_me_search(hop_ctx* ctx, int n, const hop_pu_job* jobs, hop_pu_result* results, int stage);
int hop_me_search_device(hop_ctx* ctx, int n, const hop_pu_job* d_jobs, hop_pu_result* d_results, int stage);

/* replaces: TComPrediction::motionCompensation -> xPredInterLumaBlk/xPredInterChromaBlk
 * (TLibCommon/TComPrediction.cpp:419-528, :639-720, :1235-1347).  Predictions are written into the
 * context's prediction picture at the PU position; out_* (may be NULL) additionally receive them
 * packed job after job (w*h luma, (w/2)*(h/2) per chroma plane). */
int hop_pred_inter(hop_ctx* ctx, int n, const hop_pred_job* jobs, int16_t* out_y, int16_t* out_cb, int16_t* out_cr);
int hop_pred_inter_device(hop_ctx* ctx, int n, const hop_pred_job* d_jobs);
int hop_pred_download(hop_ctx* ctx, int comp, int16_t* dst);   /* whole prediction plane, pitch pic_w (/2) */
/* device-side glue between the two calls above: what predInterSearch stores into the CU's MV/GT fields
 * (TLibEncoder/TEncSearch.cpp:3885-3922) before motionCompensation reads them (:4158).  out[i] is the predictor
 * job of PU d_index[i] (d_index == NULL: PU i): mv = (mv_final<<2) + (half_final<<1) + qter_final, the GT vectors
 * and use_gt = 1; a PU whose search was not valid becomes a zero-vector copy. */
int hop_pred_jobs_from_results_device(hop_ctx* ctx, int n, const int32_t* d_index, const hop_pu_job* d_jobs,
                                      const hop_pu_result* d_results, hop_pred_job* d_out);

/* replaces: TComRdCost::getDistPart / DistParam::DistFunc between the original and the prediction picture
 * (TLibCommon/TComRdCost.cpp:477-503). out[i] = distortion of job i. */
int hop_distortion(hop_ctx* ctx, int n, const hop_dist_job* jobs, uint32_t* out);
int hop_distortion_device(hop_ctx* ctx, int n, const hop_dist_job* d_jobs, uint32_t* d_out);            /* asynchronous, unchecked */

/* ---- transform / quantisation round trip (rows a9, a10) and intra rough search (row a7) ---- */
/* One transform unit: residual = original - prediction picture, forward DCT (DST-VII for use_dst on 4x4) or
 * transform skip, flat quantisation, dequantisation, inverse transform, reconstruction clip into the context's
 * reconstruction picture, SSE against the original.
 * replaces: TComTrQuant::transformNxN / invtransformNxN (TLibCommon/TComTrQuant.cpp:1204-1283) with xTrMxN/xITrMxN
 * (:786-863), the non-RDOQ branch of xQuant (:1071-1116, flat scaling list, with signBitHidingHDQ :868-990 if sign_hide) and xDeQuant
 * (:1124-1183), in the order TEncSearch::xIntraCodingLumaBlk uses them (TLibEncoder/TEncSearch.cpp:1082-1160).
 * qp_scaled is what setQPforQuant hands to setQpParam (:192-214); is_i_slice selects the rounding offset 171/85
 * (an ISS slice is NOT an I slice, :1079). */
typedef struct {
  int32_t x, y;            /* luma position of the TU (chroma planes use x/2, y/2) */
  int32_t comp;            /* 0 Y, 1 Cb, 2 Cr */
  int32_t log2_size;       /* 2..5 in samples of the plane */
  int32_t use_dst;         /* 4x4 intra luma: DST-VII (uiMode != REG_DCT, TComTrQuant.cpp:795-799) */
  int32_t transform_skip;
  int32_t qp_scaled;
  int32_t is_i_slice;
  int32_t sign_hide;       /* the PPS's sign_data_hiding flag: TComTrQuant::signBitHidingHDQ (TLibCommon/TComTrQuant.cpp:868-990, called :1110-1116) after the quantiser */
  int32_t scan_idx;        /* 0 diagonal, 1 horizontal, 2 vertical: the scan the hiding walks (getCoefScanIdx) */
} hop_tu_job;
typedef struct { uint32_t abs_sum; uint32_t sse; } hop_tu_result;
/* levels_out (may be NULL): quantised levels of job i at levels_out[sum_{k<i} size_k^2 ...] (TCoeff = int32) */
int hop_tu_roundtrip(hop_ctx* ctx, int n, const hop_tu_job* jobs, hop_tu_result* results, int32_t* levels_out);
/* the same on device-resident job / result arrays (asynchronous on the context stream; the jobs are not range-checked).
 * d_levels may be NULL; otherwise d_level_offsets[i] = first TCoeff of job i in d_levels. */
int hop_tu_roundtrip_device(hop_ctx* ctx, int n, const hop_tu_job* d_jobs, hop_tu_result* d_results, int32_t* d_levels,
                            const int64_t* d_level_offsets);
/* whole reconstruction planes, pitch pic_w (/2): neighbours of the intra search, output of hop_tu_roundtrip */
int hop_recon_upload(hop_ctx* ctx, int comp, const int16_t* src);
int hop_pred_upload(hop_ctx* ctx, int comp, const int16_t* src);   /* e.g. an intra prediction made elsewhere */
int hop_recon_download(hop_ctx* ctx, int comp, int16_t* dst);

/* One luma block of the 35-mode rough search.  flags[u] = availability of 4-sample neighbour unit u in the
 * reference's bNeighborFlags order (TLibCommon/TComPattern.cpp:202-210): u = 0 bottom-most below-left unit ...
 * 2*size/4 - 1 top-most left unit, 2*size/4 the corner, then above and above-right left to right.
 * replaces: initAdiPattern + predIntraLumaAng + calcHAD of TEncSearch::estIntraPredQT (TEncSearch.cpp:2430-2458);
 * satd_out[35*i + mode]; the caller adds the mode bits (:2460-2461). */
typedef struct {
  int32_t x, y, size;      /* size 4..64 */
  int32_t strong;          /* SPS strong_intra_smoothing */
  uint8_t flags[68];
} hop_intra_job;
int hop_intra_rough(hop_ctx* ctx, int n, const hop_intra_job* jobs, uint32_t* satd_out);
int hop_intra_rough_device(hop_ctx* ctx, int n, const hop_intra_job* d_jobs, uint32_t* d_satd);   /* asynchronous, unchecked */
/* replaces: initAdiPattern + predIntraLumaAng of TEncSearch::xIntraCodingLumaBlk (TLibEncoder/TEncSearch.cpp:1043-1049): the prediction
 * of mode modes[i] (0 planar, 1 DC, 2..34 angular) of block i, written into the context's prediction picture (luma). */
int hop_intra_pred(hop_ctx* ctx, int n, const hop_intra_job* jobs, const int32_t* modes);
int hop_intra_pred_device(hop_ctx* ctx, int n, const hop_intra_job* d_jobs, const int32_t* d_modes);     /* asynchronous, unchecked */
/* replaces: initAdiPatternChroma + predIntraChromaAng of TEncSearch::xIntraCodingChromaBlk (TLibEncoder/TEncSearch.cpp:1200-1215; TComPattern.cpp:315-372,
 * TComPrediction.cpp:375-390) for both chroma planes: block i has size jobs[i].size (4..32) at the chroma position (x / 2, y / 2), flags give the availability
 * per 2-sample unit in the same order as for luma (68 entries are enough: 2 * size + 1), mode modes[i] is the direction actually used (the caller resolves
 * DM_CHROMA_IDX to the luma direction).  No reference smoothing, no edge filters.  Written into the context's prediction picture (Cb, Cr). */
int hop_intra_pred_chroma(hop_ctx* ctx, int n, const hop_intra_job* jobs, const int32_t* modes);
int hop_intra_pred_chroma_device(hop_ctx* ctx, int n, const hop_intra_job* d_jobs, const int32_t* d_modes);   /* asynchronous, unchecked */

/* ---- rate-distortion optimised quantisation (row a11) ---- */
/* Image of the reference's estBitsSbacStruct (TLibCommon/TComTrQuant.h:59-70): the bit estimates (15 fractional bits)
 * TEncSbac::estBit (TLibEncoder/TEncSbac.cpp:2175-2290) derives from the CABAC context states before a TU is tested. */
typedef struct {
  int32_t significantCoeffGroupBits[2][2];
  int32_t significantBits[42][2];
  int32_t lastXBits[32];
  int32_t lastYBits[32];
  int32_t greaterOneBits[24][2];
  int32_t levelAbsBits[6][2];
  int32_t blockCbpBits[12][2];
  int32_t blockRootCbpBits[4][2];
} hop_estbits;
typedef struct {
  int32_t log2_size;       /* 2..5 (chroma: 2..4, the reference has no chroma 32x32 tables) */
  int32_t comp;            /* 0 Y, 1 Cb, 2 Cr */
  int32_t is_intra;        /* pcCU->isIntra */
  int32_t scan_idx;        /* TComDataCU::getCoefScanIdx (TLibCommon/TComDataCU.cpp:4001-4056): 0 diagonal, 1 horizontal, 2 vertical */
  int32_t tr_depth;        /* pcCU->getTransformIdx */
  int32_t qp_scaled;       /* what setQPforQuant hands to setQpParam (TComTrQuant.cpp:192-214) */
  int32_t bit_depth;       /* of the component */
  int32_t sign_hide;       /* PPS sign_data_hiding */
  double  lambda;          /* m_dLambda after selectLambda (TComTrQuant.h:153) */
  int64_t coeff_offset;    /* first coefficient of this TU in src / dst (raster N x N, Int / TCoeff) */
  int32_t estbits_index;   /* which table of `tables` */
  int32_t reserved;
} hop_rdoq_job;
/* replaces: TComTrQuant::xRateDistOptQuant (TLibCommon/TComTrQuant.cpp:1489-1999, flat scaling list) for a batch of TUs:
 * src = transform coefficients, dst = levels with sign, abs_sum[i] = the uiAbsSum the call adds for TU i. */
int hop_rdoq(hop_ctx* ctx, int n, const hop_rdoq_job* jobs, int n_tables, const hop_estbits* tables, size_t n_coeff,
             const int32_t* src, int32_t* dst, uint32_t* abs_sum);
int hop_rdoq_device(hop_ctx* ctx, int n, const hop_rdoq_job* d_jobs, const hop_estbits* d_tables, const int32_t* d_src,
                    int32_t* d_dst, uint32_t* d_abs_sum);                                           /* asynchronous, unchecked */

/* ---- CABAC bit estimator for residual coding (building block of rows a0 / a8 / a8b; feeds a11) ---- */
/* Context states (ContextModel::m_ucState = state << 1 | MPS) of the sets residual coding uses, in the reference's set
 * order (TLibEncoder/TEncSbac.cpp:76-88): qt_cbf[2][4] at 0, trans_subdiv[3] at 8, qt_root_cbf[1] at 11, sig_cg[2][2] at 12,
 * sig[27 luma + 15 chroma] at 16, last_x[2][15] at 58, last_y[2][15] at 88, one[16 + 8] at 118, abs[4 + 2] at 142,
 * transform_skip[2] at 148; state[150..151] = the fraction below one bit the counting coder carries (m_fracBits & 32767,
 * little endian): TEncBinCABAC::resetBits keeps it, so integer bit counts depend on it. */
typedef struct { uint8_t state[152]; } hop_cabac_ctx;
/* replaces: TEncSbac::resetEntropy for those sets (ContextModel3DBuffer::initBuffer + ContextModel::init,
 * TLibEncoder/TEncSbac.cpp:136-148, TLibCommon/ContextModel.cpp:56-65, tables TLibCommon/ContextTables.h:340-546).
 * slice_type: 0 B, 1 P, 2 I, 3 ISS, 4 PSS (TLibCommon/TypeDef.h:418-427). Host only. */
int hop_cabac_init(hop_cabac_ctx* ctx, int slice_type, int qp);
/* replaces: TEncSbac::estBit as TEncEntropy::estimateBit calls it (TLibEncoder/TEncSbac.cpp:2175-2370, TEncEntropy.cpp:669-674):
 * the entries of the table the reference writes for this (width, component); the others are left as they are. Host only. */
int hop_cabac_est_bits(const hop_cabac_ctx* ctx, int width, int comp, hop_estbits* est);
typedef struct {
  int32_t log2_size;       /* 2..5 (chroma 2..4) */
  int32_t comp;            /* 0 Y, 1 Cb, 2 Cr */
  int32_t scan_idx;        /* getCoefScanIdx */
  int32_t sign_hide;       /* PPS sign_data_hiding && !cu_transquant_bypass */
  int32_t use_ts;          /* PPS transform_skip_enabled */
  int32_t ts_flag;         /* the TU's transform_skip_flag */
  int32_t ctx_index;       /* which context snapshot the TU starts from */
  int32_t cbf_ctx_plus1;   /* 0: levels only; else 1 + index of the coded_block_flag context in qt_cbf[8] (component class * 4 +
                              getCtxQtCbf): the flag (= any level non-zero) is coded first, as encodeQtCbf + encodeCoeffNxN do */
  int64_t coeff_offset;    /* first level of the TU in coef (raster N x N, TCoeff) */
} hop_coeff_bits_job;
/* replaces: TEncSbac::codeCoeffNxN (TLibEncoder/TEncSbac.cpp:1829-2092) driven through the counting bin coder
 * (TEncBinCoderCABACCounter.cpp:72-108) for a batch of TUs: bits[i] = fractional bits (15 binary places, the coder's
 * m_fracBits) of the levels of TU i; ctx_out (may be NULL) receives the context states after TU i. */
int hop_coeff_bits(hop_ctx* ctx, int n, const hop_coeff_bits_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, size_t n_coeff,
                   const int32_t* coef, uint64_t* bits, hop_cabac_ctx* ctx_out);
int hop_coeff_bits_device(hop_ctx* ctx, int n, const hop_coeff_bits_job* d_jobs, const hop_cabac_ctx* d_ctx_in, const int32_t* d_coef,
                          uint64_t* d_bits, hop_cabac_ctx* d_ctx_out);                                /* asynchronous, unchecked */

/* ---- residual quadtree, leaf step (row a8b) ---- */
/* One component TU of TEncSearch::xEstimateResidualQT (TLibEncoder/TEncSearch.cpp:6896-7200), default transform:
 * residual = original - prediction picture -> xT -> estBit from the snapshot -> xRateDistOptQuant -> bits of cbf flag + levels
 * from the snapshot (integer, as getNumberOfWrittenBits) -> xDeQuant + xIT -> distortion in the residual domain -> the
 * cbf-zero decision on TComRdCost::calcRdCost values (:7008-7032).  flags select the 4x4 transform-skip variant of the retry
 * (:7210-7440); the retry's decision, the split recursion and the subtree recount are hop_rqt, the root-cbf decision hop_rqt_finish. */
#define HOP_TU_RD_TS   1   /* the 4x4 transform-skip variant (residual quadtree :7210-7440; intra retry of xRecurIntraCodingQT :1431-1470): xTransformSkip /
                              xITransformSkip instead of the transform, transform_skip_flag = 1 in the bits */
#define HOP_TU_RD_KEEP 2   /* no cbf-zero decision: levels, bits and distortion of coding the block are returned as they are */
typedef struct {
  int32_t x, y;            /* luma position of the TU (chroma planes use x/2, y/2) */
  int32_t comp;            /* 0 Y, 1 Cb, 2 Cr */
  int32_t log2_size;       /* 2..5 (chroma 2..4) */
  int32_t qp_scaled;
  int32_t tr_depth;        /* uiTrMode */
  int32_t ctx_index;       /* context snapshot (CI_QT_TRAFO_ROOT) */
  int32_t sign_hide, use_ts;
  int32_t bit_depth;       /* of the component (must equal the context's) */
  int32_t is_intra;        /* 0: the leaf of xEstimateResidualQT described above.  1: the leaf of xIntraCodingLumaBlk / ChromaBlk after the
                              prediction (TEncSearch.cpp:1082-1160): intra RDOQ tables, no cbf-zero decision, reconstruction = clip(prediction +
                              residual) written to the reconstruction picture, dist = distortion of it against the original; bits = cbf flag +
                              levels from the snapshot (what xGetIntraBitsQT counts for the block) */
  int32_t scan_idx;        /* getCoefScanIdx; 0 unless is_intra */
  int32_t use_dst;         /* 4x4 intra luma: DST-VII */
  int32_t flags;           /* HOP_TU_RD_* */
  double  lambda_rdoq;     /* TComTrQuant::m_dLambda after selectLambda */
  double  lambda_rd;       /* TComRdCost::m_dLambda */
  double  dist_weight;     /* chroma distortion weight of getDistPart (TLibCommon/TComRdCost.cpp:493-497); ignored for luma */
} hop_tu_rd_job;
typedef struct {
  uint32_t abs_sum, cbf;   /* after the decision */
  uint32_t dist;           /* distortion of the choice */
  uint32_t zero_dist, nonzero_dist;
  uint32_t bits, null_bits;/* uiSingleBits, uiNullBits (integer) */
  uint32_t pad;
  double   cost;           /* calcRdCost of the choice */
} hop_tu_rd_result;
/* levels_out: final levels of TU i at levels_out[sum_{k<i} size_k^2 ...] */
int hop_tu_rd(hop_ctx* ctx, int n, const hop_tu_rd_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, hop_tu_rd_result* results, int32_t* levels_out);
/* d_coef_offsets[i] = first level of TU i in d_levels (n_coeff entries in total) */
int hop_tu_rd_device(hop_ctx* ctx, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx_in, const int64_t* d_coef_offsets, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_results);                                 /* asynchronous, unchecked */

/* ---- the mode-decision half of the intra rough search (rest of row a7) ---- */
/* replaces: the candidate selection of TEncSearch::estIntraPredQT (TLibEncoder/TEncSearch.cpp:2440-2493) on the 35 SATDs of hop_intra_rough: per mode
 * xModeBitsIntra (:7734-7745: the luma direction through the counting coder from the CI_CURR_BEST state -- TEncSbac::codeIntraDirLumaAng, TEncSbac.cpp:770-831),
 * cost = SATD + bits * sqrt(lambda) in double, the sorted list of xUpdateCandList (:7747-7767), the most probable modes appended if missing (:2466-2488). */
typedef struct {
  int32_t preds[3], pred_num;      /* TComDataCU::getIntraDirLumaPredictor */
  int32_t mpm_cand;                /* how many of them the list must contain (numCand, :2468-2473) */
  int32_t num_full_rd;             /* g_aucIntraModeNumFast of the block size: 3 or 8 */
  int32_t ctx_state, frac_left;    /* m_ucState of the prev_intra_luma_pred_flag context and the fraction the coder carries, at CI_CURR_BEST */
  double  sqrt_lambda;             /* TComRdCost::getSqrtLambda */
} hop_intra_modes_job;
typedef struct { uint32_t n, modes[11]; double costs[8]; } hop_intra_modes_result;   /* uiRdModeList (n entries), CandCostList (num_full_rd entries) */
/* satd: 35 values per job, as hop_intra_rough returns them */
int hop_intra_modes(hop_ctx* ctx, int n, const hop_intra_modes_job* jobs, const uint32_t* satd, hop_intra_modes_result* results);
int hop_intra_modes_device(hop_ctx* ctx, int n, const hop_intra_modes_job* d_jobs, const uint32_t* d_satd, hop_intra_modes_result* d_results);   /* asynchronous, unchecked */

/* ---- residual quadtree of an SS/GT ("inter") CU (row a8b) ---- */
/* replaces: TEncSearch::xEstimateResidualQT (TLibEncoder/TEncSearch.cpp:6824-7560) with xEncodeResidualQT (:7562-7655) as
 * encodeResAndCalcRdInterCU calls it (:6700) for a batch of CUs: the full search over transform sizes -- per node Y/Cb/Cr through
 * transformNxN with RDOQ, counted bits, inverse path, cbf-zero decision, the 4x4 transform-skip retry, the node's own cost; the
 * four children on the coder state the previous one left; the subtree recounted in syntax order; the split decision.  The
 * residual is original - prediction picture of the context.  RDOQ and RDOQTS on, no lossless coding, flat scaling lists. */
typedef struct {
  int32_t x, y, log2_cu;           /* CU position (luma) and size, 3..6 */
  int32_t qp_scaled[3];            /* what setQPforQuant hands to setQpParam for Y, Cb, Cr */
  int32_t ctx_index;               /* coder state on entry (m_pcRDGoOnSbacCoder) */
  int32_t sign_hide, use_ts;       /* PPS sign_data_hiding, transform_skip_enabled */
  int32_t log2_max_tu, log2_min_tu_in_cu;   /* SPS QuadtreeTULog2MaxSize, TComDataCU::getQuadtreeTULog2MinSizeInCU */
  int32_t inter_split_flag;        /* QuadtreeTUMaxDepthInter == 1 && partition != 2Nx2N (:6831) */
  double  lambda_rd;               /* TComRdCost::m_dLambda */
  double  lambda_rdoq[3];          /* TComTrQuant::m_lambdas */
  double  dist_weight[2];          /* TComRdCost::m_cbDistortionWeight, m_crDistortionWeight */
} hop_rqt_job;
typedef struct {
  double   cost;                   /* what the call adds to rdCost, ruiBits, ruiDist, *puiZeroDist */
  uint32_t bits, dist, zero_dist, pad;
  uint8_t  tr_idx[256];            /* per 4x4 partition of the CU, z-order: TComDataCU::getTransformIdx */
  uint8_t  cbf[3][256];            /* getCbf(Y / Cb / Cr): one bit per transform depth */
  uint8_t  tskip[3][256];          /* getTransformSkip */
} hop_rqt_result;
/* coef_out (may be NULL): per CU, job after job, 1.5 * size^2 levels = the chosen transform units in the CU's coefficient layout
 * (Y, then Cb, then Cr; the TU of partition p at 16 p, chroma at 4 p; what xSetResidualQTData copies into getCoeffY/Cb/Cr).
 * ctx_out (may be NULL): the coder state after the CU. */
int hop_rqt(hop_ctx* ctx, int n, const hop_rqt_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, hop_rqt_result* results, int32_t* coef_out,
            hop_cabac_ctx* ctx_out);
/* device-resident form: all n CUs of ONE class -- the size and the transform-tree limits / flags of *cls (a host copy of any of the jobs);
 * ctx_index of a job indexes d_ctx_in; asynchronous, unchecked */
int hop_rqt_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_cabac_ctx* d_ctx_in, hop_rqt_result* d_results,
                   int32_t* d_coef_out, hop_cabac_ctx* d_ctx_out);
/* several classes at once (e.g. the CUs of every depth of a frame): class i = n[i] jobs at d_jobs[i] with the parameters of cls[i]; the
 * classes run concurrently on the context's streams and are joined on its main stream */
int hop_rqt_device_classes(hop_ctx* ctx, int n_classes, const int* n, const hop_rqt_job* const* d_jobs, const hop_rqt_job* cls, const hop_cabac_ctx* d_ctx_in,
                           hop_rqt_result* const* d_results, int32_t* const* d_coef_out, hop_cabac_ctx* const* d_ctx_out);

/* replaces: the tail of TEncSearch::encodeResAndCalcRdInterCU around the quadtree (TLibEncoder/TEncSearch.cpp:6700-6723, :6804-6812) without
 * the CU-level syntax bits in between (xAddSymbolBitsInter stays with the caller): the root-cbf-zero test -- bits of a zero rqt_root_cbf on
 * the coder as the quadtree left it (ctx_after = hop_rqt's ctx_out) against the quadtree's cost --, results / coef cleared in place where
 * the zero residual wins, the reconstruction Clip(prediction + residual of the chosen transform units) written into the context's
 * reconstruction picture (hop_recon_download), and its distortion against the original. */
typedef struct {
  uint32_t root_cbf;               /* getQtRootCbf after the test */
  uint32_t dist[3];                /* getDistPart of the reconstruction: Y, Cb, Cr (the chroma planes weighted) */
} hop_cu_final;
int hop_rqt_finish(hop_ctx* ctx, int n, const hop_rqt_job* jobs, hop_rqt_result* results, int32_t* coef, const hop_cabac_ctx* ctx_after, hop_cu_final* finals);
/* device-resident form, one class of CUs as in hop_rqt_device; asynchronous, unchecked */
int hop_rqt_finish_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, hop_rqt_result* d_results, int32_t* d_coef,
                          const hop_cabac_ctx* d_ctx_after, hop_cu_final* d_finals);

/* ---- CU-level syntax bits of an SS/GT ("inter") CU (first piece of row a0) ---- */
/* replaces: TEncSearch::xAddSymbolBitsInter (TLibEncoder/TEncSearch.cpp:7779-7810) through the counting coder: skip flag and either the merge index
 * (a merged 2Nx2N CU without residual is a skipped CU) or prediction mode, partition size (TEncSbac::codePartSize TEncSbac.cpp:469-566), per PU the merge
 * flag / index or MVD (codeMvd :944-1048), MVP index (:434-467), GT flag (:654-677) and the GT corner vectors (codeGT :1051-1330: corners 0..2, coded like
 * MVDs on their own two contexts), the root cbf and the transform tree in bitstream order (TEncEntropy::encodeCoeff :633-660, xEncodeTransform :219-394).
 * One reference list with one picture, no transquant bypass, no delta QP. */
typedef struct { uint8_t state[20]; } hop_cabac_cu_ctx;   /* m_ucState of: skip[3], merge_flag, merge_idx, part_size[4], pred_mode, mvd[2], mvp_idx, gt_flag, gt[2],
                                                              prev_intra_luma_pred_flag, chroma_pred[2], one unused byte */
typedef struct {
  int32_t part_size;               /* PartSize: 0 2Nx2N, 1 2NxN, 2 Nx2N, 3 NxN, 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N */
  int32_t n_pu, skip_flag, skip_ctx;   /* isSkipped, getCtxSkipFlag (skipped CUs left + above) */
  int32_t amp_acc, is_min_cu, max_merge_cand;   /* SPS getAMPAcc(depth), depth == max CU depth, slice MaxNumMergeCand */
  struct { int32_t merge_flag, merge_idx, mvd[2], mvp_idx, gt_flag, gt[8]; } pu[4];
} hop_cu_syntax;
/* jobs: the class fields and ctx_index of hop_rqt_job are used (ctx_index indexes ctx_in and cu_ctx_in); results / coef: the arrays and levels as
 * hop_rqt / hop_rqt_finish leave them.  bits[i]: what the call adds to ruiBits; skipped[i]: isSkipped afterwards; ctx_out / cu_ctx_out (may be NULL):
 * the coder after the CU (what the caller stores as CI_TEMP_BEST). */
/* replaces: TEncSbac::resetEntropy for those sets (initBuffer with TLibCommon/ContextTables.h:140-310, 472-482); slice_type as hop_cabac_init. Host only. */
int hop_cabac_cu_init(hop_cabac_cu_ctx* ctx, int slice_type, int qp);
int hop_inter_cu_bits(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, int n_ctx,
                      const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, uint32_t* skipped, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out);

/* replaces: TEncSearch::xGetIntraBitsQT (TLibEncoder/TEncSearch.cpp:957-980) through the counting coder: xEncIntraHeader (:887-954: skip flag, prediction mode,
 * partition size, the luma directions of the CU -- or of the PU that starts at this node --, the chroma direction), xEncSubdivCbfQT (:764-830) and xEncCoeffQT
 * (:833-884) of the asked components from the node (tr_depth, part) downwards.  coef: the levels of the current tree in the CU layout (what the layer buffers hold
 * for the transform units the tr_idx array describes).  No PCM, no transquant bypass, not an I slice (the skip flag and prediction mode are coded). */
typedef struct {
  int32_t part_nxn;                /* 0: 2Nx2N, 1: NxN */
  int32_t skip_flag, skip_ctx, is_min_cu;          /* skip_ctx < 0: the slice is an I slice -- neither cu_skip_flag nor pred_mode_flag is coded (TEncEntropy::encodeSkipFlag,
                                                      encodePredMode return at once, TLibEncoder/TEncEntropy.cpp:92-131) */
  int32_t luma_dir[4], preds[4][3], pred_num[4];   /* per PU: getLumaIntraDir, getIntraDirLumaPredictor */
  int32_t chroma_is_dm, chroma_dir;   /* chroma direction == DM_CHROMA_IDX; otherwise the direction (it selects the coefficient scan) */
  int32_t tr_depth, part;          /* the node: transform depth and first 4x4 partition (z-order inside the CU) */
  int32_t b_luma, b_chroma;        /* what to count */
} hop_intra_cu_syntax;
int hop_intra_cu_bits(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, int n_ctx,
                      const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out);
/* ---- luma transform tree of an intra PU (rest of row a8) ---- */
/* replaces: TEncSearch::xRecurIntraCodingQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:1361-1710) as estIntraPredQT calls it per candidate mode (:2524, bCheckFirst)
 * and for the chosen one (:2587), for a batch of PUs: per node xIntraCodingLumaBlk (:1003-1161: reference samples from the context's reconstruction picture,
 * prediction, residual, DST / DCT with RDOQ on the current coder state, inverse path, reconstruction written back into the picture, SSE), for 4x4 nodes also the
 * transform-skip variant (:1424-1523), the bits through xGetIntraBitsQT (:957-980), the four children, the recount and the split decision (:1576-1700).
 * jobs: CU position / size / QP / lambdas / transform-tree limits and ctx_index (into ctx_in and cu_ctx_in) of hop_rqt_job; syntax: the CU's syntax elements, with
 * (tr_depth, part) = the node the PU starts at (0, 0 for 2Nx2N; 1, k * parts / 4 for PU k of NxN); b_luma / b_chroma are not read.  RDOQ and RDOQTS on, RDpenalty 0,
 * no PCM / transquant bypass, not an I slice.  The reconstruction picture is read (neighbours) and written (the PU's blocks) as the search goes: the PUs of one call
 * must not lie in each other's neighbourhood (up to 2 * size to the right / below, one sample left / above); afterwards it holds the chosen tree's reconstruction. */
typedef struct {
  int32_t  check_first;            /* bCheckFirst: no split below a node that can be coded as one TU (HHI_RQT_INTRA_SPEEDUP) */
  int32_t  ts_fast;                /* TransformSkipFast: transform skip tried in NxN CUs only */
  int32_t  strong;                 /* SPS strong_intra_smoothing */
  int32_t  pad;
  uint64_t avail[341];             /* neighbour availability of every node of the CU's quadtree: node (transform depth d, size 2^l, first partition p) at
                                      {0, 1, 5, 21, 85}[d] + (p >> (2 * (l - 2))); bit u = flags[u] of hop_intra_job (TComPattern::initAdiPattern, TComPattern.cpp:199-211) */
} hop_intra_rqt_opt;
/* results: cost / dist = what the call adds to dRDCost / ruiDistY; the arrays hold the PU's partitions (luma).  coef_out (may be NULL): per job 1.5 * size^2 entries,
 * the chosen luma levels of the PU's partitions in the CU layout (other entries untouched).  ctx_out / cu_ctx_out (may be NULL): the coder after the PU. */
int hop_intra_rqt(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, int n_ctx, const hop_cabac_ctx* ctx_in,
                  const hop_cabac_cu_ctx* cu_ctx_in, hop_rqt_result* results, int32_t* coef_out, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out);
/* device-resident form: all n PUs of ONE class -- CU size and transform-tree limits / flags of *cls, the first transform depth, bCheckFirst; asynchronous, unchecked */
int hop_intra_rqt_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, int tr_depth, int check_first, const hop_intra_cu_syntax* d_syntax,
                         const hop_intra_rqt_opt* d_opts, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, hop_rqt_result* d_results, int32_t* d_coef_out,
                         hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out);
/* ---- luma intra search of a CU (row a8) ---- */
/* replaces: TEncSearch::estIntraPredQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:2386-2710) as xCheckRDCostIntra calls it, for a batch of CUs: per PU (one, or the four
 * of NxN in order) the most probable modes (TComDataCU::getIntraDirLumaPredictor, TComDataCU.cpp:1772-1830), the 35-mode rough search with the candidate list
 * (:2430-2493 = hop_intra_rough + hop_intra_modes), every candidate through the transform tree with bCheckFirst (:2507-2553 = hop_intra_rqt), the best one again with the
 * full tree (:2555-2590), the better result kept as xSetIntraResultQT keeps it; the decided PU's reconstruction goes into the picture unless it is the last PU
 * (:2603-2660; the last PU's picture block is left as the final pass wrote it, as in the reference), the cbf of an NxN CU is combined at depth 0 (:2667-2685).
 * jobs / opts: as for hop_intra_rqt (check_first is not read); syntax: part_nxn, skip_flag, skip_ctx, is_min_cu (directions and predictors are derived here);
 * ctx_in / cu_ctx_in[jobs[i].ctx_index]: the CI_CURR_BEST state every tree and the mode bits start from.  The CUs of one call must not lie in each other's neighbourhood. */
typedef struct {
  int32_t left_dir[4], above_dir[4];   /* per PU: the luma direction getIntraDirLumaPredictor sees left / above where that neighbour lies outside the CU (1 = DC when it
                                          is unavailable, not intra, or above the CTU row); not read where the neighbour is a PU of this CU */
  uint8_t rough_flags[4][68];          /* per PU: flags of hop_intra_job for the PU's block (the rough search) */
  double  sqrt_lambda;                 /* TComRdCost::getSqrtLambda */
  int32_t num_full_rd;                 /* g_aucIntraModeNumFast of the PU size: 8 for 4x4 and 8x8, 3 above */
  int32_t pad;
} hop_intra_search_job;
typedef struct { int32_t best_dir[4], n_cand[4]; uint32_t dist, pad; } hop_intra_search_result;   /* getLumaIntraDir per PU, candidates tested, getTotalDistortion */
/* results: the tr_idx / cbf[0] / tskip[0] arrays of the CU (dist = the same distortion); coef_out: per job 1.5 * size^2 entries, the luma part = getCoeffY;
 * reco_out: per job size^2 samples, the CU's luma reconstruction (pcRecoYuv) */
int hop_intra_luma_search(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, const hop_intra_search_job* sjobs,
                          int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, hop_intra_search_result* sresults, hop_rqt_result* results,
                          int32_t* coef_out, int16_t* reco_out);
/* device-resident form: all n CUs of ONE class -- size, transform-tree limits / flags of *cls, partition (part_nxn) and num_full_rd; asynchronous, unchecked */
int hop_intra_luma_search_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, int part_nxn, int num_full_rd, const hop_intra_cu_syntax* d_syntax,
                                 const hop_intra_rqt_opt* d_opts, const hop_intra_search_job* d_sjobs, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in,
                                 hop_intra_search_result* d_sresults, hop_rqt_result* d_results, int32_t* d_coef_out, int16_t* d_reco_out);
/* ---- chroma intra search of a CU (row a8) ---- */
/* replaces: TEncSearch::estIntraPredChromaQT (TLibEncoder/TEncSearch.cpp:2720-2785) with xRecurIntraChromaCodingQT (:2130-2277) for a batch of CUs: the five allowed
 * chroma directions (TComDataCU::getAllowedChromaDir, TComDataCU.cpp:1746-1764), each coded along the luma transform tree -- xIntraCodingChromaBlk per block and plane
 * (:1164-1330: hop_intra_pred_chroma + the intra leaf with the chroma quantiser, lambda and distortion weight), for 4x4 blocks also as transform-skip blocks, the better
 * variant by its xGetIntraBitsQTChroma cost (:2176-2247) --, then the CU's chroma bits from the CI_CURR_BEST state (xGetIntraBitsQT, chroma only) and the cost; the best
 * direction kept as xSetIntraResultChromaQT keeps it (:2280-2345).  jobs / syntax (luma directions decided, part_nxn, skip flag ...) / opts (ts_fast, avail) as for the luma
 * search; results on entry: tr_idx and tskip[0] of the luma search; on return also cbf[1..2] and tskip[1..2].  The chroma reconstruction pictures are read and written as
 * the search goes (afterwards they hold what the last direction tested left, as in the reference).  The CUs of one call must not lie in each other's neighbourhood. */
typedef struct { int32_t best_mode; uint32_t dist; } hop_intra_chroma_result;   /* getChromaIntraDir (0 / 26 / 10 / 1 / 34, or 36 = DM_CHROMA_IDX), uiBestDist */
/* coef_out: per job 1.5 * size^2 entries, the two chroma parts (getCoeffCb / getCoeffCr) written; reco_out: per job size^2 / 2 samples, Cb then Cr (pcRecoYuv) */
int hop_intra_chroma_search(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, int n_ctx, const hop_cabac_ctx* ctx_in,
                            const hop_cabac_cu_ctx* cu_ctx_in, hop_rqt_result* results, hop_intra_chroma_result* cresults, int32_t* coef_out, int16_t* reco_out);
/* device-resident form: all n CUs of ONE class (size, transform-tree limits / flags of *cls); asynchronous, unchecked */
int hop_intra_chroma_search_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_intra_cu_syntax* d_syntax, const hop_intra_rqt_opt* d_opts,
                                   const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, hop_rqt_result* d_results, hop_intra_chroma_result* d_cresults,
                                   int32_t* d_coef_out, int16_t* d_reco_out);
/* ---- the bits and cost of a finished intra CU (rows a0 / a8) ---- */
/* replaces: the counting part of TEncCu::xCheckRDCostIntra (TLibEncoder/TEncCu.cpp:1483-1503) for a batch of CUs: resetBits, encodeSkipFlag / encodePredMode /
 * encodePartSize, encodePredInfo (the luma directions of all PUs against their most probable modes, the chroma direction), encodeCoeff = xEncodeTransform
 * (TEncEntropy.cpp:219-420) on the CU's final levels with the intra rules, getTotalCost = calcRdCost(bits, getTotalDistortion).  syntax: all elements of the CU (tr_depth /
 * part / b_* not read); results: the arrays after the luma and chroma searches; coef: 1.5 * size^2 levels per CU (Y | Cb | Cr, CU layout); dist: getTotalDistortion per CU.
 * ctx_in / cu_ctx_in[jobs[i].ctx_index]: the coder the searches left (CI_CURR_BEST); ctx_out / cu_ctx_out (may be NULL): what the caller stores as CI_TEMP_BEST.
 * No PCM, no transquant bypass, no cu_qp_delta, not an I slice. */
int hop_intra_cu_total_bits(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, const uint32_t* dist,
                            int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, double* cost, hop_cabac_ctx* ctx_out,
                            hop_cabac_cu_ctx* cu_ctx_out);
/* device-resident form, one class of CUs; asynchronous, unchecked */
int hop_intra_cu_total_bits_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_intra_cu_syntax* d_syntax, const hop_rqt_result* d_results,
                                   const int32_t* d_coef, const uint32_t* d_dist, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, uint32_t* d_bits, double* d_cost,
                                   hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out);
/* ---- the residual-free candidate of an SS/GT CU (row a8b) ---- */
/* replaces: TEncSearch::encodeResAndCalcRdInterCU with bSkipRes (TLibEncoder/TEncSearch.cpp:6635-6668) for a batch of CUs: the reconstruction is the prediction picture
 * (copied into the reconstruction picture), finals[i].dist its distortion against the original per plane (chroma weighted, root_cbf 0), bits[i] the skip flag and the merge
 * index (syntax[i].skip_ctx, pu[0].merge_idx, max_merge_cand) counted from the CI_CURR_BEST state, cost[i] = calcRdCost; ctx_out / cu_ctx_out (may be NULL): the coder to
 * store as CI_TEMP_BEST.  jobs: position, size, ctx_index, lambda_rd, dist_weight are read. */
int hop_inter_cu_skip(hop_ctx* ctx, int n, const hop_rqt_job* jobs, const hop_cu_syntax* syntax, int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in,
                      hop_cu_final* finals, uint32_t* bits, double* cost, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out);
int hop_inter_cu_skip_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syntax, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in,
                             hop_cu_final* d_finals, uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out);   /* asynchronous, unchecked */
/* ---- whole SS/GT candidates with residual, device-resident (rows a0 / a8b) ---- */
/* TEncSearch::encodeResAndCalcRdInterCU without bSkipRes (TLibEncoder/TEncSearch.cpp:6670-6822) for several classes of CUs at once, no host step between the stages: per
 * class hop_rqt_device -> hop_rqt_finish_device -> hop_inter_cu_bits_device -> getTotalCost = calcRdCost(bits, final distortions); the classes on separate streams (small batches: one kernel per candidate, see
 * hop_intra_cu_device_classes).  The prediction picture holds the candidates' predictions.  All pointers device memory except
 * the descriptor array; d_ctx_after: scratch for the coder after the quadtree (n states).  Asynchronous, unchecked beyond the class fields. */
typedef struct {
  int32_t n, pad;
  hop_rqt_job cls;
  const hop_rqt_job* d_jobs; const hop_cu_syntax* d_syntax;
  hop_rqt_result* d_results; int32_t* d_coef; hop_cabac_ctx* d_ctx_after; hop_cu_final* d_finals;
  uint32_t* d_bits; uint32_t* d_skipped; double* d_cost;
  hop_cabac_ctx* d_ctx_out; hop_cabac_cu_ctx* d_cu_ctx_out;   /* may be NULL */
} hop_inter_class;
int hop_inter_cu_device_classes(hop_ctx* ctx, int n_classes, const hop_inter_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in);

/* ---- whole intra candidates, device-resident (rows a0 / a8) ---- */
/* The body of TEncCu::xCheckRDCostIntra (TLibEncoder/TEncCu.cpp:1455-1503) for several classes of CUs at once, with no host step between the stages: per class
 * hop_intra_luma_search_device -> hop_intra_chroma_search_device -> getTotalDistortion -> hop_intra_cu_total_bits_device, the classes on separate streams (their chains of
 * small batch steps fill each other's gaps).  All pointers are device memory except the descriptor array itself; d_syntax_out receives the syntax elements with the decided
 * directions (what the bit count used).  The CUs of ALL classes of one call must not lie in each other's neighbourhood.  Asynchronous, unchecked beyond the class fields. */
typedef struct {
  int32_t n, part_nxn, num_full_rd, pad;
  hop_rqt_job cls;                            /* size, transform-tree limits and flags of the class (a copy of any of its jobs) */
  const hop_rqt_job* d_jobs; const hop_intra_cu_syntax* d_syntax; const hop_intra_rqt_opt* d_opts; const hop_intra_search_job* d_sjobs;
  hop_intra_search_result* d_sresults; hop_rqt_result* d_results; hop_intra_chroma_result* d_cresults;
  int32_t* d_coef;                            /* 1.5 * size^2 levels per CU: Y | Cb | Cr */
  int16_t* d_reco_y; int16_t* d_reco_c;       /* size^2 and size^2 / 2 (Cb, Cr) samples per CU */
  hop_intra_cu_syntax* d_syntax_out; uint32_t* d_dist; uint32_t* d_bits; double* d_cost;
  hop_cabac_ctx* d_ctx_out; hop_cabac_cu_ctx* d_cu_ctx_out;   /* may be NULL */
} hop_intra_class;
int hop_intra_cu_device_classes(hop_ctx* ctx, int n_classes, const hop_intra_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in);
/* Batches of up to HOP_WALK candidates (default 4096: the RD spine's batches hold one candidate per CTU in flight) run as ONE kernel per candidate -- a workgroup walks its
 * candidate's transform tree on the device (csrc/k_walk.inl: the bodies of the batch-step kernels called in the same order) --, larger batches as batch steps over all
 * candidates, a few kernels per tree node (csrc/k_rqt.inl).  Both forms give the same bytes (tests/test_gpu_tq_intra.py runs both). */
/* device-resident form, one class of CUs as in hop_rqt_device; asynchronous, unchecked */
int hop_inter_cu_bits_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_cu_syntax* d_syntax, const hop_rqt_result* d_results,
                             const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, uint32_t* d_bits, uint32_t* d_skipped,
                             hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out);

/* ---- CTU-level host logic ---- */
/* replaces: the PU enumeration of TEncCu::xCompressCU for an ISS slice (TLibEncoder/TEncCu.cpp:451-637) with
 * TComDataCU::getPartOffset (TLibCommon/TComDataCU.cpp:2251-2296, incl. the SIZE_nLx2N offY quirk) and
 * hop_set_search_range: for every CU of the CTU's quadtree that lies inside the picture (CUs crossing the
 * picture border are split, TEncCu.cpp:407-409,755) the PUs of 2Nx2N, Nx2N, 2NxN and -- with_amp != 0 -- the
 * four AMP shapes for CUs >= 16 (what the reference tests when deriveTestModeAMP enables both directions),
 * in the reference's test order, depth first.  pred/amvp are applied to every PU (the real predictors come
 * from the caller's CU state).  cu_index_out (may be NULL) receives, per job, depth<<16 | z-order index of
 * its CU at that depth.  Returns the number of jobs the CTU has (only the first max_out are written; call with
 * max_out = 0 to size a buffer) or a negative hop_status. */
int hop_enumerate_ctu_jobs(int pic_w, int pic_h, int ctu_addr, int search_range, const int pred_qpel[2],
                           int n_amvp, const int amvp_qpel[4], uint32_t lambda_cost, int flags, int with_amp,
                           hop_pu_job* out, int32_t* cu_index_out, int max_out);

/* How many stream lanes hop_me_search_device cuts a large batch into (1..4; default 2 or HOP_LANES).  The results do not
 * depend on it.  With 1 lane every kernel runs alone on the device: the setting bench.py uses for its profiling pass, so that
 * HIP-event durations are exclusive. */
int hop_set_lanes(hop_ctx* ctx, int lanes);

/* ---- the RD spine: a whole picture through TEncCu::compressCU (row a0) ---- */
/* One 4x4 unit of the finished picture: TComDataCU's per-partition arrays (TLibCommon/TComDataCU.h:96-170) as the spine keeps them.  pred_mode: 0 inter (SS / GT),
 * 1 intra, 15 none (outside the picture); part_size: PartSize, 15 none; ref_idx: 0 or -1; cbf: one bit per transform depth. */
typedef struct {
  uint8_t depth, pred_mode, part_size, skip, merge_flag, merge_idx, gt_flag, inter_dir;
  int8_t  ref_idx, mvp_idx, mvp_num; uint8_t luma_dir, chroma_dir, tr_idx;
  uint8_t cbf[3], tskip[3];
  int16_t mv[2], mvd[2], gt[8];
} hop_cu_part;
int hop_sizeof_cu_part(void);
/* sizeof(<name>) for any struct of this header ("hop_pu_job", "hop_rqt_job", ...) as the library was compiled, -1 for an unknown name: a binding in another language
 * checks its mirror of each struct against it before its first call. */
int hop_sizeof(const char* name);
/* replaces: TComRdCost::isValidPattern (TLibCommon/TComRdCost.cpp:430-443) on the resident SS reference for n queries of 6 values: PU x, y, w, h, vector (quarter-pel, as
 * the caller clipped it): out[i] = 1 if the two probe samples below the displaced block are not the sentinel. */
int hop_valid_pattern(hop_ctx* ctx, int n, const int32_t* xywh_mv, uint8_t* out);
/* replaces: the bookkeeping of m_ppcRecoYuvBest / Temp in TEncCu::xCheckBestMode (TLibEncoder/TEncCu.cpp:1572-1575) and xCopyYuv2Pic (:1623-1662): n blocks
 * rect4 = {x, y, size, slot} of the reconstruction picture put aside into stash slots (restore = 0) or brought back (restore = 1); 1024 slots. */
int hop_recon_stash(hop_ctx* ctx, int n, const int32_t* rect4, int restore);
/* replaces: TComYuv::copyToPicYuv of an intra candidate's reconstruction (TEncCu.cpp:1476, :1640): the packed blocks hop_intra_cu_device_classes leaves in d_reco_y /
 * d_reco_c into the reconstruction picture at the jobs' positions; asynchronous on the context stream. */
int hop_recon_put_device(hop_ctx* ctx, int n, const hop_rqt_job* d_jobs, const int16_t* d_reco_y, const int16_t* d_reco_c);
/* replaces: TEncCu::xCopyYuv2SSRef (TEncCu.cpp:1677-1715) from the context's reconstruction picture: n blocks rect4 = {x, y, size, -}. */
int hop_ssref_commit_recon(hop_ctx* ctx, int n, const int32_t* rect4);
/* replaces: the CTU loop of TEncSlice::compressSlice (TLibEncoder/TEncSlice.cpp:1000-1196) with TEncCu::compressCU (TEncCu.cpp:246-264, xCompressCU :371-892) for one
 * picture of the HOP configuration (cfg/3DHencoder_intra_main.cfg: ISS slice, SS +-128 full search with FEN, GT search, AMP, RDOQ, transform skip, contexts running on
 * from CTU to CTU): every candidate is evaluated by this library's kernels, the decisions are taken by the host spine (host/hop_spine.cpp).  The original must be resident
 * (hop_upload_orig).  ctu_cost / ctu_bits / ctu_dist: getTotalCost / Bits / Distortion of every CTU (what the reference appends to cost.csv, TEncSlice.cpp:183-191);
 * parts: 256 hop_cu_part per CTU in z-order; afterwards hop_recon_download gives the reconstruction before the loop filters.  first_ctus > 0: stop after that many CTUs.
 * trace_path (may be NULL): one text line per candidate that reaches xCheckBestMode.
 * A stacked context (hop_ctx_set_stack, n pictures): wavefront_lag > 0 is required; the rows of all pictures form one wavefront whose batches serve CTUs of every picture;
 * the output arrays hold picture k's CTUs at [k * ctus_per_picture, ...), the traces go to trace_path.<k>; each picture's results are those of a context of its own. */
typedef struct {
  int32_t qp, mi_size, first_ctus;
  int32_t wpp;            /* 1: the rows' coders are synchronised as WaveFrontSynchro does (TEncSlice.cpp:1027-1051, :1158-1161): the result equals the reference run with
                             --WaveFrontSynchro=1 --WaveFrontSubstreams=<CTU rows>; 0: the shipped configuration (contexts run on in raster order, CTUs strictly serial) */
  int32_t wavefront_lag;  /* > 0 (implies wpp): the CTU rows run as a wavefront -- row r codes CTU c once row r - 1 has finished CTU c + lag - 1 -- and the candidate
                             evaluations of all rows in flight are batched into common launches.  A wavefront is not raster order: a search sees the rows above coded up to column c + lag - 1
                             only and the rows below already coded up to c - lag - 1; with predictors of ordinary length 5 is enough (every golden picture), on the 7728x5368 frame
                             5 and 6 leave the reference's decisions at CTU 3079, 8 nowhere in the whole frame; lag >= the picture's width in CTUs is raster order */
  int32_t plain_intra;    /* 1: the plain HM intra configurations (cfg/encoder_intra_main.cfg, encoder_intra_main10.cfg): an I slice without SS / GT search, at the context's
                             bit depth (8 or 10); 0: the HOP configuration (8 bit only: the GT warp clips to 255, TComPrediction.cpp:969) */
  int32_t streams;        /* wavefront mode: > 1 = that many views of the context (hop_ctx_create_view), one per CTU row in flight, each row's requests on its own stream so
                             that the rows' launch chains overlap on the device; <= 1 = the rows' requests rendezvous and are served in batches on the context's stream */
  const char* trace_path;
} hop_enc_params;
/* The quantised levels of the picture(s) hop_encode_frame coded, as TEncCu leaves them in TComDataCU::m_pcTrCoeffY / Cb / Cr (what TEncSlice::encodeSlice codes): per CTU
 * 6144 TCoeff -- 4096 luma, 1024 Cb, 1024 Cr --, a CU's block at 16 x (luma) / 4 x (chroma) its z-order partition index, a TU's coefficients in raster order inside it;
 * CTUs in raster order, the pictures of a stacked context one after the other.  out: n_ctu * 6144 int32. */
int hop_levels_download(hop_ctx* ctx, int32_t* out);
/* The one piece of state the RD search leaves behind besides the picture's data: the fraction of a bit (15 bits, TEncBinCABAC::m_fracBits & 32767) the counting coder of the
 * RD search (TEncTop::m_cRDGoOnSbacCoder over TEncBinCABACCounter) carries when a CTU's compressCU returns.  The reference never clears it (resetBits keeps it,
 * TEncBinCoderCABAC.cpp:163-170), and the SAO parameter decision that follows the CTU loop starts its rate count from it (TEncSampleAdaptiveOffset.cpp:594, :643: whole
 * bits of fraction + rate), so a caller that lets the reference's SAO encoder run after hop_encode_frame sets the coder's m_fracBits to the LAST CTU's value first
 * (oracle/enc_shim_pic.cpp); without it an SAO offset can come out different.  out: one uint16 per CTU, CTUs and pictures ordered as in hop_encode_frame's outputs. */
int hop_rd_fraction_download(hop_ctx* ctx, uint16_t* out);
int hop_encode_frame(hop_ctx* ctx, const hop_enc_params* params, double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, hop_cu_part* parts, uint64_t* n_candidates);
/* A picture takes as long as its wavefront (TEncSlice::compressSlice's CTU loop, TLibEncoder/TEncSlice.cpp:1000-1196, is the whole encode of an all-intra sequence): these two
 * may be called from ANOTHER host thread while hop_encode_frame runs in wavefront mode on ctx.  hop_encode_progress: the number of CTUs whose compressCU has returned so far
 * (all pictures of a stacked context together; 0 again when the next hop_encode_frame starts) -- a retired CTU's decisions, reconstruction and SS-reference commit are
 * complete on the device.  hop_encode_cancel: no CTU row starts another CTU; the CTUs in flight finish, hop_encode_frame returns HOP_OK, and its outputs hold the finished
 * CTUs (cost, bits and distortion of the others stay 0).  What bench.py's steps are made of: a step is a fixed number of retired CTUs of one continuously running picture. */
int hop_encode_progress(hop_ctx* ctx, int64_t* ctus_retired);
int hop_encode_cancel(hop_ctx* ctx);
/* ONE picture on several GPUs (the CTU rows of TEncSlice::compressSlice's loop under WaveFrontSynchro, TEncSlice.cpp:1027-1051, dealt to ranks): after
 * hop_encode_set_shard(ctx, rank, world, fn, user) the next hop_encode_frame (wavefront mode, one picture, the same original resident on every rank) codes the rows
 * r % world == rank.  After every wavefront step the ranks hand each other what the step finished -- per CTU its reconstruction block, partition data, costs and the row's
 * coder states -- through fn, an all-gather the CALLER provides (RCCL or gloo behind it; all ranks call it the same number of times with equal sizes): recv holds world
 * contributions of bytes_per_rank, rank k's at k * bytes_per_rank; host pointers; return 0 on success.  Every rank ends with the whole picture (reconstruction and SS
 * reference on its device, all outputs of hop_encode_frame), equal to the single-GPU result.  hop_encode_cancel on any rank is agreed on through the same exchange.
 * world = 1 (fn may be NULL) switches it off again. */
typedef int (*hop_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int hop_encode_set_shard(hop_ctx* ctx, int rank, int world, hop_allgather_fn fn, void* user);
/* diagnostics of the last hop_encode_frame of this process: host wall time (ms) and number of requests per kind -- 0 ME chain, 1 predictor, 2 distortion, 3 validity
 * probes, 4 SS/GT candidates with residual, 5 without, 6 intra candidates, 7 reconstruction stash, 8 SS-reference commits */
void hop_encode_stats(double ms[16], double calls[16]);

/* ---- after the search: the loop filter over the resident reconstruction ---- */
/* replaces: TComLoopFilter::loopFilterPic (TLibCommon/TComLoopFilter.cpp:129-153; xDeblockCU :166-227, xGetBoundaryStrengthSingle :395-519, xEdgeFilterLuma :522-632,
 * xEdgeFilterChroma :635-737) as TEncGOP::compressGOP calls it after the CTU loop (TLibEncoder/TEncGOP.cpp:1192-1197): the context's reconstruction picture (what
 * hop_encode_frame left there, or hop_recon_upload) is filtered in place -- all vertical edges of the 8x8 grid, then all horizontal ones --; hop_recon_download then gives
 * the picture SAO starts from.  parts: the per-partition data of the picture(s) as hop_encode_frame returned it (256 hop_cu_part per CTU; the pictures of a stacked context
 * one after the other): CU depth, partition shape, transform depth, prediction mode, luma cbf, vector and reference index are read.  qp: the slice QP every CU is coded at
 * (MaxDeltaQP 0); offsets: slice_beta_offset_div2 / slice_tc_offset_div2 (-6..6), pps_cb_qp_offset / pps_cr_qp_offset; disable: slice_deblocking_filter_disabled_flag
 * (the picture is left as it is).  One slice, one tile, no PCM / lossless CUs: the configurations of the path. */
typedef struct { int32_t qp, beta_offset_div2, tc_offset_div2, cb_qp_offset, cr_qp_offset, disable; } hop_deblock_params;
int hop_deblock_frame(hop_ctx* ctx, const hop_deblock_params* params, const hop_cu_part* parts);

/* Sample adaptive offset, the encoder side (TEncGOP.cpp:1748-1758): statistics of the deblocked picture against the original, the per-CTU parameter decision, the
 * offsets applied.  hop_sao_param: one component of one CTU, as SAOOffset (TLibCommon/TypeDef.h:351-368): mode 0 off, 1 new, 2 merge; type: new: 0..3 edge offset
 * 0 / 90 / 135 / 45 degrees, 4 band offset; merge: 0 left, 1 above; aux: sao_band_position; offset: per edge class (0 full valley, 1 half valley, 2 plain = 0, 3 half peak,
 * 4 full peak) or per band (32). */
typedef struct { int8_t mode, type, aux, pad; int8_t offset[32]; } hop_sao_param;
typedef struct {
  double  lambda[3];      /* TComSlice::getLambdas(): Y, Cb, Cr */
  int32_t enabled[3];     /* slice_sao_luma_flag / slice_sao_chroma_flag as decidePicParams left them (TEncSampleAdaptiveOffset.cpp:354-379) */
  int32_t slice_type;     /* as hop_cabac_init: the initial state of the two SAO contexts */
  int32_t qp;
  uint32_t rd_fraction;   /* the RD coder's carried fraction after the picture's last CTU (hop_rd_fraction_download) */
} hop_sao_params;
/* replaces: TEncSampleAdaptiveOffset::getStatistics (TLibEncoder/TEncSampleAdaptiveOffset.cpp:305-352, getBlkStats :862-1383; SAOLcuBoundary 0) between the context's
 * reconstruction picture (deblocked: hop_deblock_frame) and the resident original.  stats (host): per CTU, component, type (5) and class (32) the pair count, sum of
 * (original - reconstruction), int32: n_ctu x 3 x 5 x 32 x 2; the pictures of a stacked context one after the other. */
int hop_sao_stats(hop_ctx* ctx, int32_t* stats);
/* replaces: TEncSampleAdaptiveOffset::decideBlkParams (:754-860) for one picture of n_ctu CTUs: coded = what encodeSlice writes (SAOBlkParam per CTU: 3 hop_sao_param),
 * recon = the same with merges resolved and offsets scaled (what offsetCTU applies).  Host only. */
int hop_sao_decide(int n_ctu, int ctus_per_row, int bit_depth, const int32_t* stats, const hop_sao_params* params, hop_sao_param* coded, hop_sao_param* recon);
/* replaces: TComSampleAdaptiveOffset::offsetCTU for every CTU (TLibCommon/TComSampleAdaptiveOffset.cpp:655-707, offsetBlock :365-653): recon (host, 3 per CTU) applied
 * to the context's reconstruction picture in place (the kernel reads an untouched copy, as the reference reads its m_tempPicYuv). */
int hop_sao_apply(hop_ctx* ctx, const hop_sao_param* recon);
/* the three steps for the picture(s) of the context: coded (n_ctu x 3 per picture) out; afterwards hop_recon_download gives the final reconstruction.  The pictures of a
 * stacked context that hop_encode_frame coded on this context each start from the fraction their own last CTU left (params->rd_fraction is for a single picture).
 * replaces: TEncSampleAdaptiveOffset::SAOProcess (:251-283) after decidePicParams. */
int hop_sao_frame(hop_ctx* ctx, const hop_sao_params* params, hop_sao_param* coded);

/* replaces: the distortion loops and the PSNR formula of TEncGOP::xCalculateAddPSNR (TLibEncoder/TEncGOP.cpp:2383-2456) between the resident original and the context's
 * reconstruction picture (after hop_sao_frame: the final picture).  Per picture of the context and component (Y, Cb, Cr): ssd = the sum of squared differences, psnr in
 * dB (99.99 for an exact picture).  Either output may be NULL. */
int hop_psnr(hop_ctx* ctx, uint64_t* ssd, double* psnr);

/* ---- profiling (bench.py roofline): HIP events around every kernel launch on the context stream ---- */
#define HOP_K_SS_SEARCH 0
#define HOP_K_FRAC      1
#define HOP_K_GT_SEARCH 2
#define HOP_K_PRED      3
#define HOP_K_COMMIT    4
#define HOP_K_DIST      5
#define HOP_K_TQ        6
#define HOP_K_INTRA     7
#define HOP_K_RDOQ      8
#define HOP_K_CABAC     9
#define HOP_K_DEBLOCK   10
#define HOP_K_SAO       11
#define HOP_K_WALK_INTER 12   /* + (log2 CU size - 3): the SS/GT candidate walk of csrc/k_walk.inl, one kernel per candidate, by CU size 8 / 16 / 32 / 64 */
#define HOP_K_WALK_INTRA 16   /* + (log2 CU size - 3): the intra 2Nx2N candidate walk by CU size */
#define HOP_K_WALK_INTRA_NXN 20   /* the intra NxN candidate walk (8x8 CUs) */
#define HOP_K_COUNT     21
int hop_profile_enable(hop_ctx* ctx, int on);
/* waits for the stream, then reports launches, summed kernel time and units (PUs/CUs/jobs) since the last reset */
int hop_profile_read(hop_ctx* ctx, int kernel, uint64_t* launches, double* total_ms, uint64_t* units);
int hop_profile_reset(hop_ctx* ctx);

/* library/build identification: "hophip <version> gfx950" */
const char* hop_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HOPHIP_H */
