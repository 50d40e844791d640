// hop_dev.h -- shared host/device declarations of libhophip (gfx950 only; wave = 64).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hophip.h"

#define HOP_MARGIN_Y 80      // TLibCommon/TComPicYuv.cpp:82-85 (g_uiMaxCUWidth + 16)
#define HOP_MARGIN_C 40
#define HOP_NOT_VALID (-1)   // TLibCommon/CommonDef.h:126
#define HOP_WAVE 64
// The reference's GT patch and 8-tap reach can leave its own buffer by up to H/2+4 rows when the start vector
// points into the top margin (an out-of-bounds read there).  Guard rows keep our reads inside one allocation;
// they hold the sentinel and are not part of the reference layout.
#define HOP_GUARD_ROWS 64
#define HOP_MAX_LANES 4

struct hop_ctx {
  int pic_w, pic_h, bd_y, bd_c, device;
  int stride_y, stride_c;            // SS-ref strides (with margins)
  int slots;                         // hop_ctx_set_slots: the original, prediction and reconstruction pictures exist slots + 1 times, copy k at rows k * pic_h (0: once)
  int fused_leaf_max;                // hop_set_fused_leaf: leaf batches of up to this many TUs run as the one-kernel form (default 8192; 0 = always staged)
  int sub_h, sub_pitch;              // hop_ctx_set_stack: the picture is a stack of independent pictures of sub_h rows, origins sub_pitch rows apart (0, 0: one picture)
  hipStream_t stream;
  // device pictures
  int16_t *org_y, *org_cb, *org_cr;  // original, pitch pic_w / pic_w/2, no margins
  int16_t *ss_alloc[3];              // allocations: HOP_GUARD_ROWS sentinel rows + padded plane + HOP_GUARD_ROWS
  int16_t *ss_buf[3];                // padded SS-ref buffers (the reference's TComPicYuv layout) inside ss_alloc
  int16_t *ss00[3];                  // sample (0,0) inside them
  int16_t *pred[3];                  // prediction picture, pitch pic_w / pic_w/2
  int16_t *rec[3];                   // reconstruction picture (TU round trip output, intra neighbours), same pitch
  // scratch that grows on demand (never allocated inside a *_device call once sized)
  void*  scratch; size_t scratch_bytes;
  void*  stage;   size_t stage_bytes;   // staging for host-array entry points
  void*  rqt_buf; size_t rqt_bytes;     // state of the residual-quadtree search (k_rqt.inl); separate from scratch, which its leaf pipeline uses
  void*  walk_buf; size_t walk_bytes;   // work areas of the candidate walks (k_walk.inl): one kernel per candidate for the RD spine's small batches
  void*  xwalk_buf[HOP_MAX_LANES - 1]; size_t xwalk_bytes[HOP_MAX_LANES - 1];   // ... of the extra lanes (classes of one call on separate streams)
  int    walk_max;                      // batches of up to this many candidates take the walk kernels (HOP_WALK, default 4096; 0 = always the batch-step form)
  void*  xrqt_buf[HOP_MAX_LANES - 1]; size_t xrqt_bytes[HOP_MAX_LANES - 1];
  // extra lanes for hop_me_search_device: the parts of a batch run on separate streams so that one part's kernel tails
  // and low-occupancy phases are filled by the other parts' kernels
  hipStream_t xstream[HOP_MAX_LANES - 1]; void* xscratch[HOP_MAX_LANES - 1]; size_t xscratch_bytes[HOP_MAX_LANES - 1];
  hipEvent_t ev_fork, ev_join[HOP_MAX_LANES - 1]; int lanes;   // HOP_LANES=1..4 (default 2)
  bool   have_orig;
  int32_t* entropy_bits;             // device: the 128 fractional-bit values of the CABAC states (ContextModel::m_entropyBits)
  uint16_t* rdoq_scans;              // device: the scan tables of the RDOQ kernel (hop_rdoq_build_scans)
  bool   ss_families;                // SS search: share one pass among the five symmetric PUs of a CU (HOP_SS_FAMILIES=0 turns it off)
  // RD spine support (k_spine.hip): stash slots for reconstruction blocks (64 x 64 x 1.5 samples each), allocated on first use
  int16_t* stash; int stash_slots;
  // the levels of the pictures as the reference keeps them (TComDataCU::m_pcTrCoeffY / Cb / Cr): per 64x64 CTU of the (stacked) picture 4096 + 1024 + 1024 TCoeff, a CU's at
  // 16 x / 4 x its partition index; one such image per candidate slot; and their part of the stash slots.  Allocated by hop_encode_frame.
  int32_t* coefpic; int32_t* coef_stash;
  int shard_rank, shard_world; hop_allgather_fn shard_fn; void* shard_user;   // hop_encode_set_shard: the next hop_encode_frame codes the CTU rows r % world == rank of its picture
  std::atomic<long> enc_progress; std::atomic<int> enc_cancel;   // hop_encode_progress / hop_encode_cancel: CTUs the running hop_encode_frame has retired; a request to stop it
  uint16_t* rd_fraction; int rd_fraction_n;   // host: hop_encode_frame's per-CTU carried fraction of the RD coder (hop_rd_fraction_download)
  bool   is_view;                    // hop_ctx_create_view: pictures, tables and stash belong to the parent; stream, scratch areas and profiling are its own
  char   err[512];
  // profiling (hop_profile_*): event pairs recorded around kernel launches, folded into the sums on read
  bool   prof_on;
  struct hop_prof_rec* prof_recs; int prof_n, prof_cap;
  double   prof_ms[HOP_K_COUNT];
  uint64_t prof_launches[HOP_K_COUNT], prof_units[HOP_K_COUNT];
};
struct hop_prof_rec { hipEvent_t a, b; int kernel; uint64_t units; };
int  hop_prof_begin(hop_ctx* c, int kernel, uint64_t units);   // returns record index or -1 when profiling is off
void hop_prof_end(hop_ctx* c, int rec);

// read-only view of the pictures handed to kernels
struct hop_pics {
  const int16_t* org_y; const int16_t* org_cb; const int16_t* org_cr;
  const int16_t* ss_y;  const int16_t* ss_cb;  const int16_t* ss_cr;   // at sample (0,0)
  int16_t* pred_y; int16_t* pred_cb; int16_t* pred_cr;
  int pic_w, pic_h, stride_y, stride_c, bd_y, bd_c;
};

static inline hop_pics hop_make_pics(const hop_ctx* c) {
  hop_pics p;
  p.org_y = c->org_y; p.org_cb = c->org_cb; p.org_cr = c->org_cr;
  p.ss_y = c->ss00[0]; p.ss_cb = c->ss00[1]; p.ss_cr = c->ss00[2];
  p.pred_y = c->pred[0]; p.pred_cb = c->pred[1]; p.pred_cr = c->pred[2];
  p.pic_w = c->pic_w; p.pic_h = c->pic_h; p.stride_y = c->stride_y; p.stride_c = c->stride_c;
  p.bd_y = c->bd_y; p.bd_c = c->bd_c;
  return p;
}

#ifdef __HIPCC__
// ---- bit-cost helpers (TLibCommon/TComRdCost.cpp:270-284, TComRdCost.h:185-215, FIX203) ----
__host__ __device__ static inline uint32_t hopd_component_bits(int v) {
  uint32_t t = (v <= 0) ? (uint32_t)((-v << 1) + 1) : (uint32_t)(v << 1);
#ifdef __HIP_DEVICE_COMPILE__
  return 2u * (31u - (uint32_t)__clz((int)t)) + 1u;
#else
  uint32_t len = 1; while (t != 1) { t >>= 1; len += 2; } return len;
#endif
}
__host__ __device__ static inline uint32_t hopd_mv_cost(uint32_t lambda_cost, int x, int y, int scale, int pred_x, int pred_y) {
  uint32_t bits = hopd_component_bits(x * (1 << scale) - pred_x) + hopd_component_bits(y * (1 << scale) - pred_y);
  return (lambda_cost * bits) >> 16;
}

// ---- wave-level helpers ----
__device__ static inline int hopd_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline unsigned long long hopd_wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// SATD of one 8x8 block held one difference per lane (lane = 8*row + col): the 2-D Hadamard is six
// butterfly stages over the lane index bits; sum|coef| is invariant to the butterfly order, so this
// equals xCalcHADs8x8 (TLibCommon/TComRdCost.cpp:1481-1575).  Returns (sum + 2) >> 2 in every lane.
__device__ static inline int hopd_satd8x8_wave(int d, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_xor(d, o, 64);
    d = (lane & o) ? (t - d) : (d + t);
  }
  int s = hopd_wave_sum(d < 0 ? -d : d);
  return (s + 2) >> 2;
}
// four 4x4 blocks per wave: lane = 16*blk + 4*row + col; xCalcHADs4x4 (:1387-1479): (sum+1)>>1 per block.
// Returns this lane's block SATD (same in the 16 lanes of a block).
__device__ static inline int hopd_satd4x4_quad(int d, int lane) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    int t = __shfl_xor(d, o, 64);
    d = (lane & o) ? (t - d) : (d + t);
  }
  int s = d < 0 ? -d : d;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
  return (s + 1) >> 1;
}
#endif // __HIPCC__

// kernel launchers (each .hip file defines its own)
int hop_launch_ss_search(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res);
int hop_launch_frac(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res);
int hop_launch_gt(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res);
void hop_launch_size_classes(hop_ctx* c, int n, const hop_pu_job* d_jobs, const hop_pu_result* d_res, void* sc);
int hop_check_pu_jobs(hop_ctx* c, int n, const hop_pu_job* jobs);        // host: the checks of hop_me_search / hop_pred_inter without the transfer (the spine packs its own pinned buffers)
int hop_check_pred_jobs(hop_ctx* c, int n, const hop_pred_job* jobs);
int hop_launch_pred(hop_ctx* c, int n, const hop_pred_job* d_jobs);
int hop_launch_pred_cost(hop_ctx* c, int total, const int32_t* d_seq_of, const int32_t* d_first, const hop_pred_job* d_jobs, const int32_t* d_kinds, uint32_t* d_out);   // sequences of candidates, one launch, a workgroup per candidate (k_pred.hip)
int hop_launch_dist(hop_ctx* c, int n, const hop_dist_job* d_jobs, uint32_t* d_out);
int hop_launch_tu(hop_ctx* c, int n, const hop_tu_job* d_jobs, hop_tu_result* d_res, int32_t* d_levels, const int64_t* d_level_off);
int hop_launch_intra(hop_ctx* c, int n, const hop_intra_job* d_jobs, uint32_t* d_satd);
int hop_launch_intra_pred(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes);
int hop_launch_intra_pred_chroma(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes);
int hop_launch_rdoq(hop_ctx* c, int n, const hop_rdoq_job* d_jobs, const hop_estbits* d_tables, const int32_t* d_src, int32_t* d_dst, uint32_t* d_abs_sum,
                    void* d_work /* hop_rdoq_work_bytes(n) */);
size_t hop_rdoq_work_bytes(int n);
int hop_launch_tu_rd(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_res, int size_hint);
int hop_launch_tu_recon(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const int64_t* d_coef_off, const int32_t* d_levels, const uint32_t* d_abs_sum, uint32_t* d_sse);
size_t hop_intra_rqt_work_bytes(int log2_cu, int n);
int hop_launch_intra_rqt(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int tr_depth0, int check_first, int n, const hop_rqt_job* d_jobs,
                         const hop_intra_cu_syntax* d_syn, const hop_intra_rqt_opt* d_opt, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, hop_rqt_result* d_res,
                         int32_t* d_coef_out, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out, void* buf, size_t buf_bytes, const uint8_t* d_active);
size_t hop_intra_search_work_bytes(int log2_cu, int n);
int hop_launch_intra_search(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int nxn, int num_full_rd, int n, const hop_rqt_job* d_jobs,
                            const hop_intra_cu_syntax* d_syn_in, const hop_intra_rqt_opt* d_opt, const hop_intra_search_job* d_sj, const hop_cabac_ctx* d_ctx_in,
                            const hop_cabac_cu_ctx* d_cu_in, hop_intra_search_result* d_sres, hop_rqt_result* d_res, int32_t* d_coef_out, int16_t* d_reco_out, void* buf,
                            size_t buf_bytes, hop_intra_cu_syntax* d_syn_out);
int hop_launch_intra_dist_sum(hop_ctx* c, int n, const hop_intra_search_result* d_sres, const hop_intra_chroma_result* d_cres, uint32_t* d_dist);
size_t hop_intra_chroma_work_bytes(int log2_cu, int n);
int hop_launch_intra_chroma_search(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                                   const hop_intra_cu_syntax* d_syn_in, const hop_intra_rqt_opt* d_opt, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                                   hop_rqt_result* d_res, hop_intra_chroma_result* d_cres, int32_t* d_coef_out, int16_t* d_reco_out, void* buf, size_t buf_bytes,
                                   hop_intra_cu_syntax* d_syn_update);
int hop_launch_intra_cu_total(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs, const hop_intra_cu_syntax* d_syn,
                              const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, const uint32_t* d_dist,
                              uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out);
int hop_launch_cu_skip(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syn, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, hop_cu_final* d_fin,
                       uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out);
int hop_launch_inter_cost(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_cu_final* d_fin, const uint32_t* d_bits, double* d_cost);
size_t hop_rqt_finish_work_bytes(int log2_cu, int log2_max_tu, int log2_min_tu, int n);
int hop_launch_rqt_finish(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int use_ts, int n, const hop_rqt_job* d_jobs, hop_rqt_result* d_res, int32_t* d_coef,
                          const hop_cabac_ctx* d_after, hop_cu_final* d_fin, void* buf);
int hop_launch_cu_bits(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int inter_split, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                       const hop_cu_syntax* d_syn, const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                       uint32_t* d_bits, uint32_t* d_skipped, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out);
int hop_launch_intra_modes(hop_ctx* c, int n, const hop_intra_modes_job* d_jobs, const uint32_t* d_satd, hop_intra_modes_result* d_res);
int hop_launch_intra_cu_bits(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs, const hop_intra_cu_syntax* d_syn,
                             const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, uint32_t* d_bits,
                             hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out);
size_t hop_rqt_work_bytes(int log2_cu, int n);
size_t hop_intra_walk_bytes(const hop_ctx* c, int log2_cu, int n, int num_full_rd);
int hop_launch_intra_walk(hop_ctx* c, const hop_intra_class& q, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, void* buf, size_t buf_bytes);
size_t hop_inter_walk_bytes(int log2_cu, int log2_max_tu, int log2_min_tu, int n);
int hop_launch_inter_walk(hop_ctx* c, const hop_rqt_job* cls, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syn, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                          hop_rqt_result* d_res, int32_t* d_coef, hop_cabac_ctx* d_ctx_after, hop_cu_final* d_fin, uint32_t* d_bits, uint32_t* d_skipped, double* d_cost,
                          hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out, void* buf, size_t buf_bytes);
int hop_launch_rqt_class(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int inter_split, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                         const hop_cabac_ctx* d_ctx_in, hop_rqt_result* d_res, int32_t* d_coef_out, hop_cabac_ctx* d_ctx_out, void* buf, size_t buf_bytes);
void hop_rdoq_build_scans(uint16_t* tabs);
const int32_t* hop_entropy_bits_host(void);
static inline const int32_t* hop_entropy_bits_device(const hop_ctx* c) { return c->entropy_bits; }
int hop_launch_tu_rd(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_res);
// the same leaf step as one kernel, a workgroup per TU (k_leaf_fused.inl): hop_launch_tu_rd takes it for batches of up to c->fused_leaf_max TUs
int hop_launch_tu_rd_fused(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                           int32_t* d_levels, hop_tu_rd_result* d_res, int all_small /* every TU is 4x4 or 8x8: a wave per TU */);
int hop_launch_coeff_bits(hop_ctx* c, int n, const hop_coeff_bits_job* d_jobs, const hop_cabac_ctx* d_ctx, const int32_t* d_coef,
                          unsigned long long* d_bits, hop_cabac_ctx* d_ctx_out);
#define HOP_RDOQ_SCAN_ENTRIES (4080 + 255)
int hop_launch_ssref_reset(hop_ctx* c);
int hop_launch_ssref_commit(hop_ctx* c, int n, const int32_t* d_rect4, const int16_t* d_y, const int16_t* d_cb, const int16_t* d_cr, int packed);
int hop_set_err(hop_ctx* c, int code, const char* fmt, ...);
int hop_scratch(hop_ctx* c, size_t bytes, void** out);
