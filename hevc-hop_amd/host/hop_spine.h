// hop_spine.h -- the host RD spine of the HOP encoder: TEncCu::xCompressCU with its mode order,
// xCheckRDCost{Merge2Nx2N,Inter,Intra}, xCheckBestMode, the snapshot discipline of the RD coders, TEncSearch::predInterSearch
// (AMVP / merge candidate derivation, ME-vs-merge) and the CTU loop of TEncSlice::compressSlice -- control over candidate
// evaluations that all run behind a Backend.  The product backend (hop_spine_hip.cpp) drives libhophip's kernels; the tests
// instantiate the same spine over the CPU restatement (oracle/spine_backend_cpu.cpp) to pin it against the reference encoder.
#pragma once
#include <stdint.h>
#include <atomic>
#include <functional>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/hophip.h"

namespace hopspine {

enum { MODE_INTER = 0, MODE_INTRA = 1, MODE_NONE = 15 };                       // TLibCommon/TypeDef.h PredMode
enum { SIZE_2Nx2N = 0, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN, SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N, SIZE_NONE = 15 };
enum { DC_IDX = 1, PLANAR_IDX = 0, VER_IDX = 26, DM_CHROMA_IDX = 36 };

// the state of one RD coder (TEncSbac + its counting bin coder): residual sets + carried fraction, CU-level sets, split_cu_flag
struct Coder { hop_cabac_ctx r; hop_cabac_cu_ctx c; uint8_t split[3]; uint8_t pad; };
static inline uint32_t coder_frac(const Coder& k) { return (uint32_t)k.r.state[150] | ((uint32_t)k.r.state[151] << 8); }
static inline void coder_set_frac(Coder& k, uint32_t f) { k.r.state[150] = (uint8_t)(f & 255); k.r.state[151] = (uint8_t)((f >> 8) & 127); }

struct EncConfig {
  int pic_w, pic_h, bit_depth;
  int qp, slice_type;              // slice_type as hop_cabac_init: 3 = ISS
  int search_range, mi_size, mi_merge, max_merge_cand;
  int amp, fen, hadme, fdm, esd, cfm, ecu;
  int log2_max_tu, log2_min_tu, tu_max_depth_inter, tu_max_depth_intra;
  int sign_hide, use_ts, ts_fast, strong_intra;
  int fuse_pred;                   // 1 (default; HOP_SPINE_FUSE_PRED=0 switches it off): InterEval carries the CU's final predictions
  int spec_slots, slot_pitch;      // > 0: the SS/GT candidates of a CU are evaluated side by side, candidate k in copy k of the prediction / reconstruction pictures (rows k * slot_pitch, hop_ctx_set_slots)
  int y_origin;                    // added to the y coordinate of every request: the picture's first row in a stacked context (hop_ctx_set_stack), else 0
  int wpp;                         // 0: contexts run on from CTU to CTU in raster order (shipped configurations); 1: WaveFrontSynchro rows
  // wavefront mode, both optional: *progress counts the CTUs whose compressCU has returned (all pictures together); once *cancel is non-zero no row starts another CTU
  // (the CTUs in flight finish; the results of the finished CTUs are valid, the others stay 0) -- hop_encode_progress / hop_encode_cancel
  std::atomic<long>* progress; std::atomic<int>* cancel;
  // ONE picture's CTU rows dealt to several processes (WaveFrontSynchro semantics, TEncSlice.cpp:1027-1051, :1158-1161): rank g of shard_world codes the rows r
  // with r % shard_world == g as its part of the lag-5 wavefront.  After every wavefront step all ranks exchange the CTUs they have just finished through `shard` (an
  // all-gather: per CTU its reconstruction 64 x 64 x 1.5, its partition data, cost / bits / distortion, the coder it started from and the coder it left -- the row's context
  // snapshot after its second CTU is that --): every rank then holds the whole picture so far (reconstruction and SS reference on its device, partition data on its host), and
  // the next step's CTUs find their neighbours, search windows and contexts as in a single process.  shard_world <= 1: off.  A cancel request is agreed on through the same
  // exchange, so that all ranks leave the wavefront at the same step.
  int shard_rank, shard_world; struct ShardComm* shard;
  // derived by finish_config()
  double lambda, sqrt_lambda, lambda_rdoq[3], dist_weight[2];
  uint32_t lambda_sad;
  int qp_scaled[3];
};
void default_hop_config(EncConfig& c, int pic_w, int pic_h, int qp, int mi_size);   // cfg/3DHencoder_intra_main.cfg
void default_plain_config(EncConfig& c, int pic_w, int pic_h, int qp, int bit_depth);   // cfg/encoder_intra_main.cfg / encoder_intra_main10.cfg: I slice, no SS / GT
void finish_config(EncConfig& c);                                                 // TEncSlice::initEncSlice lambda / weights / chroma QP

// HOP_SPINE_POSTED (hop_spine.cpp, FiberPool): requests without an answer do not stop their row.  Level 1, the default, posts what touches nothing another pending request
// reads (stash, restore, commit); level 2 also posts predictions -- a backend that serves a batch of posted predictions with ONE launch cannot keep two posted predictions of
// one block in their order, so the device backend clears this flag before it runs and level 2 stays a mode of the CPU spine
extern bool posted_requests_allowed;
struct ShardComm {                 // what the caller provides for a picture coded by several ranks (RCCL / gloo / shared memory behind it)
  virtual ~ShardComm() {}
  virtual void allgather(const void* send, void* recv, size_t bytes_per_rank) = 0;     // recv: shard_world * bytes_per_rank, rank k's contribution at k * bytes_per_rank
};
enum { SPINE_LANES = 2048 };       // CTU rows in flight at once (all pictures together); a lane owns 16 stash slots
typedef hop_cu_part Part;          // one 4x4 unit of the CU data (TComDataCU's per-partition arrays)

struct CuData {                    // TComDataCU as the RD search uses it (one CU of the quadtree, or the whole CTU)
  int ctu_addr, ctu_x, ctu_y, abs_idx, depth, x, y, size, num_part;
  Part p[256];
  uint64_t fbits[256];             // at a coded CU's first partition: the fractional bits of its syntax (what the final pass of the CTU adds to the coder's fraction)
  double cost; uint32_t bits, dist;
  int slot;                        // the candidate slot its reconstruction lies in
};

// ---- candidate evaluation requests (the boundary between the spine and the kernels) ----
struct InterEval {                 // encodeResAndCalcRdInterCU of a CU whose prediction is in the prediction picture
  hop_rqt_job job; hop_cu_syntax syn; int skip_res;
  // ... or is made first: the final motion compensation of the CU's PUs (TEncSearch.cpp:4158, TEncCu.cpp:1304) travels with the evaluation it is for (EncConfig::fuse_pred)
  // instead of as a request of its own -- nothing else reads it
  int n_pred; hop_pred_job pred[4];
};
struct IntraEval {                 // the body of xCheckRDCostIntra
  hop_rqt_job job; hop_intra_cu_syntax syn; hop_intra_rqt_opt opt; hop_intra_search_job sjob; int part_nxn;
};
struct EvalResult {
  double cost; uint32_t bits, dist; int skipped, root_cbf;
  Coder after;                     // the coder the candidate leaves (CI_TEMP_BEST)
  uint8_t tr_idx[256], cbf[3][256], tskip[3][256];
  int luma_dir[4], chroma_dir;     // intra
};

// A backend owns the pictures (original, SS reference, prediction picture, reconstruction picture) and evaluates requests.
// Every call is synchronous for the calling lane; `lane` names the CTU worker (stash slots are per lane).
class Backend {
 public:
  virtual ~Backend() {}
  virtual void begin_frame() = 0;                                                        // SS reference to the sentinel
  // where the calling lane is in its CTU (a number that grows along the reference's order of operations and means the same operation in every CTU): a batching backend
  // serves the requests with the smallest tag first, so that CTUs coded side by side stay in step and their requests meet
  virtual void set_tag(int /*lane*/, uint64_t /*tag*/) {}
  virtual void me_search(int lane, int n, const hop_pu_job* jobs, hop_pu_result* res) = 0;   // SS + fractional + GT search
  virtual void pred_inter(int lane, int n, const hop_pred_job* jobs) = 0;                // into the prediction picture
  virtual void distortion(int lane, int n, const hop_dist_job* jobs, uint32_t* out) = 0; // original vs prediction picture
  virtual void valid_pattern(int lane, int n, const int32_t* xywh_mv /* 6 per item: x, y, w, h, mvx, mvy (quarter-pel) */, uint8_t* out) = 0;
  // the jobs ONE AFTER THE OTHER: predict job i into the prediction picture, then its luma distortion `kind` (HOP_DIST_*) against the original over the job's
  // rectangle into out[i]; afterwards the picture holds the last job's prediction (the candidates of one PU overwrite each other, as in the reference's m_tmpYuvPred)
  virtual void pred_cost(int lane, int n, const hop_pred_job* jobs, int kind, uint32_t* out) {
    for (int i = 0; i < n; i++) {
      pred_inter(lane, 1, jobs + i);
      hop_dist_job d; d.x = jobs[i].pu_x; d.y = jobs[i].pu_y + jobs[i].dst_row_off; d.w = jobs[i].w; d.h = jobs[i].h; d.comp = 0; d.kind = kind;
      distortion(lane, 1, &d, out + i);
    }
  }
  // candidate evaluation; afterwards the reconstruction picture holds the candidate's reconstruction in the CU's area
  virtual void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& out) = 0;
  virtual void intra_cu(int lane, const IntraEval& e, const Coder& in, EvalResult& out) = 0;
  // reconstruction picture <-> stash slot (lane, slot): rect = x, y, size
  virtual void recon_save(int lane, int slot, int x, int y, int size) = 0;
  virtual void recon_restore(int lane, int slot, int x, int y, int size) = 0;
  virtual void commit(int lane, int x, int y, int size) = 0;                             // reconstruction picture -> SS reference (xCopyYuv2SSRef)
  // the two requests a CU's decision ends with when the split did not win (xCopyYuv2Pic, xCopyYuv2SSRef: TEncCu.cpp:869-880) as one
  virtual void restore_commit(int lane, int slot, int x, int y, int size) { recon_restore(lane, slot, x, y, size); commit(lane, x, y, size); }
  // a picture coded by several ranks (EncConfig::shard): the w x h block at (x, y) of the reconstruction picture (w, h multiples of 8: a CTU's part inside the picture) out as
  // packed planes (w * h luma, then the caller's Cb and Cr arrays of w/2 * h/2), and back in on another rank -- into the reconstruction picture AND the SS reference (the
  // commit of every CU of the CTU, xCopyYuv2SSRef)
  virtual void export_block(int /*x*/, int /*y*/, int /*w*/, int /*h*/, int16_t* /*y_out*/, int16_t* /*cb_out*/, int16_t* /*cr_out*/) { throw 1; }
  virtual void import_block(int /*x*/, int /*y*/, int /*w*/, int /*h*/, const int16_t* /*y_in*/, const int16_t* /*cb_in*/, const int16_t* /*cr_in*/) { throw 1; }
  // fn(0) ... fn(n - 1), each issuing requests of its own: a backend that batches runs them side by side (their requests then meet in the batches), the others one
  // after the other -- the answers, and so the results, are the same
  virtual void fork_join(int n, const std::function<void(int)>& fn) { for (int i = 0; i < n; i++) fn(i); }
};

// The n-forms a backend may offer so that requests of several CTUs in flight are served by one launch chain.  The defaults loop over the single forms.
class BatchInner : public Backend {
 public:
  virtual void inter_n(int n, const InterEval* const* e, const Coder* const* in, EvalResult* const* out) { for (int i = 0; i < n; i++) inter_cu(0, *e[i], *in[i], *out[i]); }
  virtual void intra_n(int n, const IntraEval* const* e, const Coder* const* in, EvalResult* const* out) { for (int i = 0; i < n; i++) intra_cu(0, *e[i], *in[i], *out[i]); }
  virtual void stash_n(int n, const int32_t* rect4 /* x, y, size, lane * 16 + slot */, int restore) {
    for (int i = 0; i < n; i++) { const int32_t* r = rect4 + 4 * i; if (restore) recon_restore(r[3] >> 4, r[3] & 15, r[0], r[1], r[2]); else recon_save(r[3] >> 4, r[3] & 15, r[0], r[1], r[2]); }
  }
  virtual void commit_n(int n, const int32_t* rect4) { for (int i = 0; i < n; i++) commit(0, rect4[4 * i], rect4[4 * i + 1], rect4[4 * i + 2]); }
  // the groups of one round have all been handed over: a backend that issues them on separate streams waits for them here and fills in the answers
  virtual void end_round() {}
  // true: inter_n / intra_n only ISSUE their batch (on a stream of the backend's own) and end_round() completes it, so a batch may be handed over rounds ahead
  virtual bool can_defer() { return false; }
  // m sequences of pred_cost requests (different CTUs, so their rectangles are disjoint): step k of all sequences may run together, the steps in order.
  // jobs / kinds / out: concatenated sequence after sequence, len[s] jobs each
  virtual void pred_cost_n(int m, const int* len, const hop_pred_job* jobs, const int* kinds, uint32_t* out) {
    for (int s = 0, at = 0; s < m; at += len[s], s++) pred_cost(0, len[s], jobs + at, kinds[s], out + at);
  }
};

// A backend that forwards to another one and appends every request with its answer to a binary log (HOP_SPINE_LOG=<file>: diagnostics -- two backends serving the same
// picture write the same log up to the first request they answer differently).  Record: int32 kind, int32 n, uint32 bytes in, uint32 bytes out, then the bytes.
class LogBackend : public BatchInner {
 public:
  LogBackend(Backend* inner, const char* path);
  ~LogBackend();
  void begin_frame() { in_->begin_frame(); }
  void me_search(int lane, int n, const hop_pu_job* jobs, hop_pu_result* res) { in_->me_search(lane, n, jobs, res); rec(0, n, jobs, n * sizeof(hop_pu_job), res, n * sizeof(hop_pu_result)); }
  void pred_inter(int lane, int n, const hop_pred_job* jobs) { in_->pred_inter(lane, n, jobs); rec(1, n, jobs, n * sizeof(hop_pred_job), NULL, 0); }
  void distortion(int lane, int n, const hop_dist_job* jobs, uint32_t* out) { in_->distortion(lane, n, jobs, out); rec(2, n, jobs, n * sizeof(hop_dist_job), out, n * 4); }
  void valid_pattern(int lane, int n, const int32_t* q, uint8_t* out) { in_->valid_pattern(lane, n, q, out); rec(3, n, q, n * 24, out, n); }
  void pred_cost(int lane, int n, const hop_pred_job* jobs, int kind, uint32_t* out) { in_->pred_cost(lane, n, jobs, kind, out); rec(9, n, jobs, n * sizeof(hop_pred_job), out, n * 4); }
  // (fields of the answer the spine does not read are normalised in the log's copy: the cost, the split contexts, directions of PUs that do not exist)
  void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& out) {
    in_->inter_cu(lane, e, in, out);
    EvalResult o = out; o.cost = 0; o.after.split[0] = o.after.split[1] = o.after.split[2] = o.after.pad = 0; o.luma_dir[0] = o.luma_dir[1] = o.luma_dir[2] = o.luma_dir[3] = o.chroma_dir = 0;
    rec2(4, &e, sizeof(e), &in, sizeof(in), &o, sizeof(o));
  }
  void intra_cu(int lane, const IntraEval& e, const Coder& in, EvalResult& out) {
    in_->intra_cu(lane, e, in, out);
    EvalResult o = out; o.cost = 0; o.after.split[0] = o.after.split[1] = o.after.split[2] = o.after.pad = 0; if (!e.part_nxn) o.luma_dir[1] = o.luma_dir[2] = o.luma_dir[3] = 0;
    rec2(5, &e, sizeof(e), &in, sizeof(in), &o, sizeof(o));
  }
  void recon_save(int lane, int slot, int x, int y, int size) { in_->recon_save(lane, slot, x, y, size); }
  void recon_restore(int lane, int slot, int x, int y, int size) { in_->recon_restore(lane, slot, x, y, size); }
  void commit(int lane, int x, int y, int size) { in_->commit(lane, x, y, size); }
 private:
  Backend* in_; FILE* f_;
  void rec(int kind, int n, const void* a, size_t na, const void* b, size_t nb);
  void rec2(int kind, const void* a, size_t na, const void* a2, size_t na2, const void* b, size_t nb);
};

// the luma directions of a finished intra search and their most probable modes (TComDataCU::getIntraDirLumaPredictor, TComDataCU.cpp:1772-1830) into the CU's
// syntax elements: neighbours outside the CU from sj.left_dir / above_dir, inside it the PUs decided before
void intra_syntax_dirs(hop_intra_cu_syntax& syn, const hop_intra_search_job& sj, const int dirs[4]);

class Encoder {
 public:
  Encoder(const EncConfig& cfg, Backend* be);
  void encode_frame(int first_ctus = 0);             // the CTUs in raster order (all, or the first `first_ctus`)
  // CTU rows as a wavefront: row r starts CTU c once row r - 1 has finished CTU c + lag - 1 (lag 5 covers the reach of the SS / GT search); one thread per
  // row, their requests rendezvous and go to `inner` in batches.  With cfg.wpp the rows' coders are synchronised as WaveFrontSynchro does (TEncSlice.cpp:1027-1051,
  // :1158-1161) and the result equals the reference run with one substream per row; without it the rows would need the coder of the previous row's END (raster order),
  // which serialises them: wavefront mode requires cfg.wpp.
  void encode_frame_wavefront(BatchInner* inner, int lag = 5, int max_rows_in_flight = 0);
  // the same wavefront without batching: row r talks to lanes[r % n_lanes] directly (backends that run concurrently, e.g. one stream each); at most n_lanes rows in flight
  void encode_frame_wavefront_direct(Backend* const* lanes, int n_lanes, int lag = 5);
  // n independent pictures of equal geometry (their configurations differ in y_origin: a stacked context) coded side by side: the rows of all of them on one
  // rendezvous, so that a batch serves CTUs of every picture; each picture's result is what encode_frame_wavefront gives for it alone
  // lane_base: first lane number (stash slots are per lane) when several groups of pictures run side by side on one context, each on a backend of its own;
  // begin = false: the caller has called begin_frame() for all of them
  static void encode_pictures_wavefront(Encoder* const* encs, int n, BatchInner* inner, int lag = 5, int lane_base = 0, bool begin = true, int threads = 0);
 private:
  static void wavefront_many(Encoder* const* encs, int n_pic, BatchInner* inner, Backend* const* lanes, int n_lanes, int lag, int lane_base = 0, bool begin = true, int threads = 0);
 public:
  const EncConfig& config() const { return cfg_; }
  int n_ctu() const { return wctu_ * hctu_; }
  // results
  std::vector<double>   ctu_cost;                    // per CTU: getTotalCost of compressCU (what cost.csv records)
  std::vector<uint32_t> ctu_bits, ctu_dist;
  std::vector<uint16_t> ctu_rd_fraction;             // the go-on coder's carried fraction (m_fracBits & 32767) when each CTU's compressCU returns
  std::vector<Part>     pic;                         // 256 parts per CTU, z-order
  std::vector<Coder>    ctu_entry;                   // coder at the start of every CTU
  std::vector<Coder>    ctu_exit;                    // ... and the coder it left (a picture coded by several ranks: what travels to the others; the row's second CTU's is the WaveFrontSynchro snapshot)
  FILE* trace;                                       // optional: one line per candidate that reaches xCheckBestMode (written CTU by CTU in raster order)
  uint64_t n_candidates;
  std::vector<std::string> ctu_trace;                // the lines of each CTU while the picture is in flight
  std::vector<uint8_t>  committed;                   // per 8x8 block: its reconstruction is in the SS reference (what TComRdCost::isValidPattern's sentinel test sees)
  uint64_t batch_rounds, batch_requests;             // wavefront mode: rendezvous rounds and requests served
  // wavefront mode, the visibility check (CtuWorker::reach_check): a motion search whose window -- the predictor +- the range, plus the block and the GT patch around it --
  // covers samples that the reference, coding the CTUs in raster order, sees otherwise than the wavefront does: (below) committed samples of the CTU rows BELOW the current
  // one (the reference has coded nothing there yet), (above) samples of the rows ABOVE that the wavefront has not coded yet (the reference has).  The lag of 5 CTUs keeps
  // both away from windows around vectors of ordinary length; a long predictor can carry a window there.  Counted per search request, with the first CTU of each kind
  std::atomic<long> reach_below, reach_above; std::atomic<int> first_below, first_above;
  double batch_run_s = 0;                            // ... the wall time of the whole wavefront (rows' host work + serving)
  double batch_serve_s = 0;                          // ... and the wall time spent serving them (one thread; the rows' own host work runs between the rounds)
 private:
  friend class CtuWorker;
  EncConfig cfg_; Backend* be_; int wctu_, hctu_;
};

}  // namespace hopspine
