// hop_spine.cpp -- the host RD spine.  Citations: paths under /root/reference/source/Lib.
//   TEncCu::xCompressCU TLibEncoder/TEncCu.cpp:371-892, deriveTestModeAMP :292-356, xCheckRDCostMerge2Nx2N :1243-1395, xCheckRDCostInter :1399-1453,
//   xCheckRDCostIntra :1455-1507, xCheckBestMode :1557-1590; TEncSearch::predInterSearch TLibEncoder/TEncSearch.cpp:3141-4169 (one reference list with
//   the SS picture: the bi-predictive half is unreachable), xEstimateMvPredAMVP :4173-4262, xGetTemplateCost :4411-4477, xCheckBestMVP :4364-4409,
//   xMergeEstimation :2992-3106, xGetInterPredictionError :2951-2977, xGetBlkBits :4294-4353; TComDataCU::fillMvpCand TLibCommon/TComDataCU.cpp:3297-3478,
//   getInterMergeCandidates :2761-3204 with the micro-image candidates :2620-2748, the neighbour access :1270-1635, getPartOffset :2251-2296,
//   getIntraDirLumaPredictor :1772-1830, getCtxSplitFlag :1832, getCtxSkipFlag :1888; TEncSlice::compressSlice TLibEncoder/TEncSlice.cpp:1000-1196
//   (compressCU, then encodeCU on the CTU's entry coder: that pass is what the next CTU starts from).
// The spine keeps the reference's bookkeeping literally where results depend on it: which coder the split flag is counted on (the go-on coder as
// the last candidate left it), the fields a PU carries while its merge candidates are rated (the GT vectors of its motion search), the partition
// index handed to the micro-image candidates.
#include "hop_spine.h"
#include <map>
#include <deque>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <atomic>
#include <functional>
#include <chrono>
#include <sys/mman.h>

#if !defined(__x86_64__)
#error "hop_spine.cpp: the fiber switch (hop_fiber_switch) is written for x86-64 System V"
#endif

namespace hopspine {

bool posted_requests_allowed = true;

static const double MAX_DOUBLE = 1.7e+308;           // TLibCommon/CommonDef.h
static const uint32_t MAX_UINT = 0xFFFFFFFFu;
static const int CTU = 64;

// ---------------------------------------------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------------------------------------------
void default_hop_config(EncConfig& c, int pic_w, int pic_h, int qp, int mi_size) {
  memset(&c, 0, sizeof(c));
  c.pic_w = pic_w; c.pic_h = pic_h; c.bit_depth = 8; c.qp = qp; c.slice_type = 3;
  c.search_range = 128; c.mi_size = mi_size; c.mi_merge = 1; c.max_merge_cand = 5;
  c.amp = 1; c.fen = 1; c.hadme = 1; c.fdm = 1; c.esd = 0; c.cfm = 0; c.ecu = 0;
  c.log2_max_tu = 5; c.log2_min_tu = 2; c.tu_max_depth_inter = 3; c.tu_max_depth_intra = 3;
  c.sign_hide = 1; c.use_ts = 1; c.ts_fast = 1; c.strong_intra = 1; c.wpp = 0;
  { const char* e = getenv("HOP_SPINE_FUSE_PRED"); c.fuse_pred = e ? atoi(e) != 0 : 1; }
  finish_config(c);
}

void default_plain_config(EncConfig& c, int pic_w, int pic_h, int qp, int bit_depth) {
  default_hop_config(c, pic_w, pic_h, qp, 16);
  c.slice_type = 2; c.bit_depth = bit_depth; c.mi_merge = 0;
  finish_config(c);
}

void finish_config(EncConfig& c) {
  // TEncSlice::initEncSlice, TLibEncoder/TEncSlice.cpp:358-462: I / ISS slice at depth 0, GOP size 1, no QP offsets
  static const uint8_t chroma_scale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                            29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };   // g_aucChromaScale, TComRom.cpp
  double lambda = 0.57 * pow(2.0, (c.qp - 12) / 3.0);
  if (!c.hadme && c.slice_type != 2) lambda *= 0.95;
  c.lambda = lambda; c.sqrt_lambda = sqrt(lambda); c.lambda_sad = (uint32_t)floor(65536.0 * c.sqrt_lambda);
  const int qpc = c.qp < 0 ? 0 : c.qp > 57 ? 57 : c.qp;
  const double w = pow(2.0, (c.qp - chroma_scale[qpc]) / 3.0);
  c.dist_weight[0] = c.dist_weight[1] = w;
  c.lambda_rdoq[0] = lambda; c.lambda_rdoq[1] = c.lambda_rdoq[2] = lambda / w;
  // TComTrQuant::setQPforQuant, TLibCommon/TComTrQuant.cpp:192-214 (qpBdOffset = 6 * (bit depth - 8))
  const int off = 6 * (c.bit_depth - 8);
  c.qp_scaled[0] = c.qp + off;
  int q = c.qp < -off ? -off : c.qp > 57 ? 57 : c.qp;
  c.qp_scaled[1] = c.qp_scaled[2] = q < 0 ? q + off : chroma_scale[q] + off;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// z-order
// ---------------------------------------------------------------------------------------------------------------------------------
static inline int zidx(int x4, int y4) {            // g_auiRasterToZscan for a 16x16 grid of 4x4 units: x in the even bits
  int z = 0;
  for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1));
  return z;
}
static inline void zpos(int z, int& x4, int& y4) {
  x4 = y4 = 0;
  for (int b = 0; b < 4; b++) { x4 |= ((z >> (2 * b)) & 1) << b; y4 |= ((z >> (2 * b + 1)) & 1) << b; }
}
static inline int zpix(int px, int py) { return zidx((px & 63) >> 2, (py & 63) >> 2); }

static double calc_rd_cost(uint32_t bits, uint32_t dist, double lambda) {     // TComRdCost::calcRdCost, DF_DEFAULT (TComRdCost.cpp:59-111)
  double d = (double)dist + (double)((int)(bits * lambda + .5));
  return (double)(uint32_t)floor(d);
}
static inline uint32_t component_bits(int v) { return hop_component_bits(v); }

static void part_init(Part& p, int depth) {          // TComDataCU::initEstData, TComDataCU.cpp:586-665
  memset(&p, 0, sizeof(p));
  p.depth = (uint8_t)depth; p.pred_mode = MODE_NONE; p.part_size = SIZE_NONE; p.luma_dir = DC_IDX;
  p.ref_idx = -1; p.mvp_idx = -1; p.mvp_num = -1;
}

// PU geometry ------------------------------------------------------------------------------------------------------------------------
static int num_pus(int ps) { return ps == SIZE_2Nx2N ? 1 : ps == SIZE_NxN ? 4 : 2; }
// TComDataCU::getPartIndexAndSize (TComDataCU.cpp:2210-2249) in samples: offset of the PU inside the CU and its size
static void pu_rect(int ps, int S, int pu, int& ox, int& oy, int& w, int& h) {
  ox = oy = 0; w = h = S;
  switch (ps) {
    case SIZE_2NxN:  h = S >> 1; oy = pu ? S >> 1 : 0; break;
    case SIZE_Nx2N:  w = S >> 1; ox = pu ? S >> 1 : 0; break;
    case SIZE_NxN:   w = h = S >> 1; ox = (pu & 1) ? S >> 1 : 0; oy = (pu >> 1) ? S >> 1 : 0; break;
    case SIZE_2NxnU: h = pu ? (S >> 2) + (S >> 1) : S >> 2; oy = pu ? S >> 2 : 0; break;
    case SIZE_2NxnD: h = pu ? S >> 2 : (S >> 2) + (S >> 1); oy = pu ? (S >> 2) + (S >> 1) : 0; break;
    case SIZE_nLx2N: w = pu ? (S >> 2) + (S >> 1) : S >> 2; ox = pu ? S >> 2 : 0; break;
    case SIZE_nRx2N: w = pu ? S >> 2 : (S >> 2) + (S >> 1); ox = pu ? (S >> 2) + (S >> 1) : 0; break;
    default: break;
  }
}
// TComDataCU::getPartPosition (TComDataCU.cpp:3229-3288): only the size is used by its callers here; partIdx is whatever the caller hands over
static void part_position_size(int ps, int S, unsigned partIdx, int& w, int& h) {
  w = h = S;
  switch (ps) {
    case SIZE_2NxN:  h = S >> 1; break;
    case SIZE_Nx2N:  w = S >> 1; break;
    case SIZE_NxN:   w = h = S >> 1; break;
    case SIZE_2NxnU: h = partIdx == 0 ? S >> 2 : (S >> 2) + (S >> 1); break;
    case SIZE_2NxnD: h = partIdx == 0 ? (S >> 2) + (S >> 1) : S >> 2; break;
    case SIZE_nLx2N: w = partIdx == 0 ? S >> 2 : (S >> 2) + (S >> 1); break;
    case SIZE_nRx2N: w = partIdx == 0 ? (S >> 2) + (S >> 1) : S >> 2; break;
    default: break;
  }
}
// TComDataCU::getPartOffset (TComDataCU.cpp:2251-2296), incl. SIZE_nLx2N's offY = height
static void part_offset(int ps, int S, int pu, int& offx, int& offy) {
  offx = offy = 0;
  switch (ps) {
    case SIZE_2NxN:  offy = pu ? S >> 1 : 0; break;
    case SIZE_Nx2N:  offx = pu ? S >> 1 : 0; break;
    case SIZE_NxN:   offx = (pu & 1) ? S >> 1 : 0; offy = (pu & 2) ? S >> 1 : 0; break;
    case SIZE_2NxnU: offy = pu ? S >> 2 : 0; break;
    case SIZE_2NxnD: offy = pu ? (S >> 2) + (S >> 1) : 0; break;
    case SIZE_nLx2N: offx = pu ? S >> 2 : 0; offy = S; break;
    case SIZE_nRx2N: offx = pu ? (S >> 2) + (S >> 1) : 0; break;
    default: break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// one CTU
// ---------------------------------------------------------------------------------------------------------------------------------
struct MvField { int16_t mv[2]; int8_t ref; };
struct MergeCands { MvField f[5]; uint8_t dir[5]; int n; };
struct AmvpInfo { int n; int16_t cand[3][2]; };

class CtuWorker {
 public:
  uint64_t n_cand_ = 0;
  // request tags: (sequence number of the quadtree node visit, candidate slot, step inside the candidate)
  uint64_t tag_seq_ = 0; int tag_slot_ = 0;
  static int node_index(int d, int abs_idx) { int i = 0; if (d >= 1) i += 1 + (abs_idx >> 6) * 21; if (d >= 2) i += 1 + ((abs_idx >> 4) & 3) * 5; if (d >= 3) i += 1 + ((abs_idx >> 2) & 3); return i; }
  void tag_enter(int d, int abs_idx) { tag_seq_ = 2 * (uint64_t)node_index(d, abs_idx); tag_slot_ = 0; tag_step(0); }
  void tag_exit(int d, int abs_idx) { static const int sub[4] = { 85, 21, 5, 1 }; tag_seq_ = 2 * (uint64_t)(node_index(d, abs_idx) + sub[d] - 1) + 1; tag_slot_ = 3 - d; tag_step(0); }
  void tag_cand(int slot) { tag_slot_ = slot; tag_step(0); }
  void tag_step(int step) { be->set_tag(lane_, (tag_seq_ << 20) | ((uint64_t)tag_slot_ << 8) | (uint64_t)step); }
  // kid: a worker that evaluates ONE candidate of its parent (candidate slots): it gets the CU storage of the depth it works at only
  CtuWorker(Encoder& e, int lane, Backend* backend = NULL, bool kid = false) : E(e), cfg(e.cfg_), be(backend ? backend : e.be_), lane_(lane) {
    for (int d = 0; d < 4; d++) { store_[d][0] = store_[d][1] = NULL; best_[d] = temp_[d] = NULL; if (!kid) ensure(d); }
  }
  ~CtuWorker() { for (int d = 0; d < 4; d++) { delete store_[d][0]; delete store_[d][1]; } for (size_t i = 0; i < kids_.size(); i++) delete kids_[i]; }
  void ensure(int d) { if (!store_[d][0]) { store_[d][0] = new CuData; store_[d][1] = new CuData; best_[d] = store_[d][0]; temp_[d] = store_[d][1]; } }
  void compress_ctu(int addr, const Coder& entry, Coder& exit);
 private:
  Encoder& E; const EncConfig& cfg; Backend* be; int lane_;
  int slot_ = 0;                                                          // candidate slot this worker predicts and reconstructs in (0: the pictures themselves)
  int row_off() const { return slot_ * cfg.slot_pitch; }
  CuData* store_[4][2]; CuData* best_[4]; CuData* temp_[4];
  std::vector<CtuWorker*> kids_;                                          // the candidate workers of this worker, by slot; made once, used for every node
  CtuWorker(const CtuWorker&); CtuWorker& operator=(const CtuWorker&);    // (not copyable)
  enum { CI_CURR = 0, CI_NEXT = 1, CI_TEMP = 2 };
  Coder sb_[4][3]; Coder goon_;
  struct SpecCand; std::vector<SpecCand> spec_intra_[4];                   // candidate slots: the intra candidates of the node, evaluated with the first batch
  struct SpecSet; std::vector<SpecSet> spec_set_;                          // per depth: a CU's candidates as a list (spec_build); ready: already evaluated with the parent's
  std::vector<SpecCand> spec_amp_[4];                                      // ... and, with 24 slots or more, the AMP shapes in both of their forms (all vectors / merge only): the derivation then picks
  int ctu_addr_, ctu_x_, ctu_y_;

  // data model
  void init_cu(CuData& c, int abs_idx, int depth, int x, int y);
  void init_est(CuData& c) { for (int i = 0; i < c.num_part; i++) part_init(c.p[i], c.depth); c.cost = MAX_DOUBLE; c.bits = 0; c.dist = 0; c.slot = 0; }
  Part& pic_part(int px, int py) { return E.pic[(size_t)((py >> 6) * E.wctu_ + (px >> 6)) * 256 + zpix(px, py)]; }
  const Part* part_at(const CuData& c, int px, int py) {
    if (px >= c.x && px < c.x + c.size && py >= c.y && py < c.y + c.size) return &c.p[zpix(px, py) - c.abs_idx];
    return &pic_part(px, py);
  }
  void copy_to_pic(const CuData& c) { memcpy(&E.pic[(size_t)c.ctu_addr * 256 + c.abs_idx], c.p, sizeof(Part) * c.num_part); }
  void set_parts(CuData& c, int ox, int oy, int w, int h, void (*fn)(Part&, const void*), const void* arg) {
    for (int yy = oy; yy < oy + h; yy += 4) for (int xx = ox; xx < ox + w; xx += 4) fn(c.p[zpix(c.x + xx, c.y + yy) - c.abs_idx], arg);
  }
  // neighbours (TComDataCU::getPULeft ... getPUBelowLeftAdi): the unit the reference would return, or NULL
  const Part* nb_left(const CuData& c, int x, int y) { return ((x & 63) || x > 0) ? part_at(c, x - 4, y) : NULL; }
  const Part* nb_above(const CuData& c, int x, int y, bool planar_at_ctu_boundary = false) {
    if (y & 63) return part_at(c, x, y - 4);
    if (planar_at_ctu_boundary || y == 0) return NULL;
    return part_at(c, x, y - 4);
  }
  const Part* nb_above_left(const CuData& c, int x, int y) {
    const bool ok = (x & 63) ? ((y & 63) ? true : y > 0) : ((y & 63) ? x > 0 : (x > 0 && y > 0));
    return ok ? part_at(c, x - 4, y - 4) : NULL;
  }
  bool ar_avail(int x, int y, int k) {              // unit k to the right of (x, y), one row up (k = 1: getPUAboveRight)
    if (x + 4 * k >= cfg.pic_w) return false;
    if (((x & 63) >> 2) + k < 16) return (y & 63) ? zpix(x, y) > zpix(x + 4 * k, y - 4) : y > 0;
    return (y & 63) ? false : y > 0;
  }
  bool bl_avail(int x, int y, int k) {              // unit k below (x, y), one column to the left
    if (y + 4 * k >= cfg.pic_h) return false;
    if (((y & 63) >> 2) + k < 16) return (x & 63) ? zpix(x, y) > zpix(x - 4, y + 4 * k) : x > 0;
    return false;
  }
  const Part* nb_above_right(const CuData& c, int x, int y) { return ar_avail(x, y, 1) ? part_at(c, x + 4, y - 4) : NULL; }
  const Part* nb_below_left(const CuData& c, int x, int y) { return bl_avail(x, y, 1) ? part_at(c, x - 4, y + 4) : NULL; }

  // the spine
  void compress_cu(int d, int parent_part_size);
  void check_best_mode(int d, bool save_recon);
  void reach_check(int px, int py, int w, int h, const int* r6);
  bool raster_below_ = [] { const char* e = getenv("HOP_SPINE_RASTER_BELOW"); return e ? atoi(e) != 0 : true; }();
  struct PendingSave { bool on; int d, x, y, size; } psave_ = { false, 0, 0, 0, 0 };
  void flush_save();
  void check_merge_2Nx2N(int d, bool* early_skip);
  void check_inter(int d, int part_size, bool use_mrg);
  void check_intra(int d, int part_size);
  void eval_intra(int d, int part_size);
  void spec_run(int d, SpecCand& sc, const CuData& tmpl, int slot);
  void spec_adopt(int d, SpecCand& sc);
  void spec_inter_phase(int d, std::vector<SpecCand>& cands);
  void check_merge_and_inter_spec(int d);
  void spec_build(int d, SpecSet& S);
  void spec_slots_of(const std::vector<SpecCand>& cands, int base, int L, std::vector<int>& slot);
  // candidate slots per level: from 48 on the slots are levels of 24 -- a CU's candidates and, with them, those of the chain of first sub-CUs below it, one level each
  // (check_merge_and_inter_spec)
  int spec_level_slots() const { return cfg.spec_slots >= 48 ? 24 : cfg.spec_slots; }
  bool pred_inter_search(CuData& c, int part_size, bool use_mrg);
  void fill_mvp_cand(const CuData& c, int pu, AmvpInfo& info);
  void merge_candidates(const CuData& c, int pu, MergeCands& mc);
  bool mi_cand(const CuData& c, int which, int ux, int uy, int16_t mv[2]);
  uint32_t inter_pred_error(CuData& c, int pu);
  void pu_pred_job(const CuData& c, int pu, hop_pred_job& j);
  void motion_comp_pu(CuData& c, int pu);
  hop_pred_job held_[4]; int n_held_ = 0;      // fuse_pred: the candidate's final predictions until its evaluation is asked for
  void clip_mv(const CuData& c, int& h, int& v) const;
  bool valid_pattern(int px, int py, int w, int h, int mvx, int mvy);
  uint32_t split_flag_bits(const CuData& c, int d, Coder& k);
  void eval_inter(int d, bool skip_res);
  void fill_intra_eval(const CuData& c, int part_size, IntraEval& e);
  void set_cu_field_u8(CuData& c, size_t off, uint8_t v) { for (int i = 0; i < c.num_part; i++) ((uint8_t*)&c.p[i])[off] = v; }
  void trace_candidate(const CuData& c);
  uint64_t final_walk(int x, int y, int size, int d, Coder& k);
};

// a CU's data without the partitions it does not have (an 8x8 CU owns 4 of the 256)
static inline void cu_copy(CuData& dst, const CuData& src) {
  dst.ctu_addr = src.ctu_addr; dst.ctu_x = src.ctu_x; dst.ctu_y = src.ctu_y; dst.abs_idx = src.abs_idx; dst.depth = src.depth; dst.x = src.x; dst.y = src.y; dst.size = src.size;
  dst.num_part = src.num_part; dst.cost = src.cost; dst.bits = src.bits; dst.dist = src.dist; dst.slot = src.slot;
  memcpy(dst.p, src.p, sizeof(Part) * src.num_part); memcpy(dst.fbits, src.fbits, sizeof(uint64_t) * src.num_part);
}

void CtuWorker::init_cu(CuData& c, int abs_idx, int depth, int x, int y) {
  c.ctu_addr = ctu_addr_; c.ctu_x = ctu_x_; c.ctu_y = ctu_y_; c.abs_idx = abs_idx; c.depth = depth; c.x = x; c.y = y; c.size = CTU >> depth; c.num_part = 256 >> (2 * depth);
  for (int i = 0; i < c.num_part; i++) { part_init(c.p[i], depth); c.fbits[i] = 0; }
  c.cost = MAX_DOUBLE; c.bits = 0; c.dist = 0; c.slot = 0;
}

void CtuWorker::clip_mv(const CuData& c, int& h, int& v) const {          // TComDataCU::clipMv, TComDataCU.cpp:3492-3504 (the CU's position)
  const int hmax = (cfg.pic_w + 8 - c.x - 1) * 4, hmin = (-CTU - 8 - c.x + 1) * 4;
  const int vmax = (cfg.pic_h + 8 - c.y - 1) * 4, vmin = (-CTU - 8 - c.y + 1) * 4;
  h = std::min(hmax, std::max(hmin, h)); v = std::min(vmax, std::max(vmin, v));
}

// TComRdCost::isValidPattern (TLibCommon/TComRdCost.cpp:430-443) without touching the device: the two probes read the SS reference at linear addresses of the padded
// plane; a sample there is the sentinel unless the picture sample it copies (itself, or for a margin sample the nearest picture sample, TComPicYuv::extendPicBorder)
// belongs to a CU that has been committed -- reconstructed samples are clipped to >= 0 and never equal the sentinel.
bool CtuWorker::valid_pattern(int px, int py, int w, int h, int mvx, int mvy) {
  const int stride = cfg.pic_w + 160, W8 = cfg.pic_w >> 3;
  auto probe = [&](long off) -> bool {                // off: linear offset from picture sample (0, 0)
    long o = off + 80L * stride + 80;                 // from the first sample of the padded plane
    if (o < 0) return false;                          // guard rows hold the sentinel
    const long by = o / stride, bx = o % stride;
    if (by >= cfg.pic_h + 160) return false;
    int x = (int)bx - 80, y = (int)by - 80;
    x = x < 0 ? 0 : x >= cfg.pic_w ? cfg.pic_w - 1 : x; y = y < 0 ? 0 : y >= cfg.pic_h ? cfg.pic_h - 1 : y;
    if (raster_below_ && y >= ctu_y_ + CTU) return false;            // the reference codes in raster order: nothing below the current CTU row exists yet, whatever a CTU row that
                                                                      // runs behind this one in the wavefront has committed there (HOP_SPINE_RASTER_BELOW=0: as the map has it)
    return E.committed[(size_t)(y >> 3) * W8 + (x >> 3)] != 0;
  };
  const long lb = (long)(py + (mvy >> 2) + h + 4) * stride + (px + (mvx >> 2));
  return probe(lb) && probe(lb + w + 4);
}

// (see Encoder::reach_below): the window of a motion search of the PU (px, py, w, h) with the range r6 of hop_set_search_range (left, right, top, bottom: integer vectors
// relative to the PU) against the coded area as the wavefront has it.  The searches add at most the block's half size and the filters' reach around the displaced block
void CtuWorker::reach_check(int px, int py, int w, int h, const int* r6) {
  if (!cfg.wpp) return;
  const int W8 = cfg.pic_w >> 3, m = (w > h ? w : h) / 2 + 8;
  int x0 = px + r6[0] - m, x1 = px + r6[1] + w + m, y0 = py + r6[2] - m, y1 = py + r6[3] + h + m;
  x0 = x0 < 0 ? 0 : x0; x1 = x1 >= cfg.pic_w ? cfg.pic_w - 1 : x1;
  const int row_top = ctu_y_, row_bot = ctu_y_ + CTU;                    // the current CTU row: [row_top, row_bot)
  bool below = false, above = false;
  if (y1 >= row_bot && row_bot < cfg.pic_h)                               // rows below: committed by a CTU row that runs behind this one (it fills its CTUs from the top: the first unit row tells)
    for (int x = x0 >> 3; x <= x1 >> 3 && !below; x++) below = E.committed[(size_t)(row_bot >> 3) * W8 + x] != 0;
  if (y0 < row_top && row_top > 0)                                        // rows above: not committed yet (a CTU row ahead finishes its CTUs at the bottom: the last unit row tells)
    for (int x = x0 >> 3; x <= x1 >> 3 && !above; x++) above = E.committed[(size_t)((row_top >> 3) - 1) * W8 + x] == 0;
  if (below) { E.reach_below.fetch_add(1); int e = -1; E.first_below.compare_exchange_strong(e, ctu_addr_); }
  if (above) { E.reach_above.fetch_add(1); int e = -1; E.first_above.compare_exchange_strong(e, ctu_addr_); }
}

void CtuWorker::trace_candidate(const CuData& c) {
  n_cand_++;
  if (!E.trace) return;
  char ln[160];
  snprintf(ln, sizeof(ln), "%d %d %d %d %d %d %d %d %u %u %.17g\n", c.depth, c.x, c.y, CTU >> c.p[0].depth, c.p[0].pred_mode, c.p[0].part_size, c.p[0].skip, c.p[0].merge_flag, c.bits, c.dist, c.cost);
  E.ctu_trace[ctu_addr_] += ln;
}

// TEncCu::xCheckBestMode (:1557-1590): strict '<'; the winner's reconstruction is put aside, its coder becomes CI_NEXT_BEST
void CtuWorker::flush_save() { if (!psave_.on) return; psave_.on = false; tag_step(62); be->recon_save(lane_, psave_.d, psave_.x, psave_.y, psave_.size); }
void CtuWorker::check_best_mode(int d, bool save_recon) {
  trace_candidate(*temp_[d]);
  if (temp_[d]->cost < best_[d]->cost) {
    std::swap(best_[d], temp_[d]);
    if (save_recon) {
      // candidates side by side: the decisions are replayed over finished results, one winner after the other into the same stash slot -- only the last one has to
      // get there, and only before its candidate slot is written again (flush_save: before the next batch of candidates, the sub-CUs, the end of the node)
      if (best_[d]->slot > 0) { psave_.on = true; psave_.d = d; psave_.x = best_[d]->x; psave_.y = best_[d]->y + cfg.y_origin + best_[d]->slot * cfg.slot_pitch; psave_.size = best_[d]->size; }
      else { psave_.on = false; tag_step(62); be->recon_save(lane_, d, best_[d]->x, best_[d]->y + cfg.y_origin + best_[d]->slot * cfg.slot_pitch, best_[d]->size); }
    }
    sb_[d][CI_NEXT] = sb_[d][CI_TEMP];
  }
}

// split_cu_flag counted on coder k as the reference's resetBits / encodeSplitFlag / getNumberOfWrittenBits does (TEncCu.cpp:686-688, :803-806; TEncSbac.cpp:731-754)
uint32_t CtuWorker::split_flag_bits(const CuData& c, int d, Coder& k) {
  uint32_t frac = coder_frac(k);
  if (d < 3) {
    const Part* l = nb_left(c, c.x, c.y); const Part* a = nb_above(c, c.x, c.y);
    const int ctx = (l ? (l->depth > d) : 0) + (a ? (a->depth > d) : 0);
    frac += hop_cabac_bin_bits(&k.split[ctx], c.p[0].depth > d ? 1 : 0);
  }
  coder_set_frac(k, frac & 32767);
  return frac >> 15;
}

// ---- AMVP: TComDataCU::fillMvpCand (TComDataCU.cpp:3297-3478), one list with the SS picture ----
bool CtuWorker::mi_cand(const CuData& c, int which, int ux, int uy, int16_t mv[2]) {
  // getMILeftCand / getMIAboveCand / getMIAboveLeftCand (:2620-2748): which = 0 left, 1 above, 2 above-left; (ux, uy) = the unit whose z-index the
  // reference hands over -- also as the "partition index" of getPartPosition, so every PU but one at the CTU's origin reads the second PU's size
  if (which != 1 && (ux & 63) == 0) return false;
  if (which == 1 && (uy & 63) == 0) return false;
  int w, h; part_position_size(c.p[0].part_size, c.size, (unsigned)zpix(ux, uy), w, h);
  const int mi = cfg.mi_size;
  const double xs = which == 1 ? 0.0 : -ceil((double)w / (double)mi), ys = which == 0 ? 0.0 : -ceil((double)h / (double)mi);
  const int16_t mh = (int16_t)((int16_t)(xs * mi) << 2), mvv = (int16_t)((int16_t)(ys * mi) << 2);
  // isMvInsidePic (:2598-2610)
  const int hmax = (cfg.pic_w + 8 - c.x - 1) << 2, hmin = (-CTU - 8 - c.x + 1) * 4, vmax = (cfg.pic_h + 8 - c.y - 1) << 2, vmin = (-CTU - 8 - c.y + 1) * 4;
  if (!(mh >= hmin && mh <= hmax && mvv >= vmin && mvv <= vmax)) return false;
  mv[0] = mh; mv[1] = mvv;
  return true;
}

void CtuWorker::fill_mvp_cand(const CuData& c, int pu, AmvpInfo& info) {
  int ox, oy, w, h; pu_rect(c.p[0].part_size, c.size, pu, ox, oy, w, h);
  const int ltx = c.x + ox, lty = c.y + oy, rtx = ltx + w - 4, rty = lty, lbx = ltx, lby = lty + h - 4;
  info.n = 0;
  auto add = [&](const Part* nb) -> bool {           // xAddMVPCand / xAddMVPCandOrder (:3552-3782) with one SS reference: an inter neighbour's vector as it is
    if (!nb || nb->ref_idx < 0) return false;
    if (info.n < 3) { info.cand[info.n][0] = nb->mv[0]; info.cand[info.n][1] = nb->mv[1]; }
    info.n++;
    return true;
  };
  const Part* bl = nb_below_left(c, lbx, lby); const Part* l = nb_left(c, lbx, lby);
  bool added_smvp = bl && bl->pred_mode != MODE_INTRA;
  if (!added_smvp) added_smvp = l && l->pred_mode != MODE_INTRA;
  bool added = add(bl);
  if (!added) added = add(l);
  if (!added) { added = add(bl); if (!added) add(l); }
  const Part* ar = nb_above_right(c, rtx, rty); const Part* a = nb_above(c, rtx, rty); const Part* al = nb_above_left(c, ltx, lty);
  added = add(ar);
  if (!added) added = add(a);
  if (!added) add(al);
  if (!added_smvp) {
    added = add(ar);
    if (!added) added = add(a);
    if (!added) add(al);
  }
  if (info.n == 2 && info.cand[0][0] == info.cand[1][0] && info.cand[0][1] == info.cand[1][1]) info.n = 1;
  if (info.n > 2) info.n = 2;                        // AMVP_MAX_NUM_CANDS
  if (cfg.mi_merge) {
    if (info.n < 2) {
      int16_t mv[2];
      bool ok = mi_cand(c, 0, lbx, lby, mv);
      if (!ok) ok = mi_cand(c, 1, rtx, rty, mv);
      if (!ok) ok = mi_cand(c, 2, ltx, lty, mv);
      if (ok) { info.cand[info.n][0] = mv[0]; info.cand[info.n][1] = mv[1]; info.n++; }
    }
    if (info.n == 2 && info.cand[0][0] == info.cand[1][0] && info.cand[0][1] == info.cand[1][1]) info.n = 1;
  }
  while (info.n < 2) { info.cand[info.n][0] = 0; info.cand[info.n][1] = 0; info.n++; }
}

// ---- merge candidates: TComDataCU::getInterMergeCandidates (TComDataCU.cpp:2761-3204), P-like slice, no temporal candidate ----
void CtuWorker::merge_candidates(const CuData& c, int pu, MergeCands& mc) {
  const int ps = c.p[0].part_size, maxc = cfg.max_merge_cand;
  int ox, oy, w, h; pu_rect(ps, c.size, pu, ox, oy, w, h);
  const int ltx = c.x + ox, lty = c.y + oy, rtx = ltx + w - 4, rty = lty, lbx = ltx, lby = lty + h - 4;
  for (int i = 0; i < 5; i++) { mc.f[i].mv[0] = mc.f[i].mv[1] = 0; mc.f[i].ref = -1; mc.dir[i] = 0; }
  mc.n = maxc;
  int cnt = 0;
  auto same = [](const Part* a, const Part* b) { return a->inter_dir == b->inter_dir && (!(a->inter_dir & 1) || (a->mv[0] == b->mv[0] && a->mv[1] == b->mv[1] && a->ref_idx == b->ref_idx)); };
  auto take = [&](const Part* nb) { mc.dir[cnt] = nb->inter_dir; mc.f[cnt].mv[0] = nb->mv[0]; mc.f[cnt].mv[1] = nb->mv[1]; mc.f[cnt].ref = nb->ref_idx; cnt++; };
  // (isDiffMER is always true at parallel merge level 2: a neighbouring unit is another 4x4 block)
  const Part* A1 = nb_left(c, lbx, lby);
  const bool okA1 = A1 && !(pu == 1 && (ps == SIZE_Nx2N || ps == SIZE_nLx2N || ps == SIZE_nRx2N)) && A1->pred_mode != MODE_INTRA;
  if (okA1) take(A1);
  if (cnt == maxc) return;
  const Part* B1 = nb_above(c, rtx, rty);
  const bool okB1 = B1 && !(pu == 1 && (ps == SIZE_2NxN || ps == SIZE_2NxnU || ps == SIZE_2NxnD)) && B1->pred_mode != MODE_INTRA;
  if (okB1 && (!okA1 || !same(A1, B1))) take(B1);
  if (cnt == maxc) return;
  const Part* B0 = nb_above_right(c, rtx, rty);
  const bool okB0 = B0 && B0->pred_mode != MODE_INTRA;
  if (okB0 && (!okB1 || !same(B1, B0))) take(B0);
  if (cnt == maxc) return;
  const Part* A0 = nb_below_left(c, lbx, lby);
  const bool okA0 = A0 && A0->pred_mode != MODE_INTRA;
  if (okA0 && (!okA1 || !same(A1, A0))) take(A0);
  if (cnt == maxc) return;
  if (cnt < 4) {
    const Part* B2 = nb_above_left(c, ltx, lty);
    const bool okB2 = B2 && B2->pred_mode != MODE_INTRA;
    if (okB2 && (!okA1 || !same(A1, B2)) && (!okB1 || !same(B1, B2))) take(B2);
  }
  if (cnt == maxc) return;
  if (cfg.mi_merge) {                                // :2942-3035, all three from the PU's first unit
    int16_t mv[2];
    if (cnt < 4 && mi_cand(c, 0, ltx, lty, mv)) { mc.dir[cnt] = 1; mc.f[cnt].mv[0] = mv[0]; mc.f[cnt].mv[1] = mv[1]; mc.f[cnt].ref = 0; cnt++; }
    if (cnt == maxc) return;
    if (mi_cand(c, 1, ltx, lty, mv) && cnt < 4) { mc.dir[cnt] = 1; mc.f[cnt].mv[0] = mv[0]; mc.f[cnt].mv[1] = mv[1]; mc.f[cnt].ref = 0; cnt++; }
    if (cnt == maxc) return;
    if (mi_cand(c, 2, ltx, lty, mv) && cnt < 4) { mc.dir[cnt] = 1; mc.f[cnt].mv[0] = mv[0]; mc.f[cnt].mv[1] = mv[1]; mc.f[cnt].ref = 0; cnt++; }
    if (cnt == maxc) return;
  }
  while (cnt < maxc) { mc.dir[cnt] = 1; mc.f[cnt].mv[0] = mc.f[cnt].mv[1] = 0; mc.f[cnt].ref = 0; cnt++; }   // zero candidates (:3177-3201)
  mc.n = cnt;
}

// TComPrediction::motionCompensation for one PU (TComPrediction.cpp:419-470 -> xPredInterUni :528-552): the vector clipped, GT iff the PU is not merged and has its flag
void CtuWorker::motion_comp_pu(CuData& c, int pu) {
  hop_pred_job j; pu_pred_job(c, pu, j);
  if (cfg.fuse_pred) { if (pu == 0) n_held_ = 0; if (n_held_ < 4) held_[n_held_++] = j; else throw 1; return; }   // handed over with the candidate's evaluation (eval_inter)
  be->pred_inter(lane_, 1, &j);
}
void CtuWorker::pu_pred_job(const CuData& c, int pu, hop_pred_job& j) {
  int ox, oy, w, h; pu_rect(c.p[0].part_size, c.size, pu, ox, oy, w, h);
  const Part& p = *part_at(c, c.x + ox, c.y + oy);
  memset(&j, 0, sizeof(j));
  j.pu_x = c.x + ox; j.pu_y = c.y + oy + cfg.y_origin; j.dst_row_off = row_off(); j.w = w; j.h = h;
  int mh = p.mv[0], mvv = p.mv[1]; clip_mv(c, mh, mvv);
  j.mv_x = mh; j.mv_y = mvv; j.use_gt = (!p.merge_flag && p.gt_flag) ? 1 : 0;
  for (int k = 0; k < 8; k++) j.gt[k] = p.gt[k];
}
uint32_t CtuWorker::inter_pred_error(CuData& c, int pu) {                  // TEncSearch::xGetInterPredictionError (:2951-2977)
  hop_pred_job j; pu_pred_job(c, pu, j);
  uint32_t e = 0; be->pred_cost(lane_, 1, &j, cfg.hadme ? HOP_DIST_HADS : HOP_DIST_SAD, &e);
  return e;
}

struct PuFields { int16_t mv[2], mvd[2], gt[8]; int8_t ref, mvp_idx, mvp_num; uint8_t gt_flag, merge_flag, merge_idx, inter_dir; };
static void apply_pu_fields(Part& p, const void* a) {
  const PuFields& f = *(const PuFields*)a;
  p.mv[0] = f.mv[0]; p.mv[1] = f.mv[1]; p.mvd[0] = f.mvd[0]; p.mvd[1] = f.mvd[1]; memcpy(p.gt, f.gt, sizeof(p.gt));
  p.ref_idx = f.ref; p.mvp_idx = f.mvp_idx; p.mvp_num = f.mvp_num; p.gt_flag = f.gt_flag; p.merge_flag = f.merge_flag; p.merge_idx = f.merge_idx; p.inter_dir = f.inter_dir;
}

// TEncSearch::predInterSearch (:3141-4169) for an ISS slice: list 0 with the SS picture only
bool CtuWorker::pred_inter_search(CuData& c, int ps, bool use_mrg) {
  const int npu = num_pus(ps);
  const uint32_t lam = cfg.lambda_sad;
  for (int pu = 0; pu < npu; pu++) {
    int ox, oy, w, h; pu_rect(ps, c.size, pu, ox, oy, w, h);
    const int px = c.x + ox, py = c.y + oy;
    const bool test_normal = !(use_mrg && c.size > 8 && npu == 2);
    PuFields me; memset(&me, 0, sizeof(me)); me.ref = -1; me.mvp_idx = -1; me.mvp_num = -1;
    uint32_t me_bits = 0;
    bool not_valid = false;
    if (test_normal) {
      uint32_t bits = (ps == SIZE_2Nx2N || ps == SIZE_NxN) ? 1 : 3;      // xGetBlkBits with bPSlice (:4294-4353)
      // xEstimateMvPredAMVP (:4173-4262): the candidate with the smallest template cost
      AmvpInfo info; fill_mvp_cand(c, pu, info);
      int best_idx = 0; uint32_t best_cost = 0x7FFFFFFFu;
      {
        // xGetTemplateCost (:4411-4477) of every candidate: not valid -> MAX_INT; else luma prediction with the clipped vector (no GT), SAD, + the index bit
        hop_pred_job tj[3]; int ti[3], nt = 0; uint32_t sad[3];
        for (int i = 0; i < info.n; i++) {
          if (!valid_pattern(px, py, w, h, info.cand[i][0], info.cand[i][1])) continue;
          hop_pred_job& j = tj[nt]; memset(&j, 0, sizeof(j));
          j.pu_x = px; j.pu_y = py + cfg.y_origin; j.dst_row_off = row_off(); j.w = w; j.h = h;
          int mh = info.cand[i][0], mvv = info.cand[i][1]; clip_mv(c, mh, mvv);
          j.mv_x = mh; j.mv_y = mvv; j.use_gt = 0;
          ti[nt++] = i;
        }
        tag_step(pu * 8 + 0);
        if (nt) be->pred_cost(lane_, nt, tj, HOP_DIST_SAD, sad);
        uint32_t cost[3] = { 0x7FFFFFFFu, 0x7FFFFFFFu, 0x7FFFFFFFu };
        for (int k = 0; k < nt; k++) {
          const double rd = (double)sad[k] + (double)((int)(1 * (double)lam + .5) >> 16);   // calcRdCost(m_auiMVPIdxCost = 1 bit, SAD, false, DF_SAD)
          cost[ti[k]] = (uint32_t)(double)(uint32_t)floor(rd);
        }
        for (int i = 0; i < info.n; i++) if (best_cost > cost[i]) { best_cost = cost[i]; best_idx = i; }
      }
      int pred[2] = { info.cand[best_idx][0], info.cand[best_idx][1] };
      int mvp_idx = best_idx; const int mvp_num = info.n;
      bits += 1;                                                          // m_auiMVPIdxCost[idx][AMVP_MAX_NUM_CANDS]
      // xMotionEstimation (:4479-4683)
      hop_pu_job j; memset(&j, 0, sizeof(j));
      j.pu_x = px; j.pu_y = py + cfg.y_origin; j.w = w; j.h = h;
      int offx, offy; part_offset(ps, c.size, pu, offx, offy);
      int r6[6];
      hop_set_search_range(cfg.pic_w, cfg.pic_h, c.x, c.y, c.size, c.ctu_addr, E.wctu_, pred[0], pred[1], cfg.search_range, offx, offy, c.y == 0, c.x == 0, r6);
      j.rng_left = r6[0]; j.rng_right = r6[1]; j.rng_top = r6[2]; j.rng_bottom = r6[3]; j.off_x = r6[4]; j.off_y = r6[5];
      reach_check(px, py, w, h, r6);
      j.pred_x = pred[0]; j.pred_y = pred[1]; j.lambda_cost = lam; j.n_amvp = info.n;
      for (int i = 0; i < info.n; i++) { j.amvp[2 * i] = info.cand[i][0]; j.amvp[2 * i + 1] = info.cand[i][1]; }
      j.flags = (cfg.fen ? HOP_FLAG_FEN : 0) | (cfg.hadme ? HOP_FLAG_HADME : 0);
      hop_pu_result r; memset(&r, 0, sizeof(r));
      tag_step(pu * 8 + 1);
      be->me_search(lane_, 1, &j, &r);
      not_valid = r.not_valid != 0;
      if (!not_valid) {
        int mvq[2]; uint32_t cost = 0;
        hop_me_finish(&j, &r, HOP_STAGE_GT, bits, mvq, &bits, &cost);
        // xCheckBestMVP (:4364-4409): the predictor that makes the vector cheapest
        {
          int best = mvp_idx;
          const int org_bits = (int)(component_bits(mvq[0] - pred[0]) + component_bits(mvq[1] - pred[1])) + 1;
          int best_bits = org_bits;
          for (int i = 0; i < info.n; i++) {
            if (i == mvp_idx) continue;
            const int b = (int)(component_bits(mvq[0] - info.cand[i][0]) + component_bits(mvq[1] - info.cand[i][1])) + 1;
            if (b < best_bits) { best_bits = b; best = i; }
          }
          if (best != mvp_idx) {
            pred[0] = info.cand[best][0]; pred[1] = info.cand[best][1]; mvp_idx = best;
            const uint32_t ob = bits;
            bits = ob - org_bits + best_bits;
            cost = (cost - ((lam * ob) >> 16)) + ((lam * bits) >> 16);
          }
        }
        me.mv[0] = (int16_t)mvq[0]; me.mv[1] = (int16_t)mvq[1]; me.ref = 0;
        me.mvd[0] = (int16_t)(mvq[0] - pred[0]); me.mvd[1] = (int16_t)(mvq[1] - pred[1]);
        for (int k = 0; k < 8; k++) me.gt[k] = (int16_t)r.gt[k];
        me.gt_flag = r.gt_flag ? 1 : 0; me.inter_dir = 1; me.mvp_idx = (int8_t)mvp_idx; me.mvp_num = (int8_t)mvp_num;
        me_bits = bits;
      }
    }
    // the PU's fields cleared (:3699-3727), then the search result (:3885-3922)
    { PuFields z; memset(&z, 0, sizeof(z)); z.ref = -1; z.mvp_idx = -1; z.mvp_num = -1; z.inter_dir = part_at(c, px, py)->inter_dir; z.merge_flag = part_at(c, px, py)->merge_flag; z.merge_idx = part_at(c, px, py)->merge_idx;
      set_parts(c, ox, oy, w, h, apply_pu_fields, &z);
      c.p[pu].gt_flag = 0; }                                                 // setGTFlag(iPartIdx, false) (:3721): the partition with the PU's NUMBER, not its address
    if (test_normal) {
      if (not_valid) return false;                                         // :3964-3967
      set_parts(c, ox, oy, w, h, apply_pu_fields, &me);
    }
    if (ps != SIZE_2Nx2N) {
      // ME against merge (:3977-4146)
      uint32_t me_cost = MAX_UINT;
      tag_step(pu * 8 + 2);
      // (the searched vector's prediction error, xGetInterPredictionError :3980, is asked for together with the merge candidates' below: neither needs the other's answer)
      PuFields saved = me;
      if (!test_normal) { memset(&saved, 0, sizeof(saved)); saved.ref = -1; saved.mvp_idx = -1; saved.mvp_num = -1; }
      // xMergeEstimation (:2992-3106)
      MergeCands mc; merge_candidates(c, pu, mc);
      uint32_t mrg_cost = MAX_UINT; int mrg_idx = 0; bool val_merge = false;
      {
        // every valid candidate's prediction error: the PU keeps the GT fields and flag the motion search left in the CU while its vector is replaced (:3057-3058)
        const PuFields base = test_normal ? me : saved;
        hop_pred_job mj[6]; int mi[6], nm = 0; uint32_t err[6];
        if (test_normal) { pu_pred_job(c, pu, mj[0]); mi[0] = -1; nm = 1; }     // the searched vector first (as inter_pred_error would have asked)
        for (int k = 0; k < mc.n; k++) {
          int mh = mc.f[k].mv[0], mvv = mc.f[k].mv[1]; clip_mv(c, mh, mvv);
          if (mc.f[k].ref == 0 && !valid_pattern(px, py, w, h, mh, mvv)) continue;
          hop_pred_job& j = mj[nm]; memset(&j, 0, sizeof(j));
          j.pu_x = px; j.pu_y = py + cfg.y_origin; j.dst_row_off = row_off(); j.w = w; j.h = h; j.mv_x = mh; j.mv_y = mvv;
          j.use_gt = (!base.merge_flag && base.gt_flag) ? 1 : 0; for (int q = 0; q < 8; q++) j.gt[q] = base.gt[q];
          mi[nm++] = k;
        }
        tag_step(pu * 8 + 3);
        if (nm) be->pred_cost(lane_, nm, mj, cfg.hadme ? HOP_DIST_HADS : HOP_DIST_SAD, err);
        for (int t = 0; t < nm; t++) {
          const int k = mi[t];
          if (k < 0) { me_cost = err[t] + ((lam * me_bits) >> 16); continue; }
          val_merge = true;
          uint32_t cb = (uint32_t)k + 1; if (k == cfg.max_merge_cand - 1) cb--;
          const uint32_t cand = err[t] + ((lam * cb) >> 16);
          if (cand < mrg_cost) { mrg_cost = cand; mrg_idx = k; }
        }
      }
      if (!val_merge) { mrg_cost = MAX_UINT; if (!test_normal) return false; }
      if (mrg_cost < me_cost) {
        PuFields f; memset(&f, 0, sizeof(f));
        f.merge_flag = 1; f.merge_idx = (uint8_t)mrg_idx; f.inter_dir = mc.dir[mrg_idx]; f.mv[0] = mc.f[mrg_idx].mv[0]; f.mv[1] = mc.f[mrg_idx].mv[1]; f.ref = mc.f[mrg_idx].ref;
        f.mvp_idx = -1; f.mvp_num = -1;
        set_parts(c, ox, oy, w, h, apply_pu_fields, &f);
      } else {
        PuFields f = saved; f.merge_flag = 0;
        bool any = false; for (int k = 0; k < 8; k++) any |= f.gt[k] != 0;
        f.gt_flag = any ? 1 : 0;
        set_parts(c, ox, oy, w, h, apply_pu_fields, &f);
      }
    }
    tag_step(pu * 8 + 4);
    motion_comp_pu(c, pu);                                                 // :4158
  }
  return true;
}

// ---- candidate evaluation through the backend ----
static void fill_rqt_job(const EncConfig& cfg, const CuData& c, bool intra, int part_size, hop_rqt_job& j, int row_off = 0) {
  memset(&j, 0, sizeof(j));
  j.x = c.x; j.y = c.y + cfg.y_origin + row_off; j.log2_cu = 6 - c.depth;
  for (int k = 0; k < 3; k++) { j.qp_scaled[k] = cfg.qp_scaled[k]; j.lambda_rdoq[k] = cfg.lambda_rdoq[k]; }
  j.ctx_index = 0; j.sign_hide = cfg.sign_hide; j.use_ts = cfg.use_ts; j.log2_max_tu = cfg.log2_max_tu;
  // TComDataCU::getQuadtreeTULog2MinSizeInCU (TComDataCU.cpp:1860-1886)
  const int max_depth = intra ? cfg.tu_max_depth_intra : cfg.tu_max_depth_inter;
  const int isf = (intra && part_size == SIZE_NxN) ? 1 : 0, esf = (!intra && max_depth == 1 && part_size != SIZE_2Nx2N) ? 1 : 0;
  int m;
  if (j.log2_cu < cfg.log2_min_tu + max_depth - 1 + esf + isf) m = cfg.log2_min_tu;
  else { m = j.log2_cu - (max_depth - 1 + esf + isf); if (m > cfg.log2_max_tu) m = cfg.log2_max_tu; }
  j.log2_min_tu_in_cu = m; j.inter_split_flag = esf;
  j.lambda_rd = cfg.lambda; j.dist_weight[0] = cfg.dist_weight[0]; j.dist_weight[1] = cfg.dist_weight[1];
}

void CtuWorker::eval_inter(int d, bool skip_res) {
  CuData& c = *temp_[d];
  InterEval e; memset(&e, 0, sizeof(e));
  const int ps = c.p[0].part_size;
  fill_rqt_job(cfg, c, false, ps, e.job, row_off());
  c.slot = slot_;
  e.skip_res = skip_res ? 1 : 0;
  e.n_pred = n_held_; for (int k = 0; k < n_held_; k++) e.pred[k] = held_[k];
  n_held_ = 0;
  hop_cu_syntax& y = e.syn;
  y.part_size = ps; y.n_pu = num_pus(ps); y.skip_flag = c.p[0].skip;
  { const Part* l = nb_left(c, c.x, c.y); const Part* a = nb_above(c, c.x, c.y); y.skip_ctx = (l ? l->skip : 0) + (a ? a->skip : 0); }   // getCtxSkipFlag (:1888)
  y.amp_acc = (cfg.amp && d < 3) ? 1 : 0; y.is_min_cu = d == 3; y.max_merge_cand = cfg.max_merge_cand;
  for (int pu = 0; pu < y.n_pu; pu++) {
    int ox, oy, w, h; pu_rect(ps, c.size, pu, ox, oy, w, h);
    const Part& p = *part_at(c, c.x + ox, c.y + oy);
    y.pu[pu].merge_flag = p.merge_flag; y.pu[pu].merge_idx = p.merge_idx; y.pu[pu].mvd[0] = p.mvd[0]; y.pu[pu].mvd[1] = p.mvd[1]; y.pu[pu].mvp_idx = p.mvp_idx;
    y.pu[pu].gt_flag = p.gt_flag; for (int k = 0; k < 8; k++) y.pu[pu].gt[k] = p.gt[k];
  }
  EvalResult r; memset(&r, 0, sizeof(r));
  const Coder& in = sb_[d][CI_CURR];
  be->inter_cu(lane_, e, in, r);
  for (int i = 0; i < c.num_part; i++) {
    Part& p = c.p[i];
    p.tr_idx = r.tr_idx[i]; for (int k = 0; k < 3; k++) { p.cbf[k] = r.cbf[k][i]; p.tskip[k] = r.tskip[k][i]; }
    if (skip_res || r.skipped) p.skip = 1;
  }
  c.bits = r.bits; c.dist = r.dist; c.cost = calc_rd_cost(r.bits, r.dist, cfg.lambda);
  r.after.split[0] = in.split[0]; r.after.split[1] = in.split[1]; r.after.split[2] = in.split[2];
  goon_ = r.after; sb_[d][CI_TEMP] = r.after;
  c.fbits[0] = ((uint64_t)r.bits << 15) + coder_frac(r.after) - coder_frac(in);
}

void CtuWorker::check_inter(int d, int ps, bool use_mrg) {                 // TEncCu::xCheckRDCostInter (:1399-1453)
  CuData& c = *temp_[d];
  for (int i = 0; i < c.num_part; i++) { c.p[i].depth = (uint8_t)d; c.p[i].skip = 0; c.p[i].part_size = (uint8_t)ps; c.p[i].pred_mode = MODE_INTER; }
  tag_cand(10 + ps);
  if (!pred_inter_search(c, ps, use_mrg)) { c.cost = MAX_DOUBLE; return; }
  tag_step(60);
  eval_inter(d, false);
  check_best_mode(d, true);
}

void CtuWorker::check_merge_2Nx2N(int d, bool* early_skip) {               // TEncCu::xCheckRDCostMerge2Nx2N (:1243-1395)
  CuData* c = temp_[d];
  for (int i = 0; i < c->num_part; i++) c->p[i].part_size = SIZE_2Nx2N;
  MergeCands mc; merge_candidates(*c, 0, mc);
  int buf[5] = { 0, 0, 0, 0, 0 };
  bool best_is_skip = false;
  for (int nores = 0; nores < 2; nores++) {
    for (int k = 0; k < mc.n; k++) {
      if (nores == 1 && buf[k] == 1) continue;
      if (best_is_skip && nores == 0) continue;
      c = temp_[d];
      for (int i = 0; i < c->num_part; i++) {
        Part& p = c->p[i];
        p.pred_mode = MODE_INTER; p.part_size = SIZE_2Nx2N; p.merge_flag = 1; p.merge_idx = (uint8_t)k; p.inter_dir = mc.dir[k];
        p.mv[0] = mc.f[k].mv[0]; p.mv[1] = mc.f[k].mv[1]; p.ref_idx = mc.f[k].ref;
      }
      if (mc.f[k].ref == 0) {
        int mh = mc.f[k].mv[0], mvv = mc.f[k].mv[1]; clip_mv(*c, mh, mvv);
        if (!valid_pattern(c->x, c->y, c->size, c->size, mh, mvv)) { init_est(*c); continue; }
      }
      tag_cand(nores * 5 + k);
      motion_comp_pu(*c, 0);
      tag_step(60);
      eval_inter(d, nores != 0);
      const int root = (c->p[0].cbf[0] & 1) | (c->p[0].cbf[1] & 1) | (c->p[0].cbf[2] & 1);
      if (nores == 0 && root == 0) buf[k] = 1;
      for (int i = 0; i < c->num_part; i++) c->p[i].skip = root == 0;
      check_best_mode(d, true);
      init_est(*temp_[d]);
      if (cfg.fdm && !best_is_skip) {
        const Part& b = best_[d]->p[0];
        best_is_skip = ((b.cbf[0] & 1) | (b.cbf[1] & 1) | (b.cbf[2] & 1)) == 0;
      }
    }
    if (nores == 0 && cfg.esd) {
      const Part& b = best_[d]->p[0];
      if (((b.cbf[0] & 1) | (b.cbf[1] & 1) | (b.cbf[2] & 1)) == 0) {
        if (b.merge_flag) *early_skip = true;
        else if (abs(b.mvd[0]) + abs(b.mvd[1]) == 0) *early_skip = true;
      }
    }
  }
}

void CtuWorker::fill_intra_eval(const CuData& c, int ps, IntraEval& e) {
  memset(&e, 0, sizeof(e));
  fill_rqt_job(cfg, c, true, ps, e.job, row_off());                      // (in a candidate slot the backend first copies the CU's neighbouring row and column there)
  e.part_nxn = ps == SIZE_NxN ? 1 : 0;
  hop_intra_cu_syntax& y = e.syn;
  y.part_nxn = e.part_nxn; y.skip_flag = 0; y.is_min_cu = c.depth == 3;
  { const Part* l = nb_left(c, c.x, c.y); const Part* a = nb_above(c, c.x, c.y); y.skip_ctx = (l ? l->skip : 0) + (a ? a->skip : 0); }
  if (cfg.slice_type == 2) y.skip_ctx = -1;                               // an I slice codes neither cu_skip_flag nor pred_mode_flag
  e.opt.ts_fast = cfg.ts_fast; e.opt.strong = cfg.strong_intra;
  auto flags_of = [&](int x, int yy, int size, uint8_t* fl) {
    const int u = size / 4;
    memset(fl, 0, 4 * u + 1);
    fl[2 * u] = nb_above_left(c, x, yy) ? 1 : 0;                           // position only: part_at is never dereferenced here
    for (int i = 0; i < u; i++) {
      fl[2 * u + 1 + i] = ((yy & 63) || yy > 0) ? 1 : 0;                   // above (isAboveAvailable)
      fl[2 * u - 1 - i] = ((x & 63) || x > 0) ? 1 : 0;                     // left, upwards from the corner
      fl[3 * u + 1 + i] = ar_avail(x + size - 4, yy, i + 1) ? 1 : 0;       // above right, from the block's top-right unit
      fl[u - 1 - i] = bl_avail(x, yy + size - 4, i + 1) ? 1 : 0;           // below left, from the bottom-left unit
    }
  };
  // every node of the CU's transform tree that can be predicted (size 4..32)
  static const int base[5] = { 0, 1, 5, 21, 85 };
  const int log2_cu = 6 - c.depth;
  for (int dd = 0; dd <= 4 && log2_cu - dd >= 2; dd++) {
    const int l2 = log2_cu - dd; if (l2 > 5) continue;
    const int size = 1 << l2, np = 1 << (2 * (l2 - 2));
    for (int p = 0; p < c.num_part; p += np) {
      int x4, y4; zpos(p, x4, y4);
      uint8_t fl[68];
      flags_of(c.x + 4 * x4, c.y + 4 * y4, size, fl);
      uint64_t m = 0; for (int i = 0; i < 4 * (size / 4) + 1; i++) if (fl[i]) m |= 1ull << i;
      e.opt.avail[base[dd] + (p >> (2 * (l2 - 2)))] = m;
    }
  }
  const int npu = e.part_nxn ? 4 : 1, N = c.size >> e.part_nxn;
  for (int pu = 0; pu < npu; pu++) {
    const int x = c.x + (pu & 1) * N, yy = c.y + (pu >> 1) * N;
    const Part* l = nb_left(c, x, yy); const Part* a = nb_above(c, x, yy, true);
    e.sjob.left_dir[pu] = l ? (l->pred_mode == MODE_INTRA ? l->luma_dir : DC_IDX) : DC_IDX;
    e.sjob.above_dir[pu] = a ? (a->pred_mode == MODE_INTRA ? a->luma_dir : DC_IDX) : DC_IDX;
    flags_of(x, yy, N, e.sjob.rough_flags[pu]);
  }
  e.sjob.sqrt_lambda = cfg.sqrt_lambda;
  e.sjob.num_full_rd = N <= 8 ? 8 : 3;                                     // g_aucIntraModeNumFast (TComRom.cpp:274-282)
}

void CtuWorker::check_intra(int d, int ps) {                               // TEncCu::xCheckRDCostIntra (:1455-1507)
  eval_intra(d, ps);
  check_best_mode(d, true);
}
void CtuWorker::eval_intra(int d, int ps) {                                // ... up to the comparison
  CuData& c = *temp_[d];
  for (int i = 0; i < c.num_part; i++) { c.p[i].skip = 0; c.p[i].part_size = (uint8_t)ps; c.p[i].pred_mode = MODE_INTRA; }
  tag_cand(ps == SIZE_2Nx2N ? 20 : 21);
  IntraEval e; fill_intra_eval(c, ps, e);
  EvalResult r; memset(&r, 0, sizeof(r));
  const Coder& in = sb_[d][CI_CURR];
  be->intra_cu(lane_, e, in, r);
  const int npu = e.part_nxn ? 4 : 1, q = c.num_part >> 2;
  for (int i = 0; i < c.num_part; i++) {
    Part& p = c.p[i];
    p.tr_idx = r.tr_idx[i]; for (int k = 0; k < 3; k++) { p.cbf[k] = r.cbf[k][i]; p.tskip[k] = r.tskip[k][i]; }
    p.luma_dir = (uint8_t)r.luma_dir[npu == 4 ? i / q : 0]; p.chroma_dir = (uint8_t)r.chroma_dir;
  }
  c.bits = r.bits; c.dist = r.dist; c.cost = calc_rd_cost(r.bits, r.dist, cfg.lambda);
  r.after.split[0] = in.split[0]; r.after.split[1] = in.split[1]; r.after.split[2] = in.split[2];
  goon_ = r.after; sb_[d][CI_TEMP] = r.after;
  c.fbits[0] = ((uint64_t)r.bits << 15) + coder_frac(r.after) - coder_frac(in);
  c.slot = slot_;
}

// ---- SS/GT candidates of one CU side by side (cfg.spec_slots > 0) ----
// Every candidate of xCompressCU starts from the same coder state (CI_CURR_BEST of the depth) and from a freshly initialised temporary CU, and reads nothing another
// candidate of the same CU has written except through the prediction / reconstruction pictures: with a copy of those per candidate (a slot) the candidates are independent
// evaluations.  Which of them the reference would test, and in which order their results meet xCheckBestMode, is decided afterwards exactly as in the serial path; a
// candidate the reference would have skipped was evaluated for nothing and is dropped.  Each candidate runs as a worker of its own through Backend::fork_join.
struct CtuWorker::SpecCand {
  int merge_k, nores;                // merge candidate k (>= 0) with / without residual, or
  int ps; bool use_mrg;              // an SS/GT search of this partition size, or
  int intra_ps;                      // (>= 0) the intra candidate of this partition size
  bool ok;                           // false: predInterSearch found no valid candidate (the reference does not rate the mode then)
  MvField mf; uint8_t mdir;
  CtuWorker* w;                      // the kid that evaluated it: the result is its temporary CU of the depth, its CI_TEMP coder and its go-on coder
};
struct CtuWorker::SpecSet { MergeCands mc; bool valid[5]; int idx[5][2]; int first_inter, first_amp, first_intra; std::vector<SpecCand> cands; bool ready; SpecSet() : first_inter(0), first_amp(0), first_intra(0), ready(false) { mc.n = 0; } };

void CtuWorker::spec_run(int d, SpecCand& sc, const CuData& tmpl, int slot) {
  CtuWorker* w = kids_[slot];
  w->ensure(d);
  w->temp_[d] = w->store_[d][1]; w->best_[d] = w->store_[d][0];
  w->ctu_addr_ = ctu_addr_; w->ctu_x_ = ctu_x_; w->ctu_y_ = ctu_y_; w->slot_ = slot; w->tag_seq_ = tag_seq_; w->lane_ = lane_;
  w->sb_[d][CI_CURR] = sb_[d][CI_CURR]; w->goon_ = goon_;
  CuData* c = w->temp_[d];
  cu_copy(*c, tmpl);
  sc.w = w;
  {
    if (sc.intra_ps >= 0) {                                               // xCheckRDCostIntra up to the comparison
      w->eval_intra(d, sc.intra_ps);
      sc.ok = true;
    } else if (sc.merge_k >= 0) {                                         // one pass of the loop of xCheckRDCostMerge2Nx2N
      for (int i = 0; i < c->num_part; i++) {
        Part& p = c->p[i];
        p.pred_mode = MODE_INTER; p.part_size = SIZE_2Nx2N; p.merge_flag = 1; p.merge_idx = (uint8_t)sc.merge_k; p.inter_dir = sc.mdir;
        p.mv[0] = sc.mf.mv[0]; p.mv[1] = sc.mf.mv[1]; p.ref_idx = sc.mf.ref;
      }
      w->tag_cand(sc.nores * 5 + sc.merge_k);
      w->motion_comp_pu(*c, 0);
      w->tag_step(60);
      w->eval_inter(d, sc.nores != 0);
      const int root = (c->p[0].cbf[0] & 1) | (c->p[0].cbf[1] & 1) | (c->p[0].cbf[2] & 1);
      for (int i = 0; i < c->num_part; i++) c->p[i].skip = root == 0;
      sc.ok = true;
    } else {                                                              // xCheckRDCostInter up to the comparison
      for (int i = 0; i < c->num_part; i++) { c->p[i].depth = (uint8_t)d; c->p[i].skip = 0; c->p[i].part_size = (uint8_t)sc.ps; c->p[i].pred_mode = MODE_INTER; }
      w->tag_cand(10 + sc.ps);
      sc.ok = w->pred_inter_search(*c, sc.ps, sc.use_mrg);
      if (sc.ok) { w->tag_step(60); w->eval_inter(d, false); }
    }
  }
}

// the candidate's result as the serial path would have left it in the temporary CU, then xCheckBestMode
void CtuWorker::spec_adopt(int d, SpecCand& sc) {
  cu_copy(*temp_[d], *sc.w->temp_[d]); sb_[d][CI_TEMP] = sc.w->sb_[d][CI_TEMP]; goon_ = sc.w->goon_;
  check_best_mode(d, true);
}

// the slots of a set of candidates: base + 1, base + 2, ... for the SS/GT candidates; the two highest of the level (base + L - 1, base + L) for the intra candidates, whose
// reconstructions wait there until the decisions reach them (a second batch of AMP candidates reuses the low slots)
void CtuWorker::spec_slots_of(const std::vector<SpecCand>& cands, int base, int L, std::vector<int>& slot) {
  const int n = (int)cands.size();
  int n_inter = 0; for (int i = 0; i < n; i++) n_inter += cands[i].intra_ps < 0;
  if (n_inter > L - 2) throw 1;
  slot.resize(n);
  for (int i = 0, k = 0; i < n; i++) slot[i] = base + (cands[i].intra_ps < 0 ? ++k : (cands[i].intra_ps == SIZE_2Nx2N ? L - 1 : L));
}
void CtuWorker::spec_inter_phase(int d, std::vector<SpecCand>& cands) {
  const CuData& tmpl = *temp_[d];                                         // after init_est (this worker waits in fork_join while its kids read it)
  if ((int)kids_.size() <= cfg.spec_slots) { const size_t k0 = kids_.size(); kids_.resize(cfg.spec_slots + 1, NULL); for (size_t k = k0; k < kids_.size(); k++) kids_[k] = new CtuWorker(E, lane_, be, true); }
  std::vector<int> slot; spec_slots_of(cands, 0, spec_level_slots(), slot);
  flush_save();                                                          // (the candidates about to run reuse the slots)
  be->fork_join((int)cands.size(), [&](int i) { spec_run(d, cands[i], tmpl, slot[i]); });
}

// The candidates of the CU in temp_[d] (initialised, init_est): xCheckRDCostMerge2Nx2N's passes, then xCheckRDCostInter for 2Nx2N, Nx2N and 2NxN (the order of xCompressCU
// without early skip detection and CBF fast mode), the AMP shapes, the intra candidates -- as a list; nothing is evaluated here
void CtuWorker::spec_build(int d, SpecSet& S) {
  CuData* c = temp_[d];
  for (int i = 0; i < c->num_part; i++) c->p[i].part_size = SIZE_2Nx2N;
  merge_candidates(*c, 0, S.mc);
  init_est(*temp_[d]);
  std::vector<SpecCand>& cands = S.cands; cands.clear(); cands.reserve(24);
  for (int k = 0; k < 5; k++) { S.valid[k] = false; S.idx[k][0] = S.idx[k][1] = -1; }
  for (int k = 0; k < S.mc.n; k++) {
    S.valid[k] = true;
    if (S.mc.f[k].ref == 0) { int mh = S.mc.f[k].mv[0], mvv = S.mc.f[k].mv[1]; clip_mv(*c, mh, mvv); S.valid[k] = valid_pattern(c->x, c->y, c->size, c->size, mh, mvv); }
    if (!S.valid[k]) continue;
    for (int nores = 0; nores < 2; nores++) {
      SpecCand sc; sc.intra_ps = -1; sc.merge_k = k; sc.nores = nores; sc.ps = SIZE_2Nx2N; sc.use_mrg = false; sc.ok = false; sc.mf = S.mc.f[k]; sc.mdir = S.mc.dir[k];
      S.idx[k][nores] = (int)cands.size(); cands.push_back(sc);
    }
  }
  S.first_inter = (int)cands.size();
  static const int inter_ps[3] = { SIZE_2Nx2N, SIZE_Nx2N, SIZE_2NxN };
  for (int q = 0; q < 3; q++) { SpecCand sc; sc.intra_ps = -1; sc.merge_k = -1; sc.nores = 0; sc.ps = inter_ps[q]; sc.use_mrg = false; sc.ok = false; memset(&sc.mf, 0, sizeof(sc.mf)); sc.mdir = 0; cands.push_back(sc); }
  // The AMP shapes too, when there are slots for them: which of them xCompressCU tests, and whether with all vectors or merge only, follows from the best mode of the
  // candidates above (deriveTestModeAMP), but what each test yields does not -- every candidate starts from the same coder and an empty CU.  Both forms of all four shapes
  // run with the first batch; the derivation (compress_cu) then adopts the ones the reference would have tested, in its order, and the rest is dropped: a node's second
  // round of searches and evaluations is gone from the CTU's chain.  (64x64: deriveTestModeAMP never asks for the all-vectors form.)
  S.first_amp = (int)cands.size();
  if (cfg.amp && d < 3 && spec_level_slots() >= 24) {
    static const int amp_ps[4] = { SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N };
    for (int q = 0; q < 4; q++) for (int mrg = (c->size == 64 ? 1 : 0); mrg < 2; mrg++) {
      SpecCand sc; sc.intra_ps = -1; sc.merge_k = -1; sc.nores = 0; sc.ps = amp_ps[q]; sc.use_mrg = mrg != 0; sc.ok = false; memset(&sc.mf, 0, sizeof(sc.mf)); sc.mdir = 0; cands.push_back(sc);
    }
  }
  // the intra candidates join the same batch: in an ISS slice xCompressCU always tests them, after every SS/GT candidate (their results wait in spec_intra_)
  S.first_intra = (int)cands.size();
  const int n_intra = (d == 3 && c->size > (1 << cfg.log2_min_tu)) ? 2 : 1;
  for (int q = 0; q < n_intra; q++) { SpecCand sc; sc.intra_ps = q ? SIZE_NxN : SIZE_2Nx2N; sc.merge_k = -1; sc.nores = 0; sc.ps = 0; sc.use_mrg = false; sc.ok = false; memset(&sc.mf, 0, sizeof(sc.mf)); sc.mdir = 0; cands.push_back(sc); }
}

// The candidates of this CU evaluated side by side -- and, with two levels of slots (48), those of its FIRST sub-CU with them: that CU sits at the same corner, starts from
// the same coder state (CI_CURR_BEST of the parent, TEncCu.cpp:731-733), sees the same SS reference (nothing is committed before a decision) and the same neighbours, and is
// tested whatever the parent's candidates yield (no early CU termination in these configurations) -- so its searches and evaluations ride along in the parent's rounds
// and its node costs the CTU's chain nothing but its decisions.  Then this CU's decisions, in the serial order.
void CtuWorker::check_merge_and_inter_spec(int d) {
  if (spec_set_.empty()) spec_set_.resize(4);
  SpecSet& S = spec_set_[d];
  if (!S.ready) {
    spec_build(d, S);
    // the chain of first sub-CUs below this one, as far as there are levels of slots: each at the same corner, from the same coder state
    const int L = spec_level_slots(), levels = L >= 24 ? cfg.spec_slots / L : 1;
    std::vector<SpecSet*> sets(1, &S); std::vector<int> depth(1, d);
    for (int nd = d + 1; nd <= 3 && (int)sets.size() < levels; nd++) {
      const CuData& t = *temp_[d];
      init_cu(*best_[nd], t.abs_idx, nd, t.x, t.y); init_cu(*temp_[nd], t.abs_idx, nd, t.x, t.y);
      sb_[nd][CI_CURR] = sb_[d][CI_CURR];
      init_est(*temp_[nd]);
      spec_build(nd, spec_set_[nd]);
      sets.push_back(&spec_set_[nd]); depth.push_back(nd);
    }
    if ((int)kids_.size() <= cfg.spec_slots) { const size_t k0 = kids_.size(); kids_.resize(cfg.spec_slots + 1, NULL); for (size_t k = k0; k < kids_.size(); k++) kids_[k] = new CtuWorker(E, lane_, be, true); }
    std::vector<std::vector<int> > slot(sets.size()); std::vector<int> first(sets.size() + 1, 0);
    for (size_t q = 0; q < sets.size(); q++) { spec_slots_of(sets[q]->cands, (int)q * L, L, slot[q]); first[q + 1] = first[q] + (int)sets[q]->cands.size(); }
    flush_save();                                                          // (the candidates about to run reuse the slots)
    be->fork_join(first[sets.size()], [&](int i) {                        // (this worker waits in fork_join while its kids read the templates temp_[depth])
      size_t q = 0; while (i >= first[q + 1]) q++;
      spec_run(depth[q], sets[q]->cands[i - first[q]], *temp_[depth[q]], slot[q][i - first[q]]);
    });
    for (size_t q = 1; q < sets.size(); q++) sets[q]->ready = true;
  }
  S.ready = false;
  std::vector<SpecCand>& cands = S.cands;
  spec_intra_[d].assign(cands.begin() + S.first_intra, cands.end());
  spec_amp_[d].assign(cands.begin() + S.first_amp, cands.begin() + S.first_intra);
  // ---- the decisions, in the serial order ----
  int buf[5] = { 0, 0, 0, 0, 0 };
  bool best_is_skip = false;
  for (int nores = 0; nores < 2; nores++) {
    for (int k = 0; k < S.mc.n; k++) {
      if (nores == 1 && buf[k] == 1) continue;
      if (best_is_skip && nores == 0) continue;
      if (!S.valid[k]) { init_est(*temp_[d]); continue; }
      SpecCand& sc = cands[S.idx[k][nores]];
      const Part& q0 = sc.w->temp_[d]->p[0];
      const int root = (q0.cbf[0] & 1) | (q0.cbf[1] & 1) | (q0.cbf[2] & 1);
      if (nores == 0 && root == 0) buf[k] = 1;
      spec_adopt(d, sc);
      init_est(*temp_[d]);
      if (cfg.fdm && !best_is_skip) {
        const Part& b = best_[d]->p[0];
        best_is_skip = ((b.cbf[0] & 1) | (b.cbf[1] & 1) | (b.cbf[2] & 1)) == 0;
      }
    }
  }
  init_est(*temp_[d]);
  for (int q = 0; q < 3; q++) {
    SpecCand& sc = cands[S.first_inter + q];
    if (sc.ok) spec_adopt(d, sc);
    init_est(*temp_[d]);
  }
}

// TEncCu::xCompressCU (:371-892)
void CtuWorker::compress_cu(int d, int parent_ps) {
  const int x = best_[d]->x, y = best_[d]->y, size = best_[d]->size;
  bool sub_branch = true, do_not_block_pu = true, early_skip = false, boundary = false;
  const bool inside = (x + size <= cfg.pic_w) && (y + size <= cfg.pic_h);
  const bool not_i = cfg.slice_type != 2;
  const int node_abs = best_[d]->abs_idx;
  tag_enter(d, node_abs);
  auto root_cbf = [](const CuData* c) { return (c->p[0].cbf[0] & 1) | (c->p[0].cbf[1] & 1) | (c->p[0].cbf[2] & 1); };
  // SS/GT candidates side by side: the configurations without early skip detection / CBF fast mode (the shipped ones), whose candidate list does not depend on results
  const bool spec = cfg.spec_slots >= 15 && cfg.slice_type == 3 && not_i && !cfg.esd && !cfg.cfm && !(size != 8 && d == 3);
  if (inside) {
    init_est(*temp_[d]);
    if (spec) { check_merge_and_inter_spec(d); }
    else if (not_i) {
      if (cfg.esd) { check_inter(d, SIZE_2Nx2N, false); init_est(*temp_[d]); }
      check_merge_2Nx2N(d, &early_skip); init_est(*temp_[d]);
      if (!cfg.esd) {
        check_inter(d, SIZE_2Nx2N, false); init_est(*temp_[d]);
        if (cfg.cfm) do_not_block_pu = root_cbf(best_[d]) != 0;
      }
    }
    if (!early_skip) {
      init_est(*temp_[d]);
      if (not_i) {
        if (!spec) {
        if (size != 8 && d == 3 && do_not_block_pu) { check_inter(d, SIZE_NxN, false); init_est(*temp_[d]); }
        if (do_not_block_pu) {
          check_inter(d, SIZE_Nx2N, false); init_est(*temp_[d]);
          if (cfg.cfm && best_[d]->p[0].part_size == SIZE_Nx2N) do_not_block_pu = root_cbf(best_[d]) != 0;
        }
        if (do_not_block_pu) {
          check_inter(d, SIZE_2NxN, false); init_est(*temp_[d]);
          if (cfg.cfm && best_[d]->p[0].part_size == SIZE_2NxN) do_not_block_pu = root_cbf(best_[d]) != 0;
        }
        }
        if (cfg.amp && d < 3) {                                            // getAMPAcc(depth)
          // deriveTestModeAMP (:292-356)
          bool hor = false, ver = false, mhor = false, mver = false;
          const Part& b = best_[d]->p[0];
          if (b.part_size == SIZE_2NxN) hor = true;
          else if (b.part_size == SIZE_Nx2N) ver = true;
          else if (b.part_size == SIZE_2Nx2N && !b.merge_flag && !b.skip) { hor = true; ver = true; }
          if (parent_ps >= SIZE_2NxnU && parent_ps <= SIZE_nRx2N) { mhor = true; mver = true; }
          if (parent_ps == SIZE_NONE) {
            if (b.part_size == SIZE_2NxN) mhor = true;
            else if (b.part_size == SIZE_Nx2N) mver = true;
          }
          if (b.part_size == SIZE_2Nx2N && !b.skip) { mhor = true; mver = true; }
          if (size == 64) { hor = false; ver = false; }
          std::vector<SpecCand> amps;                                        // spec: the AMP shapes the derivation asks for, side by side
          auto amp = [&](int ps, bool mrg, bool cfm_check) {
            if (!do_not_block_pu) return;
            if (spec) {
              for (size_t q = 0; q < spec_amp_[d].size(); q++) if (spec_amp_[d][q].ps == ps && spec_amp_[d][q].use_mrg == mrg) {   // evaluated with the first batch
                if (spec_amp_[d][q].ok) spec_adopt(d, spec_amp_[d][q]);
                init_est(*temp_[d]);
                return;
              }
              SpecCand sc; sc.intra_ps = -1; sc.merge_k = -1; sc.nores = 0; sc.ps = ps; sc.use_mrg = mrg; sc.ok = false; memset(&sc.mf, 0, sizeof(sc.mf)); sc.mdir = 0; amps.push_back(sc); return;
            }
            check_inter(d, ps, mrg); init_est(*temp_[d]);
            if (cfm_check && cfg.cfm && best_[d]->p[0].part_size == ps) do_not_block_pu = root_cbf(best_[d]) != 0;
          };
          if (hor) { amp(SIZE_2NxnU, false, true); amp(SIZE_2NxnD, false, true); }
          else if (mhor) { amp(SIZE_2NxnU, true, true); amp(SIZE_2NxnD, true, true); }
          if (ver) { amp(SIZE_nLx2N, false, true); amp(SIZE_nRx2N, false, false); }
          else if (mver) { amp(SIZE_nLx2N, true, true); amp(SIZE_nRx2N, true, false); }
          if (spec && !amps.empty()) {
            spec_inter_phase(d, amps);
            for (size_t q = 0; q < amps.size(); q++) { if (amps[q].ok) spec_adopt(d, amps[q]); init_est(*temp_[d]); }
          }
          spec_amp_[d].clear();
        }
      }
      const Part& b = best_[d]->p[0];
      if (spec) {                                                          // (slice_type 3: always tested)
        for (size_t q = 0; q < spec_intra_[d].size(); q++) { spec_adopt(d, spec_intra_[d][q]); init_est(*temp_[d]); }
        spec_intra_[d].clear();
      } else if (!not_i || b.cbf[0] != 0 || b.cbf[1] != 0 || b.cbf[2] != 0 || cfg.slice_type == 3 || b.part_size == SIZE_NONE) {
        check_intra(d, SIZE_2Nx2N); init_est(*temp_[d]);
        if (d == 3 && size > (1 << cfg.log2_min_tu)) { check_intra(d, SIZE_NxN); init_est(*temp_[d]); }
      }
    }
    // the split flag of "not split", counted on the go-on coder as the last candidate left it (:686-690)
    best_[d]->bits += split_flag_bits(*best_[d], d, goon_);
    best_[d]->cost = calc_rd_cost(best_[d]->bits, best_[d]->dist, cfg.lambda);
    sub_branch = !(cfg.ecu && best_[d]->p[0].skip);
  } else boundary = true;

  bool split_is_best = false;
  flush_save();                                                          // (the sub-CUs' candidates reuse the slots)
  if (sub_branch && d < 3) {
    init_est(*temp_[d]);
    CuData* t = temp_[d];
    const int nd = d + 1, q = t->num_part >> 2, h = size >> 1;
    for (int i = 0; i < 4; i++) {
      const int sx = x + (i & 1) * h, sy = y + (i >> 1) * h;
      init_cu(*best_[nd], t->abs_idx + i * q, nd, sx, sy); init_cu(*temp_[nd], t->abs_idx + i * q, nd, sx, sy);
      if (sx < cfg.pic_w && sy < cfg.pic_h) {
        sb_[nd][CI_CURR] = i == 0 ? sb_[d][CI_CURR] : sb_[nd][CI_NEXT];
        compress_cu(nd, best_[d]->p[0].pred_mode == MODE_INTRA ? (int)SIZE_NONE : (int)best_[d]->p[0].part_size);
      } else copy_to_pic(*best_[nd]);
      // copyPartFrom (TComDataCU.cpp:1005-1096)
      t = temp_[d];
      memcpy(&t->p[i * q], best_[nd]->p, sizeof(Part) * q); memcpy(&t->fbits[i * q], best_[nd]->fbits, sizeof(uint64_t) * q);
      if (sx < cfg.pic_w && sy < cfg.pic_h) { t->bits += best_[nd]->bits; t->dist += best_[nd]->dist; }
    }
    if (!boundary) t->bits += split_flag_bits(*t, d, goon_);
    t->cost = calc_rd_cost(t->bits, t->dist, cfg.lambda);
    sb_[d][CI_TEMP] = sb_[nd][CI_NEXT];
    CuData* before = best_[d];
    check_best_mode(d, false);
    split_is_best = best_[d] != before;
  }
  copy_to_pic(*best_[d]);
  tag_exit(d, node_abs);
  if (!boundary) {
    flush_save();
    if (!split_is_best) be->restore_commit(lane_, d, x, y + cfg.y_origin, size);          // xCopyYuv2Pic (:869): the winner's reconstruction back into the picture, then
    else be->commit(lane_, x, y + cfg.y_origin, size);                                    // xCopyYuv2SSRef (:872-880)
    for (int yy = y >> 3; yy < (y + size) >> 3; yy++) memset(&E.committed[(size_t)yy * (cfg.pic_w >> 3) + (x >> 3)], 1, size >> 3);
  }
}

// the pass of TEncCu::encodeCU over the finished CTU (TEncSlice.cpp:1142 on the CTU's entry coder): split flags in coding order on the evolving
// split contexts; the CUs' own syntax was counted when they were chosen (its context updates are in CI_NEXT_BEST, its fractional bits in fbits)
uint64_t CtuWorker::final_walk(int x, int y, int size, int d, Coder& k) {
  const CuData& c = *best_[0];
  const bool inside = (x + size <= cfg.pic_w) && (y + size <= cfg.pic_h);
  const Part& p = c.p[zpix(x, y)];
  uint64_t total = 0;
  if (inside && d < 3) {
    const Part* l = nb_left(c, x, y); const Part* a = nb_above(c, x, y);
    const int ctx = (l ? (l->depth > d) : 0) + (a ? (a->depth > d) : 0);
    total += hop_cabac_bin_bits(&k.split[ctx], p.depth > d ? 1 : 0);
  }
  if ((d < p.depth && d < 3) || !inside) {
    const int h = size >> 1;
    for (int i = 0; i < 4; i++) {
      const int sx = x + (i & 1) * h, sy = y + (i >> 1) * h;
      if (sx < cfg.pic_w && sy < cfg.pic_h) total += final_walk(sx, sy, h, d + 1, k);
    }
    return total;
  }
  return total + c.fbits[zpix(x, y)];
}

void CtuWorker::compress_ctu(int addr, const Coder& entry, Coder& exit) {
  ctu_addr_ = addr; ctu_x_ = (addr % E.wctu_) * CTU; ctu_y_ = (addr / E.wctu_) * CTU;
  init_cu(*best_[0], 0, 0, ctu_x_, ctu_y_); init_cu(*temp_[0], 0, 0, ctu_x_, ctu_y_);
  sb_[0][CI_CURR] = entry; sb_[0][CI_NEXT] = entry; sb_[0][CI_TEMP] = entry; goon_ = entry;
  compress_cu(0, SIZE_NONE);
  E.ctu_cost[addr] = best_[0]->cost; E.ctu_bits[addr] = best_[0]->bits; E.ctu_dist[addr] = best_[0]->dist;
  E.ctu_rd_fraction[addr] = (uint16_t)(coder_frac(goon_) & 32767);
  // the coder the next CTU starts from
  exit = sb_[0][CI_NEXT];
  memcpy(exit.split, entry.split, 3);
  uint64_t total = coder_frac(entry);
  total += final_walk(ctu_x_, ctu_y_, CTU, 0, exit);
  if (addr != E.n_ctu() - 1) total += hop_cabac_trm_bits(0);              // TEncCu::finishCU (:940-945): no terminating bin after the slice's last CTU
  coder_set_frac(exit, (uint32_t)(total & 32767));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the picture
// ---------------------------------------------------------------------------------------------------------------------------------
Encoder::Encoder(const EncConfig& cfg, Backend* be) : trace(NULL), n_candidates(0), cfg_(cfg), be_(be) {
  wctu_ = (cfg.pic_w + 63) / 64; hctu_ = (cfg.pic_h + 63) / 64;
  ctu_cost.assign(n_ctu(), 0.0); ctu_bits.assign(n_ctu(), 0); ctu_dist.assign(n_ctu(), 0); ctu_rd_fraction.assign(n_ctu(), 0); ctu_trace.resize(n_ctu()); batch_rounds = batch_requests = 0;
  pic.resize((size_t)n_ctu() * 256); ctu_entry.resize(n_ctu()); ctu_exit.resize(n_ctu()); committed.assign((size_t)(cfg.pic_w >> 3) * (cfg.pic_h >> 3), 0);
  for (size_t i = 0; i < pic.size(); i++) part_init(pic[i], 0);
  reach_below.store(0); reach_above.store(0); first_below.store(-1); first_above.store(-1);
}

LogBackend::LogBackend(Backend* inner, const char* path) : in_(inner), f_(fopen(path, "wb")) {}
LogBackend::~LogBackend() { if (f_) fclose(f_); }
void LogBackend::rec(int kind, int n, const void* a, size_t na, const void* b, size_t nb) {
  if (!f_) return;
  const int32_t h[2] = { kind, n }; const uint32_t z[2] = { (uint32_t)na, (uint32_t)nb };
  fwrite(h, 4, 2, f_); fwrite(z, 4, 2, f_); if (na) fwrite(a, 1, na, f_); if (nb) fwrite(b, 1, nb, f_);
}
void LogBackend::rec2(int kind, const void* a, size_t na, const void* a2, size_t na2, const void* b, size_t nb) {
  if (!f_) return;
  const int32_t h[2] = { kind, 1 }; const uint32_t z[2] = { (uint32_t)(na + na2), (uint32_t)nb };
  fwrite(h, 4, 2, f_); fwrite(z, 4, 2, f_); fwrite(a, 1, na, f_); fwrite(a2, 1, na2, f_); fwrite(b, 1, nb, f_);
}

void intra_syntax_dirs(hop_intra_cu_syntax& syn, const hop_intra_search_job& sj, const int dirs[4]) {
  const int npu = syn.part_nxn ? 4 : 1;
  for (int pu = 0; pu < npu; pu++) {
    const int l = (pu & 1) ? dirs[pu - 1] : sj.left_dir[pu], a = (pu >> 1) ? dirs[pu - 2] : sj.above_dir[pu];
    int* pr = syn.preds[pu];
    if (l == a) {
      if (l > 1) { pr[0] = l; pr[1] = ((l + 29) % 32) + 2; pr[2] = ((l - 1) % 32) + 2; }
      else { pr[0] = PLANAR_IDX; pr[1] = DC_IDX; pr[2] = VER_IDX; }
    } else {
      pr[0] = l; pr[1] = a;
      pr[2] = (l && a) ? (int)PLANAR_IDX : ((l + a) < 2 ? (int)VER_IDX : (int)DC_IDX);
    }
    syn.pred_num[pu] = 3; syn.luma_dir[pu] = dirs[pu];
  }
}

void Encoder::encode_frame(int first_ctus) {
  be_->begin_frame();
  for (size_t i = 0; i < pic.size(); i++) part_init(pic[i], 0);
  std::fill(committed.begin(), committed.end(), (uint8_t)0);
  Coder k; memset(&k, 0, sizeof(k));
  hop_cabac_init(&k.r, cfg_.slice_type, cfg_.qp); hop_cabac_cu_init(&k.c, cfg_.slice_type, cfg_.qp); hop_cabac_split_init(k.split, cfg_.slice_type, cfg_.qp);
  CtuWorker* w = new CtuWorker(*this, 0);
  const int n = (first_ctus > 0 && first_ctus < n_ctu()) ? first_ctus : n_ctu();
  Coder sync = k;                                                       // WaveFrontSynchro: the coder after the second CTU of the row above
  for (int a = 0; a < n; a++) {
    const int col = a % wctu_;
    if (cfg_.wpp && col == 0 && a > 0 && wctu_ >= 2) { k = sync; coder_set_frac(k, 0); }   // as in the wavefront below (TEncSlice.cpp:1057-1090)
    ctu_entry[a] = k;
    Coder next; w->compress_ctu(a, k, next);
    k = next;
    if (cfg_.wpp && col == 1) sync = k;
    if (trace) { fputs(ctu_trace[a].c_str(), trace); ctu_trace[a].clear(); }
  }
  n_candidates += w->n_cand_;
  delete w;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wavefront: one thread per CTU row, requests of the rows in flight rendezvous and are served in batches
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct Req { int kind; int lane; int n; const void* a; const void* b; void* out; int i0, i1, i2, i3; bool done; uint64_t tag; std::condition_variable* wake; bool posted; int worker; };   // wake: the submitting thread's own (only it is woken when the request is done)
enum { RQ_ME, RQ_PRED, RQ_DIST, RQ_VALID, RQ_INTER, RQ_INTRA, RQ_SAVE, RQ_RESTORE, RQ_COMMIT, RQ_PCOST, RQ_NOP };   // RQ_NOP: nothing but the round (FiberPool::flush)

// one group of requests of the same kind (and class) as one call of the batching backend
static void run_group(BatchInner* inner_, std::vector<Req*>& g) {
  const int kind = g[0]->kind;
  if (kind == RQ_NOP) return;
  if (kind == RQ_ME) {
    std::vector<hop_pu_job> j; for (Req* r : g) j.insert(j.end(), (const hop_pu_job*)r->a, (const hop_pu_job*)r->a + r->n);
    std::vector<hop_pu_result> o(j.size());
    inner_->me_search(0, (int)j.size(), j.data(), o.data());
    size_t at = 0; for (Req* r : g) { memcpy(r->out, &o[at], sizeof(hop_pu_result) * r->n); at += r->n; }
  } else if (kind == RQ_PRED) {
    std::vector<hop_pred_job> j; for (Req* r : g) j.insert(j.end(), (const hop_pred_job*)r->a, (const hop_pred_job*)r->a + r->n);
    inner_->pred_inter(0, (int)j.size(), j.data());
  } else if (kind == RQ_DIST) {
    std::vector<hop_dist_job> j; for (Req* r : g) j.insert(j.end(), (const hop_dist_job*)r->a, (const hop_dist_job*)r->a + r->n);
    std::vector<uint32_t> o(j.size());
    inner_->distortion(0, (int)j.size(), j.data(), o.data());
    size_t at = 0; for (Req* r : g) { memcpy(r->out, &o[at], 4 * r->n); at += r->n; }
  } else if (kind == RQ_VALID) {
    std::vector<int32_t> j; for (Req* r : g) j.insert(j.end(), (const int32_t*)r->a, (const int32_t*)r->a + 6 * r->n);
    std::vector<uint8_t> o(j.size() / 6);
    inner_->valid_pattern(0, (int)o.size(), j.data(), o.data());
    size_t at = 0; for (Req* r : g) { memcpy(r->out, &o[at], r->n); at += r->n; }
  } else if (kind == RQ_PCOST) {
    std::vector<int> len, kinds; std::vector<hop_pred_job> j;
    for (Req* r : g) { len.push_back(r->n); kinds.push_back(r->i0); j.insert(j.end(), (const hop_pred_job*)r->a, (const hop_pred_job*)r->a + r->n); }
    std::vector<uint32_t> o(j.size());
    inner_->pred_cost_n((int)g.size(), len.data(), j.data(), kinds.data(), o.data());
    size_t at = 0; for (Req* r : g) { memcpy(r->out, &o[at], 4 * r->n); at += r->n; }
  } else if (kind == RQ_INTER) {
    std::vector<const InterEval*> e; std::vector<const Coder*> in; std::vector<EvalResult*> o;
    for (Req* r : g) { e.push_back((const InterEval*)r->a); in.push_back((const Coder*)r->b); o.push_back((EvalResult*)r->out); }
    inner_->inter_n((int)g.size(), e.data(), in.data(), o.data());
  } else if (kind == RQ_INTRA) {
    std::vector<const IntraEval*> e; std::vector<const Coder*> in; std::vector<EvalResult*> o;
    for (Req* r : g) { e.push_back((const IntraEval*)r->a); in.push_back((const Coder*)r->b); o.push_back((EvalResult*)r->out); }
    inner_->intra_n((int)g.size(), e.data(), in.data(), o.data());
  } else {
    std::vector<int32_t> rc; for (Req* r : g) { rc.push_back(r->i0); rc.push_back(r->i1); rc.push_back(r->i2); rc.push_back(r->i3); }
    if (kind == RQ_COMMIT) {
      std::vector<int32_t> rs; for (Req* r : g) if (r->i3 > 0) { rs.push_back(r->i0); rs.push_back(r->i1); rs.push_back(r->i2); rs.push_back(r->i3 - 1); }   // restore_commit: the stash slot + 1
      if (!rs.empty()) inner_->stash_n((int)(rs.size() / 4), rs.data(), 1);
      inner_->commit_n((int)g.size(), rc.data());
    } else inner_->stash_n((int)g.size(), rc.data(), kind == RQ_RESTORE);
  }
}

class Rendezvous : public Backend {
 public:
  Rendezvous(BatchInner* inner, int n_threads) : rounds(0), requests(0), inner_(inner), active_(n_threads), failed_(false) { memset(tag_of_, 0, sizeof(tag_of_)); }
  void set_tag(int lane, uint64_t tag) { tag_of_[lane % SPINE_LANES] = tag; }
  // a row thread is about to block on another row's progress / has been released / has finished
  std::mutex m; std::condition_variable cv;
  void begin_frame() {}
  void me_search(int lane, int n, const hop_pu_job* j, hop_pu_result* r) { Req q = { RQ_ME, lane, n, j, NULL, r, 0, 0, 0, 0, false }; submit(q); }
  void pred_inter(int lane, int n, const hop_pred_job* j) { Req q = { RQ_PRED, lane, n, j, NULL, NULL, 0, 0, 0, 0, false }; submit(q); }
  void distortion(int lane, int n, const hop_dist_job* j, uint32_t* o) { Req q = { RQ_DIST, lane, n, j, NULL, o, 0, 0, 0, 0, false }; submit(q); }
  void valid_pattern(int lane, int n, const int32_t* v, uint8_t* o) { Req q = { RQ_VALID, lane, n, v, NULL, o, 0, 0, 0, 0, false }; submit(q); }
  void pred_cost(int lane, int n, const hop_pred_job* j, int kind, uint32_t* o) { Req q = { RQ_PCOST, lane, n, j, NULL, o, kind, 0, 0, 0, false }; submit(q); }
  void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& o) { Req q = { RQ_INTER, lane, 1, &e, &in, &o, e.job.log2_cu, e.skip_res, 0, 0, false }; submit(q); }
  void intra_cu(int lane, const IntraEval& e, const Coder& in, EvalResult& o) { Req q = { RQ_INTRA, lane, 1, &e, &in, &o, e.job.log2_cu, e.part_nxn, 0, 0, false }; submit(q); }
  void recon_save(int lane, int slot, int x, int y, int size) { Req q = { RQ_SAVE, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot, false }; submit(q); }
  void recon_restore(int lane, int slot, int x, int y, int size) { Req q = { RQ_RESTORE, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot, false }; submit(q); }
  void commit(int lane, int x, int y, int size) { Req q = { RQ_COMMIT, lane, 1, NULL, NULL, NULL, x, y, size, 0, false }; submit(q); }
  void restore_commit(int lane, int slot, int x, int y, int size) { Req q = { RQ_COMMIT, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot + 1, false }; submit(q); }
  // progress waits of the row threads go through the same lock so that "nobody can run" is detected exactly
  template <class Pred> void wait_until(std::unique_lock<std::mutex>& lk, Pred p) {
    if (p()) return;
    idle(lk);
    cv.wait(lk, [&] { return p() || failed_; });
    active_++;
  }
  void thread_done(std::unique_lock<std::mutex>& lk) { idle(lk); }
  bool failed() const { return failed_; }
  uint64_t rounds, requests;
 private:
  BatchInner* inner_; int active_; bool failed_; std::vector<Req*> pending_; uint64_t tag_of_[SPINE_LANES];
  void idle(std::unique_lock<std::mutex>& lk) {       // the caller stops running; if it was the last one, it serves what is pending first
    active_--;
    while (active_ == 0 && !pending_.empty()) serve(lk);
  }
  void submit(Req& q) {
    std::unique_lock<std::mutex> lk(m);
    if (failed_) throw 1;
    q.tag = tag_of_[q.lane % SPINE_LANES];
    std::condition_variable mine; q.wake = &mine;
    pending_.push_back(&q);
    active_--;
    while (active_ == 0 && !pending_.empty() && !q.done) serve(lk);
    mine.wait(lk, [&] { return q.done || failed_; });
    if (failed_ && !q.done) { for (size_t i = 0; i < pending_.size(); i++) if (pending_[i] == &q) { pending_.erase(pending_.begin() + i); break; } throw 1; }
  }
  void serve(std::unique_lock<std::mutex>&) {         // called with the lock held and every thread parked: one batch per kind / class
    // the requests with the smallest tag (the rows coded side by side have started their CTUs together: equal tags are the same operation); the others wait their turn
    uint64_t tmin = ~0ull; for (Req* r : pending_) if (r->tag < tmin) tmin = r->tag;
    std::vector<Req*> v, rest;
    for (Req* r : pending_) (r->tag == tmin ? v : rest).push_back(r);
    pending_.swap(rest);
    rounds++; requests += v.size();
    try {
      std::vector<char> used(v.size(), 0);
      for (size_t i = 0; i < v.size(); i++) {
        if (used[i]) continue;
        std::vector<Req*> g;
        for (size_t k = i; k < v.size(); k++) {
          if (used[k] || v[k]->kind != v[i]->kind) continue;
          if ((v[i]->kind == RQ_INTER || v[i]->kind == RQ_INTRA) && (v[k]->i0 != v[i]->i0 || v[k]->i1 != v[i]->i1)) continue;
          used[k] = 1; g.push_back(v[k]);
        }
        run_group(inner_, g);
      }
    } catch (...) { failed_ = true; }
    for (size_t i = 0; i < v.size(); i++) { v[i]->done = true; v[i]->wake->notify_one(); }
    active_ += (int)v.size();
    if (failed_) { for (Req* r : pending_) r->wake->notify_one(); cv.notify_all(); }   // everybody gives up
  }
};

// The same rendezvous without a thread per CTU row: the rows are FIBERS (ucontext) spread over a few worker threads.  A row that submits a request, or waits for the
// wavefront to advance, switches back to its worker's scheduler; a worker whose rows are all waiting joins a barrier; when every worker is there the requests with the
// smallest tag are served as one batch per kind (exactly as above) and the workers go round again.  A request costs two context switches in user space instead of a futex
// sleep and wake-up per row thread -- with hundreds of rows in flight (several pictures side by side) those wake-ups were the largest single cost of the host side.
// The context switch: callee-saved registers on the leaving stack, stack pointers exchanged.  (glibc's swapcontext also saves and restores the signal mask, two
// system calls per switch: at four switches per request that was most of the rows' host time.)  x86-64 System V, like everything else here.
extern "C" void hop_fiber_switch(void** save_sp, void* load_sp);
asm(".text\n.globl hop_fiber_switch\n.type hop_fiber_switch,@function\nhop_fiber_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size hop_fiber_switch, .-hop_fiber_switch\n");

class FiberPool : public Backend {
 public:
  struct Fiber { void* sp; char* stack; std::function<void()> body; bool done; Req* req; int wait_step; int worker; FiberPool* pool;
                 Fiber* parent; int live_children; bool wait_children; uint64_t tag;      // parent: a child of fork_join (recycled when done)
                 std::atomic<int>* wait_ctr; int wait_target; };                        // waiting until *wait_ctr >= wait_target
  FiberPool(BatchInner* inner, int n_workers) : rounds(0), requests(0), steps_complete(-1), inner_(inner), T_(n_workers), failed_(false), finished_(false), arrived_(0), gen_(0), left_(0), idle_rounds_(0) {
    sched_.resize(T_); mine_.resize(T_); local_.resize(T_); free_.resize(T_); kids_.resize(T_); pstore_.resize(T_); pjobs_.resize(T_);
  }
  ~FiberPool() { for (Fiber* f : all_) { stack_free(f->stack); delete f; } for (auto& v : kids_) for (Fiber* f : v) { stack_free(f->stack); delete f; } }
  void add(std::function<void()> body) {
    Fiber* f = new Fiber(); f->stack = stack_alloc(); f->body = body; f->done = false; f->req = NULL; f->wait_step = -1; f->worker = (int)(all_.size() % T_); f->pool = this;
    f->parent = NULL; f->live_children = 0; f->wait_children = false; f->tag = 0; f->wait_ctr = NULL; f->wait_target = 0;
    all_.push_back(f); mine_[f->worker].push_back(f); left_++;
  }
  // fn(0) ... fn(n - 1) as fibers of their own on the caller's worker; the caller goes on when all have ended
  void fork_join(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (failed_) throw 1;
    Fiber* me = current();
    const int w = me->worker;
    me->live_children = n;
    for (int i = 0; i < n; i++) {
      Fiber* f;
      if (!free_[w].empty()) { f = free_[w].back(); free_[w].pop_back(); }
      else { f = new Fiber(); f->stack = stack_alloc(); kids_[w].push_back(f); }
      f->body = [&fn, i]() { fn(i); };
      f->done = false; f->req = NULL; f->wait_step = -1; f->worker = w; f->pool = this; f->parent = me; f->live_children = 0; f->wait_children = false; f->tag = me->tag;
      f->wait_ctr = NULL; f->wait_target = 0;
      start(f);
      mine_[w].push_back(f);
    }
    me->wait_children = true;
    hop_fiber_switch(&me->sp, sched_[w]);
    if (failed_) throw 1;
  }
  void run() {                                                          // all fibers to completion
    std::vector<std::thread> th;
    for (int w = 0; w < T_; w++) th.emplace_back([this, w]() { worker(w); });
    for (auto& t : th) t.join();
  }
  bool failed() const { return failed_; }
  // ---- called from inside fibers ----
  void set_tag(int, uint64_t tag) { current()->tag = tag; }            // the tag belongs to the fiber (children of fork_join carry their own)
  void begin_frame() {}
  void me_search(int lane, int n, const hop_pu_job* j, hop_pu_result* r) { Req q = { RQ_ME, lane, n, j, NULL, r, 0, 0, 0, 0, false, 0, NULL }; submit(q); }
  void pred_inter(int lane, int n, const hop_pred_job* j) { Req q = { RQ_PRED, lane, n, j, NULL, NULL, 0, 0, 0, 0, false, 0, NULL }; if (posted_preds_) post(q, j); else submit(q); }
  void distortion(int lane, int n, const hop_dist_job* j, uint32_t* o) { Req q = { RQ_DIST, lane, n, j, NULL, o, 0, 0, 0, 0, false, 0, NULL }; submit(q); }
  void valid_pattern(int lane, int n, const int32_t* v, uint8_t* o) { Req q = { RQ_VALID, lane, n, v, NULL, o, 0, 0, 0, 0, false, 0, NULL }; submit(q); }
  void pred_cost(int lane, int n, const hop_pred_job* j, int kind, uint32_t* o) { Req q = { RQ_PCOST, lane, n, j, NULL, o, kind, 0, 0, 0, false, 0, NULL }; submit(q); }
  void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& o) { Req q = { RQ_INTER, lane, 1, &e, &in, &o, e.job.log2_cu, e.skip_res, 0, 0, false, 0, NULL }; submit(q); }
  void intra_cu(int lane, const IntraEval& e, const Coder& in, EvalResult& o) { Req q = { RQ_INTRA, lane, 1, &e, &in, &o, e.job.log2_cu, e.part_nxn, 0, 0, false, 0, NULL }; submit(q); }
  void recon_save(int lane, int slot, int x, int y, int size) { Req q = { RQ_SAVE, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot, false, 0, NULL }; if (posted_mode_ && (posted_kinds_ & 1)) post(q, NULL); else submit(q); }
  void recon_restore(int lane, int slot, int x, int y, int size) { Req q = { RQ_RESTORE, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot, false, 0, NULL }; if (posted_mode_ && (posted_kinds_ & 2)) post(q, NULL); else submit(q); }
  void commit(int lane, int x, int y, int size) { Req q = { RQ_COMMIT, lane, 1, NULL, NULL, NULL, x, y, size, 0, false, 0, NULL }; if (posted_mode_ && (posted_kinds_ & 4)) post(q, NULL); else submit(q); }
  // everything this fiber has posted is on the device when this returns (a CTU is only counted as retired, and handed to other ranks, behind it)
  void flush(int lane) { if (!posted_mode_) return; Req q = { RQ_NOP, lane, 1, NULL, NULL, NULL, 0, 0, 0, 0, false, 0, NULL }; submit(q); }
  void restore_commit(int lane, int slot, int x, int y, int size) {
    if (posted_mode_) { recon_restore(lane, slot, x, y, size); commit(lane, x, y, size); return; }
    Req q = { RQ_COMMIT, lane, 1, NULL, NULL, NULL, x, y, size, lane * 16 + slot + 1, false, 0, NULL }; submit(q);
  }
  void wait_counter(std::atomic<int>* ctr, int target) {                // until *ctr >= target (another fiber counts it up)
    if (ctr->load() >= target) return;
    Fiber* f = current(); f->wait_ctr = ctr; f->wait_target = target;
    hop_fiber_switch(&f->sp, sched_[f->worker]);
    if (failed_) throw 1;
  }
  void wait_step(int st) {                                              // until every wavefront step <= st is finished
    if (steps_complete.load() >= st) return;
    Fiber* f = current(); f->wait_step = st;
    hop_fiber_switch(&f->sp, sched_[f->worker]);
    if (failed_) throw 1;
  }
  uint64_t rounds, requests;
  double serve_s = 0;                                                   // wall time inside serve(): packing, issuing, waiting for the device, unpacking
  int tag_shift = 0;
  std::atomic<int> steps_complete;
  std::mutex steps_m;                                                   // guards the callers' step counters
 private:
  enum { STACK = 512 * 1024, GUARD = 4096 };
  // a fiber's stack with an inaccessible page below it: an overflow (the recursion of compress_cu under a candidate child) faults instead of corrupting the heap
  static char* stack_alloc() {
    void* m = mmap(NULL, STACK + GUARD, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw 1;
    mprotect(m, GUARD, PROT_NONE);
    return (char*)m + GUARD;
  }
  static void stack_free(char* s) { if (s) munmap(s - GUARD, STACK + GUARD); }
  static Fiber*& current() { static thread_local Fiber* cur = NULL; return cur; }
  static void tramp() {                                                 // a fiber's first frame: entered by the first switch to it
    Fiber* f = current();
    try { f->body(); } catch (...) { f->pool->failed_ = true; }
    f->done = true;
    if (f->parent) f->parent->live_children--; else f->pool->left_--;
    hop_fiber_switch(&f->sp, f->pool->sched_[f->worker]);               // for good (the stack is reused by the next fork)
    abort();
  }
  // a fresh stack whose first switch "returns" into tramp: six callee-saved registers, the return address, and the slot a call would have pushed (alignment)
  static void start(Fiber* f) {
    uintptr_t top = ((uintptr_t)f->stack + STACK) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int i = 0; i < 6; i++) sp[i] = NULL;
    sp[6] = (void*)&tramp; sp[7] = NULL;
    f->sp = sp;
  }
  // A request without an answer (a prediction into the resident picture, a reconstruction put aside or brought back, a commit to the SS reference) need not stop its
  // row: it is handed over and the row goes on; the next time requests are served, everything posted is issued first -- the stash / commit requests in the order each
  // worker received them, then the predictions -- and only then the requests rows are waiting for.  A row's own order is kept (whatever consumes a posted request's effect
  // is a waiting request of the same row, or a row of a later wavefront step, whose first request waits), and rounds that served nothing but such requests disappear
  // (HOP_SPINE_POSTED=1; off by default until measured on the device: tests/test_spine_cpu.py runs both ways).
  void post(Req& q, const hop_pred_job* jobs) {
    if (failed_) throw 1;
    Fiber* f = current(); const int w = f->worker;
    q.tag = f->tag; q.posted = true; q.worker = w;
    if (jobs) { pjobs_[w].emplace_back(jobs, jobs + q.n); q.a = pjobs_[w].back().data(); }
    pstore_[w].push_back(q);
    local_[w].push_back(&pstore_[w].back());
  }
  void submit(Req& q) {
    if (failed_) throw 1;
    Fiber* f = current();
    q.tag = f->tag;
    local_[f->worker].push_back(&q);
    f->req = &q;
    hop_fiber_switch(&f->sp, sched_[f->worker]);
    if (!q.done) throw 1;                                               // resumed without an answer: the pool has failed
  }
  void worker(int w) {
    for (Fiber* f : mine_[w]) {
      start(f);
    }
    for (;;) {
      bool ran = false, ended = false;
      for (size_t i = 0; i < mine_[w].size(); i++) {                     // (fork_join appends to the list while this loop runs)
        Fiber* f = mine_[w][i];
        if (f->done) { ended |= f->parent != NULL; continue; }
        if (f->req) { if (!f->req->done && !failed_) continue; f->req = NULL; }
        else if (f->wait_step >= 0) { if (steps_complete.load() < f->wait_step && !failed_) continue; f->wait_step = -1; }
        else if (f->wait_children) { if (f->live_children > 0) continue; f->wait_children = false; }
        else if (f->wait_ctr) { if (f->wait_ctr->load() < f->wait_target && !failed_) continue; f->wait_ctr = NULL; }
        current() = f;
        hop_fiber_switch(&sched_[w], f->sp);                            // until it submits, waits or ends
        ran = true;
      }
      if (ended) {                                                      // children that have ended: out of the list, their stacks free for the next fork
        size_t k = 0;
        for (size_t i = 0; i < mine_[w].size(); i++) { Fiber* f = mine_[w][i]; if (f->done && f->parent) free_[w].push_back(f); else mine_[w][k++] = f; }
        mine_[w].resize(k);
      }
      if (ran) continue;                                                // what ran may have released others of this worker
      if (!barrier(w)) break;
    }
  }
  int spin_us_ = [] { const char* e = getenv("HOP_SPINE_SPIN_US"); return e ? atoi(e) : 200; }();   // how long a waiting worker watches the generation before it sleeps
  bool barrier(int) {                                                   // false: everything has finished
    std::unique_lock<std::mutex> lk(bm_);
    const uint64_t gen = gen_.load();
    if (++arrived_ == T_) {                                             // every worker is out of runnable rows: serve, or finish
      lk.unlock();                                                      // (the others only watch gen_ from here on)
      for (int k = 0; k < T_; k++) { pending_.insert(pending_.end(), local_[k].begin(), local_[k].end()); local_[k].clear(); }
      if (left_.load() == 0) { if (posted_mode_ && !pending_.empty() && !failed_) { try { serve_posted(); } catch (...) { failed_ = true; } } finished_ = true; }
      else if (failed_) { for (Req* r : pending_) (void)r; pending_.clear(); }
      else if (!pending_.empty() || !inflight_.empty()) { serve(); idle_rounds_ = 0; }
      else if (++idle_rounds_ > 100000) failed_ = true;                 // rows waiting for a step nobody can finish
      lk.lock();
      arrived_ = 0; gen_.store(gen + 1);
      lk.unlock();
      bcv_.notify_all();
      return !finished_;
    }
    lk.unlock();
    // a round takes about a millisecond: watch the generation for a moment before going to sleep (3 ms, 200 us and no watching at all measured within 3 % of each other)
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; gen_.load(std::memory_order_acquire) == gen; spin++) {
      __builtin_ia32_pause();
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) {
        lk.lock();
        bcv_.wait(lk, [&] { return gen_.load() != gen; });
        lk.unlock();
        break;
      }
    }
    return !finished_;
  }
  void serve_posted() {                                                 // everything posted since the last serve, out of pending_
    std::vector<std::vector<Req*> > seq(T_); std::vector<Req*> preds, rest;
    for (Req* r : pending_) { if (!r->posted) rest.push_back(r); else if (r->kind == RQ_PRED) preds.push_back(r); else seq[r->worker].push_back(r); }
    if (rest.size() == pending_.size()) return;
    std::vector<size_t> at(T_, 0);
    for (;;) {                                                            // stash / commit requests: waves of one kind, every worker's next run of that kind
      int kind = -1; for (int w = 0; w < T_ && kind < 0; w++) if (at[w] < seq[w].size()) kind = seq[w][at[w]]->kind;
      if (kind < 0) break;
      std::vector<Req*> g;
      for (int w = 0; w < T_; w++) while (at[w] < seq[w].size() && seq[w][at[w]]->kind == kind) g.push_back(seq[w][at[w]++]);
      run_group(inner_, g); requests += g.size();
    }
    if (!preds.empty()) { run_group(inner_, preds); requests += preds.size(); }
    pending_.swap(rest);
    for (int w = 0; w < T_; w++) { pstore_[w].clear(); pjobs_[w].clear(); }
  }
  void serve() {
    struct Clock { double& acc; std::chrono::steady_clock::time_point t0; explicit Clock(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~Clock() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } clock(serve_s);
    if (posted_mode_) {
      try { serve_posted(); } catch (...) { failed_ = true; }
      if (failed_ || (pending_.empty() && inflight_.empty())) return;
    }
    // the requests with the smallest tag; with candidates side by side (tag_shift = 20) the tag is cut down to the quadtree node: the candidates of a node are at
    // different steps of their chains at any moment, and all of them are served -- one launch chain per kind and class present
    uint64_t tmin = ~0ull; for (Req* r : pending_) if ((r->tag >> tag_shift) < tmin) tmin = r->tag >> tag_shift;
    std::vector<Req*> v, rest;
    for (Req* r : pending_) ((r->tag >> tag_shift) == tmin ? v : rest).push_back(r);
    std::vector<Req*> early;                                            // intra evaluations issued ahead (below)
    if (tag_shift) {
      // candidates side by side: their searches and predictions (short chains) first, the candidate evaluations (long chains) once nothing else of the node is
      // waiting -- then the evaluations of all candidates of the node, of every CTU in flight, are ONE chain per class instead of one per candidate.  The intra
      // evaluations need nothing from the searches: they are handed over at once (a backend with streams of its own runs them beside the short rounds) and their
      // answers are collected with the round that ends the node's evaluations.
      bool light = false; for (Req* r : v) light |= (r->kind != RQ_INTER && r->kind != RQ_INTRA);
      if (light) {
        std::vector<Req*> keep;
        for (Req* r : v) {
          if (r->kind == RQ_INTER) rest.push_back(r);
          else if (r->kind == RQ_INTRA) { if (inner_->can_defer()) early.push_back(r); else rest.push_back(r); }
          else keep.push_back(r);
        }
        v.swap(keep);
      }
      // ... and among the short chains the motion searches (a dozen dependent launches, ten times a prediction request) once nothing cheaper of the node is waiting: the
      // candidates and CTUs that are a few cheap requests behind catch up first, and their searches then share the chain instead of taking one each
      if (defer_me_) {
        bool cheap = false; for (Req* r : v) cheap |= (r->kind != RQ_ME && r->kind != RQ_INTER && r->kind != RQ_INTRA);
        if (cheap) {
          std::vector<Req*> keep;
          for (Req* r : v) (r->kind == RQ_ME ? rest : keep).push_back(r);
          v.swap(keep);
        }
      }
    }
    pending_.swap(rest);
    rounds++; requests += v.size() + early.size();
    try {
      if (!early.empty()) {                                               // one group per class, issued and left in flight
        std::vector<char> used(early.size(), 0);
        for (size_t i = 0; i < early.size(); i++) {
          if (used[i]) continue;
          std::vector<Req*> g;
          for (size_t k = i; k < early.size(); k++) if (!used[k] && early[k]->i0 == early[i]->i0 && early[k]->i1 == early[i]->i1) { used[k] = 1; g.push_back(early[k]); }
          run_group(inner_, g);
        }
        inflight_.insert(inflight_.end(), early.begin(), early.end());
      }
      { unsigned mask = 0; for (Req* r : v) mask |= 1u << r->kind; round_masks_[mask]++; }
      std::vector<char> used(v.size(), 0);
      for (size_t i = 0; i < v.size(); i++) {
        if (used[i]) continue;
        std::vector<Req*> g;
        for (size_t k = i; k < v.size(); k++) {
          if (used[k] || v[k]->kind != v[i]->kind) continue;
          if ((v[i]->kind == RQ_INTER || v[i]->kind == RQ_INTRA) && (v[k]->i0 != v[i]->i0 || v[k]->i1 != v[i]->i1)) continue;
          used[k] = 1; g.push_back(v[k]);
        }
        run_group(inner_, g);
      }
      // a round without evaluations leaves what was issued ahead in flight; one with evaluations (or one that has nothing else to wait for) collects everything
      bool heavy = v.empty(); for (Req* r : v) heavy |= (r->kind == RQ_INTER || r->kind == RQ_INTRA);
      for (Req* r : inflight_) heavy |= (r->tag >> tag_shift) < tmin;     // nothing else of that node is waiting any more
      if (heavy || inflight_.empty()) {
        inner_->end_round();
        for (Req* r : inflight_) r->done = true;
        inflight_.clear();
      }
      for (size_t i = 0; i < v.size(); i++) v[i]->done = true;
    } catch (...) { failed_ = true; }
  }
  std::map<unsigned, uint64_t> round_masks_;                            // HOP_SPINE_ROUND_STATS: how many rounds served which mix of request kinds
 public:
  void print_round_stats() { if (!getenv("HOP_SPINE_ROUND_STATS")) return; for (auto& kv : round_masks_) fprintf(stderr, "hop spine rounds: kinds %03x x %llu\n", kv.first, (unsigned long long)kv.second); }
 private:
  bool defer_me_ = [] { const char* e = getenv("HOP_SPINE_DEFER_ME"); return e ? atoi(e) != 0 : true; }();
  // HOP_SPINE_POSTED: 0 (default) -- every request waits for its round; 1 -- the requests that have no answer and touch nothing another pending request reads (a reconstruction put aside or brought back, an
  // SS-reference commit) do not stop their row: they are handed over and issued first at the next serve, every worker's in its order; 2 -- predictions without a cost too
  // (only where they still are requests of their own, HOP_SPINE_FUSE_PRED=0; on the device a batch of posted predictions is one launch and two predictions of one block would
  // lose their order, so the device backend never allows 2: posted_requests_allowed).  Level 1 with restore + commit measured on the device at one picture's breadth: 14 %
  // fewer rounds, no gain (77.5 against 78.8 - 79.3 CTU/s): off
  int posted_level_ = [] { const char* e = getenv("HOP_SPINE_POSTED"); int v = e ? atoi(e) : 0; if (v > 1 && !posted_requests_allowed) v = 1; return v < 0 ? 0 : v; }();
  bool posted_mode_ = posted_level_ >= 1, posted_preds_ = posted_level_ >= 2;
  // which of them are posted at level 1: 1 stash, 2 restore, 4 commit.  Restore and commit by default; a posted stash failed on the device -- one picture of the GPU tests
  // (448x192, --MIsize=15, 16 slots: an intra candidate rated from a stale neighbourhood), cause not found, the CPU spine is fine with it -- and stays a request of its own
  int posted_kinds_ = [this] { const char* e = getenv("HOP_SPINE_POSTED_KINDS"); return e ? atoi(e) : (posted_level_ >= 2 ? 7 : 6); }();
  std::vector<std::deque<Req> > pstore_; std::vector<std::deque<std::vector<hop_pred_job> > > pjobs_;   // per worker: what was posted since the last serve (deques: addresses stay)
  std::vector<Req*> inflight_;
  BatchInner* inner_; int T_; volatile bool failed_; bool finished_;
  std::vector<void*> sched_; std::vector<std::vector<Fiber*> > mine_; std::vector<std::vector<Req*> > local_; std::vector<Fiber*> all_; std::vector<Req*> pending_;
  std::mutex bm_; std::condition_variable bcv_; int arrived_; std::atomic<uint64_t> gen_; std::atomic<int> left_; int idle_rounds_;
  std::vector<std::vector<Fiber*> > free_, kids_;
};
}  // namespace

void Encoder::encode_frame_wavefront(BatchInner* inner, int lag, int) { Encoder* e = this; wavefront_many(&e, 1, inner, NULL, 0, lag); }
void Encoder::encode_frame_wavefront_direct(Backend* const* lanes, int n_lanes, int lag) { Encoder* e = this; wavefront_many(&e, 1, NULL, lanes, n_lanes, lag); }
void Encoder::encode_pictures_wavefront(Encoder* const* encs, int n, BatchInner* inner, int lag, int lane_base, bool begin, int threads) { wavefront_many(encs, n, inner, NULL, 0, lag, lane_base, begin, threads); }

// n pictures of equal geometry side by side (n = 1: one picture): one thread per CTU row of every picture, all of them on one rendezvous
void Encoder::wavefront_many(Encoder* const* encs, int n_pic, BatchInner* inner, Backend* const* lanes, int n_lanes, int lag, int lane_base, bool begin, int threads) {
  if (n_pic <= 0 || (!inner && n_lanes <= 0) || lag <= 0) throw 1;
  Encoder& E0 = *encs[0];
  const int rows = E0.hctu_, cols = E0.wctu_;
  for (int p = 0; p < n_pic; p++) if (!encs[p]->cfg_.wpp || encs[p]->hctu_ != rows || encs[p]->wctu_ != cols) throw 1;
  if (lag > cols) lag = cols;                                          // lag = cols is raster order already
  const int rif = (cols + lag - 1) / lag + 1;                          // rows of one picture that can be in flight together (+ 1 spare)
  if (lane_base < 0 || lane_base + (long)n_pic * (rif < rows ? rif : rows) > SPINE_LANES) throw 1;
  if (begin) { if (inner) inner->begin_frame(); else lanes[0]->begin_frame(); }
  Coder init; memset(&init, 0, sizeof(init));
  hop_cabac_init(&init.r, E0.cfg_.slice_type, E0.cfg_.qp); hop_cabac_cu_init(&init.c, E0.cfg_.slice_type, E0.cfg_.qp); hop_cabac_split_init(init.split, E0.cfg_.slice_type, E0.cfg_.qp);
  for (int p = 0; p < n_pic; p++) {
    Encoder& E = *encs[p];
    for (size_t i = 0; i < E.pic.size(); i++) part_init(E.pic[i], 0);
    std::fill(E.committed.begin(), E.committed.end(), (uint8_t)0);
  }
  if (inner) {                                                          // the batching form: rows as fibers on a few worker threads
    // more workers than the 16 CPUs a GPU box gives a job: a worker spends most of a round asleep at the barrier (measured at 384 pictures: 16 / 24 / 32 workers 275 - 283 /
    // 282 - 287 / 288 CTU/s)
    int T = (int)std::thread::hardware_concurrency(); if (T > 32) T = 32; if (T < 1) T = 1;
    // every round is a barrier over all workers: one picture's 25 rows in flight are served best by about a dozen (12 / 25 / 32 workers: 68.4 / 64.8 / 65.9 CTU/s on one
    // box, the time between the rounds 6.4 / 13.1 / 12.6 s of 54 - 59 s), hundreds of rows (a stack of pictures) by all 32
    { const int in_flight = n_pic * (rif < rows ? rif : rows), want = in_flight / 2 < 4 ? 4 : in_flight / 2; if (T > want) T = want; }
    if (threads > 0) T = threads;
    if (const char* e = getenv("HOP_SPINE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) T = v; }
    if (T > rows * n_pic) T = rows * n_pic;
    // one picture's rows dealt to several ranks (EncConfig::shard): this process runs the rows r % world == rank and a fiber that, after every wavefront step, exchanges the
    // step's finished CTUs with the other ranks
    const int world = (E0.cfg_.shard_world > 1 && E0.cfg_.shard) ? E0.cfg_.shard_world : 1, rank = world > 1 ? E0.cfg_.shard_rank : 0;
    if (world > 1 && (n_pic != 1 || rank < 0 || rank >= world)) throw 1;
    auto owned = [&](int r) { return world == 1 || r % world == rank; };
    if (world > 1) { int mine = 0; for (int r = 0; r < rows; r++) mine += owned(r); if (T > mine + 1) T = mine + 1; if (T < 1) T = 1; }
    FiberPool pool(inner, T);
    if (E0.cfg_.spec_slots > 0) pool.tag_shift = 20;
    const int n_steps = cols + lag * (rows - 1);
    std::vector<int> in_step(n_steps, 0), fin_step(n_steps, 0);
    for (int r = 0; r < rows; r++) if (owned(r)) for (int c = 0; c < cols; c++) in_step[c + lag * r] += n_pic;
    std::vector<Coder> sync((size_t)rows * n_pic);
    std::vector<uint64_t> cand((size_t)rows * n_pic, 0);
    std::vector<std::atomic<int> > fin_ctr(world > 1 ? n_steps : 0);     // sharded: the step's finished local CTUs (the exchange fiber waits on it)
    for (auto& a : fin_ctr) a.store(0);
    std::atomic<int> agreed_cancel(0);                                   // sharded: a cancel request every rank has seen (set after an exchange)
    for (int p = 0; p < n_pic; p++) for (int r = 0; r < rows; r++) {
      if (!owned(r)) continue;
      pool.add([&, p, r]() {
        Encoder& E = *encs[p];
        const int lane = lane_base + p * (rif < rows ? rif : rows) + r % rif;
        CtuWorker* w = new CtuWorker(E, lane, &pool);
        int c = 0;
        try {
          Coder k = init;
          for (; c < cols; c++) {
            const int st = c + lag * r;
            pool.wait_step(st - 1);
            if (world > 1 ? agreed_cancel.load() != 0 : (E.cfg_.cancel && E.cfg_.cancel->load())) {   // hop_encode_cancel: this row stops here; nobody waits for the CTUs it leaves uncoded
              std::lock_guard<std::mutex> g(pool.steps_m);
              if (world > 1) { for (; c < cols; c++) fin_ctr[c + lag * r].fetch_add(1); break; }
              for (; c < cols; c++) fin_step[c + lag * r]++;
              int sc = pool.steps_complete.load();
              while (sc + 1 < n_steps && fin_step[sc + 1] == in_step[sc + 1]) sc++;
              pool.steps_complete.store(sc);
              break;
            }
            if (c == 0 && r > 0 && cols >= 2) { std::lock_guard<std::mutex> g(pool.steps_m); k = sync[(size_t)p * rows + r - 1]; coder_set_frac(k, 0); }   // loadContexts (see below)
            const int a = r * cols + c;
            E.ctu_entry[a] = k;
            Coder next; w->compress_ctu(a, k, next);
            k = next;
            pool.flush(lane);                                             // (posted stash / restore / commit requests of the CTU: done before it counts)
            if (E.cfg_.progress) E.cfg_.progress->fetch_add(1);
            std::lock_guard<std::mutex> g(pool.steps_m);
            if (c == 1) sync[(size_t)p * rows + r] = k;
            if (world > 1) { E.ctu_exit[a] = k; fin_ctr[st].fetch_add(1); continue; }   // (the exchange fiber moves steps_complete on)
            fin_step[st]++;
            int sc = pool.steps_complete.load();
            while (sc + 1 < n_steps && fin_step[sc + 1] == in_step[sc + 1]) sc++;
            pool.steps_complete.store(sc);
          }
        } catch (...) {
          std::lock_guard<std::mutex> g(pool.steps_m);
          if (world > 1) { for (; c < cols; c++) fin_ctr[c + lag * r].fetch_add(1); }
          else {
            for (; c < cols; c++) fin_step[c + lag * r]++;               // after a failure: nobody waits for this row
            int sc = pool.steps_complete.load();
            while (sc + 1 < n_steps && fin_step[sc + 1] == in_step[sc + 1]) sc++;
            pool.steps_complete.store(sc);
          }
          cand[(size_t)p * rows + r] = w->n_cand_; delete w;
          throw;
        }
        cand[(size_t)p * rows + r] = w->n_cand_;
        delete w;
      });
    }
    // the exchange fiber: step after step, once this rank's CTUs of the step are finished -- their records out, everybody's in (an all-gather of equally sized slot tables:
    // slot 0 of a rank carries its cancel request), the other ranks' CTUs written into the picture on this side, then the next step may start
    struct ShardRec { int32_t addr, valid; double cost; uint32_t bits, dist; uint32_t frac, pad; Coder entry, exit; Part parts[256]; int16_t y[4096], cb[1024], cr[1024]; };
    if (world > 1) pool.add([&]() {
      Encoder& E = E0;
      const EncConfig& cfg = E.cfg_;
      auto count_of = [&](int g, int s) { int n = 0; for (int r = g; r < rows; r += world) { const int c = s - lag * r; n += (c >= 0 && c < cols); } return n; };
      std::vector<ShardRec> sendb, recvb;
      for (int s = 0; s < n_steps; s++) {
        pool.wait_counter(&fin_ctr[s], in_step[s]);
        int slots = 0; for (int g = 0; g < world; g++) slots = std::max(slots, count_of(g, s));
        sendb.assign((size_t)slots + 1, ShardRec()); recvb.resize((size_t)world * (slots + 1));
        memset(&sendb[0], 0, sizeof(ShardRec) * sendb.size());
        sendb[0].addr = -1; sendb[0].valid = (cfg.cancel && cfg.cancel->load()) ? 1 : 0;
        int q = 1;
        for (int r = rank; r < rows; r += world) {
          const int c = s - lag * r; if (c < 0 || c >= cols) continue;
          const int a = r * cols + c, x = c * CTU, y = r * CTU, w = std::min(CTU, cfg.pic_w - x), h = std::min(CTU, cfg.pic_h - y);
          ShardRec& t = sendb[q++];
          t.addr = a; t.valid = 1; t.cost = E.ctu_cost[a]; t.bits = E.ctu_bits[a]; t.dist = E.ctu_dist[a]; t.frac = E.ctu_rd_fraction[a]; t.entry = E.ctu_entry[a]; t.exit = E.ctu_exit[a];
          memcpy(t.parts, &E.pic[(size_t)a * 256], sizeof(t.parts));
          inner->export_block(x, y + cfg.y_origin, w, h, t.y, t.cb, t.cr);
        }
        cfg.shard->allgather(&sendb[0], &recvb[0], sizeof(ShardRec) * sendb.size());
        bool any_cancel = false;
        for (int g = 0; g < world; g++) {
          const ShardRec* rr = &recvb[(size_t)g * (slots + 1)];
          any_cancel |= rr[0].valid != 0;
          if (g == rank) continue;
          for (int k2 = 1; k2 <= slots; k2++) {
            const ShardRec& t = rr[k2];
            if (!t.valid) continue;
            const int a = t.addr, c = a % cols, r = a / cols, x = c * CTU, y = r * CTU, w = std::min(CTU, cfg.pic_w - x), h = std::min(CTU, cfg.pic_h - y);
            if (a < 0 || a >= E.n_ctu() || r % world != g) throw 1;
            E.ctu_cost[a] = t.cost; E.ctu_bits[a] = t.bits; E.ctu_dist[a] = t.dist; E.ctu_rd_fraction[a] = (uint16_t)t.frac; E.ctu_entry[a] = t.entry; E.ctu_exit[a] = t.exit;
            memcpy(&E.pic[(size_t)a * 256], t.parts, sizeof(t.parts));
            for (int yy = y >> 3; yy < (y + h) >> 3; yy++) memset(&E.committed[(size_t)yy * (cfg.pic_w >> 3) + (x >> 3)], 1, w >> 3);
            if (c == 1) { std::lock_guard<std::mutex> gd(pool.steps_m); sync[r] = t.exit; }
            inner->import_block(x, y + cfg.y_origin, w, h, t.y, t.cb, t.cr);
          }
        }
        if (any_cancel) { agreed_cancel.store(1); pool.steps_complete.store(n_steps - 1); break; }   // every rank has seen it at this step: the rows stop at their next CTU
        pool.steps_complete.store(s);
      }
    });
    const std::chrono::steady_clock::time_point run_t0 = std::chrono::steady_clock::now();
    pool.run(); pool.print_round_stats();
    const double run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - run_t0).count();
    for (int p = 0; p < n_pic; p++) {
      Encoder& E = *encs[p];
      for (int r = 0; r < rows; r++) E.n_candidates += cand[(size_t)p * rows + r];
      E.batch_rounds = pool.rounds; E.batch_requests = pool.requests; E.batch_serve_s = pool.serve_s; E.batch_run_s = run_s;
      if (E.trace) for (int a = 0; a < E.n_ctu(); a++) { fputs(E.ctu_trace[a].c_str(), E.trace); E.ctu_trace[a].clear(); }
    }
    if (pool.failed()) throw 1;
    return;
  }
  Rendezvous rv(inner, rows * n_pic);
  // synchronous wavefront: step s holds the CTUs (r, c) with c + lag * r == s of every picture; a step starts when the previous one has finished, so that its CTUs start
  // together (their requests then carry equal tags and meet in the batches) and every CTU finds the rows above it coded up to column c + lag - 1
  const int n_steps = cols + lag * (rows - 1);
  std::vector<int> in_step(n_steps, 0), fin_step(n_steps, 0);
  for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) in_step[c + lag * r] += n_pic;
  int steps_complete = -1;                                             // all steps <= this one are finished
  std::vector<Coder> sync((size_t)rows * n_pic);                       // the coder after the second CTU of each row (WaveFrontSynchro)
  std::vector<uint64_t> cand((size_t)rows * n_pic, 0);
  std::vector<std::thread> th;
  for (int p = 0; p < n_pic; p++) for (int r = 0; r < rows; r++) {
    th.emplace_back([&, p, r]() {
      Encoder& E = *encs[p];
      const int lane = p * (rif < rows ? rif : rows) + r % rif;
      CtuWorker* w = new CtuWorker(E, lane, inner ? (Backend*)&rv : lanes[(p * rows + r) % n_lanes]);
      int c = 0;
      try {
        Coder k = init;
        for (; c < cols; c++) {
          const int st = c + lag * r;
          {
            std::unique_lock<std::mutex> lk(rv.m);
            rv.wait_until(lk, [&] { return steps_complete >= st - 1; });
            if (rv.failed()) break;
            if (c == 0 && r > 0 && cols >= 2) { k = sync[(size_t)p * rows + r - 1]; coder_set_frac(k, 0); }   // loadContexts: the contexts of the row above after its second CTU, the row's own (fresh) bin coder
          }
          const int a = r * cols + c;
          E.ctu_entry[a] = k;
          Coder next; w->compress_ctu(a, k, next);
          k = next;
          std::unique_lock<std::mutex> lk(rv.m);
          if (c == 1) sync[(size_t)p * rows + r] = k;
          fin_step[st]++;
          while (steps_complete + 1 < n_steps && fin_step[steps_complete + 1] == in_step[steps_complete + 1]) steps_complete++;
          rv.cv.notify_all();
        }
      } catch (...) {}
      cand[(size_t)p * rows + r] = w->n_cand_;
      delete w;
      std::unique_lock<std::mutex> lk(rv.m);
      for (; c < cols; c++) { fin_step[c + lag * r]++; }               // after a failure: nobody waits for this row
      while (steps_complete + 1 < n_steps && fin_step[steps_complete + 1] == in_step[steps_complete + 1]) steps_complete++;
      rv.thread_done(lk);
      rv.cv.notify_all();
    });
  }
  for (auto& t : th) t.join();
  for (int p = 0; p < n_pic; p++) {
    Encoder& E = *encs[p];
    for (int r = 0; r < rows; r++) E.n_candidates += cand[(size_t)p * rows + r];
    E.batch_rounds = rv.rounds; E.batch_requests = rv.requests;
    if (E.trace) for (int a = 0; a < E.n_ctu(); a++) { fputs(E.ctu_trace[a].c_str(), E.trace); E.ctu_trace[a].clear(); }
  }
  if (rv.failed()) throw 1;
}

}  // namespace hopspine
