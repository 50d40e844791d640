// spine_backend_cpu.cpp -- TEST INFRASTRUCTURE ONLY: the RD spine of the product (hevc-hop_amd/host/hop_spine.cpp, compiled here unchanged) instantiated over
// the CPU restatement (oracle/hop_oracle*.c) instead of the HIP kernels.  It exists so that the spine -- host logic -- can be pinned against the reference encoder
// in the build container (tests/test_spine_cpu.py: per-CTU costs of cost.csv, every candidate of the reference's RD search, the reconstruction), and it is the
// "same work" CPU port bench.py times as cpu_baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load libhop_spine_cpu.so.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <atomic>
#include "../hevc-hop_amd/host/hop_spine.h"
extern "C" {
#include "hop_oracle.h"
}

using namespace hopspine;

namespace {

struct Plane { int w, h, stride, margin; std::vector<int16_t> buf; int16_t* p00; };

// the requests of one batch are independent of each other (different CTUs or candidate slots: disjoint rectangles of the pictures): a few threads take them side by side, so
// that the CPU tests of pictures with several CTUs or candidates in flight finish sooner.  HOP_CPU_BACKEND_THREADS=1 switches it off.
template <class F> static void parallel_batch(int n, F fn) {
  static const int T = [] { const char* e = getenv("HOP_CPU_BACKEND_THREADS"); int t = e ? atoi(e) : 6; return t < 1 ? 1 : t > 16 ? 16 : t; }();
  if (n <= 1 || T <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
  std::atomic<int> next(0);
  auto work = [&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
  std::vector<std::thread> th; const int k = (n < T ? n : T) - 1;
  for (int t = 0; t < k; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
}

class CpuBackend : public BatchInner {
 public:
  // slots: the original, prediction and reconstruction pictures exist slots + 1 times, copy k at rows k * H (what hop_ctx_set_slots makes of a context)
  CpuBackend(int W, int H, int bd, const int16_t* y, const int16_t* cb, const int16_t* cr, int slots = 0) : W(W), H(H), bd(bd) {
    hop_o_scan_init();                                                     // (lazily built tables: before any thread uses them)
    for (int c = 0; c < 3; c++) {
      const int w = c ? W / 2 : W, h = c ? H / 2 : H;
      org[c].assign((size_t)w * h * (slots + 1), 0); pred[c].assign((size_t)w * h * (slots + 1), 0); rec[c].assign((size_t)w * h * (slots + 1), 0);
      for (int k = 0; k <= slots; k++) memcpy(&org[c][(size_t)k * w * h], c == 0 ? y : c == 1 ? cb : cr, (size_t)w * h * 2);
      if (c == 0) coefpic.assign((size_t)(slots + 1) * ((W + 63) / 64) * ((H + 63) / 64) * 6144, 0);
      const int m = c ? 40 : 80, G = 64;                                   // the reference's margins + guard rows (see tests/hoputil.py:Planes)
      ss[c].w = w; ss[c].h = h; ss[c].margin = m; ss[c].stride = w + 2 * m;
      ss[c].buf.assign((size_t)ss[c].stride * (h + 2 * m + 2 * G), -1);
      ss[c].p00 = &ss[c].buf[0] + (size_t)(G + m) * ss[c].stride + m;
    }
  }
  void begin_frame() {
    for (int c = 0; c < 3; c++) { std::fill(ss[c].buf.begin(), ss[c].buf.end(), (int16_t)-1); std::fill(rec[c].begin(), rec[c].end(), (int16_t)0); }
  }
  void me_search(int, int n, const hop_pu_job* jobs, hop_pu_result* res) {
    parallel_batch(n, [&](int i) {
      const hop_pu_job& j = jobs[i]; hop_pu_result& r = res[i];
      if (getenv("HOP_SPINE_CHECK") && j.rng_right >= j.rng_left && j.rng_bottom >= j.rng_top) {   // the argument check of libhophip's hop_me_search, to see which jobs it would refuse
        const int lim = 80 + 64 - 4, stride = W + 160;
        bool bad = j.pu_y + j.rng_top - j.h / 2 - 4 < -lim || j.pu_y + j.rng_bottom + j.h + j.h / 2 + 8 > H + lim || j.pu_x + j.rng_left - j.w / 2 - 4 < -(stride - 8) || j.pu_x + j.rng_right + 2 * j.w + 8 > stride + W - 8;
        for (int k = 0; k < j.n_amvp; k++) { const int sx = (int)(int16_t)j.amvp[2 * k] >> 2, sy = (int)(int16_t)j.amvp[2 * k + 1] >> 2;
          bad = bad || j.pu_y + sy - j.h / 2 < -lim || j.pu_y + sy + j.h + j.h / 2 > H + lim || j.pu_x + sx - j.w / 2 - 4 < -(stride - 8) || j.pu_x + sx + 2 * j.w + 8 > stride + W - 8; }
        if (bad) fprintf(stderr, "refused: pu %d %d %dx%d rng %d..%d %d..%d amvp %d %d %d %d pred %d %d\n", j.pu_x, j.pu_y, j.w, j.h, j.rng_left, j.rng_right, j.rng_top, j.rng_bottom, j.amvp[0], j.amvp[1], j.amvp[2], j.amvp[3], j.pred_x, j.pred_y);
      }
      int64_t out[32]; memset(out, 0, sizeof(out));
      int amvp[4] = { j.amvp[0], j.amvp[1], j.amvp[2], j.amvp[3] };
      hop_o_me_pu(&org[0][(size_t)j.pu_y * W + j.pu_x], W, ss[0].p00, ss[0].stride, j.pu_x, j.pu_y, j.w, j.h, j.rng_left, j.rng_right, j.rng_top, j.rng_bottom, j.off_x, j.off_y,
                  j.pred_x, j.pred_y, j.n_amvp, amvp, j.lambda_cost, (j.flags & HOP_FLAG_FEN) ? 1 : 0, (j.flags & HOP_FLAG_HADME) ? 1 : 0, bd, 3, out);
      memset(&r, 0, sizeof(r));
      r.mv_int[0] = (int32_t)out[0]; r.mv_int[1] = (int32_t)out[1]; r.sad = (uint32_t)out[2]; r.not_valid = (int32_t)out[3];
      r.half[0] = (int32_t)out[4]; r.half[1] = (int32_t)out[5]; r.qter[0] = (int32_t)out[6]; r.qter[1] = (int32_t)out[7]; r.frac_cost = (uint32_t)out[8];
      r.gt_flag = (int32_t)out[9]; for (int k = 0; k < 8; k++) r.gt[k] = (int32_t)out[10 + k];
      r.cost = (uint32_t)out[18]; r.mv_final[0] = (int32_t)out[19]; r.mv_final[1] = (int32_t)out[20];
      r.half_final[0] = (int32_t)out[21]; r.half_final[1] = (int32_t)out[22]; r.qter_final[0] = (int32_t)out[23]; r.qter_final[1] = (int32_t)out[24];
    });
  }
  void inter_n(int n, const InterEval* const* e, const Coder* const* in, EvalResult* const* out) { parallel_batch(n, [&](int i) { inter_cu(0, *e[i], *in[i], *out[i]); }); }
  void intra_n(int n, const IntraEval* const* e, const Coder* const* in, EvalResult* const* out) { parallel_batch(n, [&](int i) { intra_cu(0, *e[i], *in[i], *out[i]); }); }
  void pred_inter(int, int n, const hop_pred_job* jobs) {
    for (int i = 0; i < n; i++) {
      const hop_pred_job& j = jobs[i];
      std::vector<int16_t> py((size_t)j.w * j.h), pb((size_t)j.w * j.h / 4), pr((size_t)j.w * j.h / 4);
      int gt[8]; for (int k = 0; k < 8; k++) gt[k] = j.gt[k];
      hop_o_pred_inter(ss[0].p00, ss[0].stride, ss[1].p00, ss[2].p00, ss[1].stride, j.pu_x, j.pu_y, j.w, j.h, j.mv_x, j.mv_y, j.use_gt, gt, bd, bd, &py[0], &pb[0], &pr[0]);
      const int dy = j.pu_y + j.dst_row_off;                                // the candidate slot's copy
      for (int r = 0; r < j.h; r++) memcpy(&pred[0][(size_t)(dy + r) * W + j.pu_x], &py[(size_t)r * j.w], j.w * 2);
      for (int r = 0; r < j.h / 2; r++) {
        memcpy(&pred[1][(size_t)(dy / 2 + r) * (W / 2) + j.pu_x / 2], &pb[(size_t)r * (j.w / 2)], j.w);
        memcpy(&pred[2][(size_t)(dy / 2 + r) * (W / 2) + j.pu_x / 2], &pr[(size_t)r * (j.w / 2)], j.w);
      }
    }
  }
  void distortion(int, int n, const hop_dist_job* jobs, uint32_t* out) {
    for (int i = 0; i < n; i++) {
      const hop_dist_job& j = jobs[i];
      const int s = j.comp ? 1 : 0, st = j.comp ? W / 2 : W;
      const int16_t* o = &org[j.comp][(size_t)(j.y >> s) * st + (j.x >> s)]; const int16_t* p = &pred[j.comp][(size_t)(j.y >> s) * st + (j.x >> s)];
      const int w = j.w >> s, h = j.h >> s;
      out[i] = j.kind == HOP_DIST_SAD ? hop_o_sad(o, st, p, st, w, h, bd, 0) : j.kind == HOP_DIST_SSE ? hop_o_sse(o, st, p, st, w, h, bd) : hop_o_hads(o, st, p, st, w, h, bd);
    }
  }
  void valid_pattern(int, int n, const int32_t* q, uint8_t* out) {          // TComRdCost::isValidPattern (TLibCommon/TComRdCost.cpp:430-443)
    for (int i = 0; i < n; i++, q += 6) {
      const int16_t* lb = ss[0].p00 + (ptrdiff_t)(q[1] + (q[5] >> 2) + q[3] + 4) * ss[0].stride + (q[0] + (q[4] >> 2));
      out[i] = (lb[0] != -1 && lb[q[2] + 4] != -1) ? 1 : 0;
    }
  }
  static void cfg_of(const hop_rqt_job& j, int bd, hop_o_rqt_cfg& c) {
    memset(&c, 0, sizeof(c));
    c.log2_cu = j.log2_cu; for (int k = 0; k < 3; k++) { c.qp[k] = j.qp_scaled[k]; c.lambda_rdoq[k] = j.lambda_rdoq[k]; }
    c.bit_depth_y = c.bit_depth_c = bd; c.sign_hide = j.sign_hide; c.use_ts = j.use_ts; c.log2_max_tu = j.log2_max_tu; c.log2_min_tu_in_cu = j.log2_min_tu_in_cu;
    c.inter_split_flag = j.inter_split_flag; c.lambda_rd = j.lambda_rd; c.dist_weight[0] = 1.0; c.dist_weight[1] = j.dist_weight[0]; c.dist_weight[2] = j.dist_weight[1];
  }
  static void coder_in(const Coder& k, hop_o_coder& c, uint8_t cu[20]) { memcpy(&c.ctx, k.r.state, 150); c.frac = coder_frac(k); memcpy(cu, k.c.state, 20); }
  static void coder_out(const hop_o_coder& c, const uint8_t cu[20], Coder& k) { memcpy(k.r.state, &c.ctx, 150); coder_set_frac(k, (uint32_t)(c.frac & 32767)); memcpy(k.c.state, cu, 20); }
  struct Layers {                                                            // the m_ppcQTTempCoeff / m_pcQTTempTComYuv layer buffers of one CU
    std::vector<int32_t> coef[4][3]; std::vector<int16_t> resi[4][3];
    Layers(int cu, hop_o_rqt_state& st) {
      memset(&st, 0, sizeof(st));
      for (int l = 0; l < 4; l++) for (int c = 0; c < 3; c++) {
        const size_t n = c ? (size_t)cu * cu / 4 : (size_t)cu * cu;
        coef[l][c].assign(n, 0); resi[l][c].assign(n, 0); st.coef[l][c] = &coef[l][c][0]; st.resi[l][c] = &resi[l][c][0];
      }
    }
  };
  void cu_planes(const std::vector<int16_t>* src, int x, int y, int cu, std::vector<int16_t> out[3]) {
    for (int c = 0; c < 3; c++) {
      const int s = c ? 1 : 0, w = cu >> s, st = c ? W / 2 : W;
      out[c].resize((size_t)w * w);
      for (int r = 0; r < w; r++) memcpy(&out[c][(size_t)r * w], &src[c][(size_t)((y >> s) + r) * st + (x >> s)], w * 2);
    }
  }
  static int zidx(int x4, int y4) { int z = 0; for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1)); return z; }
  int32_t* coef_ctu(int x, int y, int& abs_idx) {
    const int slot = y / H, yr = y % H, wctu = (W + 63) / 64, hctu = (H + 63) / 64;
    abs_idx = zidx((x & 63) >> 2, (yr & 63) >> 2);
    return &coefpic[((size_t)slot * wctu * hctu + (size_t)(yr >> 6) * wctu + (x >> 6)) * 6144];
  }
  void put_coef(const int32_t* coef /* Y | Cb | Cr of the CU, or NULL */, int x, int y, int cu) {
    int a; int32_t* ctu = coef_ctu(x, y, a); const int cu2 = cu * cu;
    for (int k = 0; k < cu2; k++) ctu[16 * a + k] = coef ? coef[k] : 0;
    for (int k = 0; k < cu2 / 4; k++) { ctu[4096 + 4 * a + k] = coef ? coef[cu2 + k] : 0; ctu[5120 + 4 * a + k] = coef ? coef[cu2 + cu2 / 4 + k] : 0; }
  }
  void put_rec(const std::vector<int16_t> in[3], int x, int y, int cu) {
    for (int c = 0; c < 3; c++) {
      const int s = c ? 1 : 0, w = cu >> s, st = c ? W / 2 : W;
      for (int r = 0; r < w; r++) memcpy(&rec[c][(size_t)((y >> s) + r) * st + (x >> s)], &in[c][(size_t)r * w], w * 2);
    }
  }
  void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& out) {
    if (e.n_pred > 0) pred_inter(lane, e.n_pred, e.pred);               // the CU's final motion compensation travels with its evaluation (EncConfig::fuse_pred)
    hop_o_rqt_cfg cfg; cfg_of(e.job, bd, cfg);
    const int cu = 1 << cfg.log2_cu, x = e.job.x, y = e.job.y, parts = (cu / 4) * (cu / 4);
    std::vector<int16_t> pr[3], og[3], rc[3];
    cu_planes(pred, x, y, cu, pr); cu_planes(org, x, y, cu, og);
    for (int c = 0; c < 3; c++) rc[c].assign(pr[c].size(), 0);
    const int16_t* pp[3] = { &pr[0][0], &pr[1][0], &pr[2][0] }; const int16_t* oo[3] = { &og[0][0], &og[1][0], &og[2][0] }; int16_t* rr[3] = { &rc[0][0], &rc[1][0], &rc[2][0] };
    hop_o_coder coder; uint8_t cuctx[20]; coder_in(in, coder, cuctx);
    memset(out.tr_idx, 0, sizeof(out.tr_idx)); memset(out.cbf, 0, sizeof(out.cbf)); memset(out.tskip, 0, sizeof(out.tskip));
    if (e.skip_res) {
      uint32_t d3[3]; double cost = 0;
      const uint32_t bits = hop_o_inter_cu_skip(&cfg, e.syn.skip_ctx, e.syn.pu[0].merge_idx, e.syn.max_merge_cand, pp, oo, &coder, cuctx, d3, &cost);
      out.bits = bits; out.dist = d3[0] + d3[1] + d3[2]; out.cost = cost; out.skipped = 1; out.root_cbf = 0;
      put_rec(pr, x, y, cu); put_coef(NULL, x, y, cu);
      coder_out(coder, cuctx, out.after);
      return;
    }
    std::vector<int16_t> rs[3];
    for (int c = 0; c < 3; c++) { rs[c].resize(pr[c].size()); for (size_t i = 0; i < pr[c].size(); i++) rs[c][i] = (int16_t)(og[c][i] - pr[c][i]); }
    hop_o_rqt_state st; Layers L(cu, st);
    double cost = 0; uint32_t bits = 0, dist = 0, zd = 0;
    hop_o_coder q = coder;
    hop_o_rqt(&cfg, &rs[0][0], cu, &rs[1][0], &rs[2][0], cu / 2, &q, &st, &cost, &bits, &dist, &zd);
    std::vector<int32_t> coef((size_t)cu * cu * 3 / 2, 0);
    uint32_t d3[3];
    const int root = hop_o_inter_cu_finish(&cfg, &st, &q, cost, zd, pp, oo, rr, d3, &coef[0]);
    hop_o_cu_syntax syn; memcpy(&syn, &e.syn, sizeof(syn));
    int skipped = 0;
    const uint32_t cbits = hop_o_inter_cu_bits(&cfg, &syn, &st, &coef[0], &coder, cuctx, &skipped);
    out.bits = cbits; out.dist = d3[0] + d3[1] + d3[2]; out.cost = hop_o_calc_rd_cost(cbits, out.dist, cfg.lambda_rd); out.skipped = skipped; out.root_cbf = root;
    memcpy(out.tr_idx, st.tr_idx, parts); for (int c = 0; c < 3; c++) { memcpy(out.cbf[c], st.cbf[c], parts); memcpy(out.tskip[c], st.tskip[c], parts); }
    put_rec(rc, x, y, cu); put_coef(&coef[0], x, y, cu);
    coder_out(coder, cuctx, out.after);
  }
  void intra_cu(int, const IntraEval& e, const Coder& in, EvalResult& out) {
    hop_o_rqt_cfg cfg; cfg_of(e.job, bd, cfg);
    const int cu = 1 << cfg.log2_cu, x = e.job.x, y = e.job.y, parts = (cu / 4) * (cu / 4), half = cu / 2;
    if (y >= H) {                                                          // a candidate slot: the row above the CU and the column to its left from the picture into the copy
      const int slot = y / H;
      for (int c = 0; c < 3; c++) {
        const int s = c ? 1 : 0, w = W >> s, h = H >> s, n = cu >> s, px = x >> s, py = (y % H) >> s;
        int16_t* pic = &rec[c][0]; int16_t* cpy = pic + (size_t)slot * h * w;
        if (py > 0) for (int k = 0; k <= 2 * n; k++) { const int xx = px - 1 + k; if (xx >= 0 && xx < w) cpy[(size_t)(py - 1) * w + xx] = pic[(size_t)(py - 1) * w + xx]; }
        if (px > 0) for (int k = 0; k < 2 * n; k++) { const int yy = py + k; if (yy < h) cpy[(size_t)yy * w + px - 1] = pic[(size_t)yy * w + px - 1]; }
      }
    }
    hop_o_intra_syntax syn; memset(&syn, 0, sizeof(syn));
    syn.part_nxn = e.syn.part_nxn; syn.skip_flag = e.syn.skip_flag; syn.skip_ctx = e.syn.skip_ctx; syn.is_min_cu = e.syn.is_min_cu;
    std::vector<uint8_t> avail((size_t)341 * HOP_O_AVAIL_PITCH, 0);
    for (int nidx = 0; nidx < 341; nidx++) for (int u = 0; u < 33; u++) avail[(size_t)nidx * HOP_O_AVAIL_PITCH + u] = (uint8_t)((e.opt.avail[nidx] >> u) & 1);
    hop_o_rqt_state st; Layers L(cu, st);
    hop_o_intra_rqt_in rin; memset(&rin, 0, sizeof(rin));
    rin.org = &org[0][(size_t)y * W + x]; rin.org_stride = W; rin.rec = &rec[0][(size_t)y * W + x]; rin.rec_stride = W;
    rin.avail = &avail[0]; rin.strong = e.opt.strong; rin.ts_fast = e.opt.ts_fast;
    hop_o_intra_search_in sin; memset(&sin, 0, sizeof(sin));
    uint8_t rough[4 * 68];
    for (int p = 0; p < 4; p++) { sin.left_dir[p] = e.sjob.left_dir[p]; sin.above_dir[p] = e.sjob.above_dir[p]; memcpy(rough + 68 * p, e.sjob.rough_flags[p], 68); }
    sin.rough_flags = rough; sin.sqrt_lambda = e.sjob.sqrt_lambda; sin.num_full_rd = e.sjob.num_full_rd;
    hop_o_coder coder; uint8_t cuctx[20]; coder_in(in, coder, cuctx);
    std::vector<int32_t> coef((size_t)cu * cu * 3 / 2, 0);
    std::vector<int16_t> rc[3]; rc[0].assign((size_t)cu * cu, 0); rc[1].assign((size_t)half * half, 0); rc[2].assign((size_t)half * half, 0);
    int best_dir[4] = { 0, 0, 0, 0 }, ncand[4]; uint32_t dist_y = 0;
    hop_o_intra_luma_search(&cfg, &syn, &rin, &sin, &coder, cuctx, &st, best_dir, &coef[0], &rc[0][0], &dist_y, ncand);
    // TEncCu.cpp:1476: the CU's luma reconstruction into the picture before the chroma search
    for (int r = 0; r < cu; r++) memcpy(&rec[0][(size_t)(y + r) * W + x], &rc[0][(size_t)r * cu], cu * 2);
    hop_intra_cu_syntax tmp = e.syn; intra_syntax_dirs(tmp, e.sjob, best_dir);
    for (int p = 0; p < 4; p++) { syn.luma_dir[p] = tmp.luma_dir[p]; syn.pred_num[p] = tmp.pred_num[p]; for (int k = 0; k < 3; k++) syn.preds[p][k] = tmp.preds[p][k]; }
    hop_o_intra_chroma_in cin; memset(&cin, 0, sizeof(cin));
    const size_t co = (size_t)(y / 2) * (W / 2) + x / 2;
    cin.org_cb = &org[1][co]; cin.org_cr = &org[2][co]; cin.org_stride = W / 2; cin.rec_cb = &rec[1][co]; cin.rec_cr = &rec[2][co]; cin.rec_stride = W / 2;
    cin.avail = &avail[0]; cin.ts_fast = e.opt.ts_fast;
    int best_mode = 0; uint32_t best_dist = 0;
    hop_o_intra_chroma_search(&cfg, &syn, &cin, &coder, cuctx, &st, &best_mode, &best_dist, &coef[(size_t)cu * cu], &coef[(size_t)cu * cu * 5 / 4], &rc[1][0], &rc[2][0]);
    syn.chroma_is_dm = best_mode == 36; syn.chroma_dir = best_mode;
    const uint32_t bits = hop_o_intra_cu_total_bits(&cfg, &syn, &st, &coef[0], &coder, cuctx);
    out.bits = bits; out.dist = dist_y + best_dist; out.cost = hop_o_calc_rd_cost(bits, out.dist, cfg.lambda_rd); out.skipped = 0; out.root_cbf = 1;
    memset(out.tr_idx, 0, sizeof(out.tr_idx)); memset(out.cbf, 0, sizeof(out.cbf)); memset(out.tskip, 0, sizeof(out.tskip));
    memcpy(out.tr_idx, st.tr_idx, parts); for (int c = 0; c < 3; c++) { memcpy(out.cbf[c], st.cbf[c], parts); memcpy(out.tskip[c], st.tskip[c], parts); }
    for (int p = 0; p < 4; p++) out.luma_dir[p] = best_dir[p];
    out.chroma_dir = best_mode;
    put_rec(rc, x, y, cu); put_coef(&coef[0], x, y, cu);
    coder_out(coder, cuctx, out.after);
  }
  void recon_save(int lane, int slot, int x, int y, int size) {
    std::vector<int16_t>* s = stash[lane * 16 + slot]; cu_planes(rec, x, y, size, s);
    int a; const int32_t* ctu = coef_ctu(x, y, a); std::vector<int32_t>& cs = coef_stash[lane * 16 + slot]; cs.resize((size_t)size * size * 3 / 2);
    memcpy(&cs[0], ctu + 16 * a, (size_t)size * size * 4); memcpy(&cs[(size_t)size * size], ctu + 4096 + 4 * a, (size_t)size * size); memcpy(&cs[(size_t)size * size * 5 / 4], ctu + 5120 + 4 * a, (size_t)size * size);
  }
  void recon_restore(int lane, int slot, int x, int y, int size) { put_rec(stash[lane * 16 + slot], x, y, size); put_coef(&coef_stash[lane * 16 + slot][0], x, y, size); }
  void commit(int, int x, int y, int size) {
    std::vector<int16_t> b[3]; cu_planes(rec, x, y, size, b);
    hop_o_ssref_commit_cu(ss[0].p00, ss[1].p00, ss[2].p00, W, H, x, y, size, &b[0][0], &b[1][0], &b[2][0]);
  }
  // a picture coded by several ranks: a CTU's block out of / into the reconstruction picture (and, coming in, into the SS reference: 8x8 commits cover any w x h)
  void export_block(int x, int y, int w, int h, int16_t* py, int16_t* pcb, int16_t* pcr) {
    for (int r = 0; r < h; r++) memcpy(py + (size_t)r * w, &rec[0][(size_t)(y + r) * W + x], (size_t)w * 2);
    for (int r = 0; r < h / 2; r++) { memcpy(pcb + (size_t)r * (w / 2), &rec[1][(size_t)(y / 2 + r) * (W / 2) + x / 2], (size_t)w); memcpy(pcr + (size_t)r * (w / 2), &rec[2][(size_t)(y / 2 + r) * (W / 2) + x / 2], (size_t)w); }
  }
  void import_block(int x, int y, int w, int h, const int16_t* py, const int16_t* pcb, const int16_t* pcr) {
    for (int r = 0; r < h; r++) memcpy(&rec[0][(size_t)(y + r) * W + x], py + (size_t)r * w, (size_t)w * 2);
    for (int r = 0; r < h / 2; r++) { memcpy(&rec[1][(size_t)(y / 2 + r) * (W / 2) + x / 2], pcb + (size_t)r * (w / 2), (size_t)w); memcpy(&rec[2][(size_t)(y / 2 + r) * (W / 2) + x / 2], pcr + (size_t)r * (w / 2), (size_t)w); }
    for (int yy = y; yy < y + h; yy += 8) for (int xx = x; xx < x + w; xx += 8) commit(0, xx, yy, 8);
  }
  int W, H, bd;
  std::vector<int32_t> coefpic;                                            // the levels, per slot and CTU 6144 TCoeff in the reference's layout (hop_levels_download)
  std::map<int, std::vector<int32_t> > coef_stash;
  std::vector<int16_t> org[3], pred[3], rec[3];
  Plane ss[3];
  std::map<int, std::vector<int16_t>[3]> stash;
};

}  // namespace

// HOP_SPEC_SLOTS=<n> (tests): the SS/GT candidates of a CU side by side in n candidate slots (EncConfig::spec_slots), as hop_ctx_set_slots + hop_encode_frame do on the device
static std::vector<int32_t> g_last_levels;   // the levels of the last picture(s) coded through an entry below (slot 0: the decided CUs), for hop_spine_cpu_last_levels
static std::vector<uint16_t> g_last_fraction;   // ... and the RD coder's carried fraction per CTU, for hop_spine_cpu_last_rd_fraction
static void keep_fraction(const Encoder& e, bool append = false) { if (!append) g_last_fraction.clear(); g_last_fraction.insert(g_last_fraction.end(), e.ctu_rd_fraction.begin(), e.ctu_rd_fraction.end()); }
static void keep_levels(const CpuBackend& be, bool append = false) { const size_t n = (size_t)((be.W + 63) / 64) * ((be.H + 63) / 64) * 6144; if (!append) g_last_levels.clear(); g_last_levels.insert(g_last_levels.end(), be.coefpic.begin(), be.coefpic.begin() + n); }
static int spec_slots_env() { const char* e = getenv("HOP_SPEC_SLOTS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 128 ? v : 0; }

extern "C" {

// One frame through the spine on the CPU restatement.  y / cb / cr: the original (pitch w, w/2).  Outputs (any may be NULL): ctu_cost / ctu_bits / ctu_dist per CTU,
// parts = the finished picture's per-4x4 data (sizeof(hopspine::Part) per unit, 256 per CTU, z-order), rec_* the reconstruction before the loop filters, entry = the coder
// every CTU started from (sizeof(hopspine::Coder) each).  trace_path: one line per candidate that reaches xCheckBestMode.  Returns the number of such candidates.
long hop_spine_cpu_encode_wpp(int w, int h, int qp, int mi_size, int lag, const int16_t* y, const int16_t* cb, const int16_t* cr, const char* trace_path,
                              double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr, double* rounds_requests);
long hop_spine_cpu_encode(int w, int h, int qp, int mi_size, int first_ctus, const int16_t* y, const int16_t* cb, const int16_t* cr, const char* trace_path,
                          double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr, void* entry) {
  EncConfig cfg; default_hop_config(cfg, w, h, qp, mi_size);
  const int slots = spec_slots_env(); cfg.spec_slots = slots; cfg.slot_pitch = h;
  CpuBackend be(w, h, 8, y, cb, cr, slots);
  Encoder enc(cfg, &be);
  if (trace_path && *trace_path) enc.trace = fopen(trace_path, "w");
  enc.encode_frame(first_ctus);
  if (enc.trace) fclose(enc.trace);
  const int n = enc.n_ctu();
  keep_levels(be); keep_fraction(enc);
  if (ctu_cost) memcpy(ctu_cost, &enc.ctu_cost[0], n * sizeof(double));
  if (ctu_bits) memcpy(ctu_bits, &enc.ctu_bits[0], n * 4);
  if (ctu_dist) memcpy(ctu_dist, &enc.ctu_dist[0], n * 4);
  if (parts) memcpy(parts, &enc.pic[0], enc.pic.size() * sizeof(Part));
  if (entry) memcpy(entry, &enc.ctu_entry[0], n * sizeof(Coder));
  if (rec_y) memcpy(rec_y, &be.rec[0][0], (size_t)w * h * 2);
  if (rec_cb) memcpy(rec_cb, &be.rec[1][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rec_cr) memcpy(rec_cr, &be.rec[2][0], (size_t)(w / 2) * (h / 2) * 2);
  return (long)enc.n_candidates;
}
// the same with WaveFrontSynchro semantics (one substream per CTU row): lag 0 = the CTUs in raster order on one thread, lag > 0 = the rows as a wavefront of threads whose
// requests are served in batches (what the product does on the GPU)
// the progress counter and the cancel request of the wavefront (what hop_encode_progress / hop_encode_cancel reach in the product): the next hop_spine_cpu_encode_wpp call
// is cancelled by a watcher thread as soon as `n` CTUs have been retired (n <= 0: never); hop_spine_cpu_last_progress gives the count the call ended with
static std::atomic<long> g_progress(0); static std::atomic<int> g_cancel(0); static int g_cancel_after = 0;
void hop_spine_cpu_cancel_after(int n) { g_cancel_after = n; }
long hop_spine_cpu_last_progress(void) { return g_progress.load(); }
long hop_spine_cpu_encode_wpp(int w, int h, int qp, int mi_size, int lag, const int16_t* y, const int16_t* cb, const int16_t* cr, const char* trace_path,
                              double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr, double* rounds_requests) {
  EncConfig cfg; default_hop_config(cfg, w, h, qp, mi_size); cfg.wpp = 1;
  const int slots = spec_slots_env(); cfg.spec_slots = slots; cfg.slot_pitch = h;
  g_progress.store(0); g_cancel.store(0); cfg.progress = &g_progress; cfg.cancel = &g_cancel;
  const int cancel_after = g_cancel_after; g_cancel_after = 0;
  std::atomic<int> watch_stop(0);
  std::thread watcher([&]() { while (!watch_stop.load()) { if (cancel_after > 0 && g_progress.load() >= cancel_after) g_cancel.store(1); std::this_thread::sleep_for(std::chrono::microseconds(200)); } });
  struct Join { std::thread& t; std::atomic<int>& s; ~Join() { s.store(1); t.join(); } } join{ watcher, watch_stop };
  CpuBackend be(w, h, 8, y, cb, cr, slots);
  LogBackend* lg = getenv("HOP_SPINE_LOG") ? new LogBackend(&be, getenv("HOP_SPINE_LOG")) : NULL;
  BatchInner* use = lg ? (BatchInner*)lg : (BatchInner*)&be;
  Encoder enc(cfg, use);
  if (trace_path && *trace_path) enc.trace = fopen(trace_path, "w");
  try { if (lag > 0) enc.encode_frame_wavefront(use, lag); else enc.encode_frame(0); delete lg; } catch (...) { if (enc.trace) fclose(enc.trace); return -1; }
  if (enc.trace) fclose(enc.trace);
  const int n = enc.n_ctu();
  keep_levels(be); keep_fraction(enc);
  if (ctu_cost) memcpy(ctu_cost, &enc.ctu_cost[0], n * sizeof(double));
  if (ctu_bits) memcpy(ctu_bits, &enc.ctu_bits[0], n * 4);
  if (ctu_dist) memcpy(ctu_dist, &enc.ctu_dist[0], n * 4);
  if (parts) memcpy(parts, &enc.pic[0], enc.pic.size() * sizeof(Part));
  if (rec_y) memcpy(rec_y, &be.rec[0][0], (size_t)w * h * 2);
  if (rec_cb) memcpy(rec_cb, &be.rec[1][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rec_cr) memcpy(rec_cr, &be.rec[2][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rounds_requests) { rounds_requests[0] = (double)enc.batch_rounds; rounds_requests[1] = (double)enc.batch_requests; }
  return (long)enc.n_candidates;
}
// ---- one picture's CTU rows dealt to several ranks (EncConfig::shard) ----
// (a) `world` ranks as threads of this process, each with a backend (a "device") of its own, exchanging through an in-process all-gather: every rank must end with the whole
//     picture, equal to hop_spine_cpu_encode_wpp's; out: rank `take`'s view (costs, partition data, reconstruction)
namespace {
struct LocalComm : public ShardComm {
  LocalComm(int world) : world(world), arrived(0), gen(0) {}
  struct Port : public ShardComm { LocalComm* hub; int rank; void allgather(const void* s, void* r, size_t b) { hub->gather(rank, s, r, b); } };
  void gather(int rank, const void* send, void* recv, size_t bytes) {
    std::unique_lock<std::mutex> lk(m);
    if (buf.size() < (size_t)world * bytes) buf.resize((size_t)world * bytes);
    memcpy(&buf[(size_t)rank * bytes], send, bytes);
    const unsigned long my = gen;
    if (++arrived == world) { arrived = 0; ready = buf; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != my; });
    memcpy(recv, &ready[0], (size_t)world * bytes);
  }
  void allgather(const void*, void*, size_t) {}
  int world, arrived; unsigned long gen; std::mutex m; std::condition_variable cv; std::vector<char> buf, ready;
};
}
long hop_spine_cpu_encode_sharded(int w, int h, int qp, int mi_size, int lag, int world, int take, int cancel_after, const int16_t* y, const int16_t* cb, const int16_t* cr,
                                  double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr, long* retired_per_rank) {
  LocalComm hub(world);
  std::vector<LocalComm::Port> ports(world);
  std::vector<CpuBackend*> bes; std::vector<Encoder*> encs;
  std::vector<std::atomic<long> > prog(world); std::vector<std::atomic<int> > canc(world);
  const int slots = spec_slots_env();
  for (int g = 0; g < world; g++) {
    ports[g].hub = &hub; ports[g].rank = g; prog[g].store(0); canc[g].store(0);
    EncConfig cfg; default_hop_config(cfg, w, h, qp, mi_size); cfg.wpp = 1; cfg.spec_slots = slots; cfg.slot_pitch = h;
    cfg.shard_rank = g; cfg.shard_world = world; cfg.shard = &ports[g]; cfg.progress = &prog[g]; cfg.cancel = &canc[g];
    bes.push_back(new CpuBackend(w, h, 8, y, cb, cr, slots)); encs.push_back(new Encoder(cfg, bes[g]));
  }
  std::atomic<int> stop(0);
  std::thread watcher([&]() { while (!stop.load()) { if (cancel_after > 0 && prog[0].load() >= cancel_after) canc[0].store(1); std::this_thread::sleep_for(std::chrono::microseconds(200)); } });
  std::vector<std::thread> th; std::vector<int> ok(world, 1);
  for (int g = 0; g < world; g++) th.emplace_back([&, g]() { try { encs[g]->encode_frame_wavefront(bes[g], lag); } catch (...) { ok[g] = 0; } });
  for (auto& t : th) t.join();
  stop.store(1); watcher.join();
  long rc = 0;
  for (int g = 0; g < world; g++) { if (!ok[g]) rc = -1; if (retired_per_rank) retired_per_rank[g] = prog[g].load(); }
  Encoder& e = *encs[take]; CpuBackend& b = *bes[take];
  const int n = e.n_ctu();
  if (ctu_cost) memcpy(ctu_cost, &e.ctu_cost[0], n * sizeof(double));
  if (ctu_bits) memcpy(ctu_bits, &e.ctu_bits[0], n * 4);
  if (ctu_dist) memcpy(ctu_dist, &e.ctu_dist[0], n * 4);
  if (parts) memcpy(parts, &e.pic[0], e.pic.size() * sizeof(Part));
  if (rec_y) memcpy(rec_y, &b.rec[0][0], (size_t)w * h * 2);
  if (rec_cb) memcpy(rec_cb, &b.rec[1][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rec_cr) memcpy(rec_cr, &b.rec[2][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rc == 0) for (int g = 0; g < world; g++) rc += (long)encs[g]->n_candidates;
  for (auto e2 : encs) delete e2;
  for (auto b2 : bes) delete b2;
  return rc;
}
// (b) one rank of a real job (tests/test_multi_rank.py: two processes on gloo): the all-gather is the caller's
typedef int (*hop_cpu_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
namespace { struct CallbackComm : public ShardComm { hop_cpu_allgather_fn fn; void* user; void allgather(const void* s, void* r, size_t b) { if (fn(user, s, r, b) != 0) throw 1; } }; }
long hop_spine_cpu_encode_shard_rank(int w, int h, int qp, int mi_size, int lag, int rank, int world, hop_cpu_allgather_fn fn, void* user, const int16_t* y, const int16_t* cb,
                                     const int16_t* cr, double* ctu_cost, void* parts, int16_t* rec_y) {
  CallbackComm comm; comm.fn = fn; comm.user = user;
  EncConfig cfg; default_hop_config(cfg, w, h, qp, mi_size); cfg.wpp = 1;
  cfg.shard_rank = rank; cfg.shard_world = world; cfg.shard = &comm;
  CpuBackend be(w, h, 8, y, cb, cr, 0);
  Encoder enc(cfg, &be);
  try { enc.encode_frame_wavefront(&be, lag); } catch (...) { return -1; }
  const int n = enc.n_ctu();
  if (ctu_cost) memcpy(ctu_cost, &enc.ctu_cost[0], n * sizeof(double));
  if (parts) memcpy(parts, &enc.pic[0], enc.pic.size() * sizeof(Part));
  if (rec_y) memcpy(rec_y, &be.rec[0][0], (size_t)w * h * 2);
  return (long)enc.n_candidates;
}

// A stack of independent pictures (what hop_ctx_set_stack makes of a context): every request goes to its own picture's backend, its coordinates back in that
// picture's own frame.  The default n-forms of BatchInner loop over the single forms below, so a batch that mixes pictures is split here.
class StackRouter : public BatchInner {
 public:
  StackRouter(std::vector<CpuBackend*>& b, int pitch) : be(b), pitch_(pitch) {}
  void begin_frame() { for (auto b : be) b->begin_frame(); }
  void me_search(int lane, int n, const hop_pu_job* jobs, hop_pu_result* res) { for (int i = 0; i < n; i++) { hop_pu_job j = jobs[i]; const int k = j.pu_y / pitch_; j.pu_y -= k * pitch_; be[k]->me_search(lane, 1, &j, res + i); } }
  void pred_inter(int lane, int n, const hop_pred_job* jobs) { for (int i = 0; i < n; i++) { hop_pred_job j = jobs[i]; const int k = j.pu_y / pitch_; j.pu_y -= k * pitch_; be[k]->pred_inter(lane, 1, &j); } }
  void distortion(int lane, int n, const hop_dist_job* jobs, uint32_t* out) { for (int i = 0; i < n; i++) { hop_dist_job j = jobs[i]; const int k = j.y / pitch_; j.y -= k * pitch_; be[k]->distortion(lane, 1, &j, out + i); } }
  void valid_pattern(int, int, const int32_t*, uint8_t*) { throw 1; }   // the spine answers these from its own map
  void inter_cu(int lane, const InterEval& e, const Coder& in, EvalResult& out) { InterEval q = e; const int k = q.job.y / pitch_; q.job.y -= k * pitch_; for (int i = 0; i < q.n_pred; i++) q.pred[i].pu_y -= k * pitch_; be[k]->inter_cu(lane, q, in, out); }
  void intra_cu(int lane, const IntraEval& e, const Coder& in, EvalResult& out) { IntraEval q = e; const int k = q.job.y / pitch_; q.job.y -= k * pitch_; be[k]->intra_cu(lane, q, in, out); }
  void recon_save(int lane, int slot, int x, int y, int size) { const int k = y / pitch_; be[k]->recon_save(lane, slot, x, y - k * pitch_, size); }
  void recon_restore(int lane, int slot, int x, int y, int size) { const int k = y / pitch_; be[k]->recon_restore(lane, slot, x, y - k * pitch_, size); }
  void commit(int lane, int x, int y, int size) { const int k = y / pitch_; be[k]->commit(lane, x, y - k * pitch_, size); }
 private:
  std::vector<CpuBackend*>& be; int pitch_;
};
// n_pic pictures (planes back to back: picture k's luma at y + k * w * h, ...) coded side by side as a wavefront of lag `lag`, picture k's requests carrying y + k * pitch;
// outputs per picture back to back.  Each picture's result must be what hop_spine_cpu_encode_wpp gives for it alone.
long hop_spine_cpu_encode_stack(int w, int h, int n_pic, int pitch, int qp, int mi_size, int lag, const int16_t* y, const int16_t* cb, const int16_t* cr,
                                double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, double* rounds_requests) {
  std::vector<CpuBackend*> bes; std::vector<Encoder*> encs;
  for (int k = 0; k < n_pic; k++) bes.push_back(new CpuBackend(w, h, 8, y + (size_t)k * w * h, cb + (size_t)k * (w / 2) * (h / 2), cr + (size_t)k * (w / 2) * (h / 2)));
  StackRouter router(bes, pitch);
  for (int k = 0; k < n_pic; k++) { EncConfig cfg; default_hop_config(cfg, w, h, qp, mi_size); cfg.wpp = 1; cfg.y_origin = k * pitch; encs.push_back(new Encoder(cfg, &router)); }
  long total = 0;
  try { Encoder::encode_pictures_wavefront(&encs[0], n_pic, &router, lag); } catch (...) { total = -1; }
  const int n = encs[0]->n_ctu();
  for (int k = 0; k < n_pic && total >= 0; k++) {
    Encoder& e = *encs[k];
    if (ctu_cost) memcpy(ctu_cost + (size_t)k * n, &e.ctu_cost[0], n * sizeof(double));
    if (ctu_bits) memcpy(ctu_bits + (size_t)k * n, &e.ctu_bits[0], n * 4);
    if (ctu_dist) memcpy(ctu_dist + (size_t)k * n, &e.ctu_dist[0], n * 4);
    if (parts) memcpy((char*)parts + (size_t)k * n * 256 * sizeof(Part), &e.pic[0], e.pic.size() * sizeof(Part));
    if (rec_y) memcpy(rec_y + (size_t)k * w * h, &bes[k]->rec[0][0], (size_t)w * h * 2);
    keep_levels(*bes[k], k > 0); keep_fraction(e, k > 0);
    total += (long)e.n_candidates;
  }
  if (rounds_requests) { rounds_requests[0] = (double)encs[0]->batch_rounds; rounds_requests[1] = (double)encs[0]->batch_requests; }
  for (auto e : encs) delete e;
  for (auto b : bes) delete b;
  return total;
}
// the plain intra configurations (cfg/encoder_intra_main.cfg, encoder_intra_main10.cfg): I slice, bit depth 8 or 10 (samples already at that depth), CTUs in raster order
long hop_spine_cpu_encode_plain(int w, int h, int qp, int bit_depth, const int16_t* y, const int16_t* cb, const int16_t* cr, const char* trace_path,
                                double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, void* parts, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr) {
  EncConfig cfg; default_plain_config(cfg, w, h, qp, bit_depth);
  CpuBackend be(w, h, bit_depth, y, cb, cr);
  Encoder enc(cfg, &be);
  if (trace_path && *trace_path) enc.trace = fopen(trace_path, "w");
  enc.encode_frame(0);
  if (enc.trace) fclose(enc.trace);
  const int n = enc.n_ctu();
  keep_levels(be); keep_fraction(enc);
  if (ctu_cost) memcpy(ctu_cost, &enc.ctu_cost[0], n * sizeof(double));
  if (ctu_bits) memcpy(ctu_bits, &enc.ctu_bits[0], n * 4);
  if (ctu_dist) memcpy(ctu_dist, &enc.ctu_dist[0], n * 4);
  if (parts) memcpy(parts, &enc.pic[0], enc.pic.size() * sizeof(Part));
  if (rec_y) memcpy(rec_y, &be.rec[0][0], (size_t)w * h * 2);
  if (rec_cb) memcpy(rec_cb, &be.rec[1][0], (size_t)(w / 2) * (h / 2) * 2);
  if (rec_cr) memcpy(rec_cr, &be.rec[2][0], (size_t)(w / 2) * (h / 2) * 2);
  return (long)enc.n_candidates;
}
// the levels of the picture(s) the last entry coded: per CTU 6144 TCoeff in the reference's layout, as hop_levels_download hands them out; returns the count
long hop_spine_cpu_last_levels(int32_t* out, long max_n) { const long n = (long)g_last_levels.size(); if (out && n <= max_n) memcpy(out, &g_last_levels[0], (size_t)n * 4); return n; }
long hop_spine_cpu_last_rd_fraction(uint16_t* out, long max_n) { const long n = (long)g_last_fraction.size(); if (out && n <= max_n) memcpy(out, &g_last_fraction[0], (size_t)n * 2); return n; }
int hop_spine_sizeof_part(void) { return (int)sizeof(Part); }
int hop_spine_sizeof_coder(void) { return (int)sizeof(Coder); }

}
