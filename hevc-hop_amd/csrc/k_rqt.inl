// k_rqt.inl -- residual quadtree search of SS/GT ("inter") CUs, SURVEY 8(a) row a8b (included by k_cabac.hip: it shares the counting coder).
// Replaces TEncSearch::xEstimateResidualQT (TLibEncoder/TEncSearch.cpp:6824-7560) with xEncodeResidualQT (:7562-7655) for a batch of CUs.
//
// Inside one CU the search is a chain: every node starts from the coder state its predecessor left (m_pcRDGoOnSbacCoder), the RDOQ
// tables and every bit count depend on it, and a parent recounts its whole subtree.  The parallelism is across CUs (an RD search tests
// hundreds of CU candidates per CTU): the host walks the quadtree of the CU class in the reference's order, and every step is a batch
// over all CUs of the class:
//   k_rqt_begin    store the entry state (CI_QT_TRAFO_ROOT), emit the node's Y/Cb/Cr transform units as leaf jobs (+ their 4x4
//                  transform-skip variants)
//   hop_launch_tu_rd  the leaf pipeline of k_tq.hip: residual, xT, estBit, RDOQ, bits from the entry state, inverse path, cbf-zero decision
//   k_rqt_single   transform-skip decisions (:7210-7437), the node coded as one TU from the entry state (:7439-7466), its cost; a leaf
//                  adds itself to its parent's sums, an inner node parks the state (CI_QT_TRAFO_TEST) and rewinds for its children
//   k_rqt_close    cbf of the children folded upwards, the subtree recounted in syntax order from the entry state (:7503-7511), the
//                  split decision (:7514-7522)
// One lane per CU in the k_rqt_* kernels; context states of a lane in LDS as in k_coeff_bits.  Coefficients of every evaluated node stay in
// per-CU layer buffers (layer = log2 max TU - log2 size, TU of 4x4 partition p at 16 p, chroma at (16 p) >> 2: the reference's
// m_ppcQTTempCoeff layout) until the root has decided; k_rqt_final gathers the chosen ones.

struct RqtClass { int log2_cu, log2_max_tu, log2_min_tu, inter_split, sign_hide, use_ts; };
struct RqtNode { int part, d, log2, code_chroma, check_full, check_split, add_zero, ts_y, ts_c; };
struct RqtWork {                                                      // per CU: what the reference keeps in locals of the recursion, by transform depth
  double   single_cost[4]; uint32_t single_bits[4], single_dist[4], abs_sum[4][3];
  double   sub_cost[5]; uint32_t sub_bits[5], sub_dist[5];            // what the children of depth d have added up (index 0: the CU's result)
  uint32_t zero_dist; uint8_t best_skip[4][4];
};

// The kernels that walk a CU's coder serially take one LANE per CU (64 CUs per wave: batches of thousands of CUs) or, launched with one workgroup per CU (grid = n), one
// WAVE per CU with lane 0 working: the lanes of a wave serialise wherever their CUs' paths differ, so a small batch -- the RD spine's, one CU per CTU in flight -- runs in
// the time of ONE CU only when every CU has a wave of its own.  rqt_grid() picks the launch; the kernel tells the two apart by its grid.
#define RQT_LANE_OR_BLOCK(n_) const bool spread_ = (n_) > 1 && gridDim.x == (unsigned)(n_); if (spread_ && threadIdx.x) return; \
                              const int lane = spread_ ? 0 : (int)threadIdx.x, i = spread_ ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x)
// (a wave per CU needs one column of context states: CabacLds1, 204 bytes -- a compute unit then holds 32 such CUs instead of 12)
#define RQT_LAUNCH(K, c_, n_, ...) do { if ((n_) > 1 && rqt_grid(c_, n_) == (n_)) hipLaunchKernelGGL((K<CabacLds1>), dim3(n_), dim3(64), 0, (c_)->stream, __VA_ARGS__); \
                                        else hipLaunchKernelGGL((K<CabacLds>), dim3(rqt_grid(c_, n_)), dim3(64), 0, (c_)->stream, __VA_ARGS__); } while (0)
static inline int rqt_grid(const hop_ctx* c, int n) { return (n > 1 && n <= c->fused_leaf_max) ? n : (n + 63) / 64; }
__device__ static inline int rqt_zx(int p) { int x = 0; for (int b = 0; b < 4; b++) x |= ((p >> (2 * b)) & 1) << b; return 4 * x; }
__device__ static inline int rqt_zy(int p) { int y = 0; for (int b = 0; b < 4; b++) y |= ((p >> (2 * b + 1)) & 1) << b; return 4 * y; }
__device__ static inline double rqt_cost(uint32_t bits, uint32_t dist, double lambda) { return (double)(uint32_t)floor((double)dist + (double)((int)(bits * lambda + .5))); }
__device__ static inline size_t rqt_coef_base(const RqtClass& k, int i) { return (size_t)i * (size_t)(6 << (2 * k.log2_cu)); }
__device__ static inline size_t rqt_coef_at(const RqtClass& k, int i, int layer, int comp, int part) {
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  return rqt_coef_base(k, i) + (size_t)layer * (cu2 + (cu2 >> 1)) + (comp == 0 ? 0 : comp == 1 ? cu2 : cu2 + (cu2 >> 2)) + (comp ? (size_t)((16 * part) >> 2) : (size_t)(16 * part));
}

// (the kernels of this file are thin wrappers around *_body functions: the candidate walks of k_walk.inl call the same bodies from ONE kernel per candidate)
__device__ static inline void rqt_init_body(const int i, const hop_rqt_job* jobs, const hop_cabac_ctx* ctx_in, hop_cabac_ctx* cur, RqtWork* work, hop_rqt_result* res) {
  cur[i] = ctx_in[jobs[i].ctx_index];
  RqtWork w; memset(&w, 0, sizeof(w));
  work[i] = w;
  hop_rqt_result* r = res + i;
  r->cost = 0; r->bits = r->dist = r->zero_dist = r->pad = 0;
  for (int p = 0; p < 256; p++) { r->tr_idx[p] = 0; for (int c = 0; c < 3; c++) { r->cbf[c][p] = 0; r->tskip[c][p] = 0; } }
}
__global__ void k_rqt_init(const hop_rqt_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, hop_cabac_ctx* __restrict__ cur,
                           RqtWork* __restrict__ work, hop_rqt_result* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rqt_init_body(i, jobs, ctx_in, cur, work, res);
}

// entry of a node: CI_QT_TRAFO_ROOT <- the coder; the node's transform units as leaf jobs
__device__ static inline void rqt_begin_body(const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, int bd_y, int bd_c, const hop_cabac_ctx* cur,
                                             hop_cabac_ctx* root, hop_rqt_result* res, RqtWork* work, hop_tu_rd_job* tuj, int64_t* off, hop_tu_rd_job* tuj2, int64_t* off2, size_t ts_base) {
  root[i] = cur[i];
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d);
  if (!nd.check_full) { work[i].single_cost[nd.d] = 1.7e+308; work[i].sub_dist[nd.d + 1] = 0; work[i].sub_cost[nd.d + 1] = 0; work[i].sub_bits[nd.d + 1] = 0; return; }
  const hop_rqt_job jb = jobs[i];
  hop_rqt_result* r = res + i;
  const int dC = nd.log2 == 2 ? nd.d - 1 : nd.d, npartsC = parts >> (2 * dC), log2C = nd.log2 == 2 ? 2 : nd.log2 - 1;
  for (int p = 0; p < nparts; p++) { r->tr_idx[nd.part + p] = (uint8_t)nd.d; r->tskip[0][nd.part + p] = 0; }
  if (nd.code_chroma) for (int p = 0; p < npartsC; p++) { r->tskip[1][nd.part + p] = 0; r->tskip[2][nd.part + p] = 0; }
  const int ncomp = nd.code_chroma ? 3 : 1, layer = k.log2_max_tu - nd.log2;
  const int nts = nd.ts_y + 2 * nd.ts_c;
  int t = 0;
  for (int c = 0; c < ncomp; c++) {
    hop_tu_rd_job j;
    j.x = jb.x + rqt_zx(nd.part); j.y = jb.y + rqt_zy(nd.part); j.comp = c; j.log2_size = c ? log2C : nd.log2; j.qp_scaled = jb.qp_scaled[c]; j.tr_depth = nd.d;
    j.ctx_index = i; j.sign_hide = k.sign_hide; j.use_ts = k.use_ts; j.bit_depth = c ? bd_c : bd_y; j.is_intra = 0; j.scan_idx = 0; j.use_dst = 0; j.flags = 0;
    j.lambda_rdoq = jb.lambda_rdoq[c]; j.lambda_rd = jb.lambda_rd; j.dist_weight = c ? jb.dist_weight[c - 1] : 1.0;
    tuj[ncomp * i + c] = j;
    off[ncomp * i + c] = (int64_t)rqt_coef_at(k, i, layer, c, nd.part);
    if (c == 0 ? nd.ts_y : nd.ts_c) {
      j.flags = HOP_TU_RD_TS | HOP_TU_RD_KEEP;
      tuj2[nts * i + t] = j;
      off2[nts * i + t] = (int64_t)(ts_base + (size_t)i * 48 + (size_t)c * 16);
      t++;
    }
  }
}
__global__ void k_rqt_begin(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, int n, int bd_y, int bd_c, const hop_cabac_ctx* __restrict__ cur,
                            hop_cabac_ctx* __restrict__ root, hop_rqt_result* __restrict__ res, RqtWork* __restrict__ work, hop_tu_rd_job* __restrict__ tuj,
                            int64_t* __restrict__ off, hop_tu_rd_job* __restrict__ tuj2, int64_t* __restrict__ off2, size_t ts_base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rqt_begin_body(i, k, nd, jobs, bd_y, bd_c, cur, root, res, work, tuj, off, tuj2, off2, ts_base);
}

#define RQ_LOAD(src) do { const uint8_t* s_ = (src).state; for (int q_ = 0; q_ < 152; q_++) sh.st[q_][lane] = s_[q_]; } while (0)
#define RQ_LEFT() ((unsigned)sh.st[150][lane] | ((unsigned)sh.st[151][lane] << 8))
template <class LDS>
__device__ static inline void rqt_store(LDS& sh, int lane, unsigned long long frac_total, hop_cabac_ctx* dst) {
  for (int q = 0; q < 150; q++) dst->state[q] = sh.st[q][lane];
  const unsigned left = (unsigned)(frac_total & 32767ull);
  dst->state[150] = (uint8_t)(left & 0xFF); dst->state[151] = (uint8_t)(left >> 8);
}
__device__ static inline int rqt_cbf_ctx(int comp, int d) { return CX_QT_CBF + (comp ? 4 + d : (d == 0 ? 1 : 0)); }     // getCtxQtCbf, TComDataCU.cpp:1848-1859

template <class LDS>
__device__ static void rqt_single_body(LDS& sh, const int lane, const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, hop_cabac_ctx* cur,
                                       const hop_cabac_ctx* root, hop_cabac_ctx* test, hop_rqt_result* res, RqtWork* work, const hop_tu_rd_result* tr, const hop_tu_rd_result* tr2,
                                       int32_t* coef, size_t ts_base, const uint16_t* scans) {
  const hop_rqt_job jb = jobs[i];
  hop_rqt_result* r = res + i; RqtWork* w = work + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d), d = nd.d;
  const int dC = nd.log2 == 2 ? d - 1 : d, npartsC = parts >> (2 * dC), log2C = nd.log2 == 2 ? 2 : nd.log2 - 1;
  const int ncomp = nd.code_chroma ? 3 : 1, layer = k.log2_max_tu - nd.log2, nts = nd.ts_y + 2 * nd.ts_c;
  uint32_t absSum[3] = { 0, 0, 0 }, distC[3] = { 0, 0, 0 }; double minCost[3] = { 0, 0, 0 }; int skip[3] = { 0, 0, 0 };
  for (int c = 0; c < ncomp; c++) {
    const hop_tu_rd_result t = tr[ncomp * i + c];
    absSum[c] = t.abs_sum; distC[c] = t.dist; minCost[c] = t.cost;
    if (nd.add_zero) w->zero_dist += t.zero_dist;
  }
  // transform-skip retry: the variant replaces the block unless it has no level or the best cost so far is smaller (:7258, :7389, :7421)
  int t2 = 0;
  for (int c = 0; c < ncomp; c++) {
    if (!(c == 0 ? nd.ts_y : nd.ts_c)) continue;
    const hop_tu_rd_result t = tr2[nts * i + t2]; t2++;
    if (t.abs_sum && !(minCost[c] < t.cost)) {
      int32_t* dst = coef + rqt_coef_at(k, i, layer, c, nd.part); const int32_t* src = coef + ts_base + (size_t)i * 48 + (size_t)c * 16;
      for (int q = 0; q < 16; q++) dst[q] = src[q];
      distC[c] = t.nonzero_dist; absSum[c] = t.abs_sum; skip[c] = 1;
    }
  }
  const uint8_t setCbf = (uint8_t)(1 << d);
  for (int p = 0; p < nparts; p++) { r->cbf[0][nd.part + p] = absSum[0] ? setCbf : 0; r->tskip[0][nd.part + p] = (uint8_t)skip[0]; }
  if (nd.code_chroma) for (int p = 0; p < npartsC; p++) {
    r->cbf[1][nd.part + p] = absSum[1] ? setCbf : 0; r->cbf[2][nd.part + p] = absSum[2] ? setCbf : 0;
    r->tskip[1][nd.part + p] = (uint8_t)skip[1]; r->tskip[2][nd.part + p] = (uint8_t)skip[2];
  }
  // the node as one transform unit, from the entry state (:7439-7466)
  RQ_LOAD(root[i]);
  unsigned long long frac = RQ_LEFT();                                // resetBits keeps the fraction
  if (nd.log2 > k.log2_min_tu) CBIN(CX_TRANS_SUBDIV + 5 - nd.log2, 0);
  if (nd.code_chroma) { CBIN(rqt_cbf_ctx(1, d), absSum[1] ? 1 : 0); CBIN(rqt_cbf_ctx(2, d), absSum[2] ? 1 : 0); }
  CBIN(rqt_cbf_ctx(0, d), absSum[0] ? 1 : 0);
  for (int c = 0; c < ncomp; c++)
    frac += cb_code_tu(sh, lane, coef + rqt_coef_at(k, i, layer, c, nd.part), c ? log2C : nd.log2, c != 0, 0, k.sign_hide, k.use_ts, skip[c], 0, scans);
  const uint32_t singleBits = (uint32_t)(frac >> 15), singleDist = distC[0] + distC[1] + distC[2];
  const double singleCost = rqt_cost(singleBits, singleDist, jb.lambda_rd);
  rqt_store(sh, lane, frac, cur + i);
  if (nd.check_split) {
    test[i] = cur[i]; cur[i] = root[i];
    w->single_cost[d] = singleCost; w->single_bits[d] = singleBits; w->single_dist[d] = singleDist;
    for (int c = 0; c < 3; c++) { w->abs_sum[d][c] = absSum[c]; w->best_skip[d][c] = (uint8_t)skip[c]; }
    w->sub_dist[d + 1] = 0; w->sub_cost[d + 1] = 0; w->sub_bits[d + 1] = 0;
  } else {
    w->sub_cost[d] += singleCost; w->sub_bits[d] += singleBits; w->sub_dist[d] += singleDist;
  }
}
template <class LDS>
__global__ __launch_bounds__(64) void k_rqt_single(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, int n, hop_cabac_ctx* __restrict__ cur,
                                                   const hop_cabac_ctx* __restrict__ root, hop_cabac_ctx* __restrict__ test, hop_rqt_result* __restrict__ res,
                                                   RqtWork* __restrict__ work, const hop_tu_rd_result* __restrict__ tr, const hop_tu_rd_result* __restrict__ tr2,
                                                   int32_t* __restrict__ coef, size_t ts_base, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;                                                // no barrier in this kernel
  rqt_single_body(sh, lane, i, k, nd, jobs, cur, root, test, res, work, tr, tr2, coef, ts_base, scans);
}

// xEncodeResidualQT on the lane's coder: the subtree below (part, d0) as the arrays describe it; flags (subdiv_and_cbf) or one component's levels
template <class LDS>
__device__ static unsigned long long rqt_encode_tree(LDS& sh, const int lane, const RqtClass& k, const int i, const hop_rqt_result* r, const int32_t* coef,
                                                     const int part0, const int d0, const int log2_0, const int subdiv_and_cbf, const int comp, const uint16_t* scans) {
  unsigned long long frac = 0;
  const int parts = 1 << (2 * (k.log2_cu - 2));
  int sp_part[4], sp_k[4]; int sp = 0;
  sp_part[0] = part0; sp_k[0] = -1;
  while (sp >= 0) {
    const int part = sp_part[sp], cd = d0 + sp, log2 = log2_0 - sp;
    if (sp_k[sp] < 0) {
      const int trMode = r->tr_idx[part], subdiv = cd != trMode;
      if (subdiv_and_cbf && log2 <= k.log2_max_tu && log2 > k.log2_min_tu) CBIN(CX_TRANS_SUBDIV + 5 - log2, subdiv);
      if (subdiv_and_cbf) {
        const int first = cd == 0;
        if (first || log2 > 2) {
          if (first || ((r->cbf[1][part] >> (cd - 1)) & 1)) CBIN(rqt_cbf_ctx(1, cd), (r->cbf[1][part] >> cd) & 1);
          if (first || ((r->cbf[2][part] >> (cd - 1)) & 1)) CBIN(rqt_cbf_ctx(2, cd), (r->cbf[2][part] >> cd) & 1);
        }
      }
      if (!subdiv) {
        const int layer = k.log2_max_tu - log2;
        int codeChroma = 1, log2C = log2 - 1;
        if (log2 == 2) { log2C = 2; codeChroma = (part % (parts >> (2 * (trMode - 1)))) == 0; }
        if (subdiv_and_cbf) CBIN(rqt_cbf_ctx(0, trMode), (r->cbf[0][part] >> trMode) & 1);
        else {
          if (comp == 0 && ((r->cbf[0][part] >> trMode) & 1))
            frac += cb_code_tu(sh, lane, coef + rqt_coef_at(k, i, layer, 0, part), log2, 0, 0, k.sign_hide, k.use_ts, r->tskip[0][part], 0, scans);
          if (codeChroma && comp && ((r->cbf[comp][part] >> trMode) & 1))
            frac += cb_code_tu(sh, lane, coef + rqt_coef_at(k, i, layer, comp, part), log2C, 1, 0, k.sign_hide, k.use_ts, r->tskip[comp][part], 0, scans);
        }
        sp--; continue;
      }
      if (!(subdiv_and_cbf || ((r->cbf[comp][part] >> cd) & 1))) { sp--; continue; }
      sp_k[sp] = 0;
    }
    if (sp_k[sp] < 4) {
      const int q = (parts >> (2 * cd)) >> 2, kk = sp_k[sp]++;
      sp_part[sp + 1] = part + kk * q; sp_k[sp + 1] = -1; sp++;
    } else sp--;
  }
  return frac;
}

template <class LDS>
__device__ static void rqt_close_body(LDS& sh, const int lane, const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, hop_cabac_ctx* cur, const hop_cabac_ctx* root,
                                      const hop_cabac_ctx* test, hop_rqt_result* res, RqtWork* work, const int32_t* coef, const uint16_t* scans) {
  const hop_rqt_job jb = jobs[i];
  hop_rqt_result* r = res + i; RqtWork* w = work + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), d = nd.d, nparts = parts >> (2 * d), q = nparts >> 2;
  int any[3] = { 0, 0, 0 };
  for (int c = 0; c < 3; c++) for (int kk = 0; kk < 4; kk++) any[c] |= (r->cbf[c][nd.part + kk * q] >> (d + 1)) & 1;
  for (int c = 0; c < 3; c++) for (int p = 0; p < nparts; p++) r->cbf[c][nd.part + p] |= (uint8_t)(any[c] << d);
  RQ_LOAD(root[i]);
  unsigned long long frac = RQ_LEFT();
  frac += rqt_encode_tree(sh, lane, k, i, r, coef, nd.part, d, nd.log2, 1, 0, scans);
  frac += rqt_encode_tree(sh, lane, k, i, r, coef, nd.part, d, nd.log2, 0, 0, scans);
  frac += rqt_encode_tree(sh, lane, k, i, r, coef, nd.part, d, nd.log2, 0, 1, scans);
  frac += rqt_encode_tree(sh, lane, k, i, r, coef, nd.part, d, nd.log2, 0, 2, scans);
  const uint32_t subBits = (uint32_t)(frac >> 15), subDist = w->sub_dist[d + 1];
  const double subCost = rqt_cost(subBits, subDist, jb.lambda_rd);
  if ((any[0] || any[1] || any[2] || !nd.check_full) && subCost < w->single_cost[d]) {
    w->sub_cost[d] += subCost; w->sub_bits[d] += subBits; w->sub_dist[d] += subDist;
    rqt_store(sh, lane, frac, cur + i);
    return;
  }
  const int dC = nd.log2 == 2 ? d - 1 : d, npartsC = parts >> (2 * dC);
  const uint8_t setCbf = (uint8_t)(1 << d);
  for (int p = 0; p < nparts; p++) { r->tskip[0][nd.part + p] = w->best_skip[d][0]; r->tr_idx[nd.part + p] = (uint8_t)d; r->cbf[0][nd.part + p] = w->abs_sum[d][0] ? setCbf : 0; }
  if (nd.code_chroma) for (int p = 0; p < npartsC; p++) {
    r->tskip[1][nd.part + p] = w->best_skip[d][1]; r->tskip[2][nd.part + p] = w->best_skip[d][2];
    r->cbf[1][nd.part + p] = w->abs_sum[d][1] ? setCbf : 0; r->cbf[2][nd.part + p] = w->abs_sum[d][2] ? setCbf : 0;
  }
  cur[i] = test[i];
  w->sub_cost[d] += w->single_cost[d]; w->sub_bits[d] += w->single_bits[d]; w->sub_dist[d] += w->single_dist[d];
}
template <class LDS>
__global__ __launch_bounds__(64) void k_rqt_close(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, int n, hop_cabac_ctx* __restrict__ cur,
                                                  const hop_cabac_ctx* __restrict__ root, const hop_cabac_ctx* __restrict__ test, hop_rqt_result* __restrict__ res,
                                                  RqtWork* __restrict__ work, const int32_t* __restrict__ coef, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  rqt_close_body(sh, lane, i, k, nd, jobs, cur, root, test, res, work, coef, scans);
}

// results of the root + the chosen transform units' levels in the CU's coefficient layout (what xSetResidualQTData copies, :7658-7777)
__device__ static inline void rqt_final_body(const int i, const int tid, const int nt, const RqtClass& k, const RqtWork* work, hop_rqt_result* res, const int32_t* coef, int32_t* out,
                                             const hop_cabac_ctx* cur, hop_cabac_ctx* ctx_out) {
  hop_rqt_result* r = res + i;
  if (tid == 0) { r->cost = work[i].sub_cost[0]; r->bits = work[i].sub_bits[0]; r->dist = work[i].sub_dist[0]; r->zero_dist = work[i].zero_dist; if (ctx_out) ctx_out[i] = cur[i]; }
  if (!out) return;
  const int parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  int32_t* o = out + (size_t)i * (cu2 + (cu2 >> 1));
  // a TU's block is contiguous from its first partition on (16 luma / 4 + 4 chroma coefficients per partition), so copying partition by
  // partition from the layer of the partition's transform depth copies every chosen block whole
  for (int t = tid; t < parts * 16; t += nt) {
    const int p = t >> 4, q = t & 15, layer = k.log2_max_tu - (k.log2_cu - r->tr_idx[p]);
    o[t] = coef[rqt_coef_at(k, i, layer, 0, p) + q];
    if (q < 4) {
      o[cu2 + 4 * p + q] = coef[rqt_coef_at(k, i, layer, 1, p) + q];
      o[cu2 + (cu2 >> 2) + 4 * p + q] = coef[rqt_coef_at(k, i, layer, 2, p) + q];
    }
  }
}
__global__ __launch_bounds__(64) void k_rqt_final(RqtClass k, int n, const RqtWork* __restrict__ work, hop_rqt_result* __restrict__ res, const int32_t* __restrict__ coef,
                                                  int32_t* __restrict__ out, const hop_cabac_ctx* __restrict__ cur, hop_cabac_ctx* __restrict__ ctx_out) {
  if ((int)blockIdx.x < n) rqt_final_body(blockIdx.x, threadIdx.x, 64, k, work, res, coef, out, cur, ctx_out);
}

// ---- host orchestration ----
static int rqt_run_class(hop_ctx* c, const RqtClass& k, int n, const hop_rqt_job* d_jobs, const hop_cabac_ctx* d_ctx_in, hop_rqt_result* d_res, int32_t* d_coef_out,
                         hop_cabac_ctx* d_ctx_out, char* buf, size_t buf_bytes) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu), n_coeff = (size_t)n * (6 * cu2 + 48), ts_base = (size_t)n * 6 * cu2;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  hop_cabac_ctx* cur = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx));
  hop_cabac_ctx* root[4]; hop_cabac_ctx* test[4];
  for (int d = 0; d < 4; d++) { root[d] = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); test[d] = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); }
  RqtWork* work = (RqtWork*)take((size_t)n * sizeof(RqtWork));
  hop_tu_rd_job* tuj = (hop_tu_rd_job*)take((size_t)3 * n * sizeof(hop_tu_rd_job)); hop_tu_rd_job* tuj2 = (hop_tu_rd_job*)take((size_t)3 * n * sizeof(hop_tu_rd_job));
  int64_t* off = (int64_t*)take((size_t)3 * n * 8); int64_t* off2 = (int64_t*)take((size_t)3 * n * 8);
  hop_tu_rd_result* tr = (hop_tu_rd_result*)take((size_t)3 * n * sizeof(hop_tu_rd_result)); hop_tu_rd_result* tr2 = (hop_tu_rd_result*)take((size_t)3 * n * sizeof(hop_tu_rd_result));
  int32_t* coef = (int32_t*)take(n_coeff * 4);
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "rqt: work buffer too small");
  const int g64 = (n + 63) / 64, g256 = (n + 255) / 256;
  hipLaunchKernelGGL(k_rqt_init, dim3(g256), dim3(256), 0, c->stream, d_jobs, n, d_ctx_in, cur, work, d_res);
  const int parts = 1 << (2 * (k.log2_cu - 2));
  int rc = HOP_OK;
  // the reference's recursion (:6824-7560), one batch step per node
  struct Rec { static int go(hop_ctx* c, const RqtClass& k, int n, const hop_rqt_job* d_jobs, hop_rqt_result* d_res, hop_cabac_ctx* cur, hop_cabac_ctx** root, hop_cabac_ctx** test,
                             RqtWork* work, hop_tu_rd_job* tuj, hop_tu_rd_job* tuj2, int64_t* off, int64_t* off2, hop_tu_rd_result* tr, hop_tu_rd_result* tr2, int32_t* coef,
                             size_t n_coeff, size_t ts_base, int parts, int part, int d, int log2, int zero_open) {
    RqtNode nd; nd.part = part; nd.d = d; nd.log2 = log2;
    nd.check_full = (k.inter_split && d == 0 && log2 > k.log2_min_tu) ? 0 : (log2 <= k.log2_max_tu);
    nd.check_split = log2 > k.log2_min_tu;
    nd.code_chroma = 1;
    if (log2 == 2) nd.code_chroma = (part % (parts >> (2 * (d - 1)))) == 0;
    nd.add_zero = zero_open && nd.check_full;
    nd.ts_y = (k.use_ts && nd.check_full && log2 == 2) ? 1 : 0;
    nd.ts_c = (k.use_ts && nd.check_full && nd.code_chroma && (log2 == 2 || log2 == 3)) ? 1 : 0;
    const int g256 = (n + 255) / 256, ncomp = nd.code_chroma ? 3 : 1, nts = nd.ts_y + 2 * nd.ts_c;
    hipLaunchKernelGGL(k_rqt_begin, dim3(g256), dim3(256), 0, c->stream, k, nd, d_jobs, n, c->bd_y, c->bd_c, cur, root[d], d_res, work, tuj, off, tuj2, off2, ts_base);
    if (nd.check_full) {
      const int hint = log2 <= 3 ? 1 : (log2 == 5 ? 2 : 0);          // 16x16 luma comes with 8x8 chroma
      int r = hop_launch_tu_rd(c, n * ncomp, tuj, root[d], off, n_coeff, coef, tr, hint); if (r) return r;
      if (nts) { r = hop_launch_tu_rd(c, n * nts, tuj2, root[d], off2, n_coeff, coef, tr2, 1); if (r) return r; }
      RQT_LAUNCH(k_rqt_single, c, n, k, nd, d_jobs, n, cur, root[d], test[d], d_res, work, tr, tr2, coef, ts_base, c->rdoq_scans);
    }
    if (nd.check_split) {
      const int q = (parts >> (2 * d)) >> 2;
      for (int kk = 0; kk < 4; kk++) {
        const int r = go(c, k, n, d_jobs, d_res, cur, root, test, work, tuj, tuj2, off, off2, tr, tr2, coef, n_coeff, ts_base, parts, part + kk * q, d + 1, log2 - 1, zero_open && !nd.check_full);
        if (r) return r;
      }
      RQT_LAUNCH(k_rqt_close, c, n, k, nd, d_jobs, n, cur, root[d], test[d], d_res, work, coef, c->rdoq_scans);
    }
    return HOP_OK;
  } };
  rc = Rec::go(c, k, n, d_jobs, d_res, cur, root, test, work, tuj, tuj2, off, off2, tr, tr2, coef, n_coeff, ts_base, parts, 0, 0, k.log2_cu, 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_rqt_final, dim3(n), dim3(64), 0, c->stream, k, n, work, d_res, coef, d_coef_out, cur, d_ctx_out);
  (void)g64;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "rqt launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

size_t hop_rqt_work_bytes(int log2_cu, int n) {
  const size_t cu2 = (size_t)1 << (2 * log2_cu);
  return (size_t)n * (9 * sizeof(hop_cabac_ctx) + sizeof(RqtWork) + 6 * (sizeof(hop_tu_rd_job) + 8 + sizeof(hop_tu_rd_result)) + (6 * cu2 + 48) * 4) + 64 * 256;
}

// one homogeneous class of CUs (same size and transform-tree limits); d_* device pointers, buf = hop_rqt_work_bytes
int hop_launch_rqt_class(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int inter_split, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                         const hop_cabac_ctx* d_ctx_in, hop_rqt_result* d_res, int32_t* d_coef_out, hop_cabac_ctx* d_ctx_out, void* buf, size_t buf_bytes) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = inter_split; k.sign_hide = sign_hide; k.use_ts = use_ts;
  return rqt_run_class(c, k, n, d_jobs, d_ctx_in, d_res, d_coef_out, d_ctx_out, (char*)buf, buf_bytes);
}

// =====================================================================================================================
// The tail of TEncSearch::encodeResAndCalcRdInterCU around the quadtree (TLibEncoder/TEncSearch.cpp:6700-6723, :6804-6812), without the
// CU-level syntax bits in between (xAddSymbolBitsInter stays with the caller): the root-cbf-zero test on the coder as the quadtree left it,
// the arrays and levels cleared if the zero residual wins, the reconstruction Clip(prediction + residual of the chosen transform units)
// into the context's reconstruction picture and its distortion against the original.
//   k_fin_decide   one lane per CU: bits of a zero rqt_root_cbf, calcRdCost against the quadtree's cost
//   k_fin_emit     one lane per CU: a table of transform-unit jobs, one slot per node of the CU's tree and component; the slots of the chosen
//                  nodes are live (size, cbf, transform skip), the others empty
//   k_turd_inverse(_small) in reconstruction mode: xDeQuant + xIT / xITransformSkip, Clip(prediction + residual), SSE against the original
//   k_fin_sum      one lane per CU: the three distortions (getDistPart weights the chroma planes once per plane)
// =====================================================================================================================
__device__ static inline void fin_decide_body(const int i, const RqtClass& k, const hop_rqt_job* jobs, const hop_cabac_ctx* after, const int32_t* entropy_bits, hop_rqt_result* res, hop_cu_final* fin) {
  const uint8_t* s = after[i].state;
  const unsigned left = (unsigned)s[150] | ((unsigned)s[151] << 8);
  const uint32_t zeroBits = (uint32_t)((left + (unsigned long long)entropy_bits[s[CX_ROOT_CBF] ^ 0]) >> 15);     // encodeQtRootCbfZero
  hop_rqt_result* r = res + i;
  const double zeroCost = rqt_cost(zeroBits, r->zero_dist, jobs[i].lambda_rd);
  const int root = !(zeroCost < r->cost);
  if (!root) {
    const int parts = 1 << (2 * (k.log2_cu - 2));
    for (int p = 0; p < parts; p++) { r->tr_idx[p] = 0; for (int c = 0; c < 3; c++) { r->cbf[c][p] = 0; r->tskip[c][p] = 0; } }
  }
  fin[i].root_cbf = (uint32_t)root;
}
__global__ void k_fin_decide(RqtClass k, const hop_rqt_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ after, const int32_t* __restrict__ entropy_bits,
                             hop_rqt_result* __restrict__ res, hop_cu_final* __restrict__ fin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fin_decide_body(i, k, jobs, after, entropy_bits, res, fin);
}

__device__ static inline void fin_zero_body(const int i, const int tid, const int nt, const RqtClass& k, const hop_cu_final* fin, int32_t* coef) {
  if (fin[i].root_cbf) return;
  const size_t m = ((size_t)3 << (2 * k.log2_cu)) / 2;
  for (size_t q = tid; q < m; q += nt) coef[(size_t)i * m + q] = 0;
}
__global__ __launch_bounds__(256) void k_fin_zero(RqtClass k, int n, const hop_cu_final* __restrict__ fin, int32_t* __restrict__ coef) {
  if ((int)blockIdx.x < n) fin_zero_body(blockIdx.x, threadIdx.x, 256, k, fin, coef);
}

__device__ static inline void fin_emit_body(const int i, const RqtClass& k, const hop_rqt_job* jobs, int bd_y, int bd_c, int d0, int d1, int nj, const hop_rqt_result* res,
                                            const hop_cu_final* fin, hop_tu_rd_job* tuj, int64_t* off, uint32_t* abs_flag) {
  const hop_rqt_job jb = jobs[i];
  const hop_rqt_result* r = res + i;
  const int root = (int)fin[i].root_cbf, parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu), base = (size_t)i * (cu2 + (cu2 >> 1));
  int q = 0;
  for (int d = d0; d <= d1; d++) {
    const int log2 = k.log2_cu - d, np = parts >> (2 * d);
    for (int node = 0; node < (1 << (2 * d)); node++) {
      const int part = node * np;
      const bool chosen = root ? (r->tr_idx[part] == d) : (d == d0);
      for (int c = 0; c < 3; c++, q++) {
        hop_tu_rd_job j; memset(&j, 0, sizeof(j));
        const bool have = chosen && (c == 0 || log2 > 2 || (part & 3) == 0);      // a 4x4 luma TU shares the chroma block of its 8x8 parent (first sibling)
        j.log2_size = have ? (c ? (log2 == 2 ? 2 : log2 - 1) : log2) : 0;
        j.x = jb.x + rqt_zx(part); j.y = jb.y + rqt_zy(part); j.comp = c; j.qp_scaled = jb.qp_scaled[c]; j.bit_depth = c ? bd_c : bd_y; j.is_intra = 1;
        j.flags = (root && r->tskip[c][part]) ? HOP_TU_RD_TS : 0; j.use_ts = k.use_ts;
        tuj[(size_t)i * nj + q] = j;
        off[(size_t)i * nj + q] = (int64_t)(base + (c == 0 ? (size_t)(16 * part) : cu2 + (size_t)(c - 1) * (cu2 >> 2) + (size_t)(4 * part)));
        abs_flag[(size_t)i * nj + q] = (root && have) ? (uint32_t)((r->cbf[c][part] >> d) & 1) : 0u;
      }
    }
  }
}
__global__ void k_fin_emit(RqtClass k, const hop_rqt_job* __restrict__ jobs, int n, int bd_y, int bd_c, int d0, int d1, int nj, const hop_rqt_result* __restrict__ res,
                           const hop_cu_final* __restrict__ fin, hop_tu_rd_job* __restrict__ tuj, int64_t* __restrict__ off, uint32_t* __restrict__ abs_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fin_emit_body(i, k, jobs, bd_y, bd_c, d0, d1, nj, res, fin, tuj, off, abs_flag);
}

__device__ static inline void fin_sum_body(const int i, const hop_rqt_job* jobs, int nj, const hop_tu_rd_job* tuj, const uint32_t* sse, hop_cu_final* fin) {
  unsigned long long s3[3] = { 0, 0, 0 };
  for (int q = 0; q < nj; q++) if (tuj[(size_t)i * nj + q].log2_size >= 2) s3[q % 3] += sse[(size_t)i * nj + q];
  const hop_rqt_job jb = jobs[i];
  fin[i].dist[0] = (uint32_t)s3[0];
  fin[i].dist[1] = (uint32_t)(int)(jb.dist_weight[0] * (uint32_t)s3[1]);           // getDistPart, TComRdCost.cpp:493-497
  fin[i].dist[2] = (uint32_t)(int)(jb.dist_weight[1] * (uint32_t)s3[2]);
}
__global__ void k_fin_sum(const hop_rqt_job* __restrict__ jobs, int n, int nj, const hop_tu_rd_job* __restrict__ tuj, const uint32_t* __restrict__ sse, hop_cu_final* __restrict__ fin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fin_sum_body(i, jobs, nj, tuj, sse, fin);
}

static int rqt_jobs_per_cu(int log2_cu, int log2_max_tu, int log2_min_tu, int* d0, int* d1) {
  *d0 = log2_cu > log2_max_tu ? log2_cu - log2_max_tu : 0; *d1 = log2_cu - log2_min_tu;
  int nodes = 0; for (int d = *d0; d <= *d1; d++) nodes += 1 << (2 * d);
  return 3 * nodes;
}
size_t hop_rqt_finish_work_bytes(int log2_cu, int log2_max_tu, int log2_min_tu, int n) {
  int d0, d1; const int nj = rqt_jobs_per_cu(log2_cu, log2_max_tu, log2_min_tu, &d0, &d1);
  return (size_t)n * nj * (sizeof(hop_tu_rd_job) + 8 + 4 + 4) + 4 * 256;
}
// one class of CUs; d_res / d_coef are updated in place (cleared where the zero residual wins), the reconstruction goes to the context's picture
int hop_launch_rqt_finish(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int use_ts, int n, const hop_rqt_job* d_jobs, hop_rqt_result* d_res, int32_t* d_coef,
                          const hop_cabac_ctx* d_after, hop_cu_final* d_fin, void* buf) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = 0; k.use_ts = use_ts;
  int d0, d1; const int nj = rqt_jobs_per_cu(log2_cu, log2_max_tu, log2_min_tu, &d0, &d1);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t nt = (size_t)n * nj;
  char* b = (char*)buf; size_t o = 0;
  hop_tu_rd_job* tuj = (hop_tu_rd_job*)(b + o); o = al(o + nt * sizeof(hop_tu_rd_job));
  int64_t* off = (int64_t*)(b + o); o = al(o + nt * 8);
  uint32_t* af = (uint32_t*)(b + o); o = al(o + nt * 4);
  uint32_t* sse = (uint32_t*)(b + o);
  const int g256 = (n + 255) / 256;
  hipLaunchKernelGGL(k_fin_decide, dim3(g256), dim3(256), 0, c->stream, k, d_jobs, n, d_after, hop_entropy_bits_device(c), d_res, d_fin);
  hipLaunchKernelGGL(k_fin_zero, dim3(n), dim3(256), 0, c->stream, k, n, d_fin, d_coef);
  hipLaunchKernelGGL(k_fin_emit, dim3(g256), dim3(256), 0, c->stream, k, d_jobs, n, c->bd_y, c->bd_c, d0, d1, nj, d_res, d_fin, tuj, off, af);
  int r = hop_launch_tu_recon(c, (int)nt, tuj, off, d_coef, af, sse); if (r) return r;
  hipLaunchKernelGGL(k_fin_sum, dim3(g256), dim3(256), 0, c->stream, d_jobs, n, nj, tuj, sse, d_fin);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "rqt finish launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The CU-level syntax of an SS/GT CU, counted: TEncSearch::xAddSymbolBitsInter (TLibEncoder/TEncSearch.cpp:7779-7810) = skip flag and either the
// merge index (skipped CU) or prediction mode, partition size (TEncSbac.cpp:469-566), per PU merge flag / index or MVD (:944-1048), MVP index
// (:434-467), GT flag (:654-677), GT corner vectors (codeGT :1051-1330: corners 0..2, coded like MVDs on their own two contexts), root cbf and the
// transform tree in bitstream order (TEncEntropy::encodeCoeff :633-660, xEncodeTransform :219-394).  One lane per CU; rows 152.. of the lane's LDS
// states hold the CU-level sets (hop_cabac_cu_ctx: skip[3], merge_flag, merge_idx, part_size[4], pred_mode, mvd[2], mvp_idx, gt_flag, gt[2]).
// =====================================================================================================================
#define CUX 152
#define CU_SKIP (CUX + 0)
#define CU_MERGE_FLAG (CUX + 3)
#define CU_MERGE_IDX (CUX + 4)
#define CU_PART (CUX + 5)
#define CU_PRED (CUX + 9)
#define CU_MVD (CUX + 10)
#define CU_MVP (CUX + 12)
#define CU_GTF (CUX + 13)
#define CU_GT (CUX + 14)
#define CEP(nbins) do { frac += 32768ull * (unsigned long long)(nbins); } while (0)

__device__ static inline int cu_eg_bins(unsigned sym, int kk) { int nb = 0; while (sym >= (1u << kk)) { nb++; sym -= 1u << kk; kk++; } return nb + 1 + kk; }   // xWriteEpExGolomb
template <class LDS>
__device__ static unsigned long long cu_vec(LDS& sh, const int lane, const int base, const int32_t* v, const int ncomp) {    // codeMvd / codeGT
  unsigned long long frac = 0;
  for (int i = 0; i < ncomp; i++) CBIN(base, v[i] != 0);
  for (int i = 0; i < ncomp; i++) if (v[i]) CBIN(base + 1, (v[i] < 0 ? -v[i] : v[i]) > 1);
  for (int i = 0; i < ncomp; i++) if (v[i]) { const int a = v[i] < 0 ? -v[i] : v[i]; if (a > 1) CEP(cu_eg_bins((unsigned)a - 2, 1)); CEP(1); }
  return frac;
}
template <class LDS>
__device__ static unsigned long long cu_merge_index(LDS& sh, const int lane, const int idx, const int num) {
  unsigned long long frac = 0;
  if (num <= 1) return 0;
  for (int ui = 0; ui < num - 1; ui++) { const int sym = ui == idx ? 0 : 1; if (ui == 0) CBIN(CU_MERGE_IDX, sym); else CEP(1); if (!sym) break; }
  return frac;
}

template <class LDS>
__device__ static void cu_bits_body(LDS& sh, const int lane, const int i, const RqtClass& k, const hop_rqt_job* jobs, const hop_cu_syntax* syn, const hop_rqt_result* res, const int32_t* coef,
                                    const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_in, uint32_t* bits_out, uint32_t* skipped_out, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_out,
                                    const uint16_t* scans) {
  const int ci = jobs[i].ctx_index;
  RQ_LOAD(ctx_in[ci]);
  for (int q = 0; q < 20; q++) sh.st[CUX + q][lane] = cu_in[ci].state[q];
  const hop_cu_syntax y = syn[i];
  const hop_rqt_result* r = res + i;
  const int parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  const int32_t* cf = coef + (size_t)i * (cu2 + (cu2 >> 1));
  const int root = ((r->cbf[0][0] | r->cbf[1][0] | r->cbf[2][0]) & 1);
  unsigned long long frac = RQ_LEFT();                               // resetBits keeps the fraction
  uint32_t skipped;
  if (y.pu[0].merge_flag && y.part_size == 0 && !root) {
    skipped = 1;
    CBIN(CU_SKIP + y.skip_ctx, 1);
    frac += cu_merge_index(sh, lane, y.pu[0].merge_idx, y.max_merge_cand);
  } else {
    skipped = (uint32_t)(y.skip_flag != 0);
    CBIN(CU_SKIP + y.skip_ctx, y.skip_flag ? 1 : 0);
    CBIN(CU_PRED, 0);                                                // MODE_INTER
    const int e = y.part_size;                                       // codePartSize
    if (e == 0) CBIN(CU_PART, 1);
    else if (e == 1 || e == 4 || e == 5) {
      CBIN(CU_PART, 0); CBIN(CU_PART + 1, 1);
      if (y.amp_acc) { if (e == 1) CBIN(CU_PART + 3, 1); else { CBIN(CU_PART + 3, 0); CEP(1); } }
    } else if (e == 2 || e == 6 || e == 7) {
      CBIN(CU_PART, 0); CBIN(CU_PART + 1, 0);
      if (y.is_min_cu && k.log2_cu != 3) CBIN(CU_PART + 2, 1);
      if (y.amp_acc) { if (e == 2) CBIN(CU_PART + 3, 1); else { CBIN(CU_PART + 3, 0); CEP(1); } }
    } else if (y.is_min_cu && k.log2_cu != 3) { CBIN(CU_PART, 0); CBIN(CU_PART + 1, 0); CBIN(CU_PART + 2, 0); }
    for (int p = 0; p < y.n_pu; p++) {                               // encodePUWise
      CBIN(CU_MERGE_FLAG, y.pu[p].merge_flag ? 1 : 0);
      if (y.pu[p].merge_flag) { frac += cu_merge_index(sh, lane, y.pu[p].merge_idx, y.max_merge_cand); continue; }
      frac += cu_vec(sh, lane, CU_MVD, y.pu[p].mvd, 2);
      CBIN(CU_MVP, y.pu[p].mvp_idx ? 1 : 0);
      CBIN(CU_GTF, y.pu[p].gt_flag ? 1 : 0);
      if (y.pu[p].gt_flag) frac += cu_vec(sh, lane, CU_GT, y.pu[p].gt, 6);
    }
    if (!(y.pu[0].merge_flag && y.part_size == 0)) CBIN(CX_ROOT_CBF, root);
    if (root) {
      // xEncodeTransform: pre-order walk; flags on the way down, luma cbf + levels at the leaves (chroma of four 4x4 luma TUs after the last of them)
      int sp_part[4], sp_k[4]; int sp = 0, bak = 0;
      sp_part[0] = 0; sp_k[0] = -1;
      while (sp >= 0) {
        const int part = sp_part[sp], trIdx = sp, log2 = k.log2_cu - sp;
        if (sp_k[sp] < 0) {
          const int subdiv = r->tr_idx[part] > trIdx;
          const int cbfY = (r->cbf[0][part] >> trIdx) & 1; int cbfU = (r->cbf[1][part] >> trIdx) & 1, cbfV = (r->cbf[2][part] >> trIdx) & 1;
          if (log2 == 2) {
            const int pn = parts >> (2 * (trIdx - 1));
            if (part % pn == 0) bak = part;
            else if (part % pn == pn - 1) { cbfU = (r->cbf[1][bak] >> trIdx) & 1; cbfV = (r->cbf[2][bak] >> trIdx) & 1; }
          }
          if (!((k.inter_split && trIdx == 0) || log2 > k.log2_max_tu || log2 == 2 || log2 == k.log2_min_tu)) CBIN(CX_TRANS_SUBDIV + 5 - log2, subdiv);
          const int first = trIdx == 0;
          if (first || log2 > 2) {
            if (first || ((r->cbf[1][part] >> (trIdx - 1)) & 1)) CBIN(rqt_cbf_ctx(1, trIdx), (r->cbf[1][part] >> trIdx) & 1);
            if (first || ((r->cbf[2][part] >> (trIdx - 1)) & 1)) CBIN(rqt_cbf_ctx(2, trIdx), (r->cbf[2][part] >> trIdx) & 1);
          }
          if (!subdiv) {
            if (!(trIdx == 0 && !(r->cbf[1][part] & 1) && !(r->cbf[2][part] & 1))) CBIN(rqt_cbf_ctx(0, r->tr_idx[part]), (r->cbf[0][part] >> r->tr_idx[part]) & 1);
            if (cbfY) frac += cb_code_tu(sh, lane, cf + 16 * part, log2, 0, 0, k.sign_hide, k.use_ts, r->tskip[0][part], 0, scans);
            if (log2 > 2) {
              if (cbfU) frac += cb_code_tu(sh, lane, cf + cu2 + 4 * part, log2 - 1, 1, 0, k.sign_hide, k.use_ts, r->tskip[1][part], 0, scans);
              if (cbfV) frac += cb_code_tu(sh, lane, cf + cu2 + (cu2 >> 2) + 4 * part, log2 - 1, 1, 0, k.sign_hide, k.use_ts, r->tskip[2][part], 0, scans);
            } else {
              const int pn = parts >> (2 * (trIdx - 1));
              if (part % pn == pn - 1) {
                if (cbfU) frac += cb_code_tu(sh, lane, cf + cu2 + 4 * bak, 2, 1, 0, k.sign_hide, k.use_ts, r->tskip[1][bak], 0, scans);
                if (cbfV) frac += cb_code_tu(sh, lane, cf + cu2 + (cu2 >> 2) + 4 * bak, 2, 1, 0, k.sign_hide, k.use_ts, r->tskip[2][bak], 0, scans);
              }
            }
            sp--; continue;
          }
          sp_k[sp] = 0;
        }
        if (sp_k[sp] < 4) { const int q = (parts >> (2 * trIdx)) >> 2, kk = sp_k[sp]++; sp_part[sp + 1] = part + kk * q; sp_k[sp + 1] = -1; sp++; }
        else sp--;
      }
    }
  }
  bits_out[i] = (uint32_t)(frac >> 15);
  skipped_out[i] = skipped;
  if (ctx_out) rqt_store(sh, lane, frac, ctx_out + i);
  if (cu_out) for (int q = 0; q < 20; q++) cu_out[i].state[q] = sh.st[CUX + q][lane];
}
template <class LDS>
__global__ __launch_bounds__(64) void k_cu_bits(RqtClass k, int n, const hop_rqt_job* __restrict__ jobs, const hop_cu_syntax* __restrict__ syn, const hop_rqt_result* __restrict__ res,
                                                const int32_t* __restrict__ coef, const hop_cabac_ctx* __restrict__ ctx_in, const hop_cabac_cu_ctx* __restrict__ cu_in,
                                                uint32_t* __restrict__ bits_out, uint32_t* __restrict__ skipped_out, hop_cabac_ctx* __restrict__ ctx_out,
                                                hop_cabac_cu_ctx* __restrict__ cu_out, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  cu_bits_body(sh, lane, i, k, jobs, syn, res, coef, ctx_in, cu_in, bits_out, skipped_out, ctx_out, cu_out, scans);
}

int hop_launch_cu_bits(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int inter_split, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                       const hop_cu_syntax* d_syn, const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                       uint32_t* d_bits, uint32_t* d_skipped, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = inter_split; k.sign_hide = sign_hide; k.use_ts = use_ts;
  const int pr = hop_prof_begin(c, HOP_K_CABAC, (uint64_t)n);
  RQT_LAUNCH(k_cu_bits, c, n, k, n, d_jobs, d_syn, d_res, d_coef, d_ctx_in, d_cu_in, d_bits, d_skipped, d_ctx_out, d_cu_out, c->rdoq_scans);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "cu_bits launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The bits of an intra CU's quadtree as the intra search counts them: TEncSearch::xGetIntraBitsQT (TLibEncoder/TEncSearch.cpp:957-980) = xEncIntraHeader
// (:887-954), xEncSubdivCbfQT (:764-830), xEncCoeffQT (:833-884), from the node (tr_depth, part) downwards.  One lane per job.
// =====================================================================================================================
#define CU_IPRED (CUX + 16)
#define CU_CPRED (CUX + 17)
__device__ static inline int icu_scan(const hop_intra_cu_syntax& y, int parts, int part, int log2w, int comp) {      // TComDataCU::getCoefScanIdx, intra
  const int dir = comp ? (y.chroma_is_dm ? y.luma_dir[0] : y.chroma_dir) : y.luma_dir[y.part_nxn ? part / (parts >> 2) : 0];
  const bool multi = comp ? (log2w == 2 || log2w == 1) : (log2w == 2 || log2w == 3);                                 // sizes with direction-dependent scans
  if (!multi) return 0;
  const int dv = dir - 26, dh = dir - 10;
  return (dv < 0 ? -dv : dv) < 5 ? 1 : ((dh < 0 ? -dh : dh) < 5 ? 2 : 0);
}
template <class LDS>
__device__ static unsigned long long icu_dir(LDS& sh, const int lane, const int dir, const int32_t* preds, const int pred_num) {   // codeIntraDirLumaAng, one PU
  unsigned long long frac = 0;
  int idx = -1;
  for (int q = 0; q < pred_num; q++) if (dir == preds[q]) idx = q;
  CBIN(CU_IPRED, idx != -1);
  CEP(idx == -1 ? 5 : (idx ? 2 : 1));
  return frac;
}

// the counting itself, from the node (tr0, part0) downwards; levels either in the CU layout (cf) or in the layer buffers of the quadtree searches (layered, CU i)
template <class LDS>
__device__ __noinline__ static unsigned long long icu_count(LDS& sh, const int lane, const RqtClass& k, const hop_intra_cu_syntax& y, const int tr0, const int part0, const int b_luma,
                                               const int b_chroma, const hop_rqt_result* r, const int32_t* cf, const int32_t* layered, const int i, const uint16_t* scans) {
  const int parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  unsigned long long frac = 0;
  // xEncIntraHeader
  if (b_luma) {
    if (part0 == 0) {
      if (y.skip_ctx >= 0) {                                         // skip_ctx < 0: an I slice, which codes neither the skip flag nor the prediction mode
        CBIN(CU_SKIP + y.skip_ctx, y.skip_flag ? 1 : 0);
        CBIN(CU_PRED, 1);                                            // MODE_INTRA
      }
      if (y.is_min_cu) CBIN(CU_PART, y.part_nxn ? 0 : 1);
    }
    if (!y.part_nxn) { if (part0 == 0) frac += icu_dir(sh, lane, y.luma_dir[0], y.preds[0], y.pred_num[0]); }
    else {
      const int q4 = parts >> 2;
      if (tr0 == 0) { for (int p = 0; p < 4; p++) frac += icu_dir(sh, lane, y.luma_dir[p], y.preds[p], y.pred_num[p]); }
      else if (part0 % q4 == 0) frac += icu_dir(sh, lane, y.luma_dir[part0 / q4], y.preds[part0 / q4], y.pred_num[part0 / q4]);
    }
  }
  if (b_chroma && part0 == 0) { if (y.chroma_is_dm) CBIN(CU_CPRED, 0); else { CBIN(CU_CPRED, 1); CEP(2); } }
  // xEncSubdivCbfQT, then xEncCoeffQT per component: pass 0 = flags, passes 1..3 = levels of Y, Cb, Cr
  for (int pass = 0; pass < 4; pass++) {
    if (pass == 1 && !b_luma) continue;
    if (pass >= 2 && !b_chroma) continue;
    int sp_part[4], sp_k[4]; int sp = 0;
    sp_part[0] = part0; sp_k[0] = -1;
    while (sp >= 0) {
      const int part = sp_part[sp], trDepth = tr0 + sp, log2 = k.log2_cu - trDepth;
      if (sp_k[sp] < 0) {
        const int trMode = r->tr_idx[part], subdiv = trMode > trDepth;
        if (pass == 0) {
          if (!((y.part_nxn && trDepth == 0) || log2 > k.log2_max_tu || log2 == 2 || log2 == k.log2_min_tu) && b_luma) CBIN(CX_TRANS_SUBDIV + 5 - log2, subdiv);
          if (b_chroma && log2 > 2) {
            if (trDepth == 0 || ((r->cbf[1][part] >> (trDepth - 1)) & 1)) CBIN(rqt_cbf_ctx(1, trDepth), (r->cbf[1][part] >> trDepth) & 1);
            if (trDepth == 0 || ((r->cbf[2][part] >> (trDepth - 1)) & 1)) CBIN(rqt_cbf_ctx(2, trDepth), (r->cbf[2][part] >> trDepth) & 1);
          }
        }
        if (!subdiv) {
          if (pass == 0) { if (b_luma) CBIN(rqt_cbf_ctx(0, trMode), (r->cbf[0][part] >> trMode) & 1); }
          else {
            const int comp = pass - 1;
            int d = trDepth; bool code = true;
            if (comp && log2 == 2) { d--; code = (part % (parts >> (2 * d))) == 0; }
            if (code) {
              const int lg = k.log2_cu - d - (comp ? 1 : 0);
              const int32_t* cp = layered ? layered + rqt_coef_at(k, i, k.log2_max_tu - log2, comp, part)
                                              : cf + (comp == 0 ? (size_t)(16 * part) : cu2 + (size_t)(comp - 1) * (cu2 >> 2) + (size_t)(4 * part));
              frac += cb_code_tu(sh, lane, cp, lg, comp != 0, icu_scan(y, parts, part, lg, comp), k.sign_hide, k.use_ts, r->tskip[comp][part], 0, scans);
            }
          }
          sp--; continue;
        }
        sp_k[sp] = 0;
      }
      if (sp_k[sp] < 4) { const int q = (parts >> (2 * trDepth)) >> 2, kk = sp_k[sp]++; sp_part[sp + 1] = part + kk * q; sp_k[sp + 1] = -1; sp++; }
      else sp--;
    }
  }
  return frac;
}

template <class LDS>
__global__ __launch_bounds__(64) void k_intra_cu_bits(RqtClass k, int n, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn,
                                                      const hop_rqt_result* __restrict__ res, const int32_t* __restrict__ coef, const hop_cabac_ctx* __restrict__ ctx_in,
                                                      const hop_cabac_cu_ctx* __restrict__ cu_in, uint32_t* __restrict__ bits_out, hop_cabac_ctx* __restrict__ ctx_out,
                                                      hop_cabac_cu_ctx* __restrict__ cu_out, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  const int ci = jobs[i].ctx_index;
  RQ_LOAD(ctx_in[ci]);
  for (int q = 0; q < 20; q++) sh.st[CUX + q][lane] = cu_in[ci].state[q];
  const hop_intra_cu_syntax y = syn[i];
  const hop_rqt_result* r = res + i;
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  const int32_t* cf = coef + (size_t)i * (cu2 + (cu2 >> 1));
  unsigned long long frac = RQ_LEFT();
  frac += icu_count(sh, lane, k, y, y.tr_depth, y.part, y.b_luma, y.b_chroma, r, cf, nullptr, i, scans);
  bits_out[i] = (uint32_t)(frac >> 15);
  if (ctx_out) rqt_store(sh, lane, frac, ctx_out + i);
  if (cu_out) for (int q = 0; q < 20; q++) cu_out[i].state[q] = sh.st[CUX + q][lane];
}

int hop_launch_intra_cu_bits(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs, const hop_intra_cu_syntax* d_syn,
                             const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, uint32_t* d_bits,
                             hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = sign_hide; k.use_ts = use_ts;
  const int pr = hop_prof_begin(c, HOP_K_CABAC, (uint64_t)n);
  RQT_LAUNCH(k_intra_cu_bits, c, n, k, n, d_jobs, d_syn, d_res, d_coef, d_ctx_in, d_cu_in, d_bits, d_ctx_out, d_cu_out, c->rdoq_scans);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_cu_bits launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The luma transform tree of an intra PU: TEncSearch::xRecurIntraCodingQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:1361-1710), for a batch of PUs of one
// class (CU size, transform-tree limits, first transform depth, bCheckFirst).  The same shape as the residual quadtree above - a chain inside a PU, a batch
// across PUs, the host walking the tree - with the intra leaf per node (xIntraCodingLumaBlk :1003-1161):
//   k_irqt_begin   entry state stored (CI_QT_TRAFO_ROOT), depth / transform-skip arrays set, the node's prediction job (reference samples from the context's
//                  reconstruction picture by the caller's neighbour flags) and leaf job(s) emitted
//   hop_launch_intra_pred, hop_launch_tu_rd (is_intra: residual, DST / DCT or transform skip, estBit on the entry state, RDOQ, inverse path, Clip(prediction +
//                  residual) into the reconstruction picture, SSE); for 4x4 nodes first the transform-skip variant, its block parked by k_irqt_copy
//   k_irqt_single  one lane per PU: the node's bits through xGetIntraBitsQT (icu_count) from the entry state, its cost; the 4x4 transform-skip decision
//                  (:1424-1523); a node without children adds itself to its parent's sums, one with children parks the state (CI_QT_TRAFO_TEST) and rewinds
//   k_irqt_close   cbf of the children folded upwards, the subtree recounted from the entry state, the split decision (:1576-1700); k_irqt_copy puts the
//                  single block's reconstruction (kept in a per-PU layer plane, m_pcQTTempTComYuv) back into the picture where it wins
// The PUs of one call must not lie in each other's neighbourhood (the picture is read and written as the search goes, as in the reference).
// =====================================================================================================================
struct IrqWork { double single_cost[4]; double sub_cost[5]; uint32_t single_dist[4], single_cbf[4], sub_dist[5]; uint8_t best[4], restore[4]; };
__host__ __device__ static inline int irq_node_index(int d, int log2, int part) { const int first[5] = { 0, 1, 5, 21, 85 }; return first[d] + (part >> (2 * (log2 - 2))); }
#define IRQ_CU_LOAD(src) do { for (int q_ = 0; q_ < 20; q_++) sh.st[CUX + q_][lane] = (src).state[q_]; } while (0)
#define IRQ_CU_STORE(dst) do { for (int q_ = 0; q_ < 20; q_++) (dst).state[q_] = sh.st[CUX + q_][lane]; } while (0)

__device__ static inline void irqt_init_body(const int i, const hop_rqt_job* jobs, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_in, hop_cabac_ctx* cur, hop_cabac_cu_ctx* cucur,
                                             IrqWork* work, hop_rqt_result* res) {
  cur[i] = ctx_in[jobs[i].ctx_index]; cucur[i] = cu_in[jobs[i].ctx_index];
  IrqWork w; memset(&w, 0, sizeof(w));
  work[i] = w;
  hop_rqt_result* r = res + i;
  r->cost = 0; r->bits = r->dist = r->zero_dist = r->pad = 0;
  for (int p = 0; p < 256; p++) { r->tr_idx[p] = 0; for (int c = 0; c < 3; c++) { r->cbf[c][p] = 0; r->tskip[c][p] = 0; } }
}
__global__ void k_irqt_init(const hop_rqt_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const hop_cabac_cu_ctx* __restrict__ cu_in,
                            hop_cabac_ctx* __restrict__ cur, hop_cabac_cu_ctx* __restrict__ cucur, IrqWork* __restrict__ work, hop_rqt_result* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) irqt_init_body(i, jobs, ctx_in, cu_in, cur, cucur, work, res);
}

__device__ static inline void irqt_begin_body(const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn, const hop_intra_rqt_opt* opt, int bd_y,
                                              const hop_cabac_ctx* cur, const hop_cabac_cu_ctx* cucur, hop_cabac_ctx* root, hop_cabac_cu_ctx* curoot, hop_rqt_result* res, IrqWork* work,
                                              hop_intra_job* pj, int32_t* modes, hop_tu_rd_job* tuj, int64_t* off, hop_tu_rd_job* tuj2, int64_t* off2, size_t ts_base, const uint8_t* active) {
  if (active && !active[i]) {                                           // an idle PU of this pass: empty slots in the prediction and leaf batches
    hop_intra_job q0; memset(&q0, 0, sizeof(q0)); pj[i] = q0; modes[i] = 0;
    hop_tu_rd_job j0; memset(&j0, 0, sizeof(j0)); tuj[i] = j0; off[i] = 0; if (nd.ts_y) { tuj2[i] = j0; off2[i] = 0; }
    return;
  }
  root[i] = cur[i]; curoot[i] = cucur[i];
  IrqWork* w = work + i;
  if (nd.check_split) { w->sub_cost[nd.d + 1] = 0; w->sub_dist[nd.d + 1] = 0; }
  w->restore[nd.d] = 0;
  if (!nd.check_full) { w->single_cost[nd.d] = 1.7e+308; return; }
  const hop_rqt_job jb = jobs[i];
  const hop_intra_cu_syntax y = syn[i];
  hop_rqt_result* r = res + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d), part = y.part + nd.part, N = 1 << nd.log2;
  for (int p = 0; p < nparts; p++) { r->tr_idx[part + p] = (uint8_t)nd.d; r->tskip[0][part + p] = 0; }
  const int dir = y.luma_dir[y.part_nxn ? part / (parts >> 2) : 0];
  hop_intra_job q;
  q.x = jb.x + rqt_zx(part); q.y = jb.y + rqt_zy(part); q.size = N; q.strong = opt[i].strong;
  const unsigned long long av = opt[i].avail[irq_node_index(nd.d, nd.log2, part)];
  for (int u = 0; u < 68; u++) q.flags[u] = (u < 4 * (N / 4) + 1) ? (uint8_t)((av >> u) & 1ull) : 0;
  pj[i] = q; modes[i] = dir;
  hop_tu_rd_job j;
  j.x = q.x; j.y = q.y; j.comp = 0; j.log2_size = nd.log2; j.qp_scaled = jb.qp_scaled[0]; j.tr_depth = nd.d; j.ctx_index = i; j.sign_hide = k.sign_hide; j.use_ts = k.use_ts;
  j.bit_depth = bd_y; j.is_intra = 1; j.scan_idx = icu_scan(y, parts, part, nd.log2, 0); j.use_dst = 1; j.flags = 0;
  j.lambda_rdoq = jb.lambda_rdoq[0]; j.lambda_rd = jb.lambda_rd; j.dist_weight = 1.0;
  tuj[i] = j; off[i] = (int64_t)rqt_coef_at(k, i, k.log2_max_tu - nd.log2, 0, part);
  if (nd.ts_y) {
    j.flags = HOP_TU_RD_TS;
    if (opt[i].ts_fast && !y.part_nxn) j.log2_size = 0;                 // TransformSkipFast: not tried for this CU (an empty slot of the leaf batch)
    tuj2[i] = j; off2[i] = (int64_t)(ts_base + (size_t)i * 16);
  }
}
__global__ void k_irqt_begin(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn, const hop_intra_rqt_opt* __restrict__ opt,
                             int n, int bd_y, const hop_cabac_ctx* __restrict__ cur, const hop_cabac_cu_ctx* __restrict__ cucur, hop_cabac_ctx* __restrict__ root,
                             hop_cabac_cu_ctx* __restrict__ curoot, hop_rqt_result* __restrict__ res, IrqWork* __restrict__ work, hop_intra_job* __restrict__ pj,
                             int32_t* __restrict__ modes, hop_tu_rd_job* __restrict__ tuj, int64_t* __restrict__ off, hop_tu_rd_job* __restrict__ tuj2,
                             int64_t* __restrict__ off2, size_t ts_base, const uint8_t* __restrict__ active) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) irqt_begin_body(i, k, nd, jobs, syn, opt, bd_y, cur, cucur, root, curoot, res, work, pj, modes, tuj, off, tuj2, off2, ts_base, active);
}

// mode 0: the node's block, picture -> layer plane; 1: the 4x4 block, picture -> transform-skip park; 2: layer plane -> picture where the single block has won
__device__ static inline void irqt_copy_body(const int i, const int tid, const int nt, int mode, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn,
                                             const IrqWork* work, int16_t* rec, int pitch, int16_t* recl, int16_t* park, const uint8_t* active) {
  if (active && !active[i]) return;
  if (mode == 2 && !work[i].restore[nd.d]) return;
  const int part = syn[i].part + nd.part, N = 1 << nd.log2, cu = 1 << k.log2_cu, x0 = rqt_zx(part), y0 = rqt_zy(part);
  int16_t* pic = rec + (size_t)(jobs[i].y + y0) * pitch + jobs[i].x + x0;
  int16_t* lay = recl + ((size_t)i * 4 + (size_t)(k.log2_max_tu - nd.log2)) * ((size_t)cu * cu) + (size_t)y0 * cu + x0;
  for (int e = tid; e < N * N; e += nt) {
    const int rr = e >> nd.log2, cc = e & (N - 1);
    if (mode == 0) lay[(size_t)rr * cu + cc] = pic[(size_t)rr * pitch + cc];
    else if (mode == 1) park[(size_t)i * 16 + e] = pic[(size_t)rr * pitch + cc];
    else pic[(size_t)rr * pitch + cc] = lay[(size_t)rr * cu + cc];
  }
}
__global__ __launch_bounds__(256) void k_irqt_copy(int mode, RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn, int n,
                                                   const IrqWork* __restrict__ work, int16_t* __restrict__ rec, int pitch, int16_t* __restrict__ recl, int16_t* __restrict__ park,
                                                   const uint8_t* __restrict__ active) {
  irqt_copy_body(blockIdx.x, threadIdx.x, 256, mode, k, nd, jobs, syn, work, rec, pitch, recl, park, active);
}

template <class LDS>
__device__ static void irqt_single_body(LDS& sh, const int lane, const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn,
                                        const hop_intra_rqt_opt* opt, hop_cabac_ctx* cur, hop_cabac_cu_ctx* cucur, const hop_cabac_ctx* root, const hop_cabac_cu_ctx* curoot,
                                        hop_cabac_ctx* test, hop_cabac_cu_ctx* cutest, hop_rqt_result* res, IrqWork* work, const hop_tu_rd_result* tr, const hop_tu_rd_result* tr2,
                                        int32_t* coef, size_t ts_base, int16_t* rec, int pitch, const int16_t* park, const uint16_t* scans, const uint8_t* active) {
  if (active && !active[i]) return;
  const hop_intra_cu_syntax y = syn[i];
  const double lambda = jobs[i].lambda_rd;
  hop_rqt_result* r = res + i;
  IrqWork* w = work + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d), part = y.part + nd.part;
  // the block as the transform left it
  const uint32_t dist0 = tr[i].dist, cbf0 = tr[i].abs_sum ? 1u : 0u;
  for (int p = 0; p < nparts; p++) r->cbf[0][part + p] = (uint8_t)(cbf0 << nd.d);
  RQ_LOAD(root[i]); IRQ_CU_LOAD(curoot[i]);
  unsigned long long frac = RQ_LEFT();
  frac += icu_count(sh, lane, k, y, nd.d, part, 1, 0, r, nullptr, coef, i, scans);
  double cost = rqt_cost((uint32_t)(frac >> 15), dist0, lambda);
  uint32_t dist = dist0, cbf = cbf0; int best = 0;
  const bool ts_on = nd.ts_y && !(opt[i].ts_fast && !y.part_nxn);
  if (ts_on) {
    const uint32_t cbf1 = tr2[i].abs_sum ? 1u : 0u;
    if (cbf1) {                                                         // modeId 1 with a zero block is never taken (:1463-1466)
      rqt_store(sh, lane, frac, cur + i); IRQ_CU_STORE(cucur[i]);        // CI_TEMP_BEST
      int32_t* lv = coef + rqt_coef_at(k, i, k.log2_max_tu - 2, 0, part); int32_t* tv = coef + ts_base + (size_t)i * 16;
      for (int e = 0; e < 16; e++) { const int32_t t = lv[e]; lv[e] = tv[e]; tv[e] = t; }
      for (int p = 0; p < nparts; p++) { r->tskip[0][part + p] = 1; r->cbf[0][part + p] = (uint8_t)(1u << nd.d); }
      RQ_LOAD(root[i]); IRQ_CU_LOAD(curoot[i]);
      unsigned long long f1 = RQ_LEFT();
      f1 += icu_count(sh, lane, k, y, nd.d, part, 1, 0, r, nullptr, coef, i, scans);
      const double cost1 = rqt_cost((uint32_t)(f1 >> 15), tr2[i].dist, lambda);
      if (cost1 < cost) {
        cost = cost1; dist = tr2[i].dist; cbf = 1; best = 1; frac = f1;
        int16_t* pic = rec + (size_t)(jobs[i].y + rqt_zy(part)) * pitch + jobs[i].x + rqt_zx(part);
        for (int e = 0; e < 16; e++) pic[(size_t)(e >> 2) * pitch + (e & 3)] = park[(size_t)i * 16 + e];
      } else {                                                          // xLoadIntraResultQT: levels, arrays and the coder of the first variant
        for (int e = 0; e < 16; e++) { const int32_t t = lv[e]; lv[e] = tv[e]; tv[e] = t; }
        for (int p = 0; p < nparts; p++) { r->tskip[0][part + p] = 0; r->cbf[0][part + p] = (uint8_t)(cbf0 << nd.d); }
        RQ_LOAD(cur[i]); IRQ_CU_LOAD(cucur[i]); frac = RQ_LEFT();
      }
    }
  }
  w->single_cost[nd.d] = cost; w->single_dist[nd.d] = dist; w->single_cbf[nd.d] = cbf; w->best[nd.d] = (uint8_t)best;
  if (nd.check_split) { rqt_store(sh, lane, frac, test + i); IRQ_CU_STORE(cutest[i]); cur[i] = root[i]; cucur[i] = curoot[i]; }
  else { rqt_store(sh, lane, frac, cur + i); IRQ_CU_STORE(cucur[i]); w->sub_cost[nd.d] += cost; w->sub_dist[nd.d] += dist; }
}
template <class LDS>
__global__ __launch_bounds__(64) void k_irqt_single(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn,
                                                    const hop_intra_rqt_opt* __restrict__ opt, int n, hop_cabac_ctx* __restrict__ cur, hop_cabac_cu_ctx* __restrict__ cucur,
                                                    const hop_cabac_ctx* __restrict__ root, const hop_cabac_cu_ctx* __restrict__ curoot, hop_cabac_ctx* __restrict__ test,
                                                    hop_cabac_cu_ctx* __restrict__ cutest, hop_rqt_result* __restrict__ res, IrqWork* __restrict__ work,
                                                    const hop_tu_rd_result* __restrict__ tr, const hop_tu_rd_result* __restrict__ tr2, int32_t* __restrict__ coef, size_t ts_base,
                                                    int16_t* __restrict__ rec, int pitch, const int16_t* __restrict__ park, const uint16_t* __restrict__ scans,
                                                    const uint8_t* __restrict__ active) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  irqt_single_body(sh, lane, i, k, nd, jobs, syn, opt, cur, cucur, root, curoot, test, cutest, res, work, tr, tr2, coef, ts_base, rec, pitch, park, scans, active);
}

template <class LDS>
__device__ static void irqt_close_body(LDS& sh, const int lane, const int i, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn, hop_cabac_ctx* cur,
                                       hop_cabac_cu_ctx* cucur, const hop_cabac_ctx* root, const hop_cabac_cu_ctx* curoot, const hop_cabac_ctx* test, const hop_cabac_cu_ctx* cutest,
                                       hop_rqt_result* res, IrqWork* work, const int32_t* coef, const uint16_t* scans, const uint8_t* active) {
  if (active && !active[i]) return;
  const hop_intra_cu_syntax y = syn[i];
  hop_rqt_result* r = res + i;
  IrqWork* w = work + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d), part = y.part + nd.part, q = nparts >> 2;
  uint32_t scbf = 0;
  for (int kk = 0; kk < 4; kk++) scbf |= (r->cbf[0][part + kk * q] >> (nd.d + 1)) & 1u;
  for (int p = 0; p < nparts; p++) r->cbf[0][part + p] |= (uint8_t)(scbf << nd.d);
  RQ_LOAD(root[i]); IRQ_CU_LOAD(curoot[i]);
  unsigned long long frac = RQ_LEFT();
  frac += icu_count(sh, lane, k, y, nd.d, part, 1, 0, r, nullptr, coef, i, scans);
  const double split = rqt_cost((uint32_t)(frac >> 15), w->sub_dist[nd.d + 1], jobs[i].lambda_rd);
  if (split < w->single_cost[nd.d]) {
    rqt_store(sh, lane, frac, cur + i); IRQ_CU_STORE(cucur[i]);
    w->sub_cost[nd.d] += split; w->sub_dist[nd.d] += w->sub_dist[nd.d + 1];
    return;
  }
  cur[i] = test[i]; cucur[i] = cutest[i];
  for (int p = 0; p < nparts; p++) { r->tr_idx[part + p] = (uint8_t)nd.d; r->cbf[0][part + p] = (uint8_t)(w->single_cbf[nd.d] << nd.d); r->tskip[0][part + p] = w->best[nd.d]; }
  w->restore[nd.d] = 1;
  w->sub_cost[nd.d] += w->single_cost[nd.d]; w->sub_dist[nd.d] += w->single_dist[nd.d];
}
template <class LDS>
__global__ __launch_bounds__(64) void k_irqt_close(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn, int n,
                                                   hop_cabac_ctx* __restrict__ cur, hop_cabac_cu_ctx* __restrict__ cucur, const hop_cabac_ctx* __restrict__ root,
                                                   const hop_cabac_cu_ctx* __restrict__ curoot, const hop_cabac_ctx* __restrict__ test, const hop_cabac_cu_ctx* __restrict__ cutest,
                                                   hop_rqt_result* __restrict__ res, IrqWork* __restrict__ work, const int32_t* __restrict__ coef, const uint16_t* __restrict__ scans,
                                                   const uint8_t* __restrict__ active) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  irqt_close_body(sh, lane, i, k, nd, jobs, syn, cur, cucur, root, curoot, test, cutest, res, work, coef, scans, active);
}

__device__ static inline void irqt_final_body(const int i, const int t, const int nt, const RqtClass& k, int d0, const hop_intra_cu_syntax* syn, const IrqWork* work, hop_rqt_result* res,
                                              const int32_t* coef, int32_t* coef_out, const hop_cabac_ctx* cur, const hop_cabac_cu_ctx* cucur, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_out,
                                              const uint8_t* active) {
  if (active && !active[i]) return;
  const int parts = 1 << (2 * (k.log2_cu - 2)), np = parts >> (2 * d0), p0 = syn[i].part;
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  hop_rqt_result* r = res + i;
  if (t == 0) { r->cost = work[i].sub_cost[d0]; r->dist = work[i].sub_dist[d0]; if (ctx_out) ctx_out[i] = cur[i]; if (cu_out) cu_out[i] = cucur[i]; }
  if (!coef_out) return;
  for (int e = t; e < 16 * np; e += nt) {
    const int p = p0 + (e >> 4), layer = k.log2_max_tu - (k.log2_cu - r->tr_idx[p]);
    coef_out[(size_t)i * (cu2 + (cu2 >> 1)) + (size_t)16 * p0 + e] = coef[rqt_coef_at(k, i, layer, 0, p) + (e & 15)];
  }
}
__global__ __launch_bounds__(64) void k_irqt_final(RqtClass k, int d0, int n, const hop_intra_cu_syntax* __restrict__ syn, const IrqWork* __restrict__ work,
                                                   hop_rqt_result* __restrict__ res, const int32_t* __restrict__ coef, int32_t* __restrict__ coef_out,
                                                   const hop_cabac_ctx* __restrict__ cur, const hop_cabac_cu_ctx* __restrict__ cucur, hop_cabac_ctx* __restrict__ ctx_out,
                                                   hop_cabac_cu_ctx* __restrict__ cu_out, const uint8_t* __restrict__ active) {
  irqt_final_body(blockIdx.x, threadIdx.x, 64, k, d0, syn, work, res, coef, coef_out, cur, cucur, ctx_out, cu_out, active);
}

size_t hop_intra_rqt_work_bytes(int log2_cu, int n) {
  const size_t cu2 = (size_t)1 << (2 * log2_cu);
  return (size_t)n * (9 * (sizeof(hop_cabac_ctx) + sizeof(hop_cabac_cu_ctx) + 16) + sizeof(IrqWork) + sizeof(hop_intra_job) + 4 + 2 * (sizeof(hop_tu_rd_job) + 8 + sizeof(hop_tu_rd_result)) +
                     (6 * cu2 + 16) * 4 + 4 * cu2 * 2 + 32) + 64 * 256;
}

// one class of PUs: CU size, transform-tree limits, the transform depth the PU starts at (0: 2Nx2N, 1: NxN) and bCheckFirst; buf = hop_intra_rqt_work_bytes
int hop_launch_intra_rqt(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int tr_depth0, int check_first, int n, const hop_rqt_job* d_jobs,
                         const hop_intra_cu_syntax* d_syn, const hop_intra_rqt_opt* d_opt, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, hop_rqt_result* d_res,
                         int32_t* d_coef_out, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out, void* vbuf, size_t buf_bytes,
                         const uint8_t* d_active /* may be NULL: PUs that sit this call out */) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = sign_hide; k.use_ts = use_ts;
  char* buf = (char*)vbuf;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu), n_coeff = (size_t)n * (6 * cu2 + 16), ts_base = (size_t)n * 6 * cu2;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  struct St { hop_cabac_ctx* a; hop_cabac_cu_ctx* b; };
  auto take_st = [&]() { St s; s.a = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); s.b = (hop_cabac_cu_ctx*)take((size_t)n * sizeof(hop_cabac_cu_ctx)); return s; };
  struct Bufs {
    St cur, root[4], test[4]; IrqWork* work; hop_intra_job* pj; int32_t* modes; hop_tu_rd_job *tuj, *tuj2; int64_t *off, *off2; hop_tu_rd_result *tr, *tr2; int32_t* coef;
    int16_t *recl, *park; size_t n_coeff, ts_base; const uint8_t* active;
  } B;
  B.cur = take_st();
  for (int d = 0; d < 4; d++) { B.root[d] = take_st(); B.test[d] = take_st(); }
  B.work = (IrqWork*)take((size_t)n * sizeof(IrqWork));
  B.pj = (hop_intra_job*)take((size_t)n * sizeof(hop_intra_job)); B.modes = (int32_t*)take((size_t)n * 4);
  B.tuj = (hop_tu_rd_job*)take((size_t)n * sizeof(hop_tu_rd_job)); B.tuj2 = (hop_tu_rd_job*)take((size_t)n * sizeof(hop_tu_rd_job));
  B.off = (int64_t*)take((size_t)n * 8); B.off2 = (int64_t*)take((size_t)n * 8);
  B.tr = (hop_tu_rd_result*)take((size_t)n * sizeof(hop_tu_rd_result)); B.tr2 = (hop_tu_rd_result*)take((size_t)n * sizeof(hop_tu_rd_result));
  B.coef = (int32_t*)take(n_coeff * 4);
  B.recl = (int16_t*)take((size_t)n * 4 * cu2 * 2); B.park = (int16_t*)take((size_t)n * 32);
  B.n_coeff = n_coeff; B.ts_base = ts_base; B.active = d_active;
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "intra rqt: work buffer too small");
  hipLaunchKernelGGL(k_irqt_init, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, d_ctx_in, d_cu_in, B.cur.a, B.cur.b, B.work, d_res);
  struct Rec { static int go(hop_ctx* c, const RqtClass& k, int n, int check_first, const hop_rqt_job* d_jobs, const hop_intra_cu_syntax* d_syn, const hop_intra_rqt_opt* d_opt,
                             hop_rqt_result* d_res, Bufs& B, int rel, int d, int log2) {
    RqtNode nd; memset(&nd, 0, sizeof(nd));
    nd.part = rel; nd.d = d; nd.log2 = log2;
    nd.check_full = log2 <= k.log2_max_tu;
    nd.check_split = log2 > k.log2_min_tu && !(check_first && nd.check_full);
    nd.ts_y = (k.use_ts && nd.check_full && log2 == 2) ? 1 : 0;
    const int g256 = (n + 255) / 256, pitch = c->pic_w;
    hipLaunchKernelGGL(k_irqt_begin, dim3(g256), dim3(256), 0, c->stream, k, nd, d_jobs, d_syn, d_opt, n, c->bd_y, B.cur.a, B.cur.b, B.root[d].a, B.root[d].b, d_res, B.work,
                       B.pj, B.modes, B.tuj, B.off, B.tuj2, B.off2, B.ts_base, B.active);
    if (nd.check_full) {
      int r = hop_launch_intra_pred(c, n, B.pj, B.modes); if (r) return r;
      const int hint = log2 <= 3 ? 1 : (log2 == 5 ? 2 : 0);
      if (nd.ts_y) {
        r = hop_launch_tu_rd(c, n, B.tuj2, B.root[d].a, B.off2, B.n_coeff, B.coef, B.tr2, 1); if (r) return r;
        hipLaunchKernelGGL(k_irqt_copy, dim3(n), dim3(256), 0, c->stream, 1, k, nd, d_jobs, d_syn, n, B.work, c->rec[0], pitch, B.recl, B.park, B.active);
      }
      r = hop_launch_tu_rd(c, n, B.tuj, B.root[d].a, B.off, B.n_coeff, B.coef, B.tr, hint); if (r) return r;
      if (nd.check_split) hipLaunchKernelGGL(k_irqt_copy, dim3(n), dim3(256), 0, c->stream, 0, k, nd, d_jobs, d_syn, n, B.work, c->rec[0], pitch, B.recl, B.park, B.active);
      RQT_LAUNCH(k_irqt_single, c, n, k, nd, d_jobs, d_syn, d_opt, n, B.cur.a, B.cur.b, B.root[d].a, B.root[d].b, B.test[d].a, B.test[d].b,
                         d_res, B.work, B.tr, B.tr2, B.coef, B.ts_base, c->rec[0], pitch, B.park, c->rdoq_scans, B.active);
    }
    if (nd.check_split) {
      const int q = ((1 << (2 * (k.log2_cu - 2))) >> (2 * d)) >> 2;
      for (int kk = 0; kk < 4; kk++) { const int r = go(c, k, n, check_first, d_jobs, d_syn, d_opt, d_res, B, rel + kk * q, d + 1, log2 - 1); if (r) return r; }
      RQT_LAUNCH(k_irqt_close, c, n, k, nd, d_jobs, d_syn, n, B.cur.a, B.cur.b, B.root[d].a, B.root[d].b, B.test[d].a, B.test[d].b, d_res,
                         B.work, B.coef, c->rdoq_scans, B.active);
      if (nd.check_full) hipLaunchKernelGGL(k_irqt_copy, dim3(n), dim3(256), 0, c->stream, 2, k, nd, d_jobs, d_syn, n, B.work, c->rec[0], pitch, B.recl, B.park, B.active);
    }
    return HOP_OK;
  } };
  const int rc = Rec::go(c, k, n, check_first, d_jobs, d_syn, d_opt, d_res, B, 0, tr_depth0, k.log2_cu - tr_depth0);
  if (rc) return rc;
  hipLaunchKernelGGL(k_irqt_final, dim3(n), dim3(64), 0, c->stream, k, tr_depth0, n, d_syn, B.work, d_res, B.coef, d_coef_out, B.cur.a, B.cur.b, d_ctx_out, d_cu_out, d_active);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra rqt launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The luma intra search of a CU: TEncSearch::estIntraPredQT with bLumaOnly (TLibEncoder/TEncSearch.cpp:2386-2710) for a batch of CUs of one class, PU after PU:
//   k_is_prep     most probable modes (getIntraDirLumaPredictor, TComDataCU.cpp:1772-1830: a decided PU of this CU or the caller's direction), the rough-search
//                 and candidate-list jobs of the PU
//   hop_launch_intra + hop_launch_intra_modes   35 SATDs, mode bits from the CI_CURR_BEST state, the candidate list (:2430-2493)
//   per candidate (bCheckFirst) and once more for the best (:2507-2590): k_is_pick (the direction of this pass), hop_launch_intra_rqt from the CI_CURR_BEST
//                 state, k_is_keep (a better cost: arrays, levels and the PU's picture block kept aside - xSetIntraResultQT)
//   k_is_commit   the kept arrays back, the decided block into the picture unless it is the last PU (:2603-2660), the cbf of an NxN CU combined (:2667-2685)
// CUs with fewer candidates than the class maximum sit the spare passes out (empty slots in every batch of the pass).
// =====================================================================================================================
struct IsWork { double best_cost; uint32_t best_dist; int32_t best_mode; uint8_t tr[256], cbf[256], ts[256]; };

__device__ static inline void is_prep_body(const int i, const RqtClass& k, int pu, int nxn, const hop_rqt_job* jobs, const hop_intra_search_job* sj, const hop_intra_rqt_opt* opt,
                                           const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_in, const hop_intra_search_result* sres, hop_intra_cu_syntax* syn, hop_intra_job* rj,
                                           hop_intra_modes_job* mj, IsWork* work) {
  const int parts = 1 << (2 * (k.log2_cu - 2)), q = parts >> (2 * nxn), part = pu * q, N = (1 << k.log2_cu) >> nxn;
  const int left = (nxn && (pu & 1)) ? sres[i].best_dir[pu - 1] : sj[i].left_dir[pu], above = (nxn && (pu & 2)) ? sres[i].best_dir[pu - 2] : sj[i].above_dir[pu];
  int p0, p1, p2, mpm;
  if (left == above) { mpm = 1; if (left > 1) { p0 = left; p1 = ((left + 29) % 32) + 2; p2 = ((left - 1) % 32) + 2; } else { p0 = 0; p1 = 1; p2 = 26; } }
  else { mpm = 2; p0 = left; p1 = above; p2 = (left && above) ? 0 : ((left + above) < 2 ? 26 : 1); }
  hop_intra_cu_syntax* y = syn + i;
  y->preds[pu][0] = p0; y->preds[pu][1] = p1; y->preds[pu][2] = p2; y->pred_num[pu] = 3;
  y->tr_depth = nxn; y->part = part; y->b_luma = 1; y->b_chroma = 0;
  hop_intra_job r;
  r.x = jobs[i].x + rqt_zx(part); r.y = jobs[i].y + rqt_zy(part); r.size = N; r.strong = opt[i].strong;
  for (int u = 0; u < 68; u++) r.flags[u] = sj[i].rough_flags[pu][u];
  rj[i] = r;
  hop_intra_modes_job m;
  m.preds[0] = p0; m.preds[1] = p1; m.preds[2] = p2; m.pred_num = 3; m.mpm_cand = mpm; m.num_full_rd = sj[i].num_full_rd;
  const int ci = jobs[i].ctx_index;
  m.ctx_state = cu_in[ci].state[16]; m.frac_left = (int)ctx_in[ci].state[150] | ((int)ctx_in[ci].state[151] << 8); m.sqrt_lambda = sj[i].sqrt_lambda;
  mj[i] = m;
  work[i].best_cost = 1.7e+308; work[i].best_dist = 0; work[i].best_mode = 0;
}
__global__ void k_is_prep(RqtClass k, int pu, int nxn, const hop_rqt_job* __restrict__ jobs, const hop_intra_search_job* __restrict__ sj, const hop_intra_rqt_opt* __restrict__ opt, int n,
                          const hop_cabac_ctx* __restrict__ ctx_in, const hop_cabac_cu_ctx* __restrict__ cu_in, const hop_intra_search_result* __restrict__ sres,
                          hop_intra_cu_syntax* __restrict__ syn, hop_intra_job* __restrict__ rj, hop_intra_modes_job* __restrict__ mj, IsWork* __restrict__ work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) is_prep_body(i, k, pu, nxn, jobs, sj, opt, ctx_in, cu_in, sres, syn, rj, mj, work);
}

// pass < n_max: candidate `pass` of the list (CUs whose list is shorter sit the pass out); pass == n_max: the best mode so far
__device__ static inline void is_pick_body(const int i, int pu, int pass, int n_max, const hop_intra_modes_result* mres, const IsWork* work, hop_intra_cu_syntax* syn, uint8_t* active) {
  const int cnt = (int)mres[i].n;
  active[i] = (pass == n_max || pass < cnt) ? 1 : 0;
  syn[i].luma_dir[pu] = pass == n_max ? work[i].best_mode : (int)mres[i].modes[pass < cnt ? pass : cnt - 1];
}
__global__ void k_is_pick(int pu, int pass, int n_max, int n, const hop_intra_modes_result* __restrict__ mres, const IsWork* __restrict__ work, hop_intra_cu_syntax* __restrict__ syn,
                          uint8_t* __restrict__ active) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) is_pick_body(i, pu, pass, n_max, mres, work, syn, active);
}

// (every thread of the workgroup calls: a barrier inside)
__device__ static inline void is_keep_body(const int i, const int t, const int nt, const RqtClass& k, int pu, int nxn, int pass, int n_max, const hop_rqt_job* jobs,
                                           const hop_intra_modes_result* mres, const hop_intra_cu_syntax* syn, const hop_rqt_result* tmp, const int32_t* coef_tmp, const int16_t* rec, int pitch,
                                           IsWork* work, int32_t* coef_out, int16_t* reco_out) {
  const bool active = pass == n_max || pass < (int)mres[i].n;
  if (!active || !(tmp[i].cost < work[i].best_cost)) return;           // uniform per block; best_cost is written after the barrier below
  const int cu = 1 << k.log2_cu, parts = 1 << (2 * (k.log2_cu - 2)), q = parts >> (2 * nxn), part = pu * q, N = cu >> nxn, x0 = rqt_zx(part), y0 = rqt_zy(part);
  const size_t cu2 = (size_t)cu * cu, cb = (size_t)i * (cu2 + (cu2 >> 1));
  for (int e = t; e < q; e += nt) { work[i].tr[part + e] = tmp[i].tr_idx[part + e]; work[i].cbf[part + e] = tmp[i].cbf[0][part + e]; work[i].ts[part + e] = tmp[i].tskip[0][part + e]; }
  for (int e = t; e < 16 * q; e += nt) coef_out[cb + (size_t)16 * part + e] = coef_tmp[cb + (size_t)16 * part + e];
  const int16_t* pic = rec + (size_t)(jobs[i].y + y0) * pitch + jobs[i].x + x0;
  for (int e = t; e < N * N; e += nt) { const int rr = e / N, cc = e % N; reco_out[(size_t)i * cu2 + (size_t)(y0 + rr) * cu + x0 + cc] = pic[(size_t)rr * pitch + cc]; }
  __syncthreads();
  if (t == 0) { work[i].best_cost = tmp[i].cost; work[i].best_dist = tmp[i].dist; work[i].best_mode = syn[i].luma_dir[pu]; }
}
__global__ __launch_bounds__(64) void k_is_keep(RqtClass k, int pu, int nxn, int pass, int n_max, const hop_rqt_job* __restrict__ jobs, int n,
                                                const hop_intra_modes_result* __restrict__ mres, const hop_intra_cu_syntax* __restrict__ syn, const hop_rqt_result* __restrict__ tmp,
                                                const int32_t* __restrict__ coef_tmp, const int16_t* __restrict__ rec, int pitch, IsWork* __restrict__ work,
                                                int32_t* __restrict__ coef_out, int16_t* __restrict__ reco_out) {
  is_keep_body(blockIdx.x, threadIdx.x, 64, k, pu, nxn, pass, n_max, jobs, mres, syn, tmp, coef_tmp, rec, pitch, work, coef_out, reco_out);
}

// (every thread of the workgroup calls: barriers inside)
__device__ static inline void is_commit_body(const int i, const int t, const int nt, const RqtClass& k, int pu, int nxn, const hop_rqt_job* jobs, const IsWork* work,
                                             const hop_intra_modes_result* mres, hop_intra_cu_syntax* syn, hop_rqt_result* res, hop_intra_search_result* sres, int16_t* rec, int pitch,
                                             const int16_t* reco_out) {
  const int cu = 1 << k.log2_cu, parts = 1 << (2 * (k.log2_cu - 2)), q = parts >> (2 * nxn), part = pu * q, N = cu >> nxn, x0 = rqt_zx(part), y0 = rqt_zy(part), npu = nxn ? 4 : 1;
  for (int e = t; e < q; e += nt) { res[i].tr_idx[part + e] = work[i].tr[part + e]; res[i].cbf[0][part + e] = work[i].cbf[part + e]; res[i].tskip[0][part + e] = work[i].ts[part + e]; }
  if (pu != npu - 1) {
    int16_t* pic = rec + (size_t)(jobs[i].y + y0) * pitch + jobs[i].x + x0;
    for (int e = t; e < N * N; e += nt) { const int rr = e / N, cc = e % N; pic[(size_t)rr * pitch + cc] = reco_out[(size_t)i * ((size_t)cu * cu) + (size_t)(y0 + rr) * cu + x0 + cc]; }
  }
  if (t == 0) {
    sres[i].best_dir[pu] = work[i].best_mode; sres[i].n_cand[pu] = (int)mres[i].n; syn[i].luma_dir[pu] = work[i].best_mode;
    sres[i].dist = (pu ? sres[i].dist : 0u) + work[i].best_dist;
    res[i].dist = sres[i].dist; res[i].cost = 0; res[i].bits = 0;
  }
  if (npu > 1 && pu == npu - 1) {
    __syncthreads();
    unsigned comb = 0;
    for (int p = 0; p < 4; p++) comb |= (res[i].cbf[0][p * q] >> 1) & 1u;
    __syncthreads();
    for (int e = t; e < parts; e += nt) res[i].cbf[0][e] |= (uint8_t)comb;
  }
}
__global__ __launch_bounds__(64) void k_is_commit(RqtClass k, int pu, int nxn, const hop_rqt_job* __restrict__ jobs, int n, const IsWork* __restrict__ work,
                                                  const hop_intra_modes_result* __restrict__ mres, hop_intra_cu_syntax* __restrict__ syn, hop_rqt_result* __restrict__ res,
                                                  hop_intra_search_result* __restrict__ sres, int16_t* __restrict__ rec, int pitch, const int16_t* __restrict__ reco_out) {
  is_commit_body(blockIdx.x, threadIdx.x, 64, k, pu, nxn, jobs, work, mres, syn, res, sres, rec, pitch, reco_out);
}

size_t hop_intra_search_work_bytes(int log2_cu, int n) {
  const size_t cu2 = (size_t)1 << (2 * log2_cu);
  return hop_intra_rqt_work_bytes(log2_cu, n) + (size_t)n * (sizeof(hop_intra_cu_syntax) + sizeof(hop_intra_job) + sizeof(hop_intra_modes_job) + sizeof(hop_intra_modes_result) + 35 * 4 +
                                                             sizeof(IsWork) + sizeof(hop_rqt_result) + (cu2 + (cu2 >> 1)) * 4 + 1) + 16 * 256;
}

// one class of CUs: size, transform-tree limits / flags, partition (2Nx2N or NxN) and the candidate count of the PU size; buf = hop_intra_search_work_bytes
int hop_launch_intra_search(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int nxn, int num_full_rd, int n, const hop_rqt_job* d_jobs,
                            const hop_intra_cu_syntax* d_syn_in, const hop_intra_rqt_opt* d_opt, const hop_intra_search_job* d_sj, const hop_cabac_ctx* d_ctx_in,
                            const hop_cabac_cu_ctx* d_cu_in, hop_intra_search_result* d_sres, hop_rqt_result* d_res, int32_t* d_coef_out, int16_t* d_reco_out, void* vbuf,
                            size_t buf_bytes, hop_intra_cu_syntax* d_syn_out /* may be NULL: the syntax elements with the decided directions and their predictors */) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = sign_hide; k.use_ts = use_ts;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu2 = (size_t)1 << (2 * log2_cu), rq = al(hop_intra_rqt_work_bytes(log2_cu, n));
  char* buf = (char*)vbuf;
  size_t o = rq;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  hop_intra_cu_syntax* syn = (hop_intra_cu_syntax*)take((size_t)n * sizeof(hop_intra_cu_syntax));
  hop_intra_job* rj = (hop_intra_job*)take((size_t)n * sizeof(hop_intra_job));
  hop_intra_modes_job* mj = (hop_intra_modes_job*)take((size_t)n * sizeof(hop_intra_modes_job));
  hop_intra_modes_result* mres = (hop_intra_modes_result*)take((size_t)n * sizeof(hop_intra_modes_result));
  uint32_t* satd = (uint32_t*)take((size_t)n * 35 * 4);
  IsWork* work = (IsWork*)take((size_t)n * sizeof(IsWork));
  hop_rqt_result* tmp = (hop_rqt_result*)take((size_t)n * sizeof(hop_rqt_result));
  int32_t* coef_tmp = (int32_t*)take((size_t)n * (cu2 + (cu2 >> 1)) * 4);
  uint8_t* active = (uint8_t*)take((size_t)n);
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "intra search: work buffer too small");
  hipError_t e = hipMemcpyAsync(syn, d_syn_in, (size_t)n * sizeof(hop_intra_cu_syntax), hipMemcpyDeviceToDevice, c->stream);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra search: %s", hipGetErrorString(e));
  const int g256 = (n + 255) / 256, npu = nxn ? 4 : 1, n_max = num_full_rd + 2, pitch = c->pic_w;
  e = hipMemsetAsync(d_res, 0, (size_t)n * sizeof(hop_rqt_result), c->stream);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra search: %s", hipGetErrorString(e));
  for (int pu = 0; pu < npu; pu++) {
    hipLaunchKernelGGL(k_is_prep, dim3(g256), dim3(256), 0, c->stream, k, pu, nxn, d_jobs, d_sj, d_opt, n, d_ctx_in, d_cu_in, d_sres, syn, rj, mj, work);
    int r = hop_launch_intra(c, n, rj, satd); if (r) return r;
    r = hop_launch_intra_modes(c, n, mj, satd, mres); if (r) return r;
    for (int pass = 0; pass <= n_max; pass++) {
      hipLaunchKernelGGL(k_is_pick, dim3(g256), dim3(256), 0, c->stream, pu, pass, n_max, n, mres, work, syn, active);
      r = hop_launch_intra_rqt(c, log2_cu, log2_max_tu, log2_min_tu, sign_hide, use_ts, nxn, pass < n_max ? 1 : 0, n, d_jobs, syn, d_opt, d_ctx_in, d_cu_in, tmp, coef_tmp, nullptr,
                               nullptr, buf, rq, active);
      if (r) return r;
      hipLaunchKernelGGL(k_is_keep, dim3(n), dim3(64), 0, c->stream, k, pu, nxn, pass, n_max, d_jobs, n, mres, syn, tmp, coef_tmp, c->rec[0], pitch, work, d_coef_out, d_reco_out);
    }
    hipLaunchKernelGGL(k_is_commit, dim3(n), dim3(64), 0, c->stream, k, pu, nxn, d_jobs, n, work, mres, syn, d_res, d_sres, c->rec[0], pitch, d_reco_out);
  }
  if (d_syn_out) {
    e = hipMemcpyAsync(d_syn_out, syn, (size_t)n * sizeof(hop_intra_cu_syntax), hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra search: %s", hipGetErrorString(e));
  }
  e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The chroma intra search of a CU: TEncSearch::estIntraPredChromaQT (TLibEncoder/TEncSearch.cpp:2720-2785) with xRecurIntraChromaCodingQT (:2130-2277) for a batch of
// CUs of one class.  Per allowed direction (k_ic_mode) the host walks every node a luma tree of the class can have; at a node the CUs whose tree has a transform unit
// there take part (the others are empty slots of the batches):
//   k_ic_begin     prediction job (both planes; flags of the node, per 2-sample unit) and the leaf jobs of the component - for blocks that may skip the transform also
//                  the transform-skip variant -, the coder of the CU as snapshot (CI_QT_TRAFO_ROOT)
//   hop_launch_intra_pred_chroma (once per node), hop_launch_tu_rd (is_intra; the transform-skip variant first, its block parked by k_ic_park)
//   k_ic_single    one lane per CU: cbf, distortion; for transform-skip candidates the two costs from xGetIntraBitsQTChroma (levels of the block through the counting
//                  coder), the better variant kept and the coder moved on (:2176-2247) - Cb first, Cr from the state Cb left
//   k_ic_fold      cbf of the children folded upwards (:2262-2275)
// then k_ic_bits (the CU's chroma bits from the CI_CURR_BEST state, the cost, the comparison) and k_ic_keep (xSetIntraResultChromaQT: levels, reconstruction, arrays).
// =====================================================================================================================
struct IcWork { double best_cost; uint32_t best_dist, dist; int32_t best_mode, mode, keep, dir; uint8_t cbf[2][256], ts[2][256]; };

__device__ static inline bool ic_leaf_here(const RqtClass& k, const hop_rqt_result* r, int part, int d) {   // does the CU's luma tree have the chroma block of this node?
  if (r->tr_idx[part] != d) return false;
  if (k.log2_cu - d == 2) { const int parts = 1 << (2 * (k.log2_cu - 2)); return (part % (parts >> (2 * (d - 1)))) == 0; }
  return true;
}
__device__ static inline bool ic_ts_here(const RqtClass& k, const hop_rqt_result* r, int part, int d, int ts_fast) {   // checkTransformSkip, :2158-2174
  const int log2 = k.log2_cu - d;
  if (!k.use_ts || log2 > 3) return false;
  if (!ts_fast) return true;
  if (log2 >= 3) return false;
  int nb = 0; for (int p = part; p < part + 4; p++) nb += r->tskip[0][p];
  return nb > 0;
}

__device__ static inline void ic_mode_body(const int i, int m, const hop_rqt_job* jobs, const hop_cabac_ctx* ctx_in, hop_cabac_ctx* cur, hop_intra_cu_syntax* syn, IcWork* work) {
  int list[5] = { 0, 26, 10, 1, 36 };                                   // getAllowedChromaDir, TComDataCU.cpp:1746-1764
  for (int q = 0; q < 4; q++) if (syn[i].luma_dir[0] == list[q]) { list[q] = 34; break; }
  const int mode = list[m];
  syn[i].chroma_is_dm = mode == 36; syn[i].chroma_dir = mode;
  cur[i] = ctx_in[jobs[i].ctx_index];
  IcWork* w = work + i;
  if (m == 0) { w->best_cost = 1.7e+308; w->best_dist = 0; w->best_mode = 0; }
  w->dist = 0; w->mode = mode; w->keep = 0; w->dir = mode == 36 ? syn[i].luma_dir[0] : mode;
}
__global__ void k_ic_mode(int m, const hop_rqt_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, hop_cabac_ctx* __restrict__ cur,
                          hop_intra_cu_syntax* __restrict__ syn, IcWork* __restrict__ work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ic_mode_body(i, m, jobs, ctx_in, cur, syn, work);
}

__device__ static inline void ic_begin_body(const int i, const RqtClass& k, const RqtNode& nd, int comp, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn, const hop_intra_rqt_opt* opt,
                                            int bd_c, const hop_cabac_ctx* cur, hop_cabac_ctx* root, const hop_rqt_result* res, const IcWork* work, hop_intra_job* pj, int32_t* modes,
                                            hop_tu_rd_job* tuj, int64_t* off, hop_tu_rd_job* tuj2, int64_t* off2, size_t ts_base) {
  hop_intra_job q0; memset(&q0, 0, sizeof(q0));
  hop_tu_rd_job j0; memset(&j0, 0, sizeof(j0));
  const hop_rqt_result* r = res + i;
  if (!ic_leaf_here(k, r, nd.part, nd.d)) { if (comp == 1) { pj[i] = q0; modes[i] = 0; } tuj[i] = j0; off[i] = 0; tuj2[i] = j0; off2[i] = 0; return; }
  root[i] = cur[i];
  const hop_rqt_job jb = jobs[i];
  const int parts = 1 << (2 * (k.log2_cu - 2));
  const int da = nd.log2 == 2 ? nd.d - 1 : nd.d, lgc = k.log2_cu - da - 1, N = 1 << lgc;
  if (comp == 1) {
    hop_intra_job q = q0;
    q.x = jb.x + rqt_zx(nd.part); q.y = jb.y + rqt_zy(nd.part); q.size = N; q.strong = 0;
    const unsigned long long av = opt[i].avail[irq_node_index(da, k.log2_cu - da, nd.part)];
    for (int u = 0; u < 4 * (N / 2) + 1 && u < 68; u++) q.flags[u] = (uint8_t)((av >> u) & 1ull);
    pj[i] = q; modes[i] = work[i].dir;
  }
  hop_tu_rd_job j = j0;
  j.x = jb.x + rqt_zx(nd.part); j.y = jb.y + rqt_zy(nd.part); j.comp = comp; j.log2_size = lgc; j.qp_scaled = jb.qp_scaled[comp]; j.tr_depth = nd.d; j.ctx_index = i;
  j.sign_hide = k.sign_hide; j.use_ts = k.use_ts; j.bit_depth = bd_c; j.is_intra = 1; j.scan_idx = icu_scan(syn[i], parts, nd.part, lgc, comp); j.use_dst = 0; j.flags = 0;
  j.lambda_rdoq = jb.lambda_rdoq[comp]; j.lambda_rd = jb.lambda_rd; j.dist_weight = jb.dist_weight[comp - 1];
  tuj[i] = j; off[i] = (int64_t)rqt_coef_at(k, i, k.log2_max_tu - nd.log2, comp, nd.part);
  if (ic_ts_here(k, r, nd.part, nd.d, opt[i].ts_fast)) { j.flags = HOP_TU_RD_TS; tuj2[i] = j; off2[i] = (int64_t)(ts_base + (size_t)i * 16); }
  else { tuj2[i] = j0; off2[i] = 0; }
}
__global__ void k_ic_begin(RqtClass k, RqtNode nd, int comp, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn, const hop_intra_rqt_opt* __restrict__ opt,
                           int n, int bd_c, const hop_cabac_ctx* __restrict__ cur, hop_cabac_ctx* __restrict__ root, const hop_rqt_result* __restrict__ res,
                           const IcWork* __restrict__ work, hop_intra_job* __restrict__ pj, int32_t* __restrict__ modes, hop_tu_rd_job* __restrict__ tuj, int64_t* __restrict__ off,
                           hop_tu_rd_job* __restrict__ tuj2, int64_t* __restrict__ off2, size_t ts_base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ic_begin_body(i, k, nd, comp, jobs, syn, opt, bd_c, cur, root, res, work, pj, modes, tuj, off, tuj2, off2, ts_base);
}

// the 4x4 block of the transform-skip variant, picture -> park
__device__ static inline void ic_park_body(const int i, const int e, const RqtClass& k, const RqtNode& nd, const hop_rqt_job* jobs, const hop_intra_rqt_opt* opt, const hop_rqt_result* res,
                                           const int16_t* rec, int pitch, int16_t* park) {
  if (!ic_leaf_here(k, res + i, nd.part, nd.d) || !ic_ts_here(k, res + i, nd.part, nd.d, opt[i].ts_fast)) return;
  park[(size_t)i * 16 + e] = rec[(size_t)(((jobs[i].y + rqt_zy(nd.part)) >> 1) + (e >> 2)) * pitch + ((jobs[i].x + rqt_zx(nd.part)) >> 1) + (e & 3)];
}
__global__ __launch_bounds__(64) void k_ic_park(RqtClass k, RqtNode nd, const hop_rqt_job* __restrict__ jobs, const hop_intra_rqt_opt* __restrict__ opt, int n,
                                                const hop_rqt_result* __restrict__ res, const int16_t* __restrict__ rec, int pitch, int16_t* __restrict__ park) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 4), e = threadIdx.x & 15;
  if (i < n) ic_park_body(i, e, k, nd, jobs, opt, res, rec, pitch, park);
}

template <class LDS>
__device__ static void ic_single_body(LDS& sh, const int lane, const int i, const RqtClass& k, const RqtNode& nd, int comp, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn,
                                      const hop_intra_rqt_opt* opt, hop_cabac_ctx* cur, const hop_cabac_ctx* root, hop_rqt_result* res, IcWork* work, const hop_tu_rd_result* tr,
                                      const hop_tu_rd_result* tr2, int32_t* coef, size_t ts_base, int16_t* rec, int pitch, const int16_t* park, const uint16_t* scans) {
  hop_rqt_result* r = res + i;
  if (!ic_leaf_here(k, r, nd.part, nd.d)) return;
  IcWork* w = work + i;
  const int parts = 1 << (2 * (k.log2_cu - 2)), da = nd.log2 == 2 ? nd.d - 1 : nd.d, nparts = parts >> (2 * da), lgc = k.log2_cu - da - 1, part = nd.part;
  const uint32_t cbf0 = tr[i].abs_sum ? 1u : 0u;
  for (int p = 0; p < nparts; p++) { r->cbf[comp][part + p] = (uint8_t)(cbf0 << nd.d); r->tskip[comp][part + p] = 0; }
  if (!ic_ts_here(k, r, nd.part, nd.d, opt[i].ts_fast)) { w->dist += tr[i].dist; return; }
  const double lambda = jobs[i].lambda_rd;
  const int scan = icu_scan(syn[i], parts, part, lgc, comp);
  int32_t* lv = coef + rqt_coef_at(k, i, k.log2_max_tu - nd.log2, comp, part); int32_t* tv = coef + ts_base + (size_t)i * 16;
  RQ_LOAD(root[i]);
  unsigned long long frac = RQ_LEFT();
  frac += cb_code_tu(sh, lane, lv, lgc, 1, scan, k.sign_hide, k.use_ts, 0, 0, scans);
  double cost = rqt_cost((uint32_t)(frac >> 15), tr[i].dist, lambda);
  uint32_t dist = tr[i].dist;
  if (tr2[i].abs_sum) {
    rqt_store(sh, lane, frac, cur + i);                                 // CI_TEMP_BEST
    RQ_LOAD(root[i]);
    unsigned long long f1 = RQ_LEFT();
    f1 += cb_code_tu(sh, lane, tv, lgc, 1, scan, k.sign_hide, k.use_ts, 1, 0, scans);
    const double cost1 = rqt_cost((uint32_t)(f1 >> 15), tr2[i].dist, lambda);
    if (cost1 < cost) {
      dist = tr2[i].dist;
      for (int e = 0; e < 16; e++) lv[e] = tv[e];
      for (int p = 0; p < nparts; p++) { r->cbf[comp][part + p] = (uint8_t)(1u << nd.d); r->tskip[comp][part + p] = 1; }
      int16_t* pic = rec + (size_t)((jobs[i].y + rqt_zy(part)) >> 1) * pitch + ((jobs[i].x + rqt_zx(part)) >> 1);
      for (int e = 0; e < 16; e++) pic[(size_t)(e >> 2) * pitch + (e & 3)] = park[(size_t)i * 16 + e];
      rqt_store(sh, lane, f1, cur + i);
    }
  } else rqt_store(sh, lane, frac, cur + i);
  w->dist += dist;
}
template <class LDS>
__global__ __launch_bounds__(64) void k_ic_single(RqtClass k, RqtNode nd, int comp, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn,
                                                  const hop_intra_rqt_opt* __restrict__ opt, int n, hop_cabac_ctx* __restrict__ cur, const hop_cabac_ctx* __restrict__ root,
                                                  hop_rqt_result* __restrict__ res, IcWork* __restrict__ work, const hop_tu_rd_result* __restrict__ tr,
                                                  const hop_tu_rd_result* __restrict__ tr2, int32_t* __restrict__ coef, size_t ts_base, int16_t* __restrict__ rec, int pitch,
                                                  const int16_t* __restrict__ park, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  ic_single_body(sh, lane, i, k, nd, comp, jobs, syn, opt, cur, root, res, work, tr, tr2, coef, ts_base, rec, pitch, park, scans);
}

__device__ static inline void ic_fold_body(const int i, const RqtClass& k, const RqtNode& nd, hop_rqt_result* res) {
  hop_rqt_result* r = res + i;
  if (r->tr_idx[nd.part] <= nd.d) return;
  const int parts = 1 << (2 * (k.log2_cu - 2)), nparts = parts >> (2 * nd.d), q = nparts >> 2;
  unsigned su = 0, sv = 0;
  for (int kk = 0; kk < 4; kk++) { su |= (r->cbf[1][nd.part + kk * q] >> (nd.d + 1)) & 1u; sv |= (r->cbf[2][nd.part + kk * q] >> (nd.d + 1)) & 1u; }
  for (int p = 0; p < nparts; p++) { r->cbf[1][nd.part + p] |= (uint8_t)(su << nd.d); r->cbf[2][nd.part + p] |= (uint8_t)(sv << nd.d); }
}
__global__ void k_ic_fold(RqtClass k, RqtNode nd, int n, hop_rqt_result* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ic_fold_body(i, k, nd, res);
}

template <class LDS>
__device__ static void ic_bits_body(LDS& sh, const int lane, const int i, const RqtClass& k, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn, const hop_cabac_ctx* ctx_in,
                                    const hop_cabac_cu_ctx* cu_in, const hop_rqt_result* res, IcWork* work, const int32_t* coef, const uint16_t* scans) {
  const int ci = jobs[i].ctx_index;
  RQ_LOAD(ctx_in[ci]); IRQ_CU_LOAD(cu_in[ci]);
  unsigned long long frac = RQ_LEFT();
  frac += icu_count(sh, lane, k, syn[i], 0, 0, 0, 1, res + i, nullptr, coef, i, scans);
  IcWork* w = work + i;
  const double cost = rqt_cost((uint32_t)(frac >> 15), w->dist, jobs[i].lambda_rd);
  if (cost < w->best_cost) { w->best_cost = cost; w->best_dist = w->dist; w->best_mode = w->mode; w->keep = 1; }
}
template <class LDS>
__global__ __launch_bounds__(64) void k_ic_bits(RqtClass k, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn, int n,
                                                const hop_cabac_ctx* __restrict__ ctx_in, const hop_cabac_cu_ctx* __restrict__ cu_in, const hop_rqt_result* __restrict__ res,
                                                IcWork* __restrict__ work, const int32_t* __restrict__ coef, const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  ic_bits_body(sh, lane, i, k, jobs, syn, ctx_in, cu_in, res, work, coef, scans);
}

// xSetIntraResultChromaQT (:2280-2345) for the CUs whose direction of this pass is the best so far
__device__ static inline void ic_keep_body(const int i, const int t, const int nt, const RqtClass& k, const hop_rqt_job* jobs, const hop_rqt_result* res, IcWork* work, const int32_t* coef,
                                           const int16_t* rec_cb, const int16_t* rec_cr, int pitch, int32_t* coef_out, int16_t* reco_out) {
  if (!work[i].keep) return;
  const int cu = 1 << k.log2_cu, half = cu >> 1, parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)cu * cu, h2 = cu2 >> 2;
  const hop_rqt_result* r = res + i;
  for (int e = t; e < parts; e += nt) { work[i].cbf[0][e] = r->cbf[1][e]; work[i].cbf[1][e] = r->cbf[2][e]; work[i].ts[0][e] = r->tskip[1][e]; work[i].ts[1][e] = r->tskip[2][e]; }
  for (int comp = 1; comp <= 2; comp++) {
    for (int e = t; e < (int)h2; e += nt) {                            // chroma level e of the CU layout: partition e / 4; its block starts at the first partition of the chroma TU
      const int p = e >> 2, d = r->tr_idx[p], log2 = k.log2_cu - d, dd = log2 == 2 ? d - 1 : d, np = parts >> (2 * dd), first = p - p % np;
      coef_out[(size_t)i * (cu2 + 2 * h2) + cu2 + (size_t)(comp - 1) * h2 + e] = coef[rqt_coef_at(k, i, k.log2_max_tu - log2, comp, first) + (size_t)(e - 4 * first)];
    }
    const int16_t* pic = (comp == 1 ? rec_cb : rec_cr) + (size_t)(jobs[i].y >> 1) * pitch + (jobs[i].x >> 1);
    for (int e = t; e < (int)h2; e += nt) { const int rr = e / half, cc = e % half; reco_out[(size_t)i * 2 * h2 + (size_t)(comp - 1) * h2 + e] = pic[(size_t)rr * pitch + cc]; }
  }
}
__global__ __launch_bounds__(64) void k_ic_keep(RqtClass k, const hop_rqt_job* __restrict__ jobs, int n, const hop_rqt_result* __restrict__ res, IcWork* __restrict__ work,
                                                const int32_t* __restrict__ coef, const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr, int pitch,
                                                int32_t* __restrict__ coef_out, int16_t* __restrict__ reco_out) {
  ic_keep_body(blockIdx.x, threadIdx.x, 64, k, jobs, res, work, coef, rec_cb, rec_cr, pitch, coef_out, reco_out);
}

__device__ static inline void ic_commit_body(const int i, const int t, const int nt, const RqtClass& k, const IcWork* work, hop_rqt_result* res, hop_intra_chroma_result* cres,
                                             hop_intra_cu_syntax* syn_update) {
  const int parts = 1 << (2 * (k.log2_cu - 2));
  for (int e = t; e < parts; e += nt) { res[i].cbf[1][e] = work[i].cbf[0][e]; res[i].cbf[2][e] = work[i].cbf[1][e]; res[i].tskip[1][e] = work[i].ts[0][e]; res[i].tskip[2][e] = work[i].ts[1][e]; }
  if (t == 0) {
    cres[i].best_mode = work[i].best_mode; cres[i].dist = work[i].best_dist;
    if (syn_update) { syn_update[i].chroma_is_dm = work[i].best_mode == 36; syn_update[i].chroma_dir = work[i].best_mode; }   // setChromIntraDirSubParts (:2779)
  }
}
__global__ __launch_bounds__(64) void k_ic_commit(RqtClass k, int n, const IcWork* __restrict__ work, hop_rqt_result* __restrict__ res, hop_intra_chroma_result* __restrict__ cres,
                                                  hop_intra_cu_syntax* __restrict__ syn_update) {
  ic_commit_body(blockIdx.x, threadIdx.x, 64, k, work, res, cres, syn_update);
}

size_t hop_intra_chroma_work_bytes(int log2_cu, int n) {
  const size_t cu2 = (size_t)1 << (2 * log2_cu);
  return (size_t)n * (2 * sizeof(hop_cabac_ctx) + sizeof(hop_intra_cu_syntax) + sizeof(IcWork) + sizeof(hop_intra_job) + 4 + 2 * (sizeof(hop_tu_rd_job) + 8 + sizeof(hop_tu_rd_result)) +
                     (6 * cu2 + 16) * 4 + 32) + 32 * 256;
}

// one class of CUs (size, transform-tree limits / flags); d_res: tr_idx and tskip[0] as the luma search left them (read), cbf[1..2] / tskip[1..2] (written)
int hop_launch_intra_chroma_search(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs,
                                   const hop_intra_cu_syntax* d_syn_in, const hop_intra_rqt_opt* d_opt, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                                   hop_rqt_result* d_res, hop_intra_chroma_result* d_cres, int32_t* d_coef_out, int16_t* d_reco_out, void* vbuf, size_t buf_bytes,
                                   hop_intra_cu_syntax* d_syn_update /* may be NULL: receives the decided chroma direction */) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = sign_hide; k.use_ts = use_ts;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu2 = (size_t)1 << (2 * log2_cu), n_coeff = (size_t)n * (6 * cu2 + 16), ts_base = (size_t)n * 6 * cu2;
  char* buf = (char*)vbuf; size_t o = 0;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  struct Bufs { hop_cabac_ctx *cur, *root; hop_intra_cu_syntax* syn; IcWork* work; hop_intra_job* pj; int32_t* modes; hop_tu_rd_job *tuj, *tuj2; int64_t *off, *off2;
                hop_tu_rd_result *tr, *tr2; int32_t* coef; int16_t* park; size_t n_coeff, ts_base; } B;
  B.cur = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); B.root = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx));
  B.syn = (hop_intra_cu_syntax*)take((size_t)n * sizeof(hop_intra_cu_syntax)); B.work = (IcWork*)take((size_t)n * sizeof(IcWork));
  B.pj = (hop_intra_job*)take((size_t)n * sizeof(hop_intra_job)); B.modes = (int32_t*)take((size_t)n * 4);
  B.tuj = (hop_tu_rd_job*)take((size_t)n * sizeof(hop_tu_rd_job)); B.tuj2 = (hop_tu_rd_job*)take((size_t)n * sizeof(hop_tu_rd_job));
  B.off = (int64_t*)take((size_t)n * 8); B.off2 = (int64_t*)take((size_t)n * 8);
  B.tr = (hop_tu_rd_result*)take((size_t)n * sizeof(hop_tu_rd_result)); B.tr2 = (hop_tu_rd_result*)take((size_t)n * sizeof(hop_tu_rd_result));
  B.coef = (int32_t*)take(n_coeff * 4); B.park = (int16_t*)take((size_t)n * 32);
  B.n_coeff = n_coeff; B.ts_base = ts_base;
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "intra chroma search: work buffer too small");
  hipError_t e = hipMemcpyAsync(B.syn, d_syn_in, (size_t)n * sizeof(hop_intra_cu_syntax), hipMemcpyDeviceToDevice, c->stream);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra chroma search: %s", hipGetErrorString(e));
  e = hipMemsetAsync(B.coef, 0, n_coeff * 4, c->stream);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra chroma search: %s", hipGetErrorString(e));
  const int g256 = (n + 255) / 256, pitch = c->pic_w >> 1;
  struct Rec { static int go(hop_ctx* c, const RqtClass& k, int n, const hop_rqt_job* d_jobs, const hop_intra_rqt_opt* d_opt, hop_rqt_result* d_res, Bufs& B, int part, int d, int log2) {
    RqtNode nd; memset(&nd, 0, sizeof(nd));
    nd.part = part; nd.d = d; nd.log2 = log2;
    const int g64 = (n + 63) / 64, g256 = (n + 255) / 256, pitch = c->pic_w >> 1;
    const bool may_ts = k.use_ts && log2 <= 3;
    if (log2 <= k.log2_max_tu && !(log2 == 2 && (part & 3))) {          // a transform unit can sit here (4x4 luma: the chroma block belongs to the first of four)
      for (int comp = 1; comp <= 2; comp++) {
        hipLaunchKernelGGL(k_ic_begin, dim3(g256), dim3(256), 0, c->stream, k, nd, comp, d_jobs, B.syn, d_opt, n, c->bd_c, B.cur, B.root, d_res, B.work, B.pj, B.modes, B.tuj, B.off,
                           B.tuj2, B.off2, B.ts_base);
        int r;
        if (comp == 1) { r = hop_launch_intra_pred_chroma(c, n, B.pj, B.modes); if (r) return r; }
        if (may_ts) {
          r = hop_launch_tu_rd(c, n, B.tuj2, B.cur, B.off2, B.n_coeff, B.coef, B.tr2, 1); if (r) return r;
          hipLaunchKernelGGL(k_ic_park, dim3((n + 3) / 4), dim3(64), 0, c->stream, k, nd, d_jobs, d_opt, n, d_res, c->rec[comp], pitch, B.park);
        }
        const int lgc = log2 == 2 ? 2 : log2 - 1;
        r = hop_launch_tu_rd(c, n, B.tuj, B.cur, B.off, B.n_coeff, B.coef, B.tr, lgc <= 3 ? 1 : 0); if (r) return r;
        RQT_LAUNCH(k_ic_single, c, n, k, nd, comp, d_jobs, B.syn, d_opt, n, B.cur, B.root, d_res, B.work, B.tr, B.tr2, B.coef, B.ts_base,
                           c->rec[comp], pitch, B.park, c->rdoq_scans);
      }
    }
    if (log2 > k.log2_min_tu) {
      const int q = ((1 << (2 * (k.log2_cu - 2))) >> (2 * d)) >> 2;
      for (int kk = 0; kk < 4; kk++) { const int r = go(c, k, n, d_jobs, d_opt, d_res, B, part + kk * q, d + 1, log2 - 1); if (r) return r; }
      hipLaunchKernelGGL(k_ic_fold, dim3(g256), dim3(256), 0, c->stream, k, nd, n, d_res);
    }
    (void)g64;
    return HOP_OK;
  } };
  for (int m = 0; m < 5; m++) {
    hipLaunchKernelGGL(k_ic_mode, dim3(g256), dim3(256), 0, c->stream, m, d_jobs, n, d_ctx_in, B.cur, B.syn, B.work);
    const int rc = Rec::go(c, k, n, d_jobs, d_opt, d_res, B, 0, 0, k.log2_cu);
    if (rc) return rc;
    RQT_LAUNCH(k_ic_bits, c, n, k, d_jobs, B.syn, n, d_ctx_in, d_cu_in, d_res, B.work, B.coef, c->rdoq_scans);
    hipLaunchKernelGGL(k_ic_keep, dim3(n), dim3(64), 0, c->stream, k, d_jobs, n, d_res, B.work, B.coef, c->rec[1], c->rec[2], pitch, d_coef_out, d_reco_out);
  }
  hipLaunchKernelGGL(k_ic_commit, dim3(n), dim3(64), 0, c->stream, k, n, B.work, d_res, d_cres, d_syn_update);
  e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra chroma search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The bits of a finished intra CU as TEncCu::xCheckRDCostIntra counts them (TLibEncoder/TEncCu.cpp:1483-1503): skip flag, prediction mode, partition size, the luma
// directions of all PUs and the chroma direction (encodePredInfo), then encodeCoeff = xEncodeTransform (TEncEntropy.cpp:219-420) on the CU's final levels with the
// intra rules (the split of an NxN CU inferred, the luma cbf always coded, scans by direction).  One lane per CU; cost = calcRdCost(bits, distortion).
// =====================================================================================================================
template <class LDS>
__device__ static void intra_cu_total_body(LDS& sh, const int lane, const int i, const RqtClass& k, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syn, const hop_rqt_result* res,
                                           const int32_t* coef, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_in, const uint32_t* dist, uint32_t* bits_out, double* cost_out,
                                           hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_out, const uint16_t* scans) {
  const int ci = jobs[i].ctx_index;
  RQ_LOAD(ctx_in[ci]); IRQ_CU_LOAD(cu_in[ci]);
  const hop_intra_cu_syntax y = syn[i];
  const hop_rqt_result* r = res + i;
  const int parts = 1 << (2 * (k.log2_cu - 2));
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  const int32_t* cf = coef + (size_t)i * (cu2 + (cu2 >> 1));
  unsigned long long frac = RQ_LEFT();
  if (y.skip_ctx >= 0) {                                             // skip_ctx < 0: an I slice
    CBIN(CU_SKIP + y.skip_ctx, y.skip_flag ? 1 : 0);
    CBIN(CU_PRED, 1);
  }
  if (y.is_min_cu) CBIN(CU_PART, y.part_nxn ? 0 : 1);
  for (int p = 0; p < (y.part_nxn ? 4 : 1); p++) frac += icu_dir(sh, lane, y.luma_dir[p], y.preds[p], y.pred_num[p]);
  if (y.chroma_is_dm) CBIN(CU_CPRED, 0); else { CBIN(CU_CPRED, 1); CEP(2); }
  int sp_part[4], sp_k[4]; int sp = 0, bak = 0;
  sp_part[0] = 0; sp_k[0] = -1;
  while (sp >= 0) {
    const int part = sp_part[sp], trIdx = sp, log2 = k.log2_cu - sp;
    if (sp_k[sp] < 0) {
      const int subdiv = r->tr_idx[part] > trIdx;
      const int cbfY = (r->cbf[0][part] >> trIdx) & 1; int cbfU = (r->cbf[1][part] >> trIdx) & 1, cbfV = (r->cbf[2][part] >> trIdx) & 1;
      if (log2 == 2) {
        const int pn = parts >> (2 * (trIdx - 1));
        if (part % pn == 0) bak = part;
        else if (part % pn == pn - 1) { cbfU = (r->cbf[1][bak] >> trIdx) & 1; cbfV = (r->cbf[2][bak] >> trIdx) & 1; }
      }
      if (!((y.part_nxn && trIdx == 0) || log2 > k.log2_max_tu || log2 == 2 || log2 == k.log2_min_tu)) CBIN(CX_TRANS_SUBDIV + 5 - log2, subdiv);
      const int first = trIdx == 0;
      if (first || log2 > 2) {
        if (first || ((r->cbf[1][part] >> (trIdx - 1)) & 1)) CBIN(rqt_cbf_ctx(1, trIdx), (r->cbf[1][part] >> trIdx) & 1);
        if (first || ((r->cbf[2][part] >> (trIdx - 1)) & 1)) CBIN(rqt_cbf_ctx(2, trIdx), (r->cbf[2][part] >> trIdx) & 1);
      }
      if (!subdiv) {
        CBIN(rqt_cbf_ctx(0, r->tr_idx[part]), (r->cbf[0][part] >> r->tr_idx[part]) & 1);
        if (cbfY) frac += cb_code_tu(sh, lane, cf + 16 * part, log2, 0, icu_scan(y, parts, part, log2, 0), k.sign_hide, k.use_ts, r->tskip[0][part], 0, scans);
        if (log2 > 2) {
          if (cbfU) frac += cb_code_tu(sh, lane, cf + cu2 + 4 * part, log2 - 1, 1, icu_scan(y, parts, part, log2 - 1, 1), k.sign_hide, k.use_ts, r->tskip[1][part], 0, scans);
          if (cbfV) frac += cb_code_tu(sh, lane, cf + cu2 + (cu2 >> 2) + 4 * part, log2 - 1, 1, icu_scan(y, parts, part, log2 - 1, 2), k.sign_hide, k.use_ts, r->tskip[2][part], 0, scans);
        } else {
          const int pn = parts >> (2 * (trIdx - 1));
          if (part % pn == pn - 1) {
            if (cbfU) frac += cb_code_tu(sh, lane, cf + cu2 + 4 * bak, 2, 1, icu_scan(y, parts, bak, 2, 1), k.sign_hide, k.use_ts, r->tskip[1][bak], 0, scans);
            if (cbfV) frac += cb_code_tu(sh, lane, cf + cu2 + (cu2 >> 2) + 4 * bak, 2, 1, icu_scan(y, parts, bak, 2, 2), k.sign_hide, k.use_ts, r->tskip[2][bak], 0, scans);
          }
        }
        sp--; continue;
      }
      sp_k[sp] = 0;
    }
    if (sp_k[sp] < 4) { const int q = (parts >> (2 * trIdx)) >> 2, kk = sp_k[sp]++; sp_part[sp + 1] = part + kk * q; sp_k[sp + 1] = -1; sp++; }
    else sp--;
  }
  const uint32_t bits = (uint32_t)(frac >> 15);
  bits_out[i] = bits;
  if (cost_out) cost_out[i] = rqt_cost(bits, dist ? dist[i] : 0u, jobs[i].lambda_rd);
  if (ctx_out) rqt_store(sh, lane, frac, ctx_out + i);
  if (cu_out) IRQ_CU_STORE(cu_out[i]);
}
template <class LDS>
__global__ __launch_bounds__(64) void k_intra_cu_total(RqtClass k, int n, const hop_rqt_job* __restrict__ jobs, const hop_intra_cu_syntax* __restrict__ syn,
                                                       const hop_rqt_result* __restrict__ res, const int32_t* __restrict__ coef, const hop_cabac_ctx* __restrict__ ctx_in,
                                                       const hop_cabac_cu_ctx* __restrict__ cu_in, const uint32_t* __restrict__ dist, uint32_t* __restrict__ bits_out,
                                                       double* __restrict__ cost_out, hop_cabac_ctx* __restrict__ ctx_out, hop_cabac_cu_ctx* __restrict__ cu_out,
                                                       const uint16_t* __restrict__ scans) {
  __shared__ LDS sh;
  RQT_LANE_OR_BLOCK(n);
  if (i >= n) return;
  intra_cu_total_body(sh, lane, i, k, jobs, syn, res, coef, ctx_in, cu_in, dist, bits_out, cost_out, ctx_out, cu_out, scans);
}

int hop_launch_intra_cu_total(hop_ctx* c, int log2_cu, int log2_max_tu, int log2_min_tu, int sign_hide, int use_ts, int n, const hop_rqt_job* d_jobs, const hop_intra_cu_syntax* d_syn,
                              const hop_rqt_result* d_res, const int32_t* d_coef, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, const uint32_t* d_dist,
                              uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out) {
  RqtClass k; k.log2_cu = log2_cu; k.log2_max_tu = log2_max_tu; k.log2_min_tu = log2_min_tu; k.inter_split = 0; k.sign_hide = sign_hide; k.use_ts = use_ts;
  const int pr = hop_prof_begin(c, HOP_K_CABAC, (uint64_t)n);
  RQT_LAUNCH(k_intra_cu_total, c, n, k, n, d_jobs, d_syn, d_res, d_coef, d_ctx_in, d_cu_in, d_dist, d_bits, d_cost, d_ctx_out, d_cu_out,
                     c->rdoq_scans);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_cu_total launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// The residual-free candidate of an SS/GT CU: TEncSearch::encodeResAndCalcRdInterCU with bSkipRes (TLibEncoder/TEncSearch.cpp:6635-6668).  One workgroup per CU: the
// three SSEs of the prediction picture against the original (getDistPart: chroma weighted), the prediction copied into the reconstruction picture, then lane 0 counts
// the skip flag and the merge index from the CI_CURR_BEST state and forms calcRdCost.
// =====================================================================================================================
__global__ __launch_bounds__(64) void k_cu_skip(int n, const hop_rqt_job* __restrict__ jobs, const hop_cu_syntax* __restrict__ syn, hop_pics pic, int16_t* __restrict__ rec_y,
                                                int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr, const hop_cabac_ctx* __restrict__ ctx_in,
                                                const hop_cabac_cu_ctx* __restrict__ cu_in, hop_cu_final* __restrict__ fin, uint32_t* __restrict__ bits_out, double* __restrict__ cost_out,
                                                hop_cabac_ctx* __restrict__ ctx_out, hop_cabac_cu_ctx* __restrict__ cu_out) {
  __shared__ CabacLds sh;
  __shared__ unsigned long long s_sse[3];
  const int i = blockIdx.x, lane = threadIdx.x;
  const hop_rqt_job jb = jobs[i];
  const int cu = 1 << jb.log2_cu;
  if (lane < 3) s_sse[lane] = 0;
  __syncthreads();
  for (int c = 0; c < 3; c++) {
    const int w = c ? cu >> 1 : cu, pitch = c ? pic.pic_w >> 1 : pic.pic_w, x0 = c ? jb.x >> 1 : jb.x, y0 = c ? jb.y >> 1 : jb.y, bd = c ? pic.bd_c : pic.bd_y;
    const int16_t* org = (c == 0 ? pic.org_y : c == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
    const int16_t* prd = (c == 0 ? pic.pred_y : c == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
    int16_t* rec = (c == 0 ? rec_y : c == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0;
    const int shift = bd > 8 ? (bd - 8) << 1 : 0;                      // DISTORTION_PRECISION_ADJUSTMENT, as k_distortion
    unsigned long long acc = 0;
    for (int e = lane; e < w * w; e += 64) {
      const int r = e / w, q = e % w; const int d = (int)org[(size_t)r * pitch + q] - (int)prd[(size_t)r * pitch + q];
      acc += (unsigned long long)((unsigned)(d * d) >> shift);
      rec[(size_t)r * pitch + q] = prd[(size_t)r * pitch + q];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) s_sse[c] = acc;
  }
  __syncthreads();
  if (lane != 0) return;
  const uint32_t dY = (uint32_t)s_sse[0], dU = (uint32_t)(int)(jb.dist_weight[0] * (uint32_t)s_sse[1]), dV = (uint32_t)(int)(jb.dist_weight[1] * (uint32_t)s_sse[2]);
  const int ci = jb.ctx_index;
  RQ_LOAD(ctx_in[ci]);
  for (int q = 0; q < 20; q++) sh.st[CUX + q][lane] = cu_in[ci].state[q];
  unsigned long long frac = RQ_LEFT();
  CBIN(CU_SKIP + syn[i].skip_ctx, 1);
  frac += cu_merge_index(sh, lane, syn[i].pu[0].merge_idx, syn[i].max_merge_cand);
  const uint32_t bits = (uint32_t)(frac >> 15);
  bits_out[i] = bits;
  cost_out[i] = rqt_cost(bits, dY + dU + dV, jb.lambda_rd);
  fin[i].root_cbf = 0; fin[i].dist[0] = dY; fin[i].dist[1] = dU; fin[i].dist[2] = dV;
  if (ctx_out) rqt_store(sh, lane, frac, ctx_out + i);
  if (cu_out) for (int q = 0; q < 20; q++) cu_out[i].state[q] = sh.st[CUX + q][lane];
}

int hop_launch_cu_skip(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syn, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, hop_cu_final* d_fin,
                       uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out) {
  const int pr = hop_prof_begin(c, HOP_K_CABAC, (uint64_t)n);
  hipLaunchKernelGGL(k_cu_skip, dim3(n), dim3(64), 0, c->stream, n, d_jobs, d_syn, hop_make_pics(c), c->rec[0], c->rec[1], c->rec[2], d_ctx_in, d_cu_in, d_fin, d_bits, d_cost, d_ctx_out,
                     d_cu_out);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "cu_skip launch: %s", hipGetErrorString(e));
  return HOP_OK;
}


// getTotalDistortion of an intra candidate: luma search + chroma search
__global__ void k_intra_dist_sum(int n, const hop_intra_search_result* __restrict__ sres, const hop_intra_chroma_result* __restrict__ cres, uint32_t* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dist[i] = sres[i].dist + cres[i].dist;
}
int hop_launch_intra_dist_sum(hop_ctx* c, int n, const hop_intra_search_result* d_sres, const hop_intra_chroma_result* d_cres, uint32_t* d_dist) {
  hipLaunchKernelGGL(k_intra_dist_sum, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_sres, d_cres, d_dist);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_dist_sum launch: %s", hipGetErrorString(e));
  return HOP_OK;
}


// getTotalCost of an SS/GT candidate with residual: calcRdCost(bits of xAddSymbolBitsInter, the three final distortions) (TEncSearch.cpp:6804-6812)
__global__ void k_inter_cost(int n, const hop_rqt_job* __restrict__ jobs, const hop_cu_final* __restrict__ fin, const uint32_t* __restrict__ bits, double* __restrict__ cost) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cost[i] = rqt_cost(bits[i], fin[i].dist[0] + fin[i].dist[1] + fin[i].dist[2], jobs[i].lambda_rd);
}
int hop_launch_inter_cost(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_cu_final* d_fin, const uint32_t* d_bits, double* d_cost) {
  hipLaunchKernelGGL(k_inter_cost, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_jobs, d_fin, d_bits, d_cost);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "inter_cost launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
