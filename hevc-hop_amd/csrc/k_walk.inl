// k_walk.inl -- candidate evaluations as ONE kernel per candidate (included by k_cabac.hip after k_leaf_fused.inl and k_rqt.inl, whose device bodies it calls).
//
// The batch-step orchestration of k_rqt.inl lets the HOST walk a candidate's transform tree and launches a few small kernels per node, each over all candidates of the
// batch: right for batches of thousands of candidates (the node's kernels are wide), wrong for the RD spine (host/hop_spine.cpp), whose batches hold ONE candidate per CTU
// in flight and whose time is the length of the dependent launch chain -- about 100 launches for an SS/GT candidate, 250 for an intra candidate, each with one working lane.
// Here a WORKGROUP takes its candidate through the whole evaluation: the tree is walked on the device (the same recursion, the same node descriptors, the same order), the
// per-node stages are the bodies of the batch kernels called one after the other with workgroup barriers in between, and everything a candidate needs stays where the batch
// form keeps it (per-candidate work areas in HBM, indexed by the candidate), so the results are the batch form's by construction.
//   k_inter_walk   TEncSearch::encodeResAndCalcRdInterCU of a candidate whose prediction is in the prediction picture (TLibEncoder/TEncSearch.cpp:6622-6822): xEstimateResidualQT
//                  (:6824-7560) = rqt_init / begin / leaf / single / close / final, the root-cbf-zero test and the reconstruction (:6700-6723, :6804-6812) = fin_*, the CU's
//                  syntax bits (xAddSymbolBitsInter :7779-7810) = cu_bits, calcRdCost
// (the intra candidate walk follows the same scheme)

// The walks are serial code on a few lanes: what they must not do is keep the short kernels of the spine's other requests (predictions, searches) off the compute units.
// At the compiler's free choice they take 256 registers per lane -- two workgroups fill a compute unit's register files, and a few hundred candidates in flight the whole
// chip; capped at 128 (four waves per SIMD) a compute unit holds four and still has room.
#ifndef WALK_WAVES_PER_SIMD
#define WALK_WAVES_PER_SIMD 4
#endif
struct InterWalk {
  RqtClass k; int n, bd_y, bd_c, wave_leaves; hop_pics pic;
  const hop_rqt_job* jobs; const hop_cu_syntax* syn; const hop_cabac_ctx* ctx_in; const hop_cabac_cu_ctx* cu_in;
  hop_rqt_result* res; int32_t* coef_out; hop_cabac_ctx* ctx_after; hop_cu_final* fin; uint32_t* bits; uint32_t* skipped; double* cost; hop_cabac_ctx* ctx_out; hop_cabac_cu_ctx* cu_out;
  // the quadtree's state (rqt_run_class)
  hop_cabac_ctx* cur; hop_cabac_ctx* root[4]; hop_cabac_ctx* test[4]; RqtWork* work; hop_tu_rd_job* tuj; hop_tu_rd_job* tuj2; int64_t* off; int64_t* off2;
  hop_tu_rd_result* tr; hop_tu_rd_result* tr2; int32_t* coef; size_t n_coeff, ts_base;
  // the leaf step's scratch (hop_launch_tu_rd_fused): one slot per transform unit of a node, 3 per candidate
  int32_t* lcoef; uint32_t* zs; uint32_t* ns; uint32_t* as; unsigned long long* fr; hop_rdoq_job* rq; hop_coeff_bits_job* cb; char* lwork;
  // the wrapper's tail (hop_launch_rqt_finish)
  int d0, d1, nj; hop_tu_rd_job* ftuj; int64_t* foff; uint32_t* faf; uint32_t* fsse;
  const int32_t* entropy_bits; const uint16_t* scans; int16_t* rec_y; int16_t* rec_cb; int16_t* rec_cr;
};

__device__ static inline RqtNode walk_rqt_node(const RqtClass& k, int parts, int part, int d, int log2, int zero_open) {      // as Rec::go of rqt_run_class
  RqtNode nd; nd.part = part; nd.d = d; nd.log2 = log2;
  nd.check_full = (k.inter_split && d == 0 && log2 > k.log2_min_tu) ? 0 : (log2 <= k.log2_max_tu);
  nd.check_split = log2 > k.log2_min_tu;
  nd.code_chroma = 1;
  if (log2 == 2) nd.code_chroma = (part % (parts >> (2 * (d - 1)))) == 0;
  nd.add_zero = zero_open && nd.check_full;
  nd.ts_y = (k.use_ts && nd.check_full && log2 == 2) ? 1 : 0;
  nd.ts_c = (k.use_ts && nd.check_full && nd.code_chroma && (log2 == 2 || log2 == 3)) ? 1 : 0;
  return nd;
}

// the inverse path in reconstruction mode over a candidate's table of nj jobs (hop_launch_tu_recon): the large transform units one after the other on the whole workgroup,
// the 4x4 / 8x8 ones a wave each
__device__ static void walk_recon_jobs(LeafShared& L, const int first, const int nj, const int total, const hop_tu_rd_job* jobs, hop_pics pic, const int64_t* off, const int32_t* levels,
                                       const uint32_t* abs_flag, uint32_t* sse, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int q = 0; q < nj; q++) {
    if (jobs[first + q].log2_size <= 3) continue;                      // (uniform over the workgroup)
    turd_inverse_body(L.t.big, first + q, jobs, pic, off, levels, abs_flag, sse, rec_y, rec_cb, rec_cr);
    __syncthreads();
  }
  for (int q0 = 0; q0 < nj; q0 += 4) {
    const int q = q0 + wave;
    if (q < nj) turd_inverse_small_body(L.t.small, wave, lane, first + q, jobs, total, pic, off, levels, abs_flag, sse, rec_y, rec_cb, rec_cr);
  }
  __syncthreads();
}

// the transform units of one node of an SS/GT candidate (its Y / Cb / Cr blocks in tuj, their 4x4 transform-skip variants in tuj2; candidate i's slots 3 i ..): the 16x16 /
// 32x32 ones one after the other on the whole workgroup, the 4x4 / 8x8 ones four at a time, a wave each.  None of them writes a picture (residual-domain distortion), so they
// are independent; the transform-skip variants use the second half of the scratch tables
struct InterLeafShared { union { LeafShared big; LeafSmallShared small[4]; } u; };
__device__ static void walk_inter_leaves(const InterWalk& A, InterLeafShared& L, const int i, const int d, const int ncomp, const int nts) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n3 = 3 * A.n;
  for (int c = 0; c < ncomp; c++) {
    if (A.tuj[3 * i + c].log2_size <= 3) continue;                      // (uniform over the workgroup)
    turd_fused_body(L.u.big, 3 * i + c, A.tuj, n3, A.pic, A.root[d], A.off, A.entropy_bits, A.scans, A.lcoef, A.coef, A.zs, A.ns, A.as, A.fr, A.rq, A.cb, A.lwork, A.tr,
                    A.rec_y, A.rec_cb, A.rec_cr);
    __syncthreads();
  }
  if (!A.wave_leaves) {
  // one after the other (HOP_WALK_WAVE_LEAVES=0)
  for (int c = 0; c < ncomp; c++) {
    if (A.tuj[3 * i + c].log2_size > 3) continue;
    turd_fused_body(L.u.big, 3 * i + c, A.tuj, n3, A.pic, A.root[d], A.off, A.entropy_bits, A.scans, A.lcoef, A.coef, A.zs, A.ns, A.as, A.fr, A.rq, A.cb, A.lwork, A.tr,
                    A.rec_y, A.rec_cb, A.rec_cr);
    __syncthreads();
  }
  for (int t = 0; t < nts; t++) {
    turd_fused_body(L.u.big, 3 * i + t, A.tuj2, n3, A.pic, A.root[d], A.off2, A.entropy_bits, A.scans, A.lcoef, A.coef, A.zs + n3, A.ns + n3, A.as + n3, A.fr + n3, A.rq + n3, A.cb + n3,
                    A.lwork, A.tr2, A.rec_y, A.rec_cb, A.rec_cr);
    __syncthreads();
  }
  return;
  }
  // the small ones: list position q = 0 .. ncomp + nts - 1 -> (table, slot)
  const int total = ncomp + nts;
  for (int q0 = 0; q0 < total; q0 += 4) {
    const int q = q0 + wave;
    const bool ts = q >= ncomp;
    int j = -1;
    if (q < total) { j = 3 * i + (ts ? q - ncomp : q); if (!ts && A.tuj[j].log2_size > 3) j = -1; }
    // ONE call site that every wave reaches with all its lanes (the tables picked per wave): the first form -- two calls under `if (ts)`, the body a function of its own
    // called from wave-divergent control flow -- faulted on the device (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION)
    const int sh = ts ? n3 : 0;
    turd_fused_small_wave_body(L.u.small[wave], lane, j, ts ? A.tuj2 : A.tuj, n3, A.pic, A.root[d], ts ? A.off2 : A.off, A.entropy_bits, A.scans, A.lcoef, A.coef, A.zs + sh, A.ns + sh,
                               A.as + sh, A.fr + sh, A.rq + sh, A.cb + sh, ts ? A.tr2 : A.tr, A.rec_y, A.rec_cb, A.rec_cr);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_inter_walk(InterWalk A) {
  __shared__ InterLeafShared IL;
  LeafShared& L = IL.u.big;
  __shared__ CabacLds1 sh1;
  const int i = blockIdx.x, tid = threadIdx.x;
  const RqtClass k = A.k;
  const int parts = 1 << (2 * (k.log2_cu - 2));
  if (tid == 0) rqt_init_body(i, A.jobs, A.ctx_in, A.cur, A.work, A.res);
  __syncthreads();
  // ---- xEstimateResidualQT: the reference's recursion (rqt_run_class), node by node ----
  int s_part[4], s_log2[4], s_zero[4], s_child[4];
  int sp = 0;
  s_part[0] = 0; s_log2[0] = k.log2_cu; s_zero[0] = 1; s_child[0] = -1;
  while (sp >= 0) {
    const int d = sp;
    const RqtNode nd = walk_rqt_node(k, parts, s_part[sp], d, s_log2[sp], s_zero[sp]);
    if (s_child[sp] < 0) {                                               // entering the node
      // The batch form keeps a node's transform units at [ncomp * candidate + component] (every candidate is at the same node); here candidates are at different nodes at any
      // moment, so every candidate owns three fixed slots of the job / result / scratch tables: the tables are handed to the bodies shifted so that their index lands there
      const int ncomp = nd.code_chroma ? 3 : 1, nts = nd.ts_y + 2 * nd.ts_c;
      const ptrdiff_t sh_c = (ptrdiff_t)(3 - ncomp) * i, sh_t = (ptrdiff_t)(3 - nts) * i;
      if (tid == 0) rqt_begin_body(i, k, nd, A.jobs, A.bd_y, A.bd_c, A.cur, A.root[d], A.res, A.work, A.tuj + sh_c, A.off + sh_c, A.tuj2 + sh_t, A.off2 + sh_t, A.ts_base);
      __syncthreads();
      if (nd.check_full) {
        walk_inter_leaves(A, IL, i, d, ncomp, nts);
        if (tid == 0) rqt_single_body(sh1, 0, i, k, nd, A.jobs, A.cur, A.root[d], A.test[d], A.res, A.work, A.tr + sh_c, A.tr2 + sh_t, A.coef, A.ts_base, A.scans);
        __syncthreads();
      }
      if (!nd.check_split) { sp--; continue; }
      s_child[sp] = 0;
    }
    if (s_child[sp] < 4) {
      const int q = (parts >> (2 * d)) >> 2, kk = s_child[sp]++;
      s_part[sp + 1] = s_part[sp] + kk * q; s_log2[sp + 1] = s_log2[sp] - 1; s_zero[sp + 1] = s_zero[sp] && !nd.check_full; s_child[sp + 1] = -1;
      sp++;
      continue;
    }
    if (tid == 0) rqt_close_body(sh1, 0, i, k, nd, A.jobs, A.cur, A.root[d], A.test[d], A.res, A.work, A.coef, A.scans);
    __syncthreads();
    sp--;
  }
  rqt_final_body(i, tid, 256, k, A.work, A.res, A.coef, A.coef_out, A.cur, A.ctx_after);
  __syncthreads();
  // ---- the wrapper's tail: root-cbf-zero test, the reconstruction and its distortion ----
  if (tid == 0) fin_decide_body(i, k, A.jobs, A.ctx_after, A.entropy_bits, A.res, A.fin);
  __syncthreads();
  fin_zero_body(i, tid, 256, k, A.fin, A.coef_out);
  if (tid == 0) fin_emit_body(i, k, A.jobs, A.bd_y, A.bd_c, A.d0, A.d1, A.nj, A.res, A.fin, A.ftuj, A.foff, A.faf);
  __syncthreads();
  walk_recon_jobs(L, i * A.nj, A.nj, A.n * A.nj, A.ftuj, A.pic, A.foff, A.coef_out, A.faf, A.fsse, A.rec_y, A.rec_cb, A.rec_cr);
  if (tid == 0) {
    fin_sum_body(i, A.jobs, A.nj, A.ftuj, A.fsse, A.fin);
    // ---- the CU's syntax bits from the CI_CURR_BEST state, the cost ----
    cu_bits_body(sh1, 0, i, k, A.jobs, A.syn, A.res, A.coef_out, A.ctx_in, A.cu_in, A.bits, A.skipped, A.ctx_out, A.cu_out, A.scans);
    A.cost[i] = rqt_cost(A.bits[i], A.fin[i].dist[0] + A.fin[i].dist[1] + A.fin[i].dist[2], A.jobs[i].lambda_rd);
  }
}

size_t hop_inter_walk_bytes(int log2_cu, int log2_max_tu, int log2_min_tu, int n) {
  const size_t cu2 = (size_t)1 << (2 * log2_cu), n_coeff = (size_t)n * (6 * cu2 + 48);
  int d0, d1; const int nj = rqt_jobs_per_cu(log2_cu, log2_max_tu, log2_min_tu, &d0, &d1);
  return hop_rqt_work_bytes(log2_cu, n) + (size_t)n * nj * (sizeof(hop_tu_rd_job) + 8 + 4 + 4) + n_coeff * 4 +
         (size_t)3 * n * (2 * (4 * 4 + 8 + sizeof(hop_rdoq_job) + sizeof(hop_coeff_bits_job)) + LEAF_WORK_PER_TU) + 64 * 256;
}

// n candidates of one class through k_inter_walk; every pointer a device pointer; buf = hop_inter_walk_bytes
int hop_launch_inter_walk(hop_ctx* c, const hop_rqt_job* cls, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syn, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in,
                          hop_rqt_result* d_res, int32_t* d_coef, hop_cabac_ctx* d_ctx_after, hop_cu_final* d_fin, uint32_t* d_bits, uint32_t* d_skipped, double* d_cost,
                          hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_out, void* vbuf, size_t buf_bytes) {
  InterWalk A; memset(&A, 0, sizeof(A));
  RqtClass& k = A.k;
  k.log2_cu = cls->log2_cu; k.log2_max_tu = cls->log2_max_tu; k.log2_min_tu = cls->log2_min_tu_in_cu; k.inter_split = cls->inter_split_flag ? 1 : 0; k.sign_hide = cls->sign_hide ? 1 : 0;
  k.use_ts = cls->use_ts ? 1 : 0;
  A.n = n; A.bd_y = c->bd_y; A.bd_c = c->bd_c; A.pic = hop_make_pics(c);
  { const char* e = getenv("HOP_WALK_WAVE_LEAVES"); A.wave_leaves = e ? atoi(e) : 1; }   // the 4x4 / 8x8 transform units of a node a wave each (walk_inter_leaves)
  A.jobs = d_jobs; A.syn = d_syn; A.ctx_in = d_ctx_in; A.cu_in = d_cu_in; A.res = d_res; A.coef_out = d_coef; A.ctx_after = d_ctx_after; A.fin = d_fin; A.bits = d_bits; A.skipped = d_skipped;
  A.cost = d_cost; A.ctx_out = d_ctx_out; A.cu_out = d_cu_out;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu2 = (size_t)1 << (2 * k.log2_cu);
  A.n_coeff = (size_t)n * (6 * cu2 + 48); A.ts_base = (size_t)n * 6 * cu2;
  char* buf = (char*)vbuf; size_t o = 0;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  A.cur = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx));
  for (int d = 0; d < 4; d++) { A.root[d] = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); A.test[d] = (hop_cabac_ctx*)take((size_t)n * sizeof(hop_cabac_ctx)); }
  A.work = (RqtWork*)take((size_t)n * sizeof(RqtWork));
  A.tuj = (hop_tu_rd_job*)take((size_t)3 * n * sizeof(hop_tu_rd_job)); A.tuj2 = (hop_tu_rd_job*)take((size_t)3 * n * sizeof(hop_tu_rd_job));
  A.off = (int64_t*)take((size_t)3 * n * 8); A.off2 = (int64_t*)take((size_t)3 * n * 8);
  A.tr = (hop_tu_rd_result*)take((size_t)3 * n * sizeof(hop_tu_rd_result)); A.tr2 = (hop_tu_rd_result*)take((size_t)3 * n * sizeof(hop_tu_rd_result));
  A.coef = (int32_t*)take(A.n_coeff * 4);
  A.lcoef = (int32_t*)take(A.n_coeff * 4);
  A.zs = (uint32_t*)take((size_t)6 * n * 4); A.ns = (uint32_t*)take((size_t)6 * n * 4); A.as = (uint32_t*)take((size_t)6 * n * 4); A.fr = (unsigned long long*)take((size_t)6 * n * 8);
  A.rq = (hop_rdoq_job*)take((size_t)6 * n * sizeof(hop_rdoq_job)); A.cb = (hop_coeff_bits_job*)take((size_t)6 * n * sizeof(hop_coeff_bits_job));   // (second halves: the transform-skip variants)
  A.lwork = take((size_t)3 * n * LEAF_WORK_PER_TU);
  A.nj = rqt_jobs_per_cu(k.log2_cu, k.log2_max_tu, k.log2_min_tu, &A.d0, &A.d1);
  const size_t nt = (size_t)n * A.nj;
  A.ftuj = (hop_tu_rd_job*)take(nt * sizeof(hop_tu_rd_job)); A.foff = (int64_t*)take(nt * 8); A.faf = (uint32_t*)take(nt * 4); A.fsse = (uint32_t*)take(nt * 4);
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "inter walk: work buffer too small");
  A.entropy_bits = hop_entropy_bits_device(c); A.scans = c->rdoq_scans; A.rec_y = c->rec[0]; A.rec_cb = c->rec[1]; A.rec_cr = c->rec[2];
  const int pr = hop_prof_begin(c, HOP_K_WALK_INTER + (k.log2_cu - 3), (uint64_t)n);
  hipLaunchKernelGGL(k_inter_walk, dim3(n), dim3(256), 0, c->stream, A);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "inter walk launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// =====================================================================================================================
// k_intra_walk: the body of TEncCu::xCheckRDCostIntra for one candidate CU (TLibEncoder/TEncCu.cpp:1455-1507) = hop_launch_intra_search (estIntraPredQT with bLumaOnly,
// TEncSearch.cpp:2386-2710, over hop_launch_intra_rqt = xRecurIntraCodingQT :1361-1710) -> hop_launch_intra_chroma_search (estIntraPredChromaQT :2720-2785) ->
// getTotalDistortion -> hop_launch_intra_cu_total (the CU's bits, calcRdCost), every stage the batch form's body.  The chroma walk visits only the nodes the CU's own luma
// tree has (the batch form visits every node a tree of the class can have and lets the CUs without a transform unit there sit the step out: the same work).
// =====================================================================================================================
struct IntraWalk {
  RqtClass k; int n, nxn, num_full_rd, bd_y, bd_c; hop_pics pic;
  const hop_rqt_job* jobs; const hop_intra_cu_syntax* syn_in; const hop_intra_rqt_opt* opt; const hop_intra_search_job* sj; const hop_cabac_ctx* ctx_in; const hop_cabac_cu_ctx* cu_in;
  hop_intra_search_result* sres; hop_rqt_result* res; hop_intra_chroma_result* cres; int32_t* coef_out; int16_t* reco_y; int16_t* reco_c; hop_intra_cu_syntax* syn_out;
  uint32_t* dist; uint32_t* bits; double* cost; hop_cabac_ctx* ctx_out; hop_cabac_cu_ctx* cu_out;
  // the luma search (hop_launch_intra_search)
  hop_intra_cu_syntax* syn; hop_intra_job* rj; hop_intra_modes_job* mj; hop_intra_modes_result* mres; uint32_t* satd; IsWork* iswork; hop_rqt_result* tmp; int32_t* coef_tmp; uint8_t* active;
  // the luma tree of one pass (hop_launch_intra_rqt)
  hop_cabac_ctx* cur; hop_cabac_cu_ctx* cucur; hop_cabac_ctx* root[4]; hop_cabac_cu_ctx* curoot[4]; hop_cabac_ctx* test[4]; hop_cabac_cu_ctx* cutest[4]; IrqWork* irwork;
  hop_intra_job* pj; int32_t* modes; hop_tu_rd_job* tuj; hop_tu_rd_job* tuj2; int64_t* off; int64_t* off2; hop_tu_rd_result* tr; hop_tu_rd_result* tr2; int32_t* coef;
  int16_t* recl; int16_t* park; size_t n_coeff, ts_base;
  // the chroma search (hop_launch_intra_chroma_search)
  hop_cabac_ctx* ccur; hop_cabac_ctx* croot; hop_intra_cu_syntax* csyn; IcWork* icwork; int32_t* ccoef; int16_t* cpark;
  // the leaf step's scratch: one transform unit per candidate at a time
  int32_t* lcoef; uint32_t* zs; uint32_t* ns; uint32_t* as; unsigned long long* fr; hop_rdoq_job* rq; hop_coeff_bits_job* cb; char* lwork;
  const int32_t* entropy_bits; const uint16_t* scans; int16_t* rec_y; int16_t* rec_cb; int16_t* rec_cr;
  // candidates side by side (k_iw_*): P candidate passes per candidate CU, virtual index v = i * P + pass.  The per-CU work arrays above are sized for V = n * P entries;
  // a virtual candidate works on copies of its CU's inputs and predicts / reconstructs in a BAND of its own: planes of the picture's pitch in which candidate v owns the
  // rows v * hb .. (v + 1) * hb - 1 -- handed to the bodies shifted so that the CU's picture coordinates land there, its neighbouring row and column copied in first (what
  // k_slot_prepare does for a candidate slot)
  int P, V, hb, hbc, pic_h;
  hop_rqt_job* vjobs; hop_intra_cu_syntax* vsyn; hop_intra_rqt_opt* vopt; hop_rqt_result* vtmp; int32_t* vcoef_tmp; hop_rqt_result* vres;
  int16_t* band_pred[3]; int16_t* band_rec[3];
};
union WalkShared { LeafShared leaf; IntraShared intra; };

#define IW_LEAF(JOBS, CTX, OFF, LEVELS, RES) turd_fused_body(L.leaf, i, JOBS, A.n, A.pic, CTX, OFF, A.entropy_bits, A.scans, A.lcoef, LEVELS, A.zs, A.ns, A.as, A.fr, A.rq, A.cb, A.lwork, RES, \
                                                             A.rec_y, A.rec_cb, A.rec_cr)

// one pass of xRecurIntraCodingQT over the PU (hop_launch_intra_rqt with the pass's direction in syn): results into A.tmp / A.coef_tmp
__device__ static void walk_intra_rqt(const IntraWalk& A, WalkShared& L, CabacLds1& sh1, const int i, const int tr_depth0, const int check_first) {
  const int tid = threadIdx.x;
  const RqtClass k = A.k;
  const int parts = 1 << (2 * (k.log2_cu - 2)), pitch = A.pic.pic_w;
  if (tid == 0) irqt_init_body(i, A.jobs, A.ctx_in, A.cu_in, A.cur, A.cucur, A.irwork, A.tmp);
  __syncthreads();
  int s_rel[4], s_child[4];
  int sp = 0;
  s_rel[0] = 0; s_child[0] = -1;
  while (sp >= 0) {
    const int d = tr_depth0 + sp, log2 = k.log2_cu - d;
    RqtNode nd; nd.part = s_rel[sp]; nd.d = d; nd.log2 = log2; nd.code_chroma = 0; nd.add_zero = 0; nd.ts_c = 0;
    nd.check_full = log2 <= k.log2_max_tu;
    nd.check_split = log2 > k.log2_min_tu && !(check_first && nd.check_full);
    nd.ts_y = (k.use_ts && nd.check_full && log2 == 2) ? 1 : 0;
    if (s_child[sp] < 0) {
      if (tid == 0) irqt_begin_body(i, k, nd, A.jobs, A.syn, A.opt, A.bd_y, A.cur, A.cucur, A.root[d], A.curoot[d], A.tmp, A.irwork, A.pj, A.modes, A.tuj, A.off, A.tuj2, A.off2, A.ts_base, nullptr);
      __syncthreads();
      if (nd.check_full) {
        intra_pred_body(L.intra, A.pj + i, A.modes[i], A.pic, A.rec_y);
        __syncthreads();
        if (nd.ts_y) {
          IW_LEAF(A.tuj2, A.root[d], A.off2, A.coef, A.tr2);
          __syncthreads();
          irqt_copy_body(i, tid, 256, 1, k, nd, A.jobs, A.syn, A.irwork, A.rec_y, pitch, A.recl, A.park, nullptr);
          __syncthreads();
        }
        IW_LEAF(A.tuj, A.root[d], A.off, A.coef, A.tr);
        __syncthreads();
        if (nd.check_split) { irqt_copy_body(i, tid, 256, 0, k, nd, A.jobs, A.syn, A.irwork, A.rec_y, pitch, A.recl, A.park, nullptr); __syncthreads(); }
        if (tid == 0) irqt_single_body(sh1, 0, i, k, nd, A.jobs, A.syn, A.opt, A.cur, A.cucur, A.root[d], A.curoot[d], A.test[d], A.cutest[d], A.tmp, A.irwork, A.tr, A.tr2, A.coef, A.ts_base,
                                       A.rec_y, pitch, A.park, A.scans, nullptr);
        __syncthreads();
      }
      if (!nd.check_split) { sp--; continue; }
      s_child[sp] = 0;
    }
    if (s_child[sp] < 4) {
      const int q = (parts >> (2 * d)) >> 2, kk = s_child[sp]++;
      s_rel[sp + 1] = s_rel[sp] + kk * q; s_child[sp + 1] = -1;
      sp++;
      continue;
    }
    if (tid == 0) irqt_close_body(sh1, 0, i, k, nd, A.jobs, A.syn, A.cur, A.cucur, A.root[d], A.curoot[d], A.test[d], A.cutest[d], A.tmp, A.irwork, A.coef, A.scans, nullptr);
    __syncthreads();
    if (nd.check_full) { irqt_copy_body(i, tid, 256, 2, k, nd, A.jobs, A.syn, A.irwork, A.rec_y, pitch, A.recl, A.park, nullptr); __syncthreads(); }
    sp--;
  }
  irqt_final_body(i, tid, 256, k, tr_depth0, A.syn, A.irwork, A.tmp, A.coef, A.coef_tmp, A.cur, A.cucur, nullptr, nullptr, nullptr);
  __syncthreads();
}

// ---- the phases of an intra candidate, each on the whole workgroup, index i into the arrays of A ----
// estIntraPredQT of PU pu up to the candidate list (:2430-2493)
__device__ static void iw_luma_prep(const IntraWalk& A, WalkShared& L, const int i, const int pu) {
  if (threadIdx.x == 0) is_prep_body(i, A.k, pu, A.nxn, A.jobs, A.sj, A.opt, A.ctx_in, A.cu_in, A.sres, A.syn, A.rj, A.mj, A.iswork);
  __syncthreads();
  intra_rough_body(L.intra, A.rj + i, A.pic, A.rec_y, A.satd + (size_t)i * 35);
  __syncthreads();
  if (threadIdx.x == 0) intra_modes_body(i, A.mj, A.satd, A.mres);
  __syncthreads();
}
// the best mode again with the full tree, the better result kept, the decided PU into the picture (:2555-2660)
__device__ static void iw_luma_final(const IntraWalk& A, WalkShared& L, CabacLds1& sh1, const int i, const int pu) {
  const int tid = threadIdx.x, n_max = A.num_full_rd + 2, pitch = A.pic.pic_w;
  if (tid == 0) is_pick_body(i, pu, n_max, n_max, A.mres, A.iswork, A.syn, A.active);
  __syncthreads();
  walk_intra_rqt(A, L, sh1, i, A.nxn, 0);
  is_keep_body(i, tid, 256, A.k, pu, A.nxn, n_max, n_max, A.jobs, A.mres, A.syn, A.tmp, A.coef_tmp, A.rec_y, pitch, A.iswork, A.coef_out, A.reco_y);
  __syncthreads();
  is_commit_body(i, tid, 256, A.k, pu, A.nxn, A.jobs, A.iswork, A.mres, A.syn, A.res, A.sres, A.rec_y, pitch, A.reco_y);
  __syncthreads();
}
// one direction of estIntraPredChromaQT along the luma tree in A.res[i] (xRecurIntraChromaCodingQT :2130-2277), then the CU's chroma bits and cost (ic_bits)
__device__ static void iw_chroma_mode(const IntraWalk& A, WalkShared& L, CabacLds1& sh1, const int i, const int m) {
  const int tid = threadIdx.x;
  const RqtClass k = A.k;
  const int parts = 1 << (2 * (k.log2_cu - 2)), pitch_c = A.pic.pic_w >> 1;
  const hop_rqt_result* r = A.res + i;
  if (tid == 0) ic_mode_body(i, m, A.jobs, A.ctx_in, A.ccur, A.csyn, A.icwork);
  __syncthreads();
  int s_part[4], s_child[4];
  int sp = 0;
  s_part[0] = 0; s_child[0] = -1;
  while (sp >= 0) {
    const int d = sp, log2 = k.log2_cu - d, part = s_part[sp];
    RqtNode nd; nd.part = part; nd.d = d; nd.log2 = log2; nd.code_chroma = 0; nd.add_zero = 0; nd.ts_c = 0; nd.ts_y = 0; nd.check_full = 0; nd.check_split = 0;
    const int tr = r->tr_idx[part];
    if (s_child[sp] < 0) {
      if (tr == d && log2 <= k.log2_max_tu && !(log2 == 2 && (part & 3))) {       // a transform unit of this CU sits here (ic_leaf_here)
        const bool may_ts = k.use_ts && log2 <= 3;
        for (int comp = 1; comp <= 2; comp++) {
          if (tid == 0) ic_begin_body(i, k, nd, comp, A.jobs, A.csyn, A.opt, A.bd_c, A.ccur, A.croot, A.res, A.icwork, A.pj, A.modes, A.tuj, A.off, A.tuj2, A.off2, A.ts_base);
          __syncthreads();
          if (comp == 1) {
            intra_pred_chroma_body(L.intra, A.pj + i, A.modes[i], 1, A.pic, A.rec_cb, A.rec_cr);
            __syncthreads();
            intra_pred_chroma_body(L.intra, A.pj + i, A.modes[i], 2, A.pic, A.rec_cb, A.rec_cr);
            __syncthreads();
          }
          int16_t* recc = comp == 1 ? A.rec_cb : A.rec_cr;
          if (may_ts) {
            IW_LEAF(A.tuj2, A.ccur, A.off2, A.ccoef, A.tr2);
            __syncthreads();
            if (tid < 16) ic_park_body(i, tid, k, nd, A.jobs, A.opt, A.res, recc, pitch_c, A.cpark);
            __syncthreads();
          }
          IW_LEAF(A.tuj, A.ccur, A.off, A.ccoef, A.tr);
          __syncthreads();
          if (tid == 0) ic_single_body(sh1, 0, i, k, nd, comp, A.jobs, A.csyn, A.opt, A.ccur, A.croot, A.res, A.icwork, A.tr, A.tr2, A.ccoef, A.ts_base, recc, pitch_c, A.cpark, A.scans);
          __syncthreads();
        }
      }
      if (!(tr > d && log2 > k.log2_min_tu)) { sp--; continue; }                   // the tree goes no deeper here
      s_child[sp] = 0;
    }
    if (s_child[sp] < 4) {
      const int q = (parts >> (2 * d)) >> 2, kk = s_child[sp]++;
      s_part[sp + 1] = part + kk * q; s_child[sp + 1] = -1;
      sp++;
      continue;
    }
    if (tid == 0) ic_fold_body(i, k, nd, A.res);
    __syncthreads();
    sp--;
  }
  if (tid == 0) ic_bits_body(sh1, 0, i, k, A.jobs, A.csyn, A.ctx_in, A.cu_in, A.res, A.icwork, A.ccoef, A.scans);
  __syncthreads();
}
__device__ static inline void iw_zero_ccoef(const IntraWalk& A, const int i) {
  const size_t cu2 = (size_t)1 << (2 * A.k.log2_cu);
  int32_t* z = A.ccoef + (size_t)i * 6 * cu2;
  for (size_t e = threadIdx.x; e < 6 * cu2; e += 256) z[e] = 0;
  if (threadIdx.x < 16) A.ccoef[A.ts_base + (size_t)i * 16 + threadIdx.x] = 0;
}
// getTotalDistortion, the CU's bits from the CI_CURR_BEST state, calcRdCost
__device__ static inline void iw_total(const IntraWalk& A, CabacLds1& sh1, const int i) {
  if (threadIdx.x == 0) {
    A.dist[i] = A.sres[i].dist + A.cres[i].dist;
    intra_cu_total_body(sh1, 0, i, A.k, A.jobs, A.syn_out, A.res, A.coef_out, A.ctx_in, A.cu_in, A.dist, A.bits, A.cost, A.ctx_out, A.cu_out, A.scans);
  }
}

// ---- everything on one workgroup per candidate CU, the candidate passes one after the other ----
__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_intra_walk(IntraWalk A) {
  __shared__ WalkShared L;
  __shared__ CabacLds1 sh1;
  const int i = blockIdx.x, tid = threadIdx.x;
  const RqtClass k = A.k;
  const int pitch = A.pic.pic_w, pitch_c = A.pic.pic_w >> 1;
  if (tid == 0) A.syn[i] = A.syn_in[i];
  { uint32_t* z = (uint32_t*)(A.res + i); for (int e = tid; e < (int)(sizeof(hop_rqt_result) / 4); e += 256) z[e] = 0; }
  __syncthreads();
  const int npu = A.nxn ? 4 : 1, n_max = A.num_full_rd + 2;
  for (int pu = 0; pu < npu; pu++) {
    iw_luma_prep(A, L, i, pu);
    for (int pass = 0; pass < n_max; pass++) {
      if (tid == 0) is_pick_body(i, pu, pass, n_max, A.mres, A.iswork, A.syn, A.active);
      __syncthreads();
      if (!A.active[i]) continue;                                        // this CU's list is shorter (uniform over the workgroup)
      walk_intra_rqt(A, L, sh1, i, A.nxn, 1);
      is_keep_body(i, tid, 256, k, pu, A.nxn, pass, n_max, A.jobs, A.mres, A.syn, A.tmp, A.coef_tmp, A.rec_y, pitch, A.iswork, A.coef_out, A.reco_y);
      __syncthreads();
    }
    iw_luma_final(A, L, sh1, i, pu);
  }
  if (tid == 0) { A.syn_out[i] = A.syn[i]; A.csyn[i] = A.syn[i]; }
  iw_zero_ccoef(A, i);
  __syncthreads();
  for (int m = 0; m < 5; m++) {
    iw_chroma_mode(A, L, sh1, i, m);
    ic_keep_body(i, tid, 256, k, A.jobs, A.res, A.icwork, A.ccoef, A.rec_cb, A.rec_cr, pitch_c, A.coef_out, A.reco_c);
    __syncthreads();
  }
  ic_commit_body(i, tid, 256, k, A.icwork, A.res, A.cres, A.syn_out);
  __syncthreads();
  iw_total(A, sh1, i);
}

// ---- the candidate passes side by side: k_iw_begin -> per PU (k_iw_cand over (CU, pass) -> k_iw_pick) -> k_iw_chroma over (CU, direction) -> k_iw_finish ----
// the s x s block at (bx, by) of a plane with everything intra prediction inside it can reach -- the row above and the column to the left, each 2 s long, the corner, and the
// (2 s)^2 samples they span (a transform unit inside the block reads, besides those, samples of the block itself: the candidate's own reconstruction where it has been
// written, otherwise what the picture held before, exactly as when the candidate works in the picture) -- from the picture (its slot's copy: rows are counted inside the copy
// of height ph) into a band
__device__ static inline void iw_band_prepare(const int16_t* pic, int16_t* band, const int pitch, const int pw, const int ph, const int bx, const int by, const int s) {
  const int yl = by % ph, side = 2 * s + 1;
  for (int e = threadIdx.x; e < side * side; e += 256) {
    const int r = e / side - 1, q = e % side - 1;                          // row / column relative to the block: -1 .. 2 s - 1
    if (yl + r < 0 || yl + r >= ph || bx + q < 0 || bx + q >= pw) continue;
    band[(ptrdiff_t)(by + r) * pitch + bx + q] = pic[(ptrdiff_t)(by + r) * pitch + bx + q];
  }
}
// A with the arrays and pictures of virtual candidate v of CU i (whose job sits at picture row y0): the bodies called with index v then work on v's copies and band
__device__ static inline IntraWalk iw_virtual(const IntraWalk& A, const int v, const int y0) {
  IntraWalk B = A;
  B.n = A.V; B.jobs = A.vjobs; B.syn = A.vsyn; B.csyn = A.vsyn; B.opt = A.vopt; B.tmp = A.vtmp; B.coef_tmp = A.vcoef_tmp; B.res = A.vres;
  const ptrdiff_t pitch = A.pic.pic_w, pitch_c = A.pic.pic_w >> 1;
  const ptrdiff_t sy = ((ptrdiff_t)v * A.hb + 1 - y0) * pitch, sc = ((ptrdiff_t)v * A.hbc + 1 - (y0 >> 1)) * pitch_c;
  B.pic.pred_y = A.band_pred[0] + sy; B.pic.pred_cb = A.band_pred[1] + sc; B.pic.pred_cr = A.band_pred[2] + sc;
  B.rec_y = A.band_rec[0] + sy; B.rec_cb = A.band_rec[1] + sc; B.rec_cr = A.band_rec[2] + sc;
  return B;
}

__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_iw_begin(IntraWalk A) {
  __shared__ WalkShared L;
  const int i = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) A.syn[i] = A.syn_in[i];
  { uint32_t* z = (uint32_t*)(A.res + i); for (int e = tid; e < (int)(sizeof(hop_rqt_result) / 4); e += 256) z[e] = 0; }
  __syncthreads();
  iw_luma_prep(A, L, i, 0);
}

// grid (n, P): candidate `pass` of PU pu of CU i with bCheckFirst (:2507-2553), in a band of its own; the result stays in vtmp / vcoef_tmp / the band
__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_iw_cand(IntraWalk A, int pu) {
  __shared__ WalkShared L;
  __shared__ CabacLds1 sh1;
  const int i = blockIdx.x, pass = blockIdx.y, tid = threadIdx.x;
  const int cnt = (int)A.mres[i].n, n_max = A.num_full_rd + 2;
  if (pass >= cnt || pass >= n_max) return;
  const int v = i * A.P + pass;
  const hop_rqt_job jb = A.jobs[i];
  if (tid == 0) { A.vjobs[v] = jb; hop_intra_cu_syntax y = A.syn[i]; y.luma_dir[pu] = (int)A.mres[i].modes[pass]; A.vsyn[v] = y; }
  { const uint32_t* so = (const uint32_t*)(A.opt + i); uint32_t* dq = (uint32_t*)(A.vopt + v); for (int e = tid; e < (int)(sizeof(hop_intra_rqt_opt) / 4); e += 256) dq[e] = so[e]; }
  const IntraWalk B = iw_virtual(A, v, jb.y);
  const int cu = 1 << A.k.log2_cu, N = cu >> A.nxn, parts = 1 << (2 * (A.k.log2_cu - 2)), part = pu * (parts >> (2 * A.nxn));
  iw_band_prepare(A.rec_y, B.rec_y, A.pic.pic_w, A.pic.pic_w, A.pic_h, jb.x + rqt_zx(part), jb.y + rqt_zy(part), N);
  __syncthreads();
  walk_intra_rqt(B, L, sh1, v, A.nxn, 1);
}

// the decisions over the candidate passes in their order (strict '<': the first of equal costs stays), the winner's arrays, levels and block kept (xSetIntraResultQT), then
// the best mode again with the full tree and the PU's commit; the next PU's candidate list, or the chroma search's start
__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_iw_pick(IntraWalk A, int pu) {
  __shared__ WalkShared L;
  __shared__ CabacLds1 sh1;
  __shared__ int s_best;
  const int i = blockIdx.x, tid = threadIdx.x;
  const RqtClass k = A.k;
  const int n_max = A.num_full_rd + 2, npu = A.nxn ? 4 : 1;
  const int cnt = min((int)A.mres[i].n, n_max);
  if (tid == 0) {
    int best = 0; double bc = 1.7e+308;
    for (int p = 0; p < cnt; p++) { const double c = A.vtmp[i * A.P + p].cost; if (c < bc) { bc = c; best = p; } }
    s_best = best;
  }
  __syncthreads();
  {
    const int v = i * A.P + s_best;
    const hop_rqt_job jb = A.jobs[i];
    const IntraWalk B = iw_virtual(A, v, jb.y);
    const int cu = 1 << k.log2_cu, parts = 1 << (2 * (k.log2_cu - 2)), q = parts >> (2 * A.nxn), part = pu * q, N = cu >> A.nxn, x0 = rqt_zx(part), y0 = rqt_zy(part);
    const size_t cu2 = (size_t)cu * cu, cbi = (size_t)i * (cu2 + (cu2 >> 1)), cbv = (size_t)v * (cu2 + (cu2 >> 1));
    IsWork* w = A.iswork + i; const hop_rqt_result* t = A.vtmp + v;
    for (int e = tid; e < q; e += 256) { w->tr[part + e] = t->tr_idx[part + e]; w->cbf[part + e] = t->cbf[0][part + e]; w->ts[part + e] = t->tskip[0][part + e]; }
    for (int e = tid; e < 16 * q; e += 256) A.coef_out[cbi + (size_t)16 * part + e] = A.vcoef_tmp[cbv + (size_t)16 * part + e];
    const int16_t* pic = B.rec_y + (ptrdiff_t)(jb.y + y0) * A.pic.pic_w + jb.x + x0;
    for (int e = tid; e < N * N; e += 256) { const int rr = e / N, cc = e % N; A.reco_y[(size_t)i * cu2 + (size_t)(y0 + rr) * cu + x0 + cc] = pic[(ptrdiff_t)rr * A.pic.pic_w + cc]; }
    __syncthreads();
    if (tid == 0) { w->best_cost = t->cost; w->best_dist = t->dist; w->best_mode = (int)A.mres[i].modes[s_best]; }
    __syncthreads();
  }
  iw_luma_final(A, L, sh1, i, pu);
  if (pu + 1 < npu) { iw_luma_prep(A, L, i, pu + 1); return; }
  if (tid == 0) A.syn_out[i] = A.syn[i];
}

// grid (n, 5): direction m of the chroma search of CU i in bands of its own (both chroma planes), on a copy of the CU's arrays
__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_iw_chroma(IntraWalk A) {
  __shared__ WalkShared L;
  __shared__ CabacLds1 sh1;
  const int i = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
  const int v = i * A.P + m;
  const hop_rqt_job jb = A.jobs[i];
  if (tid == 0) { A.vjobs[v] = jb; A.vsyn[v] = A.syn_out[i]; }
  { const uint32_t* so = (const uint32_t*)(A.opt + i); uint32_t* dq = (uint32_t*)(A.vopt + v); for (int e = tid; e < (int)(sizeof(hop_intra_rqt_opt) / 4); e += 256) dq[e] = so[e]; }
  { const uint32_t* so = (const uint32_t*)(A.res + i); uint32_t* dq = (uint32_t*)(A.vres + v); for (int e = tid; e < (int)(sizeof(hop_rqt_result) / 4); e += 256) dq[e] = so[e]; }
  const IntraWalk B = iw_virtual(A, v, jb.y);
  const int cu = 1 << A.k.log2_cu, pw = A.pic.pic_w >> 1;
  iw_band_prepare(A.rec_cb, B.rec_cb, pw, pw, A.pic_h >> 1, jb.x >> 1, jb.y >> 1, cu >> 1);
  iw_band_prepare(A.rec_cr, B.rec_cr, pw, pw, A.pic_h >> 1, jb.x >> 1, jb.y >> 1, cu >> 1);
  iw_zero_ccoef(B, v);
  if (tid == 0) A.icwork[v].best_cost = 1.7e+308;                           // (a private record per direction: ic_bits_body leaves this direction's cost there)
  __syncthreads();
  iw_chroma_mode(B, L, sh1, v, m);                                         // (ic_mode_body resets the best cost for m == 0 only: the comparison below reads icwork[v].dist / the cost it forms)
}

// the decisions over the five directions in their order, xSetIntraResultChromaQT for the winner, the CU's totals
__global__ __launch_bounds__(256, WALK_WAVES_PER_SIMD) void k_iw_finish(IntraWalk A) {
  __shared__ CabacLds1 sh1;
  __shared__ int s_best;
  const int i = blockIdx.x, tid = threadIdx.x;
  const RqtClass k = A.k;
  if (tid == 0) {
    int best = 0; double bc = 1.7e+308;
    for (int m = 0; m < 5; m++) { const double c = A.icwork[i * A.P + m].best_cost; if (c < bc) { bc = c; best = m; } }
    s_best = best;
  }
  __syncthreads();
  const int v = i * A.P + s_best;
  const hop_rqt_job jb = A.jobs[i];
  const IntraWalk B = iw_virtual(A, v, jb.y);
  const int cu = 1 << k.log2_cu, half = cu >> 1, parts = 1 << (2 * (k.log2_cu - 2)), pitch_c = A.pic.pic_w >> 1;
  const size_t cu2 = (size_t)cu * cu, h2 = cu2 >> 2;
  const hop_rqt_result* r = A.vres + v;
  hop_rqt_result* ro = A.res + i;
  for (int e = tid; e < parts; e += 256) { ro->cbf[1][e] = r->cbf[1][e]; ro->cbf[2][e] = r->cbf[2][e]; ro->tskip[1][e] = r->tskip[1][e]; ro->tskip[2][e] = r->tskip[2][e]; }
  for (int comp = 1; comp <= 2; comp++) {
    for (int e = tid; e < (int)h2; e += 256) {                            // as ic_keep_body: chroma level e of the CU layout from the layer of its transform unit
      const int p = e >> 2, d = r->tr_idx[p], log2 = k.log2_cu - d, dd = log2 == 2 ? d - 1 : d, np = parts >> (2 * dd), first = p - p % np;
      A.coef_out[(size_t)i * (cu2 + 2 * h2) + cu2 + (size_t)(comp - 1) * h2 + e] = A.ccoef[rqt_coef_at(k, v, k.log2_max_tu - log2, comp, first) + (size_t)(e - 4 * first)];
    }
    const int16_t* pic = (comp == 1 ? B.rec_cb : B.rec_cr) + (ptrdiff_t)(jb.y >> 1) * pitch_c + (jb.x >> 1);
    for (int e = tid; e < (int)h2; e += 256) { const int rr = e / half, cc = e % half; A.reco_c[(size_t)i * 2 * h2 + (size_t)(comp - 1) * h2 + e] = pic[(ptrdiff_t)rr * pitch_c + cc]; }
  }
  if (tid == 0) {
    const IcWork* w = A.icwork + v;
    A.cres[i].best_mode = w->mode; A.cres[i].dist = w->dist;
    A.syn_out[i].chroma_is_dm = w->mode == 36; A.syn_out[i].chroma_dir = w->mode;
  }
  __syncthreads();
  iw_total(A, sh1, i);
}

// candidates side by side (k_iw_*) unless HOP_WALK_CAND=0 or the bands would not fit the budget: P passes per candidate CU
static int iw_passes(int num_full_rd) { const int n_max = num_full_rd + 2; return n_max > 5 ? n_max : 5; }
static bool iw_parallel(const hop_ctx* c, int log2_cu, int n, int num_full_rd) {
  const char* e = getenv("HOP_WALK_CAND");                               // developer switch (read per call: the tests flip it)
  if (e && e[0] == '0') return false;
  const size_t V = (size_t)n * iw_passes(num_full_rd), cu = (size_t)1 << log2_cu;
  const size_t bands = V * ((2 * cu + 2) * (size_t)c->pic_w * 2 * 2 + (cu + 2) * (size_t)(c->pic_w >> 1) * 2 * 4);
  return bands <= ((size_t)6 << 30);
}
size_t hop_intra_walk_bytes(const hop_ctx* c, int log2_cu, int n, int num_full_rd) {
  const bool par = iw_parallel(c, log2_cu, n, num_full_rd);
  const size_t V = par ? (size_t)n * iw_passes(num_full_rd) : (size_t)n, cu = (size_t)1 << log2_cu, cu2 = cu * cu, n_coeff = V * (6 * cu2 + 16);
  size_t b = hop_intra_search_work_bytes(log2_cu, (int)V) + hop_intra_chroma_work_bytes(log2_cu, (int)V) + n_coeff * 4 +
             V * (4 * 4 + 8 + sizeof(hop_rdoq_job) + sizeof(hop_coeff_bits_job) + LEAF_WORK_PER_TU) + 128 * 256;
  if (par) b += V * (sizeof(hop_rqt_job) + sizeof(hop_intra_cu_syntax) + sizeof(hop_intra_rqt_opt) + 2 * sizeof(hop_rqt_result) + (cu2 + (cu2 >> 1)) * 4) +
                V * ((2 * cu + 2) * (size_t)c->pic_w * 2 * 2 + (cu + 2) * (size_t)(c->pic_w >> 1) * 2 * 4) + 16 * 256;
  return b;
}

int hop_launch_intra_walk(hop_ctx* c, const hop_intra_class& q, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_in, void* vbuf, size_t buf_bytes) {
  IntraWalk A; memset(&A, 0, sizeof(A));
  const hop_rqt_job* cls = &q.cls;
  RqtClass& k = A.k;
  k.log2_cu = cls->log2_cu; k.log2_max_tu = cls->log2_max_tu; k.log2_min_tu = cls->log2_min_tu_in_cu; k.inter_split = 0; k.sign_hide = cls->sign_hide ? 1 : 0; k.use_ts = cls->use_ts ? 1 : 0;
  const int n = q.n;
  const bool par = iw_parallel(c, k.log2_cu, n, q.num_full_rd);
  A.P = par ? iw_passes(q.num_full_rd) : 1; A.V = n * A.P;
  const size_t V = (size_t)A.V;
  A.n = n; A.nxn = q.part_nxn ? 1 : 0; A.num_full_rd = q.num_full_rd; A.bd_y = c->bd_y; A.bd_c = c->bd_c; A.pic = hop_make_pics(c); A.pic_h = c->pic_h;
  A.jobs = q.d_jobs; A.syn_in = q.d_syntax; A.opt = q.d_opts; A.sj = q.d_sjobs; A.ctx_in = d_ctx_in; A.cu_in = d_cu_in;
  A.sres = q.d_sresults; A.res = q.d_results; A.cres = q.d_cresults; A.coef_out = q.d_coef; A.reco_y = q.d_reco_y; A.reco_c = q.d_reco_c; A.syn_out = q.d_syntax_out;
  A.dist = q.d_dist; A.bits = q.d_bits; A.cost = q.d_cost; A.ctx_out = q.d_ctx_out; A.cu_out = q.d_cu_ctx_out;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t cu = (size_t)1 << k.log2_cu, cu2 = cu * cu;
  A.n_coeff = V * (6 * cu2 + 16); A.ts_base = V * 6 * cu2;
  char* buf = (char*)vbuf; size_t o = 0;
  auto take = [&](size_t bytes) { char* p = buf + o; o = al(o + bytes); return p; };
  A.syn = (hop_intra_cu_syntax*)take((size_t)n * sizeof(hop_intra_cu_syntax)); A.rj = (hop_intra_job*)take((size_t)n * sizeof(hop_intra_job));
  A.mj = (hop_intra_modes_job*)take((size_t)n * sizeof(hop_intra_modes_job)); A.mres = (hop_intra_modes_result*)take((size_t)n * sizeof(hop_intra_modes_result));
  A.satd = (uint32_t*)take((size_t)n * 35 * 4); A.iswork = (IsWork*)take((size_t)n * sizeof(IsWork)); A.tmp = (hop_rqt_result*)take((size_t)n * sizeof(hop_rqt_result));
  A.coef_tmp = (int32_t*)take((size_t)n * (cu2 + (cu2 >> 1)) * 4); A.active = (uint8_t*)take((size_t)n);
  A.csyn = (hop_intra_cu_syntax*)take((size_t)n * sizeof(hop_intra_cu_syntax));
  // per (virtual) candidate
  A.cur = (hop_cabac_ctx*)take(V * sizeof(hop_cabac_ctx)); A.cucur = (hop_cabac_cu_ctx*)take(V * sizeof(hop_cabac_cu_ctx));
  for (int d = 0; d < 4; d++) {
    A.root[d] = (hop_cabac_ctx*)take(V * sizeof(hop_cabac_ctx)); A.curoot[d] = (hop_cabac_cu_ctx*)take(V * sizeof(hop_cabac_cu_ctx));
    A.test[d] = (hop_cabac_ctx*)take(V * sizeof(hop_cabac_ctx)); A.cutest[d] = (hop_cabac_cu_ctx*)take(V * sizeof(hop_cabac_cu_ctx));
  }
  A.irwork = (IrqWork*)take(V * sizeof(IrqWork)); A.pj = (hop_intra_job*)take(V * sizeof(hop_intra_job)); A.modes = (int32_t*)take(V * 4);
  A.tuj = (hop_tu_rd_job*)take(V * sizeof(hop_tu_rd_job)); A.tuj2 = (hop_tu_rd_job*)take(V * sizeof(hop_tu_rd_job));
  A.off = (int64_t*)take(V * 8); A.off2 = (int64_t*)take(V * 8);
  A.tr = (hop_tu_rd_result*)take(V * sizeof(hop_tu_rd_result)); A.tr2 = (hop_tu_rd_result*)take(V * sizeof(hop_tu_rd_result));
  A.coef = (int32_t*)take(A.n_coeff * 4); A.recl = (int16_t*)take(V * 4 * cu2 * 2); A.park = (int16_t*)take(V * 32);
  A.ccur = (hop_cabac_ctx*)take(V * sizeof(hop_cabac_ctx)); A.croot = (hop_cabac_ctx*)take(V * sizeof(hop_cabac_ctx));
  A.icwork = (IcWork*)take(V * sizeof(IcWork));
  A.ccoef = (int32_t*)take(A.n_coeff * 4); A.cpark = (int16_t*)take(V * 32);
  A.lcoef = (int32_t*)take(A.n_coeff * 4);
  A.zs = (uint32_t*)take(V * 4); A.ns = (uint32_t*)take(V * 4); A.as = (uint32_t*)take(V * 4); A.fr = (unsigned long long*)take(V * 8);
  A.rq = (hop_rdoq_job*)take(V * sizeof(hop_rdoq_job)); A.cb = (hop_coeff_bits_job*)take(V * sizeof(hop_coeff_bits_job));
  A.lwork = take(V * LEAF_WORK_PER_TU);
  if (par) {
    A.vjobs = (hop_rqt_job*)take(V * sizeof(hop_rqt_job)); A.vsyn = (hop_intra_cu_syntax*)take(V * sizeof(hop_intra_cu_syntax)); A.vopt = (hop_intra_rqt_opt*)take(V * sizeof(hop_intra_rqt_opt));
    A.vtmp = (hop_rqt_result*)take(V * sizeof(hop_rqt_result)); A.vres = (hop_rqt_result*)take(V * sizeof(hop_rqt_result)); A.vcoef_tmp = (int32_t*)take(V * (cu2 + (cu2 >> 1)) * 4);
    A.hb = (int)(2 * cu + 2); A.hbc = (int)(cu + 2);
    const size_t by = V * A.hb * (size_t)c->pic_w * 2, bc = V * A.hbc * (size_t)(c->pic_w >> 1) * 2;
    A.band_pred[0] = (int16_t*)take(by); A.band_pred[1] = (int16_t*)take(bc); A.band_pred[2] = (int16_t*)take(bc);
    A.band_rec[0] = (int16_t*)take(by); A.band_rec[1] = (int16_t*)take(bc); A.band_rec[2] = (int16_t*)take(bc);
  }
  if (o > buf_bytes) return hop_set_err(c, HOP_ERR_STATE, "intra walk: work buffer too small (%zu > %zu)", o, buf_bytes);
  A.entropy_bits = hop_entropy_bits_device(c); A.scans = c->rdoq_scans; A.rec_y = c->rec[0]; A.rec_cb = c->rec[1]; A.rec_cr = c->rec[2];
  const int pr = hop_prof_begin(c, A.nxn ? HOP_K_WALK_INTRA_NXN : HOP_K_WALK_INTRA + (k.log2_cu - 3), (uint64_t)n);
  if (!par) hipLaunchKernelGGL(k_intra_walk, dim3(n), dim3(256), 0, c->stream, A);
  else {
    hipLaunchKernelGGL(k_iw_begin, dim3(n), dim3(256), 0, c->stream, A);
    for (int pu = 0; pu < (A.nxn ? 4 : 1); pu++) {
      hipLaunchKernelGGL(k_iw_cand, dim3(n, A.num_full_rd + 2), dim3(256), 0, c->stream, A, pu);
      hipLaunchKernelGGL(k_iw_pick, dim3(n), dim3(256), 0, c->stream, A, pu);
    }
    hipLaunchKernelGGL(k_iw_chroma, dim3(n, 5), dim3(256), 0, c->stream, A);
    hipLaunchKernelGGL(k_iw_finish, dim3(n), dim3(256), 0, c->stream, A);
  }
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra walk launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
