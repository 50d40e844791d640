// k_leaf_fused.inl -- the transform-unit leaf step (row a8b: residual, xT, estBit, RDOQ, counted bits, inverse path, cbf-zero decision; the stages of hop_launch_tu_rd)
// as ONE kernel, one workgroup per TU.  Included by k_cabac.hip.
// The staged pipeline of k_tq.hip / k_rdoq.hip is 13 launches per batch and lays its serial stages (RDOQ, bit counting) out one LANE per TU: right for the thousands of
// TUs of a frozen-reference search, wrong for the RD spine, whose batches hold one TU per CTU in flight and which walks a chain of thousands of such batches per CTU -- there
// the launches ARE the time.  Here a workgroup takes its TU through all stages: the transforms on all 256 threads (or one wave for 4x4 / 8x8), the serial stages on thread 0
// in between.  The arithmetic is the staged kernels' own (the shared bodies of k_turd_dev.inl, k_rdoq_dev.inl, cb_code_tu), so the results are identical by construction;
// tests/test_gpu_tq_intra.py runs both.
#include "k_turd_dev.inl"
#include "k_rdoq_dev.inl"

#define LEAF_WORK_PER_TU ((size_t)1024 * RQ_WORK_PER_COEF)           // the RDOQ work area of a 32x32 TU at ws = 1

struct LeafShared {
  union { TurdFwdShared big; TurdSmallShared small; } t;
  CabacLds1 cab;                                                      // the TU's context states
  uint16_t scan[1024]; uint16_t scanCG[64]; double cgSig[64];
  // everything the serial walk touches (a lane's dependent loads from HBM were most of its time): coefficients, levels, bit-estimate table, context states, the
  // entropy table, and the RDOQ work area of TUs up to 16x16 (a 32x32 TU's 40 KB stay in HBM)
  int32_t src[1024], lev[1024], ebits[128]; hop_estbits eb; uint8_t ctx[152]; double work[256 * RQ_WORK_PER_COEF / 8];
};

// one TU through all stages on the 256 threads of the calling workgroup (every thread calls; barriers inside)
__device__ static void turd_fused_body(LeafShared& L, const int j, const hop_tu_rd_job* jobs, int n, hop_pics pic, const hop_cabac_ctx* ctx_in, const int64_t* coef_off,
                                       const int32_t* entropy_bits, const uint16_t* scans, int32_t* coef, int32_t* levels, uint32_t* zs, uint32_t* ns, uint32_t* as,
                                       unsigned long long* fr, hop_rdoq_job* rq, hop_coeff_bits_job* cb, char* work, hop_tu_rd_result* res, int16_t* rec_y, int16_t* rec_cb, int16_t* rec_cr) {
  const int tid = threadIdx.x;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size < 2 || jb.log2_size > 5) return;                   // an empty slot of a job table (k_rqt.inl); uniform over the workgroup
  const bool small = jb.log2_size <= 3;
  const int LOG2 = jb.log2_size, N2 = 1 << (2 * LOG2), CGN = N2 >> 4;
  // residual, forward transform (or the transform-skip scaling), distortion of the zero block
  if (small) { if (tid < 64) turd_forward_small_body(L.t.small, 0, tid, j, jobs, n, pic, coef_off, coef, zs); }
  else turd_forward_body(L.t.big, j, jobs, pic, coef_off, coef, zs);
  { const uint16_t* s0 = rq_scan(scans, jb.scan_idx, LOG2); const uint16_t* s1 = rq_scan_cg(scans, jb.scan_idx, LOG2);
    for (int i = tid; i < N2; i += 256) L.scan[i] = s0[i];
    for (int i = tid; i < CGN; i += 256) L.scanCG[i] = s1[i]; }
  __threadfence_block();
  __syncthreads();
  const int64_t off = coef_off[j];
  for (int i = tid; i < N2; i += 256) L.src[i] = coef[off + i];
  { const uint8_t* st = ctx_in[jb.ctx_index].state; for (int i = tid; i < 152; i += 256) { const uint8_t v = st[i]; L.ctx[i] = v; L.cab.st[i][0] = v; } }
  for (int i = tid; i < 128; i += 256) L.ebits[i] = entropy_bits[i];
  __syncthreads();
  if (tid == 0) {                                                     // the serial stages
    turd_setup_body(j, jobs, n, ctx_in, coef_off, L.ebits, &L.eb, rq, cb, L.ctx);
    const hop_rdoq_job rj = rq[j];
    double* wd = LOG2 <= 4 ? L.work : (double*)(work + (size_t)j * LEAF_WORK_PER_TU);
    if (LOG2 == 2) rdoq_tu<2>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, wd, 1, 0);
    else if (LOG2 == 3) rdoq_tu<3>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, wd, 1, 0);
    else if (LOG2 == 4) rdoq_tu<4>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, wd, 1, 0);
    else rdoq_tu<5>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, wd, 1, 0);
    const hop_coeff_bits_job bj = cb[j];
    fr[j] = cb_code_tu_at(L.cab, 0, L.lev, bj.log2_size, bj.comp != 0, bj.scan_idx, bj.sign_hide, bj.use_ts, bj.ts_flag, bj.cbf_ctx_plus1, L.scan, L.scanCG);
  }
  __syncthreads();
  for (int i = tid; i < N2; i += 256) levels[off + i] = L.lev[i];
  __threadfence_block();
  __syncthreads();
  // dequantisation, inverse transform, reconstruction (intra) and the distortion of the coded block
  if (small) { if (tid < 64) turd_inverse_small_body(L.t.small, 0, tid, j, jobs, n, pic, coef_off, levels, as, ns, rec_y, rec_cb, rec_cr); }
  else turd_inverse_body(L.t.big, j, jobs, pic, coef_off, levels, as, ns, rec_y, rec_cb, rec_cr);
  __threadfence_block();
  __syncthreads();
  if (tid == 0) turd_decide_body(j, jobs, n, ctx_in, coef_off, entropy_bits, as, fr, zs, ns, levels, res);
}
__global__ __launch_bounds__(256) void k_turd_fused(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const hop_cabac_ctx* __restrict__ ctx_in,
                                                    const int64_t* __restrict__ coef_off, const int32_t* __restrict__ entropy_bits, const uint16_t* __restrict__ scans,
                                                    int32_t* __restrict__ coef, int32_t* __restrict__ levels, uint32_t* __restrict__ zs, uint32_t* __restrict__ ns,
                                                    uint32_t* __restrict__ as, unsigned long long* __restrict__ fr, hop_estbits* __restrict__ tables,
                                                    hop_rdoq_job* __restrict__ rq, hop_coeff_bits_job* __restrict__ cb, char* __restrict__ work,
                                                    hop_tu_rd_result* __restrict__ res, int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  __shared__ LeafShared L;
  turd_fused_body(L, blockIdx.x, jobs, n, pic, ctx_in, coef_off, entropy_bits, scans, coef, levels, zs, ns, as, fr, rq, cb, work, res, rec_y, rec_cb, rec_cr);
}

// The same for batches whose TUs are all 4x4 or 8x8 (size_hint 1: most of the quadtree steps of 8x8 CUs, every chroma step of CUs up to 16x16): ONE wave per TU and
// 1.4 KB of LDS, so that a compute unit holds 32 TUs instead of 7 -- in the large batches of many CTUs in flight the leaf step is bound by how many serial walks the
// GPU holds at once.
struct LeafSmallShared { TurdSmallShared t; CabacLds1 cab; uint16_t scan[64]; uint16_t scanCG[4]; double cgSig[4];
                         double work[64 * RQ_WORK_PER_COEF / 8]; hop_estbits eb; int32_t src[64], lev[64], ebits[128]; uint8_t ctx[152]; };   // everything the serial walk touches

// The same body for the candidate walks (k_walk.inl): up to four small transform units of one tree node side by side, one per WAVE of the 256-thread workgroup, each with
// its own LeafSmallShared.  Every wave calls (the barriers are the workgroup's); a wave without a unit passes j < 0 and only keeps step.
__device__ static __forceinline__ void turd_fused_small_wave_body(LeafSmallShared& L, const int lane, const int j, const hop_tu_rd_job* jobs, int n, hop_pics pic, const hop_cabac_ctx* ctx_in,
                                                  const int64_t* coef_off, const int32_t* entropy_bits, const uint16_t* scans, int32_t* coef, int32_t* levels, uint32_t* zs, uint32_t* ns,
                                                  uint32_t* as, unsigned long long* fr, hop_rdoq_job* rq, hop_coeff_bits_job* cb, hop_tu_rd_result* res, int16_t* rec_y, int16_t* rec_cb,
                                                  int16_t* rec_cr) {
  hop_tu_rd_job jb; jb.log2_size = 0; jb.scan_idx = 0; jb.ctx_index = 0;
  if (j >= 0) jb = jobs[j];
  const bool live = j >= 0 && jb.log2_size >= 2 && jb.log2_size <= 3;      // (uniform over the wave)
  const int LOG2 = live ? jb.log2_size : 2, N2 = 1 << (2 * LOG2), CGN = N2 >> 4;
  int64_t off = 0;
  if (live) {
    turd_forward_small_body(L.t, 0, lane, j, jobs, n, pic, coef_off, coef, zs);
    const uint16_t* s0 = rq_scan(scans, jb.scan_idx, LOG2); const uint16_t* s1 = rq_scan_cg(scans, jb.scan_idx, LOG2);
    if (lane < N2) L.scan[lane] = s0[lane];
    if (lane < CGN) L.scanCG[lane] = s1[lane];
  }
  __threadfence_block();
  __syncthreads();
  if (live) {
    off = coef_off[j];
    if (lane < N2) L.src[lane] = coef[off + lane];
    const uint8_t* st = ctx_in[jb.ctx_index].state; for (int i = lane; i < 152; i += 64) { const uint8_t v = st[i]; L.ctx[i] = v; L.cab.st[i][0] = v; }
    for (int i = lane; i < 128; i += 64) L.ebits[i] = entropy_bits[i];
  }
  __syncthreads();
  if (live && lane == 0) {
    turd_setup_body(j, jobs, n, ctx_in, coef_off, L.ebits, &L.eb, rq, cb, L.ctx);
    const hop_rdoq_job rj = rq[j];
    if (LOG2 == 2) rdoq_tu<2>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, L.work, 1, 0);
    else rdoq_tu<3>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, L.work, 1, 0);
    const hop_coeff_bits_job bj = cb[j];
    fr[j] = cb_code_tu_at(L.cab, 0, L.lev, bj.log2_size, bj.comp != 0, bj.scan_idx, bj.sign_hide, bj.use_ts, bj.ts_flag, bj.cbf_ctx_plus1, L.scan, L.scanCG);
  }
  __syncthreads();
  if (live && lane < N2) levels[off + lane] = L.lev[lane];
  __threadfence_block();
  __syncthreads();
  if (live) turd_inverse_small_body(L.t, 0, lane, j, jobs, n, pic, coef_off, levels, as, ns, rec_y, rec_cb, rec_cr);
  __threadfence_block();
  __syncthreads();
  if (live && lane == 0) turd_decide_body(j, jobs, n, ctx_in, coef_off, entropy_bits, as, fr, zs, ns, levels, res);
}

__global__ __launch_bounds__(64) void k_turd_fused_small(const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const hop_cabac_ctx* __restrict__ ctx_in,
                                                         const int64_t* __restrict__ coef_off, const int32_t* __restrict__ entropy_bits, const uint16_t* __restrict__ scans,
                                                         int32_t* __restrict__ coef, int32_t* __restrict__ levels, uint32_t* __restrict__ zs, uint32_t* __restrict__ ns,
                                                         uint32_t* __restrict__ as, unsigned long long* __restrict__ fr, hop_estbits* __restrict__ tables,
                                                         hop_rdoq_job* __restrict__ rq, hop_coeff_bits_job* __restrict__ cb, char* __restrict__ work,
                                                         hop_tu_rd_result* __restrict__ res, int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  __shared__ LeafSmallShared L;
  const int j = blockIdx.x, tid = threadIdx.x;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size < 2 || jb.log2_size > 3) return;                   // an empty slot; (larger TUs never come here: size_hint 1)
  const int LOG2 = jb.log2_size, N2 = 1 << (2 * LOG2), CGN = N2 >> 4;
  turd_forward_small_body(L.t, 0, tid, j, jobs, n, pic, coef_off, coef, zs);
  { const uint16_t* s0 = rq_scan(scans, jb.scan_idx, LOG2); const uint16_t* s1 = rq_scan_cg(scans, jb.scan_idx, LOG2);
    if (tid < N2) L.scan[tid] = s0[tid];
    if (tid < CGN) L.scanCG[tid] = s1[tid]; }
  __threadfence_block();
  __syncthreads();
  // the coefficients, the context states and the entropy table into LDS (a lane's dependent loads from HBM would be most of the serial walk's time)
  const int64_t off = coef_off[j];
  if (tid < N2) L.src[tid] = coef[off + tid];
  { const uint8_t* st = ctx_in[jb.ctx_index].state; for (int i = tid; i < 152; i += 64) { const uint8_t v = st[i]; L.ctx[i] = v; L.cab.st[i][0] = v; } }
  for (int i = tid; i < 128; i += 64) L.ebits[i] = entropy_bits[i];
  __syncthreads();
  if (tid == 0) {
    turd_setup_body(j, jobs, n, ctx_in, coef_off, L.ebits, &L.eb, rq, cb, L.ctx);
    const hop_rdoq_job rj = rq[j];
    if (LOG2 == 2) rdoq_tu<2>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, L.work, 1, 0);
    else rdoq_tu<3>(rj, &L.eb, L.scan, L.scanCG, L.cgSig, 1, L.src, L.lev, as + j, L.work, 1, 0);
    const hop_coeff_bits_job bj = cb[j];
    fr[j] = cb_code_tu_at(L.cab, 0, L.lev, bj.log2_size, bj.comp != 0, bj.scan_idx, bj.sign_hide, bj.use_ts, bj.ts_flag, bj.cbf_ctx_plus1, L.scan, L.scanCG);
  }
  __syncthreads();
  if (tid < N2) levels[off + tid] = L.lev[tid];
  __threadfence_block();
  __syncthreads();
  turd_inverse_small_body(L.t, 0, tid, j, jobs, n, pic, coef_off, levels, as, ns, rec_y, rec_cb, rec_cr);
  __threadfence_block();
  __syncthreads();
  if (tid == 0) turd_decide_body(j, jobs, n, ctx_in, coef_off, entropy_bits, as, fr, zs, ns, levels, res);
}

size_t hop_tu_rd_fused_scratch(int n, size_t n_coeff) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  return al(n_coeff * 4) + 4 * al((size_t)n * 4) + al((size_t)n * 8) + al((size_t)n * sizeof(hop_estbits)) + al((size_t)n * sizeof(hop_rdoq_job)) +
         al((size_t)n * sizeof(hop_coeff_bits_job)) + (size_t)n * LEAF_WORK_PER_TU + 256;
}

int hop_launch_tu_rd_fused(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx, const int64_t* d_coef_off, size_t n_coeff,
                           int32_t* d_levels, hop_tu_rd_result* d_res, int all_small) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_coef = 0, o_zs = al(o_coef + n_coeff * 4), o_ns = al(o_zs + (size_t)n * 4), o_as = al(o_ns + (size_t)n * 4), o_fr = al(o_as + (size_t)n * 4);
  const size_t o_tab = al(o_fr + (size_t)n * 8), o_rq = al(o_tab + (size_t)n * sizeof(hop_estbits)), o_cb = al(o_rq + (size_t)n * sizeof(hop_rdoq_job));
  const size_t o_wk = al(o_cb + (size_t)n * sizeof(hop_coeff_bits_job));
  void* sc; int r = hop_scratch(c, o_wk + (size_t)n * LEAF_WORK_PER_TU + 256, &sc); if (r) return r;
  char* b = (char*)sc;
  const int pr = hop_prof_begin(c, HOP_K_TQ, (uint64_t)n);
  if (all_small)
    hipLaunchKernelGGL(k_turd_fused_small, dim3(n), dim3(64), 0, c->stream, d_jobs, n, hop_make_pics(c), d_ctx, d_coef_off, hop_entropy_bits_device(c), c->rdoq_scans,
                       (int32_t*)(b + o_coef), d_levels, (uint32_t*)(b + o_zs), (uint32_t*)(b + o_ns), (uint32_t*)(b + o_as), (unsigned long long*)(b + o_fr),
                       (hop_estbits*)(b + o_tab), (hop_rdoq_job*)(b + o_rq), (hop_coeff_bits_job*)(b + o_cb), b + o_wk, d_res, c->rec[0], c->rec[1], c->rec[2]);
  else
  hipLaunchKernelGGL(k_turd_fused, dim3(n), dim3(256), 0, c->stream, d_jobs, n, hop_make_pics(c), d_ctx, d_coef_off, hop_entropy_bits_device(c), c->rdoq_scans,
                     (int32_t*)(b + o_coef), d_levels, (uint32_t*)(b + o_zs), (uint32_t*)(b + o_ns), (uint32_t*)(b + o_as), (unsigned long long*)(b + o_fr),
                     (hop_estbits*)(b + o_tab), (hop_rdoq_job*)(b + o_rq), (hop_coeff_bits_job*)(b + o_cb), b + o_wk, d_res, c->rec[0], c->rec[1], c->rec[2]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "tu_rd (fused) launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
