// k_turd_dev.inl -- device side of the transform-unit leaf step (row a8b, see k_tq.hip): residual and forward transform, bit-estimate tables, inverse path and
// distortion, the cbf-zero decision.  Shared by the staged kernels of k_tq.hip (batches of thousands of TUs) and the fused leaf kernel of k_cabac.hip (one TU per workgroup).
// first column of the 32-point core transform; every entry of every size follows from the cosine symmetries
static __constant__ int16_t c_dct_a[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                     61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
static __constant__ int16_t c_dst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
static __constant__ int c_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                    // TComRom.cpp:169-172

__device__ static inline int dct_coef(int step32, int k, int n) {     // T_N[k][n] = T32[k * 32/N][n]
  const int k32 = k * step32;
  if (k32 == 0) return 64;
  const int th = ((2 * n + 1) * k32) & 127;
  if (th <= 32) return c_dct_a[th];
  if (th <= 64) return -c_dct_a[64 - th];
  if (th <= 96) return -c_dct_a[th - 64];
  return c_dct_a[128 - th];
}
__device__ static inline int clip16(int v) { return min(32767, max(-32768, v)); }


struct TurdFwdShared { int16_t a[32 * 32]; int16_t b[32 * 32]; int16_t T[32 * 32]; unsigned int acc; };

__device__ static void turd_forward_body(TurdFwdShared& sh, const int bid, const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                      int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) {
  const hop_tu_rd_job jb = jobs[bid];
  if (jb.log2_size <= 3) return;                                   // k_turd_forward_small takes it
  const int tid = threadIdx.x, log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  if (tid == 0) sh.acc = 0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  for (int i = tid; i < NN; i += 256) { int k = i >> log2N, n = i & (N - 1); sh.T[i] = (int16_t)(dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n)); }
  __syncthreads();
  unsigned part = 0;
  const unsigned sshift = (unsigned)((bd - 8) << 1);
  for (int i = tid; i < NN; i += 256) {
    int r = i >> log2N, c = i & (N - 1);
    const int e = (int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c];
    sh.a[i] = (int16_t)e;
    part += (unsigned)(e * e) >> sshift;                          // getDistPart(zero block, residual), SSE
  }
  part = (unsigned)hopd_wave_sum((int)part);
  if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
  __syncthreads();
  int32_t* out = coef + coef_off[bid];
  if (jb.flags & HOP_TU_RD_TS) {                                     // xTransformSkip, TComTrQuant.cpp:1402-1420 (shift >= 0 for bit depths <= 13)
    const int shift = 15 - bd - log2N;
    for (int i = tid; i < NN; i += 256) out[i] = (int)sh.a[i] * (1 << shift);
    if (tid == 0) zero_sse[bid] = sh.acc;
    return;
  }
  const int s1 = log2N - 1 + bd - 8, s2 = log2N + 6;               // xTrMxN :788-789
  for (int i = tid; i < NN; i += 256) {
    int k = i >> log2N, j = i & (N - 1), sum = 0;
    for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.a[j * N + n];
    sh.b[k * N + j] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
  }
  __syncthreads();
  for (int i = tid; i < NN; i += 256) {
    int k = i >> log2N, j = i & (N - 1), sum = 0;
    for (int n = 0; n < N; n++) sum += sh.T[k * N + n] * sh.b[j * N + n];
    out[k * N + j] = (int)(int16_t)((sum + (1 << (s2 - 1))) >> s2);
  }
  if (tid == 0) zero_sse[bid] = sh.acc;
}

__device__ static void turd_inverse_body(TurdFwdShared& sh, const int bid, const hop_tu_rd_job* __restrict__ jobs, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                      const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,
                                                      int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  const hop_tu_rd_job jb = jobs[bid];
  if (jb.log2_size <= 3) return;                                   // k_turd_inverse_small takes it
  if (abs_sum[bid] == 0) {
    if (!jb.is_intra) { if (threadIdx.x == 0) nz_sse[bid] = 0; return; }
    // reconstruction = prediction (TEncSearch.cpp:1118-1127,1133-1152); its distortion against the original is reported like a coded block's
    const bool ch = jb.comp != 0; const int pt = ch ? pic.pic_w >> 1 : pic.pic_w, xx = ch ? jb.x >> 1 : jb.x, yy = ch ? jb.y >> 1 : jb.y, NNz = 1 << jb.log2_size;
    const int16_t* pp = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)yy * pt + xx;
    const int16_t* oo = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)yy * pt + xx;
    int16_t* rr = (jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)yy * pt + xx;
    if (threadIdx.x == 0) sh.acc = 0;
    __syncthreads();
    unsigned part0 = 0; const unsigned ss0 = (unsigned)(((ch ? pic.bd_c : pic.bd_y) - 8) << 1);
    for (int i = threadIdx.x; i < NNz * NNz; i += 256) {
      const int r = i >> jb.log2_size, c2 = i & (NNz - 1);
      const int v = pp[(size_t)r * pt + c2], e = v - (int)oo[(size_t)r * pt + c2];
      rr[(size_t)r * pt + c2] = (int16_t)v; part0 += (unsigned)(e * e) >> ss0;
    }
    part0 = (unsigned)hopd_wave_sum((int)part0);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh.acc, part0);
    __syncthreads();
    if (threadIdx.x == 0) nz_sse[bid] = sh.acc;
    return;
  }
  const int tid = threadIdx.x, log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  if (tid == 0) sh.acc = 0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  for (int i = tid; i < NN; i += 256) { int k = i >> log2N, n = i & (N - 1); sh.T[i] = (int16_t)(dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n)); }
  {
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
    const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;     // xDeQuant :1171-1182
    const int32_t* lv = levels + coef_off[bid];
    for (int i = tid; i < NN; i += 256) sh.a[i] = (int16_t)clip16((clip16(lv[i]) * scale + dadd) >> dshift);
  }
  __syncthreads();
  if ((jb.flags & HOP_TU_RD_TS) && !jb.is_intra) {                   // xITransformSkip, TComTrQuant.cpp:1442-1460; the dequantised value is an Int here (no 16-bit clip in between)
    const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
    const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;
    const int32_t* lv = levels + coef_off[bid];
    unsigned part = 0; const unsigned sshift = (unsigned)((bd - 8) << 1);
    for (int i = tid; i < NN; i += 256) {
      const int j = i >> log2N, n = i & (N - 1);
      const int dq = clip16((clip16(lv[i]) * scale + dadd) >> dshift);
      const int rr = (int)(int16_t)((dq + (1 << (transformShift - 1))) >> transformShift);
      const int e = rr - ((int)org[(size_t)j * pitch + n] - (int)prd[(size_t)j * pitch + n]);
      part += (unsigned)(e * e) >> sshift;
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
    __syncthreads();
    if (tid == 0) nz_sse[bid] = sh.acc;
    return;
  }
  const int s1 = 7, s2 = 12 - (bd - 8);                            // SHIFT_INV_1ST / SHIFT_INV_2ND
  for (int i = tid; i < NN; i += 256) {
    int j = i >> log2N, n = i & (N - 1), sum = 0;
    for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * sh.a[k * N + j];
    sh.b[j * N + n] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
  }
  __syncthreads();
  unsigned part = 0;
  const unsigned sshift = (unsigned)((bd - 8) << 1);
  for (int i = tid; i < NN; i += 256) {
    int j = i >> log2N, n = i & (N - 1), sum = 0;
    for (int k = 0; k < N; k++) sum += sh.T[k * N + n] * sh.b[k * N + j];
    const int rr = clip16((sum + (1 << (s2 - 1))) >> s2);          // reconstructed residual sample (j, n)
    int e;
    if (jb.is_intra) {                                             // ClipY / ClipC(prediction + residual), distortion against the original (:1133-1158)
      const int v = min((1 << bd) - 1, max(0, (int)prd[(size_t)j * pitch + n] + rr));
      ((jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0)[(size_t)j * pitch + n] = (int16_t)v;
      e = v - (int)org[(size_t)j * pitch + n];
    } else e = rr - ((int)org[(size_t)j * pitch + n] - (int)prd[(size_t)j * pitch + n]);
    part += (unsigned)(e * e) >> sshift;
  }
  part = (unsigned)hopd_wave_sum((int)part);
  if ((tid & 63) == 0) atomicAdd(&sh.acc, part);
  __syncthreads();
  if (tid == 0) nz_sse[bid] = sh.acc;
}

// ---- the same two stages for 4x4 and 8x8 TUs: one WAVE per TU (a lane per sample), four TUs per workgroup, no workgroup barrier ----
// (a 256-thread workgroup per 16-sample TU spends its time being scheduled: the residual quadtree of 8x8 CUs is 2 M such TUs per frame)
struct TurdSmallShared { int16_t a[4][64]; int16_t b[4][64]; };
__device__ static inline int turd_t(bool dst, int log2N, int k, int n) { return dst ? c_dst4[k][n] : dct_coef(32 >> log2N, k, n); }

__device__ static void turd_forward_small_body(TurdSmallShared& sh, const int w, const int lane, const int j, const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                            int32_t* __restrict__ coef, uint32_t* __restrict__ zero_sse) {
  if (j >= n) return;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size > 3 || jb.log2_size < 2) return;                // < 2: an empty slot of a job table (k_rqt.inl)
  const int log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0, live = lane < NN;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  const bool dst = jb.use_dst && N == 4 && !chroma;
  const int r = lane >> log2N, c = lane & (N - 1);                 // as sample: row r, column c; as coefficient: k = r, j = c
  int e = 0;
  if (live) e = (int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c];
  sh.a[w][lane] = (int16_t)e;
  const unsigned zs = (unsigned)hopd_wave_sum((int)((unsigned)(e * e) >> (unsigned)((bd - 8) << 1)));
  if (lane == 0) zero_sse[j] = zs;
  int32_t* out = coef + coef_off[j];
  if (jb.flags & HOP_TU_RD_TS) { if (live) out[lane] = e * (1 << (15 - bd - log2N)); return; }      // xTransformSkip
  __builtin_amdgcn_wave_barrier();
  const int s1 = log2N - 1 + bd - 8, s2 = log2N + 6;
  int t[8];
  for (int q = 0; q < N; q++) t[q] = turd_t(dst, log2N, r, q);     // row k = r of the transform matrix
  int sum = 0;
  if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.a[w][c * N + q];
  sh.b[w][r * N + c] = (int16_t)((sum + (1 << (s1 - 1))) >> s1);
  __builtin_amdgcn_wave_barrier();
  sum = 0;
  if (live) { for (int q = 0; q < N; q++) sum += t[q] * sh.b[w][c * N + q]; out[r * N + c] = (int)(int16_t)((sum + (1 << (s2 - 1))) >> s2); }
}

__device__ static void turd_inverse_small_body(TurdSmallShared& sh, const int w, const int lane, const int j, const hop_tu_rd_job* __restrict__ jobs, int n, hop_pics pic, const int64_t* __restrict__ coef_off,
                                                            const int32_t* __restrict__ levels, const uint32_t* __restrict__ abs_sum, uint32_t* __restrict__ nz_sse,
                                                            int16_t* __restrict__ rec_y, int16_t* __restrict__ rec_cb, int16_t* __restrict__ rec_cr) {
  if (j >= n) return;
  const hop_tu_rd_job jb = jobs[j];
  if (jb.log2_size > 3 || jb.log2_size < 2) return;                // < 2: an empty slot of a job table (k_rqt.inl)
  const int log2N = jb.log2_size, N = 1 << log2N, NN = N * N;
  const bool chroma = jb.comp != 0, live = lane < NN;
  const int bd = chroma ? pic.bd_c : pic.bd_y, pitch = chroma ? pic.pic_w >> 1 : pic.pic_w;
  const int x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int16_t* org = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int16_t* prd = (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  int16_t* rec = (jb.comp == 0 ? rec_y : jb.comp == 1 ? rec_cb : rec_cr) + (size_t)y0 * pitch + x0;
  const int r = lane >> log2N, c = lane & (N - 1);
  if (abs_sum[j] == 0) {
    if (!jb.is_intra) { if (lane == 0) nz_sse[j] = 0; return; }
    int e0 = 0;                                                    // reconstruction = prediction; its distortion against the original
    if (live) { const int v = prd[(size_t)r * pitch + c]; rec[(size_t)r * pitch + c] = (int16_t)v; e0 = v - (int)org[(size_t)r * pitch + c]; }
    const unsigned z0 = (unsigned)hopd_wave_sum((int)((unsigned)(e0 * e0) >> (unsigned)((bd - 8) << 1)));
    if (lane == 0) nz_sse[j] = z0;
    return;
  }
  const bool dst = jb.use_dst && N == 4 && !chroma;
  const int per = jb.qp_scaled / 6, rem = jb.qp_scaled % 6, transformShift = 15 - bd - log2N;
  const int dshift = 20 - 14 - transformShift, dadd = 1 << (dshift - 1), scale = c_inv_quant_scales[rem] << per;     // xDeQuant :1171-1182
  const int32_t* lv = levels + coef_off[j];
  const int dq = live ? clip16((clip16(lv[lane]) * scale + dadd) >> dshift) : 0;
  int rr;
  if (jb.flags & HOP_TU_RD_TS) rr = (int)(int16_t)((dq + (1 << (transformShift - 1))) >> transformShift);      // xITransformSkip
  else {
    sh.a[w][lane] = (int16_t)dq;
    __builtin_amdgcn_wave_barrier();
    const int s1 = 7, s2 = 12 - (bd - 8);
    // lane (j2 = r, n2 = c): b[j2][n2] = sum_k T[k][n2] a[k][j2], then rr(j2, n2) = sum_k T[k][n2] b[k][j2]
    int t[8];
    for (int q = 0; q < N; q++) t[q] = turd_t(dst, log2N, q, c);   // column n = c of the transform matrix
    int sum = 0;
    if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.a[w][q * N + r];
    sh.b[w][r * N + c] = (int16_t)clip16((sum + (1 << (s1 - 1))) >> s1);
    __builtin_amdgcn_wave_barrier();
    sum = 0;
    if (live) for (int q = 0; q < N; q++) sum += t[q] * sh.b[w][q * N + r];
    rr = clip16((sum + (1 << (s2 - 1))) >> s2);                    // reconstructed residual sample (row r, column c)
  }
  int e = 0;
  if (live) {
    if (jb.is_intra) {
      const int v = min((1 << bd) - 1, max(0, (int)prd[(size_t)r * pitch + c] + rr));
      rec[(size_t)r * pitch + c] = (int16_t)v;
      e = v - (int)org[(size_t)r * pitch + c];
    } else e = rr - ((int)org[(size_t)r * pitch + c] - (int)prd[(size_t)r * pitch + c]);
  }
  const unsigned ns = (unsigned)hopd_wave_sum((int)((unsigned)(e * e) >> (unsigned)((bd - 8) << 1)));
  if (lane == 0) nz_sse[j] = ns;
}

// one thread per TU: the bit-estimate table of its snapshot (TEncSbac::estBit as hop_cabac_est_bits) and the job records
__device__ static void turd_setup_body(const int i, const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,
                             const int32_t* __restrict__ entropy_bits, hop_estbits* __restrict__ eb, hop_rdoq_job* __restrict__ rq, hop_coeff_bits_job* __restrict__ cb,
                             const uint8_t* __restrict__ state_at = nullptr /* the TU's context states, if the caller holds a copy (LDS) */) {
  if (i >= n) return;
  const hop_tu_rd_job jb = jobs[i];
  const uint8_t* s = state_at ? state_at : ctx_in[jb.ctx_index].state;
  const int width = 1 << jb.log2_size, chroma = jb.comp != 0;
  int32_t* w = (int32_t*)eb;
  for (int k = 0; k < (int)(sizeof(hop_estbits) / 4); k++) w[k] = 0;
  for (int k = 0; k < 12; k++) { eb->blockCbpBits[k][0] = entropy_bits[s[k] ^ 0]; eb->blockCbpBits[k][1] = entropy_bits[s[k] ^ 1]; }
  for (int k = 0; k < 4; k++) { eb->blockRootCbpBits[k][0] = entropy_bits[s[11 + k] ^ 0]; eb->blockRootCbpBits[k][1] = entropy_bits[s[11 + k] ^ 1]; }
  for (int k = 0; k < 2; k++) for (int b = 0; b < 2; b++) eb->significantCoeffGroupBits[k][b] = entropy_bits[s[12 + 2 * chroma + k] ^ b];
  int firstCtx = 1, numCtx = 8;
  if (width >= 16) { firstCtx = chroma ? 12 : 21; numCtx = chroma ? 3 : 6; }
  else if (width == 8) { firstCtx = 9; numCtx = chroma ? 3 : 12; }
  const int base = 16 + (chroma ? 27 : 0);
  for (int b = 0; b < 2; b++) eb->significantBits[0][b] = entropy_bits[s[base] ^ b];
  for (int k = firstCtx; k < firstCtx + numCtx; k++) for (int b = 0; b < 2; b++) eb->significantBits[k][b] = entropy_bits[s[base + k] ^ b];
  const int cbt = jb.log2_size - 2;
  const int off = chroma ? 0 : (cbt * 3 + ((cbt + 1) >> 2)), shf = chroma ? cbt : ((cbt + 3) >> 2);
  const uint8_t* px = s + 58 + 15 * chroma; const uint8_t* py = s + 88 + 15 * chroma;
  const int gmax = (width == 4) ? 3 : (width == 8) ? 5 : (width == 16) ? 7 : 9;       // g_uiGroupIdx[width - 1]
  int bitsX = 0, bitsY = 0, c;
  for (c = 0; c < gmax; c++) { const int o = off + (c >> shf); eb->lastXBits[c] = bitsX + entropy_bits[px[o] ^ 0]; bitsX += entropy_bits[px[o] ^ 1]; }
  eb->lastXBits[c] = bitsX;
  for (c = 0; c < gmax; c++) { const int o = off + (c >> shf); eb->lastYBits[c] = bitsY + entropy_bits[py[o] ^ 0]; bitsY += entropy_bits[py[o] ^ 1]; }
  eb->lastYBits[c] = bitsY;
  const int no = chroma ? 8 : 16, na = chroma ? 2 : 4, oo = 118 + (chroma ? 16 : 0), oa = 142 + (chroma ? 4 : 0);
  for (int k = 0; k < no; k++) { eb->greaterOneBits[k][0] = entropy_bits[s[oo + k] ^ 0]; eb->greaterOneBits[k][1] = entropy_bits[s[oo + k] ^ 1]; }
  for (int k = 0; k < na; k++) { eb->levelAbsBits[k][0] = entropy_bits[s[oa + k] ^ 0]; eb->levelAbsBits[k][1] = entropy_bits[s[oa + k] ^ 1]; }
  hop_rdoq_job r;
  r.log2_size = jb.log2_size; r.comp = jb.comp; r.is_intra = jb.is_intra; r.scan_idx = jb.scan_idx; r.tr_depth = jb.tr_depth; r.qp_scaled = jb.qp_scaled;
  r.bit_depth = jb.bit_depth; r.sign_hide = jb.sign_hide; r.lambda = jb.lambda_rdoq; r.coeff_offset = coef_off[i]; r.estbits_index = i; r.reserved = 0;
  rq[i] = r;
  hop_coeff_bits_job b;
  b.log2_size = jb.log2_size; b.comp = jb.comp; b.scan_idx = jb.scan_idx; b.sign_hide = jb.sign_hide; b.use_ts = jb.use_ts; b.ts_flag = (jb.flags & HOP_TU_RD_TS) ? 1 : 0; b.ctx_index = jb.ctx_index;
  b.cbf_ctx_plus1 = 1 + 4 * chroma + (chroma ? jb.tr_depth : (jb.tr_depth == 0 ? 1 : 0));           // getCtxQtCbf, TComDataCU.cpp:1848-1859
  b.coeff_offset = coef_off[i];
  cb[i] = b;
}

// one thread per TU: integer bits, the two RD costs, the choice (TEncSearch.cpp:7004-7032; calcRdCost TComRdCost.cpp:59-111)
__device__ static void turd_decide_body(const int i, const hop_tu_rd_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in, const int64_t* __restrict__ coef_off,
                              const int32_t* __restrict__ entropy_bits, const uint32_t* __restrict__ abs_sum, const unsigned long long* __restrict__ frac,
                              const uint32_t* __restrict__ zero_sse, const uint32_t* __restrict__ nz_sse, int32_t* __restrict__ levels,
                              hop_tu_rd_result* __restrict__ res) {
  if (i >= n) return;
  const hop_tu_rd_job jb = jobs[i];
  const uint8_t* s = ctx_in[jb.ctx_index].state;
  const uint32_t left = (uint32_t)s[150] | ((uint32_t)s[151] << 8);                                  // fraction below one bit the coder carries
  const int chroma = jb.comp != 0;
  const uint32_t zeroDist = chroma ? (uint32_t)(int)(jb.dist_weight * zero_sse[i]) : zero_sse[i];   // getDistPart, TComRdCost.cpp:493-502
  const uint32_t nzDist = chroma ? (uint32_t)(int)(jb.dist_weight * nz_sse[i]) : nz_sse[i];
  const uint32_t singleBits = (uint32_t)((left + frac[i]) >> 15);
  hop_tu_rd_result r;
  r.abs_sum = abs_sum[i]; r.zero_dist = zeroDist; r.nonzero_dist = 0; r.bits = singleBits; r.null_bits = 0; r.dist = zeroDist; r.pad = 0;
  if (jb.is_intra) {                                               // xIntraCodingLumaBlk has no cbf-zero test: the block is what RDOQ made of it
    r.dist = r.abs_sum ? nzDist : zeroDist;
    r.nonzero_dist = r.abs_sum ? nzDist : 0;
    r.cost = (double)(uint32_t)floor((double)r.dist + (double)((int)(singleBits * jb.lambda_rd + .5)));
  } else if (r.abs_sum) {
    r.nonzero_dist = nzDist;
    const double singleCost = (double)(uint32_t)floor((double)nzDist + (double)((int)(singleBits * jb.lambda_rd + .5)));
    const int cbfCtx = 4 * chroma + (chroma ? jb.tr_depth : (jb.tr_depth == 0 ? 1 : 0));
    r.null_bits = (uint32_t)((left + (unsigned long long)entropy_bits[s[cbfCtx] ^ 0]) >> 15);       // encodeQtCbfZero from the snapshot
    const double nullCost = (double)(uint32_t)floor((double)zeroDist + (double)((int)(r.null_bits * jb.lambda_rd + .5)));
    if (jb.flags & HOP_TU_RD_KEEP) { r.dist = nzDist; r.cost = singleCost; }       // the transform-skip retry compares this cost itself (:7258-7262)
    else if (nullCost < singleCost) {
      r.abs_sum = 0; r.cost = nullCost;
      int32_t* lv = levels + coef_off[i];
      for (int k = 0; k < (1 << (2 * jb.log2_size)); k++) lv[k] = 0;
    } else { r.dist = nzDist; r.cost = singleCost; }
  } else {
    r.cost = (double)(uint32_t)floor((double)zeroDist + (double)((int)(singleBits * jb.lambda_rd + .5)));
  }
  r.cbf = r.abs_sum != 0;
  res[i] = r;
}

