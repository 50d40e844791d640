// k_intra_dev.inl -- device side of the intra predictor and rough search (see k_intra.hip for what it replaces); included by k_intra.hip and k_cabac.hip (k_walk.inl)
static __constant__ uint8_t c_intra_filter[5] = { 10, 7, 1, 0, 10 };        // TComPattern.cpp:49-56
static __constant__ int c_ang[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static __constant__ int c_inv_ang[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

struct IntraShared {
  int L[2][4 * 64 + 1 + 3];       // [0] unfiltered, [1] smoothed; L[2N] = corner, L[2N-1-i] = left row i, L[2N+1+i] = top col i
  int line[4 * 64 + 8];           // unit-ordered line buffer of fillReferenceSamples (the corner owns a whole unit)
  int16_t org[64 * 64];
  unsigned int satd[35];
  int dc;
};

// one predicted sample of mode `mode` at column x, row y
// luma: the smoothed line where the mode asks for it and the edge filters of small blocks; chroma (predIntraChromaAng, TComPrediction.cpp:375-390): neither
__device__ static inline int intra_sample(const IntraShared& sh, int N, int log2N, int mode, int x, int y, int maxVal, bool luma = true) {
  int diff = min(abs(mode - 10), abs(mode - 26));
  const bool filt = luma && (mode != 1) && diff > c_intra_filter[log2N - 2];
  const int* L = sh.L[filt ? 1 : 0];
  const int* top = L + 2 * N + 1;                       // top[i], i = -1 .. 2N-1
#define LEFT(i) (L[2 * N - 1 - (i)])
  if (mode == 0)                                        // planar (closed form of :1486-1503)
    return ((N - 1 - x) * LEFT(y) + (x + 1) * top[N] + (N - 1 - y) * top[x] + (y + 1) * LEFT(N) + N) >> (log2N + 1);
  const bool edge = luma && N <= 16;                    // bFilter, :358-366
  if (mode == 1) {                                      // DC + xDCPredFiltering
    const int dcv = sh.dc;
    if (!edge) return dcv;
    if (x == 0 && y == 0) return (top[0] + LEFT(0) + 2 * dcv + 2) >> 2;
    if (y == 0) return (top[x] + 3 * dcv + 2) >> 2;
    if (x == 0) return (LEFT(y) + 3 * dcv + 2) >> 2;
    return dcv;
  }
  const bool modeVer = mode >= 18;
  int ang = modeVer ? mode - 26 : -(mode - 10);
  const int aabs = abs(ang), sign = ang < 0 ? -1 : 1;
  const int invAngle = c_inv_ang[aabs];
  ang = sign * c_ang[aabs];
  // (k,l) = (row, column) of the vertical-mode formulation; horizontal modes are its transpose
  const int k = modeVer ? y : x, l = modeVer ? x : y;
  // refMain[i] (i >= -N .. 2N): main = top for vertical modes, left for horizontal; index 0 = corner
  auto refMain = [&](int i) -> int {
    if (i >= 0) return modeVer ? top[i - 1] : LEFT(i - 1);
    const int s = (128 + (-i) * invAngle) >> 8;         // projection of the side reference, :257-262
    return modeVer ? LEFT(s - 1) : top[s - 1];
  };
  if (ang == 0) {
    int v = refMain(l + 1);
    if (edge && l == 0) {                               // first column (before the transpose), :287-293
      const int side_k = modeVer ? LEFT(k) : top[k], side_0 = L[2 * N];
      v = min(maxVal, max(0, v + ((side_k - side_0) >> 1)));
    }
    return v;
  }
  const int deltaPos = (k + 1) * ang, di = deltaPos >> 5, df = deltaPos & 31;
  const int idx = l + di + 1;
  if (df) return ((32 - df) * refMain(idx) + df * refMain(idx + 1) + 16) >> 5;
  return refMain(idx);
#undef LEFT
}

// reference line of one block: fillReferenceSamples + smoothing + DC value into sh (all 256 threads; ends with a barrier).
// rec / pitch / bd: the plane the neighbours come from; us: samples per availability flag (4 luma, 2 chroma: TComPattern.cpp:325-331);
// org: the original plane for the rough search (may be null); luma: build the smoothed line too.
__device__ static inline void intra_setup_plane(IntraShared& sh, const hop_intra_job* jp, const int16_t* __restrict__ rec_plane, int pitch, int bd, int us, int x0, int y0,
                                                const int16_t* __restrict__ org_plane, bool luma, int tid) {
  const int N = jp->size, U = N / us, units = 4 * U + 1;
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int dcDefault = 1 << (bd - 1);
  const int16_t* rec = rec_plane + (size_t)y0 * pitch + x0;
  if (tid < 35) sh.satd[tid] = 0;
  if (org_plane) for (int i = tid; i < N * N; i += 256) { int r = i / N, c = i - r * N; sh.org[i] = org_plane[(size_t)(y0 + r) * pitch + x0 + c]; }
  // ---- fillReferenceSamples :374-558 : gather the available units, DC elsewhere ----
  for (int i = tid; i < units * us; i += 256) {
    const int u = i / us, s = i - u * us;
    int v = dcDefault;
    if (jp->flags[u]) {
      if (u < 2 * U) { const int j = 2 * U - 1 - u; v = rec[(ptrdiff_t)(us * j + (us - 1 - s)) * pitch - 1]; }   // left / below-left, stored upwards
      else if (u == 2 * U) v = rec[-(ptrdiff_t)pitch - 1];                                                     // corner (a whole unit of copies)
      else v = rec[-(ptrdiff_t)pitch + us * (u - 2 * U - 1) + s];                                              // above / above-right
    }
    sh.line[i] = v;
  }
  __syncthreads();
  if (tid == 0) {                                       // substitution :505-545 (sequential over <= 65 units)
    int navail = 0;
    for (int u = 0; u < units; u++) navail += jp->flags[u] ? 1 : 0;
    if (navail != 0 && navail != units) {
      int cur = 0;
      while (cur < units) {
        if (!jp->flags[cur]) {
          if (cur == 0) {
            int nxt = 1;
            while (nxt < units && !jp->flags[nxt]) nxt++;
            const int ref = sh.line[nxt * us];
            while (cur < nxt) { for (int i = 0; i < us; i++) sh.line[cur * us + i] = ref; cur++; }
          } else {
            const int ref = sh.line[cur * us - 1];
            for (int i = 0; i < us; i++) sh.line[cur * us + i] = ref;
            cur++;
          }
        } else cur++;
      }
    }
  }
  __syncthreads();
  const int n = 4 * N + 1;
  for (int i = tid; i < n; i += 256)                    // copy out :547-556
    sh.L[0][i] = i < 2 * N ? sh.line[i] : i == 2 * N ? sh.line[2 * U * us] : sh.line[(2 * U + 1) * us + (i - 2 * N - 1)];
  __syncthreads();
  // ---- smoothing, TComPattern.cpp:237-299 ----
  {
    const int* L = sh.L[0];
    bool strong = false;
    if (luma && jp->strong && N >= 32) {
      const int bl = L[0], tl = L[2 * N], tr = L[n - 1], thr = 1 << (bd - 5);
      strong = abs(bl + tl - 2 * L[N]) < thr && abs(tl + tr - 2 * L[3 * N]) < thr;
    }
    if (luma) for (int i = tid; i < n; i += 256) {
      int v;
      if (i == 0 || i == n - 1) v = L[i];
      else if (strong) {
        const int shift = log2N + 1;
        if (i == 2 * N) v = L[i];
        else if (i < 2 * N) v = ((2 * N - i) * L[0] + i * L[2 * N] + N) >> shift;
        else v = ((2 * N - (i - 2 * N)) * L[2 * N] + (i - 2 * N) * L[n - 1] + N) >> shift;
      } else v = (L[i - 1] + 2 * L[i] + L[i + 1] + 2) >> 2;
      sh.L[1][i] = v;
    }
    if (tid == 0) {                                     // predIntraGetPredValDC with bAbove && bLeft, :130-157
      int sum = 0;
      for (int i = 0; i < N; i++) sum += L[2 * N + 1 + i] + L[2 * N - 1 - i];
      sh.dc = (sum + N) / (2 * N);
    }
  }
  __syncthreads();
}
__device__ static inline void intra_setup(IntraShared& sh, const hop_intra_job* jp, const hop_pics& pic, const int16_t* __restrict__ rec_y, int tid) {
  intra_setup_plane(sh, jp, rec_y, pic.pic_w, pic.bd_y, 4, jp->x, jp->y, pic.org_y, true, tid);
}

// ---- the three kernels' bodies (every thread of a 256-thread workgroup calls; barriers inside): shared by k_intra.hip and the candidate walks of k_walk.inl ----
// 35 predictions + calcHAD of one block into satd35[0..34]
__device__ static inline void intra_rough_body(IntraShared& sh, const hop_intra_job* jp, hop_pics pic, const int16_t* rec_y, uint32_t* satd35) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = jp->size;
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int bd = pic.bd_y, maxVal = (1 << bd) - 1;
  intra_setup(sh, jp, pic, rec_y, tid);
  if (N >= 8) {
    const int bw = N >> 3, nblk = bw * bw;
    for (int it = wave; it < 35 * nblk; it += 4) {
      const int mode = it / nblk, blk = it - mode * nblk;
      const int px = (blk % bw) * 8 + (lane & 7), py = (blk / bw) * 8 + (lane >> 3);
      const int d = (int)sh.org[py * N + px] - intra_sample(sh, N, log2N, mode, px, py, maxVal);
      const int s = hopd_satd8x8_wave(d, lane);
      if (lane == 0) atomicAdd(&sh.satd[mode], (unsigned)s);
    }
  } else {                                              // 4x4: four modes per wave, one per 16 lanes
    for (int m0 = wave * 4; m0 < 35; m0 += 16) {
      const int mode = m0 + (lane >> 4);
      const bool act = mode < 35;
      const int px = lane & 3, py = (lane >> 2) & 3;
      const int d = act ? (int)sh.org[py * 4 + px] - intra_sample(sh, N, log2N, act ? mode : 0, px, py, maxVal) : 0;
      const int s = hopd_satd4x4_quad(d, lane);
      if (act && (lane & 15) == 0) atomicAdd(&sh.satd[mode], (unsigned)s);
    }
  }
  __syncthreads();
  if (tid < 35) satd35[tid] = sh.satd[tid] >> (bd - 8);
}
// the prediction of one mode into the prediction picture (TEncSearch::xIntraCodingLumaBlk, TLibEncoder/TEncSearch.cpp:1046-1049); size 0 = an empty slot
__device__ static inline void intra_pred_body(IntraShared& sh, const hop_intra_job* jp, const int mode, hop_pics pic, const int16_t* rec_y) {
  const int tid = threadIdx.x;
  const int N = jp->size, x0 = jp->x, y0 = jp->y;
  if (N == 0) return;
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int maxVal = (1 << pic.bd_y) - 1;
  intra_setup(sh, jp, pic, rec_y, tid);
  int16_t* dst = pic.pred_y + (size_t)y0 * pic.pic_w + x0;
  for (int i = tid; i < N * N; i += 256) { const int r = i >> log2N, c = i & (N - 1); dst[(size_t)r * pic.pic_w + c] = (int16_t)intra_sample(sh, N, log2N, mode, c, r, maxVal); }
}
// the chroma prediction of xIntraCodingChromaBlk (:1200-1215) for plane comp (1 / 2)
__device__ static inline void intra_pred_chroma_body(IntraShared& sh, const hop_intra_job* jp, const int mode, const int comp, hop_pics pic, const int16_t* rec_cb, const int16_t* rec_cr) {
  const int tid = threadIdx.x;
  const int N = jp->size, x0 = jp->x >> 1, y0 = jp->y >> 1, pitch = pic.pic_w >> 1;
  if (N == 0) return;
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  const int maxVal = (1 << pic.bd_c) - 1;
  intra_setup_plane(sh, jp, comp == 1 ? rec_cb : rec_cr, pitch, pic.bd_c, 2, x0, y0, nullptr, false, tid);
  int16_t* dst = (comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  for (int i = tid; i < N * N; i += 256) { const int r = i >> log2N, c = i & (N - 1); dst[(size_t)r * pitch + c] = (int16_t)intra_sample(sh, N, log2N, mode, c, r, maxVal, false); }
}
