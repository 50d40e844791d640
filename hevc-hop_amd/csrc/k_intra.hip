// k_intra.hip -- the distortion half of the 35-mode intra rough search.
// Replaces, for one luma block of TEncSearch::estIntraPredQT (TLibEncoder/TEncSearch.cpp:2430-2461):
//   TComPattern::initAdiPattern  (TLibCommon/TComPattern.cpp:179-313): fillReferenceSamples (:374-558) + the
//     [1 2 1] / strong-32 reference smoothing (:237-299),
//   TComPattern::getPredictorPtr (:583-607, filter decision table :49-56),
//   TComPrediction::predIntraLumaAng (TLibCommon/TComPrediction.cpp:340-372): xPredIntraPlanar (:1468-1505),
//     predIntraGetPredValDC (:130-167) + xDCPredFiltering (:1521-1541), xPredIntraAng (:192-338),
//   TComRdCost::calcHAD (TLibCommon/TComRdCost.cpp:391-425)
// for all 35 modes; the caller adds xModeBitsIntra * sqrt(lambda) (:2460-2461) on the host.
// Neighbour samples come from the context's reconstruction picture; availability is given per 4-sample unit in
// the reference's bNeighborFlags order (it depends on the CU structure and coding order, which the caller owns).
//
// One workgroup per block.  The reference line (4N+1 samples) and its smoothed copy live in LDS; every predicted
// sample is a closed form of the line (planar and DC included), so a work item is (mode, 8x8 block) = one wave:
// lane = sample, Hadamard across the wave, one LDS atomic per (mode, block).  No prediction buffer, no barriers
// between modes.
#include "hop_dev.h"

__constant__ uint8_t c_intra_filter[5] = { 10, 7, 1, 0, 10 };        // TComPattern.cpp:49-56
__constant__ int c_ang[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int c_inv_ang[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

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

__global__ __launch_bounds__(256) void k_intra_rough(const hop_intra_job* __restrict__ jobs, hop_pics pic, const int16_t* __restrict__ rec_y,
                                                     uint32_t* __restrict__ satd_out) {
  __shared__ IntraShared sh;
  const hop_intra_job* jp = jobs + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = jp->size;
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int bd = pic.bd_y, maxVal = (1 << bd) - 1;
  intra_setup(sh, jp, pic, rec_y, tid);
  // ---- 35 predictions + calcHAD ----
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
  if (tid < 35) satd_out[(size_t)blockIdx.x * 35 + tid] = sh.satd[tid] >> (bd - 8);
}


// the prediction of ONE mode per block, written into the context's prediction picture (the first step of
// TEncSearch::xIntraCodingLumaBlk, TLibEncoder/TEncSearch.cpp:1046-1049: initAdiPattern + predIntraLumaAng)
__global__ __launch_bounds__(256) void k_intra_pred(const hop_intra_job* __restrict__ jobs, const int32_t* __restrict__ modes, hop_pics pic,
                                                    const int16_t* __restrict__ rec_y) {
  __shared__ IntraShared sh;
  const hop_intra_job* jp = jobs + blockIdx.x;
  const int tid = threadIdx.x;
  const int N = jp->size, x0 = jp->x, y0 = jp->y;
  if (N == 0) return;                                                  // an empty slot of a batch (hop_intra_luma_search: a CU without a candidate in this pass)
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
  const int maxVal = (1 << pic.bd_y) - 1, mode = modes[blockIdx.x];
  intra_setup(sh, jp, pic, rec_y, tid);
  int16_t* dst = pic.pred_y + (size_t)y0 * pic.pic_w + x0;
  for (int i = tid; i < N * N; i += 256) { const int r = i >> log2N, c = i & (N - 1); dst[(size_t)r * pic.pic_w + c] = (int16_t)intra_sample(sh, N, log2N, mode, c, r, maxVal); }
}

// the chroma prediction of xIntraCodingChromaBlk (TLibEncoder/TEncSearch.cpp:1200-1215: initAdiPatternChroma + predIntraChromaAng) for both planes: block i of
// size jobs[i].size at the chroma position (x / 2, y / 2), availability per 2-sample unit, mode modes[i] (the caller resolves DM_CHROMA_IDX to the luma mode)
__global__ __launch_bounds__(256) void k_intra_pred_chroma(const hop_intra_job* __restrict__ jobs, const int32_t* __restrict__ modes, hop_pics pic,
                                                           const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr) {
  __shared__ IntraShared sh;
  const int bi = blockIdx.x >> 1, comp = 1 + (blockIdx.x & 1);
  const hop_intra_job* jp = jobs + bi;
  const int tid = threadIdx.x;
  const int N = jp->size, x0 = jp->x >> 1, y0 = jp->y >> 1, pitch = pic.pic_w >> 1;
  if (N == 0) return;                                                  // an empty slot of a batch (hop_intra_chroma_search: a CU without a transform unit at this node)
  const int log2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  const int maxVal = (1 << pic.bd_c) - 1, mode = modes[bi];
  intra_setup_plane(sh, jp, comp == 1 ? rec_cb : rec_cr, pitch, pic.bd_c, 2, x0, y0, nullptr, false, tid);
  int16_t* dst = (comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  for (int i = tid; i < N * N; i += 256) { const int r = i >> log2N, c = i & (N - 1); dst[(size_t)r * pitch + c] = (int16_t)intra_sample(sh, N, log2N, mode, c, r, maxVal, false); }
}

int hop_launch_intra_pred_chroma(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_pred_chroma, dim3(2 * n), dim3(256), 0, c->stream, d_jobs, d_modes, hop_make_pics(c), c->rec[1], c->rec[2]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_pred_chroma launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

int hop_launch_intra_pred(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_pred, dim3(n), dim3(256), 0, c->stream, d_jobs, d_modes, hop_make_pics(c), c->rec[0]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_pred launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

int hop_launch_intra(hop_ctx* c, int n, const hop_intra_job* d_jobs, uint32_t* d_satd) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_rough, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), c->rec[0], d_satd);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_rough launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
