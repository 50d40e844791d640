// k_pred.hip -- the final (normative, decoder-shared) inter/GT predictor and the
// distortion kernels (row a12).
// Replaces TComPrediction::xPredInterLumaBlk / xPredInterChromaBlk incl. the GT branch
// (TLibCommon/TComPrediction.cpp:639-720, :1235-1347), xPredGTLuma / xPredGTChroma (:723-805, :1351-1420),
// calcParamProjective(C) (:807-859) and ProjectiveTransform (:904-1030); DCT-IF from
// TComInterpolationFilter.cpp:55-75 (taps), :92-152 (filterCopy), :170-245 (filter<>).
// The result must be Pel-exact: the decoder runs the same double arithmetic (TDecCu.cpp:487).
// One workgroup per PU and colour plane; the doubled patch lives in LDS; IEEE double, no contraction.
#include "hop_dev.h"

__constant__ int16_t c_taps8[4][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int16_t c_taps4[8][4] = {
  { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

// Uni-prediction sample at integer (x,y) of `src`, phase (xf,yf): TComPrediction.cpp:662-678 / :1272-1289.
// NT = 8 (luma) or 4 (chroma).
template <int NT>
__device__ static inline int mc_sample(const int16_t* __restrict__ src, int stride, int x, int y, int xf, int yf, int bd) {
  const int16_t* cx = NT == 8 ? c_taps8[xf] : c_taps4[xf];
  const int16_t* cy = NT == 8 ? c_taps8[yf] : c_taps4[yf];
  constexpr int HALF = NT / 2 - 1;
  const int headRoom = 14 - bd, maxVal = (1 << bd) - 1;
  const int16_t* p = src + (ptrdiff_t)y * stride + x;
  if (xf == 0 && yf == 0) return p[0];                        // filterCopy(isFirst == isLast): plain copy, no clip
  if (yf == 0 || xf == 0) {                                   // single pass, filter<N,*,true,true>: shift 6, offset 32, clip
    int sum = 0;
    if (yf == 0) { for (int k = 0; k < NT; k++) sum += p[k - HALF] * cx[k]; }
    else { for (int k = 0; k < NT; k++) sum += p[(ptrdiff_t)(k - HALF) * stride] * cy[k]; }
    int16_t val = (int16_t)((sum + 32) >> 6);
    if (val < 0) val = 0;
    if (val > maxVal) val = (int16_t)maxVal;
    return val;
  }
  int sum2 = 0;
  const int shift1 = 6 - headRoom, off1 = -8192 * (1 << shift1);
  for (int k = 0; k < NT; k++) {                              // horizontal (isFirst, !isLast) then vertical (!isFirst, isLast)
    const int16_t* q = p + (ptrdiff_t)(k - HALF) * stride;
    int sum = 0;
    for (int j = 0; j < NT; j++) sum += q[j - HALF] * cx[j];
    int16_t t = (int16_t)((sum + off1) >> shift1);
    sum2 += t * cy[k];
  }
  const int shift2 = 6 + headRoom;
  int16_t val = (int16_t)((sum2 + (1 << (shift2 - 1)) + (8192 << 6)) >> shift2);
  if (val < 0) val = 0;
  if (val > maxVal) val = (int16_t)maxVal;
  return val;
}

struct PredShared { int16_t patch[128 * 130]; };

// one job, one plane, on the 256 threads of the calling workgroup (a barrier inside the GT branch: uniform over the workgroup)
// tile != nullptr: the block goes to `tile` (pitch = the block's width in this plane) instead of the prediction picture
__device__ static void pred_inter_body(PredShared& sh, const hop_pred_job& jb, const int comp, const hop_pics& pic, int16_t* tile = nullptr) {
  const int tid = threadIdx.x;
  const bool chroma = comp != 0;
  const int bw = chroma ? jb.w >> 1 : jb.w, bh = chroma ? jb.h >> 1 : jb.h;       // block size in this plane
  const int16_t* ref = comp == 0 ? pic.ss_y : comp == 1 ? pic.ss_cb : pic.ss_cr;
  int16_t* dst = comp == 0 ? pic.pred_y : comp == 1 ? pic.pred_cb : pic.pred_cr;
  const int rstride = chroma ? pic.stride_c : pic.stride_y, dpitch = tile ? bw : (chroma ? pic.pic_w >> 1 : pic.pic_w);
  const int bx = chroma ? jb.pu_x >> 1 : jb.pu_x, by = chroma ? jb.pu_y >> 1 : jb.pu_y;
  const int bd = chroma ? pic.bd_c : pic.bd_y;
  const int ish = chroma ? 3 : 2, fmask = chroma ? 7 : 3;
  const int ix = jb.mv_x >> ish, iy = jb.mv_y >> ish, xf = jb.mv_x & fmask, yf = jb.mv_y & fmask;
  bool any = false;
  for (int k = 0; k < 8; k++) any = any || jb.gt[k] != 0;
  if (tile) dst = tile;
  else dst += (size_t)(by + (chroma ? jb.dst_row_off >> 1 : jb.dst_row_off)) * dpitch + bx;   // dst_row_off: the candidate slot's copy of the prediction picture
  if (!jb.use_gt || !any) {                                   // plain motion compensation, :650-678 / :1246-1289
    const int16_t* r = ref + (ptrdiff_t)(by + iy) * rstride + bx + ix;
    for (int i = tid; i < bw * bh; i += 256) {
      int y = i / bw, x = i - y * bw;
      dst[(size_t)y * dpitch + x] = (int16_t)(chroma ? mc_sample<4>(r, rstride, x, y, xf, yf, bd) : mc_sample<8>(r, rstride, x, y, xf, yf, bd));
    }
    return;
  }
  // ---- GT branch: doubled patch at (mv_int) - (bw/2, bh/2), interpolated at the MV's phase (:683-713 / :1295-1339) ----
  const int PP = 2 * bw + 2;
  {
    const int16_t* r = ref + (ptrdiff_t)(by + iy - bh / 2) * rstride + bx + ix - bw / 2;
    for (int i = tid; i < 4 * bw * bh; i += 256) {
      int y = i / (2 * bw), x = i - y * (2 * bw);
      sh.patch[y * PP + x] = (int16_t)(chroma ? mc_sample<4>(r, rstride, x, y, xf, yf, bd) : mc_sample<8>(r, rstride, x, y, xf, yf, bd));
    }
  }
  __syncthreads();
  // ---- homography: xPredGTLuma :729-788 / xPredGTChroma :1357-1403 ----
  const int nssWindow = (min(bh, bw) >> 1) * 2;
  int lastStepI = nssWindow >> 6; if (lastStepI == 0) lastStepI = 1;
  double h[9];
  {
    const double Wd = (double)(2 * bw) - 1.0, Hd = (double)(2 * bh) - 1.0;
    double x0, x1, x2, x3, y0, y1, y2, y3;
    if (!chroma) {                                            // integer corners, calcParamProjective :807-832
      const int cx0 = jb.gt[0] * lastStepI, cx1 = jb.gt[2] * lastStepI + 2 * bw - 1, cx2 = jb.gt[4] * lastStepI + 2 * bw - 1, cx3 = jb.gt[6] * lastStepI;
      const int cy0 = jb.gt[1] * lastStepI, cy1 = jb.gt[3] * lastStepI, cy2 = jb.gt[5] * lastStepI + 2 * bh - 1, cy3 = jb.gt[7] * lastStepI + 2 * bh - 1;
      const double dx1 = (double)cx1 - cx2, dx2 = (double)cx3 - cx2, dx3 = (double)cx0 - cx1 + cx2 - cx3;
      const double dy1 = (double)cy1 - cy2, dy2 = (double)cy3 - cy2, dy3 = (double)cy0 - cy1 + cy2 - cy3;
      h[2] = ((dx3 * dy2 - dx2 * dy3) / (dx1 * dy2 - dx2 * dy1)) / Wd;
      h[5] = ((dx1 * dy3 - dx3 * dy1) / (dx1 * dy2 - dx2 * dy1)) / Hd;
      h[0] = (double)(cx1 - cx0) / Wd + h[2] * cx1;
      h[3] = (double)(cx3 - cx0) / Hd + h[5] * cx3;
      h[6] = (double)cx0;
      h[1] = (double)(cy1 - cy0) / Wd + h[2] * cy1;
      h[4] = (double)(cy3 - cy0) / Hd + h[5] * cy3;
      h[7] = (double)cy0;
    } else {                                                  // double corners = GT/2, calcParamProjectiveC :834-859
      const double ls = (double)lastStepI;
      x0 = ((double)jb.gt[0] / 2) * ls;                  y0 = ((double)jb.gt[1] / 2) * ls;
      x1 = (((double)jb.gt[2] / 2) * ls) + 2 * bw - 1;   y1 = ((double)jb.gt[3] / 2) * ls;
      x2 = (((double)jb.gt[4] / 2) * ls) + 2 * bw - 1;   y2 = (((double)jb.gt[5] / 2) * ls) + 2 * bh - 1;
      x3 = ((double)jb.gt[6] / 2) * ls;                  y3 = (((double)jb.gt[7] / 2) * ls) + 2 * bh - 1;
      const double dx1 = x1 - x2, dx2 = x3 - x2, dx3 = x0 - x1 + x2 - x3;
      const double dy1 = y1 - y2, dy2 = y3 - y2, dy3 = y0 - y1 + y2 - y3;
      h[2] = ((dx3 * dy2 - dx2 * dy3) / (dx1 * dy2 - dx2 * dy1)) / Wd;
      h[5] = ((dx1 * dy3 - dx3 * dy1) / (dx1 * dy2 - dx2 * dy1)) / Hd;
      h[0] = (x1 - x0) / Wd + h[2] * x1;
      h[3] = (x3 - x0) / Hd + h[5] * x3;
      h[6] = x0;
      h[1] = (y1 - y0) / Wd + h[2] * y1;
      h[4] = (y3 - y0) / Hd + h[5] * y3;
      h[7] = y0;
    }
    h[8] = 1.0;
  }
  // ---- warp: ProjectiveTransform :919-1028 with W = 2bw, H = 2bh ----
  const int offX = bw - bw / 2, offY = bh - bh / 2;           // W/2 - (W/2/2)
  const int m = nssWindow / 2;
  const int16_t* centre = sh.patch + (bh / 2) * PP + bw / 2;  // dst1 += width/2 + (height/2)*stride, :776 / :1391
  for (int i = tid; i < bw * bh; i += 256) {
    const int py = i / bw, px = i - py * bw;
    const int x = px + offX, y = py + offY;
    double Fx = (h[0] * x + h[3] * y + h[6]) / (h[2] * x + h[5] * y + h[8]);
    double Fy = (h[1] * x + h[4] * y + h[7]) / (h[2] * x + h[5] * y + h[8]);
    int Y = (int)Fy - offY, X = (int)Fx - offX;
    double q = (Fy - offY - (double)Y), p = (Fx - offX - (double)X);
    if (Y < -m) Y = -m;
    if (X < -m) X = -m;
    if (Y > m + bh - 1) Y = m + bh - 1;
    if (X > m + bw - 1) X = m + bw - 1;
    if (Y + 1 > m + bh - 1) Y = m + bh - 2;
    if (X + 1 > m + bw - 1) X = m + bw - 2;
    const int16_t* pa = centre + Y * PP + X;
    double v = (1.0 - q) * ((1.0 - p) * (double)pa[0] + p * (double)pa[1]);
    v += q * ((1.0 - p) * (double)pa[PP] + p * (double)pa[PP + 1]);
    if (v > 255) v = 255;                                     // hard-coded 8-bit clip, :969-972
    if (v < 0) v = 0;
    dst[(size_t)py * dpitch + px] = (int16_t)(v + 0.5);
  }
}
// grid = (n jobs, 3 planes)
__global__ __launch_bounds__(256) void k_pred_inter(const hop_pred_job* __restrict__ jobs, hop_pics pic) {
  __shared__ PredShared sh;
  const hop_pred_job jb = jobs[blockIdx.x];
  pred_inter_body(sh, jb, blockIdx.y, pic);
}

int hop_launch_pred(hop_ctx* c, int n, const hop_pred_job* d_jobs) {
  const int pr = hop_prof_begin(c, HOP_K_PRED, (uint64_t)n);
  hipLaunchKernelGGL(k_pred_inter, dim3(n, 3), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c));
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "pred_inter launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

__global__ void k_pred_jobs_from_results(int n, const int32_t* __restrict__ index, const hop_pu_job* __restrict__ jobs,
                                         const hop_pu_result* __restrict__ res, hop_pred_job* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = index ? index[i] : i;
  const hop_pu_job jb = jobs[k];
  const hop_pu_result r = res[k];
  hop_pred_job p;
  p.pu_x = jb.pu_x; p.pu_y = jb.pu_y; p.w = jb.w; p.h = jb.h; p.dst_row_off = 0;
  if (r.not_valid) { p.mv_x = 0; p.mv_y = 0; p.use_gt = 0; for (int q = 0; q < 8; q++) p.gt[q] = 0; }
  else {
    p.mv_x = (r.mv_final[0] << 2) + (r.half_final[0] << 1) + r.qter_final[0];      // TEncSearch.cpp:4654-4656
    p.mv_y = (r.mv_final[1] << 2) + (r.half_final[1] << 1) + r.qter_final[1];
    p.use_gt = 1;
    for (int q = 0; q < 8; q++) p.gt[q] = r.gt[q];
  }
  out[i] = p;
}

extern "C" int hop_pred_jobs_from_results_device(hop_ctx* c, int n, const int32_t* d_index, const hop_pu_job* d_jobs,
                                                 const hop_pu_result* d_results, hop_pred_job* d_out) {
  if (!c || n < 0 || (n && (!d_jobs || !d_results || !d_out))) return hop_set_err(c, HOP_ERR_ARG, "hop_pred_jobs_from_results_device: bad argument");
  if (n == 0) return HOP_OK;
  hipLaunchKernelGGL(k_pred_jobs_from_results, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_index, d_jobs, d_results, d_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "pred_jobs_from_results launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------
// distortion between the original and the prediction picture (row a12):
// SAD (TComRdCost.cpp:513-1011), SSE (:1018-1360), HADs (:1641-1708).  One workgroup per job.
// ---------------------------------------------------------------------------------------------
// one job on the 256 threads of the calling workgroup; thread 0 returns the value (barriers inside)
// tile != nullptr: the prediction is read from `tile` (pitch = the block's width) instead of the prediction picture
__device__ static uint32_t distortion_body(unsigned int& acc, const hop_dist_job& jb, const hop_pics& pic, const int16_t* tile = nullptr) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool chroma = jb.comp != 0;
  const int w = chroma ? jb.w >> 1 : jb.w, h = chroma ? jb.h >> 1 : jb.h, x0 = chroma ? jb.x >> 1 : jb.x, y0 = chroma ? jb.y >> 1 : jb.y;
  const int pitch = chroma ? pic.pic_w >> 1 : pic.pic_w, bd = chroma ? pic.bd_c : pic.bd_y;
  const int16_t* o = (jb.comp == 0 ? pic.org_y : jb.comp == 1 ? pic.org_cb : pic.org_cr) + (size_t)y0 * pitch + x0;
  const int ppitch = tile ? w : pitch;
  const int16_t* p = tile ? tile : (jb.comp == 0 ? pic.pred_y : jb.comp == 1 ? pic.pred_cb : pic.pred_cr) + (size_t)y0 * pitch + x0;
  if (tid == 0) acc = 0;
  __syncthreads();
  unsigned int part = 0;
  if (jb.kind == HOP_DIST_SAD || jb.kind == HOP_DIST_SSE) {
    const unsigned sshift = (unsigned)((bd - 8) << 1);
    for (int i = tid; i < w * h; i += 256) {
      int r = i / w, c = i - r * w;
      int d = (int)o[(size_t)r * pitch + c] - (int)p[(size_t)r * ppitch + c];
      part += jb.kind == HOP_DIST_SAD ? (unsigned)(d < 0 ? -d : d) : ((unsigned)(d * d) >> sshift);
    }
    part = (unsigned)hopd_wave_sum((int)part);
    if (lane == 0) atomicAdd(&acc, part);
  } else if (((w & 7) == 0) && ((h & 7) == 0)) {
    const int nblk = (w * h) >> 6, bwb = w >> 3;
    for (int blk = wave; blk < nblk; blk += 4) {
      const int px = (blk % bwb) * 8 + (lane & 7), py = (blk / bwb) * 8 + (lane >> 3);
      int d = (int)o[(size_t)py * pitch + px] - (int)p[(size_t)py * ppitch + px];
      int s = hopd_satd8x8_wave(d, lane);
      if (lane == 0) atomicAdd(&acc, (unsigned)s);
    }
  } else {
    const int nb4 = (w >> 2) * (h >> 2), bw4 = w >> 2;
    for (int b0 = wave * 4; b0 < nb4; b0 += 16) {
      const int blk = b0 + (lane >> 4);
      const bool act = blk < nb4;
      const int bb = act ? blk : 0;
      const int px = (bb % bw4) * 4 + (lane & 3), py = (bb / bw4) * 4 + ((lane >> 2) & 3);
      int d = (int)o[(size_t)py * pitch + px] - (int)p[(size_t)py * ppitch + px];
      int sb = hopd_satd4x4_quad(act ? d : 0, lane);
      int s = hopd_wave_sum((act && (lane & 15) == 0) ? sb : 0);
      if (lane == 0) atomicAdd(&acc, (unsigned)s);
    }
  }
  __syncthreads();
  return (jb.kind == HOP_DIST_SSE) ? acc : (acc >> (bd - 8));
}
__global__ __launch_bounds__(256) void k_distortion(const hop_dist_job* __restrict__ jobs, hop_pics pic, uint32_t* __restrict__ out) {
  __shared__ unsigned int acc;
  const hop_dist_job jb = jobs[blockIdx.x];
  const uint32_t v = distortion_body(acc, jb, pic);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// Sequences of candidates rated "one after the other" (TEncSearch::xMergeEstimation TLibEncoder/TEncSearch.cpp:2992-3106, xGetTemplateCost :4411-4477 with xGetInterPredictionError
// :2951-2977: every candidate of a PU is predicted into the same temporary block and its luma distortion taken): sequence s = jobs first[s] .. first[s + 1] - 1, all on the
// PU's rectangle.  The candidates do not depend on each other -- they only share the block they are predicted into --, so every candidate gets a workgroup of its own:
// grid (jobs, 3).  Plane 0: the luma prediction into an LDS tile, its distortion against the original -> out[job]; the LAST candidate of a sequence also writes its tile to
// the prediction picture.  Planes 1 / 2: the chroma prediction of the last candidate of each sequence into the picture (the other workgroups return at once) -- afterwards
// the prediction picture holds the last candidate's prediction in all planes, as after the reference's loop.
__global__ __launch_bounds__(256) void k_pred_cost(const int32_t* __restrict__ seq_of, const int32_t* __restrict__ first, const hop_pred_job* __restrict__ jobs,
                                                   const int32_t* __restrict__ kinds, hop_pics pic, uint32_t* __restrict__ out) {
  __shared__ PredShared sh;
  __shared__ int16_t tile[64 * 64];
  __shared__ unsigned int acc;
  const int j = blockIdx.x, comp = blockIdx.y, s = seq_of[j];
  const bool last = j == first[s + 1] - 1;
  const hop_pred_job jb = jobs[j];
  if (comp) { if (last) pred_inter_body(sh, jb, comp, pic); return; }
  pred_inter_body(sh, jb, 0, pic, tile);
  __syncthreads();
  hop_dist_job d; d.x = jb.pu_x; d.y = jb.pu_y + jb.dst_row_off; d.w = jb.w; d.h = jb.h; d.comp = 0; d.kind = kinds[s];
  const uint32_t v = distortion_body(acc, d, pic, tile);
  if (threadIdx.x == 0) out[j] = v;
  if (last) {
    int16_t* dst = pic.pred_y + (size_t)(jb.pu_y + jb.dst_row_off) * pic.pic_w + jb.pu_x;
    for (int i = threadIdx.x; i < jb.w * jb.h; i += 256) { const int y = i / jb.w, x = i - y * jb.w; dst[(size_t)y * pic.pic_w + x] = tile[i]; }
  }
}
int hop_launch_pred_cost(hop_ctx* c, int total, const int32_t* d_seq_of, const int32_t* d_first, const hop_pred_job* d_jobs, const int32_t* d_kinds, uint32_t* d_out) {
  const int pr = hop_prof_begin(c, HOP_K_PRED, (uint64_t)total);
  hipLaunchKernelGGL(k_pred_cost, dim3(total, 3), dim3(256), 0, c->stream, d_seq_of, d_first, d_jobs, d_kinds, hop_make_pics(c), d_out);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "pred_cost launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

int hop_launch_dist(hop_ctx* c, int n, const hop_dist_job* d_jobs, uint32_t* d_out) {
  const int pr = hop_prof_begin(c, HOP_K_DIST, (uint64_t)n);
  hipLaunchKernelGGL(k_distortion, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), d_out);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "distortion launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
