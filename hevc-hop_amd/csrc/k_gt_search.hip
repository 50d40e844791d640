// k_gt_search.hip -- GT / HOP 4-corner diamond search.
// Replaces TEncSearch::xPatternSearchGT, active branch IT_GT_SEARCH == 2
// (TLibEncoder/TEncSearch.cpp:4686-4790 set-up, :5093-5467 search) with
// TComPrediction::calcParamProjective (TLibCommon/TComPrediction.cpp:807-832),
// ProjectiveTransform GRID_SIZE-2 / bilinear branch (:904-1030) and xGetHADs / SAD cost.
//
// Mapping to CDNA4
//   * one workgroup (4 waves) per PU; the 2Wx2H search patch (the reference's m_filteredBlock[0][0]:
//     the SS reference around the start vector clamped to [0,maxVal]) and the original block live in LDS.
//   * the <=625 corner combinations of one iteration are enumerated in the reference's visit order by
//     the threads, filtered with the exact integer form of the reference's double test
//     "h[2]==0 && h[5]==0" and compacted in order (56 survive when the centres form a parallelogram).
//   * a wave evaluates one candidate at a time: lane = one sample of an 8x8 block (or of one of four
//     4x4 blocks), the warp is IEEE double in the reference's operation order (compiled with
//     -ffp-contract=off), the Hadamard runs across the 64 lanes, the block SATDs accumulate in a register.
//   * first-best in visit order = min over cost<<16|order, strict '<' against the incumbent.
// FP64 VALU bound (about 35 double ops per warped sample); no HBM traffic beyond the patch loads.
#include "hop_dev.h"

#define GT_MAXC 640

struct GtShared {
  int16_t patch[128 * 130];      // 2H rows x (2W + 2) pitch
  int16_t org[64 * 64];
  uint32_t cand_cost[GT_MAXC];
  uint16_t cand_list[GT_MAXC];
  uint8_t  flag[GT_MAXC];
  int n_cand;
  unsigned long long best;
};

// offsets of one corner in the reference's visit order (y outer +s,0,-s; x inner +s,0,-s; diamond mask)
// index: 0:(0,+s) 1:(+s,0) 2:(0,0) 3:(-s,0) 4:(0,-s)      (dx,dy)
__device__ static inline void corner_off(int idx, int s, int& dx, int& dy) {
  dx = (idx == 1) ? s : (idx == 3) ? -s : 0;
  dy = (idx == 0) ? s : (idx == 4) ? -s : 0;
}

__global__ __launch_bounds__(256) void k_gt_search(const hop_pu_job* __restrict__ jobs, hop_pics pic, hop_pu_result* __restrict__ res) {
  __shared__ GtShared sh;
  const hop_pu_job jb = jobs[blockIdx.x];
  hop_pu_result rr = res[blockIdx.x];
  if (rr.not_valid) return;
  const int W = jb.w, H = jb.h, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int maxVal = (1 << pic.bd_y) - 1;
  const bool use_had = (jb.flags & HOP_FLAG_HADME) != 0;
  const int PP = 2 * W + 2;                                   // patch pitch
  // original block -> LDS
  for (int i = tid; i < W * H; i += 256) {
    int r = i / W, c = i - r * W;
    sh.org[i] = pic.org_y[(size_t)(jb.pu_y + r) * pic.pic_w + jb.pu_x + c];
  }
  const int nssWindow = (min(H, W) >> 1) * 2;                 // :4756-4759
  int lastStep = nssWindow >> 6; if (lastStep == 0) lastStep = 1;   // :4763-4765 (IT_MAX_NSS_Iteration 6)
  const int m = nssWindow / 2;
  uint32_t distBest = rr.frac_cost;                           // incumbent = cost after the fractional search, :4769
  int bestCX[4] = {0, 0, 0, 0}, bestCY[4] = {0, 0, 0, 0};
  int bestNX[4], bestNY[4], curNX[4], curNY[4];
  int bestSSX = 0, bestSSY = 0;
  const int rx[4] = {0, 2 * W - 1, 2 * W - 1, 0}, ry[4] = {0, 0, 2 * H - 1, 2 * H - 1};   // rest corners, :4786-4789
  for (int k = 0; k < 4; k++) { bestNX[k] = curNX[k] = rx[k]; bestNY[k] = curNY[k] = ry[k]; }
  const bool had8 = ((W & 7) == 0) && ((H & 7) == 0);
  const bool had4 = !had8 && ((W & 3) == 0) && ((H & 3) == 0);

  for (int b = 0; b < 1 + jb.n_amvp; b++) {                   // start vectors, :5106-5178
    int sx, sy;
    if (b == 0) { sx = rr.mv_int[0]; sy = rr.mv_int[1]; if (sx == 0 && sy == 0) continue; }   // ssBestCand[0] == best integer MV
    else {
      int ax = jb.amvp[2 * (b - 1)], ay = jb.amvp[2 * (b - 1) + 1];
      if (ax == 0 && ay == 0) continue;
      sx = (int)(int16_t)ax >> 2; sy = (int)(int16_t)ay >> 2;
    }
    const int Hor = (int)(int16_t)(sx * 4), Ver = (int)(int16_t)(sy * 4);
    __syncthreads();                                          // previous start's readers are done with the patch
    // patch: rows -H/2 .. 3H/2-1, cols -W/2 .. 3W/2-1 around the displaced PU, clamped to [0,maxVal]
    // (filterCopy twice, TComInterpolationFilter.cpp:92-152 via TEncSearch.cpp:5161-5165,:7832,:7837)
    {
      const int16_t* src = pic.ss_y + (ptrdiff_t)(jb.pu_y + sy - H / 2) * pic.stride_y + (jb.pu_x + sx - W / 2);
      for (int i = tid; i < 4 * W * H; i += 256) {
        int r = i / (2 * W), c = i - r * (2 * W);
        int v = src[(ptrdiff_t)r * pic.stride_y + c];
        sh.patch[r * PP + c] = (int16_t)min(maxVal, max(0, v));
      }
    }
    __syncthreads();
    const int16_t* centre = sh.patch + (H / 2) * PP + W / 2;
    const uint32_t mvc = hopd_mv_cost(jb.lambda_cost, Hor, Ver, 0, jb.pred_x, jb.pred_y);    // :5345, cost scale 0
    int iter = 1;
    for (int j0 = nssWindow; (j0 > 1) && (iter <= 6); j0 /= 2) {    // :5181
      iter++;
      if (j0 == nssWindow) { for (int k = 0; k < 4; k++) { curNX[k] = bestNX[k] = rx[k]; curNY[k] = bestNY[k] = ry[k]; } }
      else { for (int k = 0; k < 4; k++) { curNX[k] = bestNX[k]; curNY[k] = bestNY[k]; } }
      const int s = j0 / 2;
      // ---- enumerate + filter the 625 combinations (visit order = index) ----
      if (tid == 0) { sh.n_cand = 0; sh.best = ~0ull; }
      for (int idx = tid; idx < 625; idx += 256) {
        int i3 = idx % 5, i2 = (idx / 5) % 5, i1 = (idx / 25) % 5, i0 = idx / 125;
        int dx[4], dy[4];
        corner_off(i0, s, dx[0], dy[0]); corner_off(i1, s, dx[1], dy[1]); corner_off(i2, s, dx[2], dy[2]); corner_off(i3, s, dx[3], dy[3]);
        bool ok = !(i0 == i1 && i0 == i2 && i0 == i3);                     // not a pure translation, :5289
        // affine test :5323 on calcParamProjective's h[2], h[5]: numerators and denominator are products of
        // small integers (exact in double); h==0.0 <=> numerator == 0 and denominator != 0 (0/0 = NaN, x/0 = inf)
        int x0 = curNX[0] + dx[0], x1 = curNX[1] + dx[1], x2 = curNX[2] + dx[2], x3 = curNX[3] + dx[3];
        int y0 = curNY[0] + dy[0], y1 = curNY[1] + dy[1], y2 = curNY[2] + dy[2], y3 = curNY[3] + dy[3];
        int ddx1 = x1 - x2, ddx2 = x3 - x2, ddx3 = x0 - x1 + x2 - x3;
        int ddy1 = y1 - y2, ddy2 = y3 - y2, ddy3 = y0 - y1 + y2 - y3;
        int num2 = ddx3 * ddy2 - ddx2 * ddy3, num5 = ddx1 * ddy3 - ddx3 * ddy1, den = ddx1 * ddy2 - ddx2 * ddy1;
        ok = ok && (num2 == 0) && (num5 == 0) && (den != 0);
        sh.flag[idx] = ok ? 1 : 0;
      }
      __syncthreads();
      if (wave == 0) {                                        // ordered compaction by one wave
        int base = 0;
        for (int c0 = 0; c0 < 625; c0 += 64) {
          int idx = c0 + lane;
          bool f = idx < 625 && sh.flag[idx];
          unsigned long long mask = __ballot(f);
          int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
          if (f) sh.cand_list[pos] = (uint16_t)idx;
          base += __popcll(mask);
        }
        if (lane == 0) sh.n_cand = base;
      }
      __syncthreads();
      const int ncand = sh.n_cand;
      // ---- evaluate: one candidate per wave at a time ----
      for (int ci = wave; ci < ncand; ci += 4) {
        const int idx = sh.cand_list[ci];
        int i3 = idx % 5, i2 = (idx / 5) % 5, i1 = (idx / 25) % 5, i0 = idx / 125;
        int cx[4], cy[4], ddx, ddy;
        corner_off(i0, s, ddx, ddy); cx[0] = curNX[0] + ddx; cy[0] = curNY[0] + ddy;
        corner_off(i1, s, ddx, ddy); cx[1] = curNX[1] + ddx; cy[1] = curNY[1] + ddy;
        corner_off(i2, s, ddx, ddy); cx[2] = curNX[2] + ddx; cy[2] = curNY[2] + ddy;
        corner_off(i3, s, ddx, ddy); cx[3] = curNX[3] + ddx; cy[3] = curNY[3] + ddy;
        // calcParamProjective on the doubled grid, TComPrediction.cpp:807-832.  For the candidates that reach
        // this point h[2] = h[5] = +/-0, so h[2]*x adds +/-0 and the denominator of the warp is exactly 1.0.
        const double Wd = (double)(2 * W) - 1.0, Hd = (double)(2 * H) - 1.0;
        const double h0 = (double)(cx[1] - cx[0]) / Wd, h3 = (double)(cx[3] - cx[0]) / Hd, h6 = (double)cx[0];
        const double h1 = (double)(cy[1] - cy[0]) / Wd, h4 = (double)(cy[3] - cy[0]) / Hd, h7 = (double)cy[0];
        const int offX = W / 2, offY = H / 2;                 // offsetX/Y of the doubled grid, :919-920
        // one warped sample of the PU at (px,py): ProjectiveTransform, TComPrediction.cpp:919-1025
        auto warp = [&](int px, int py) -> int {
          const int gx = px + offX, gy = py + offY;
          double Fx = (h0 * gx + h3 * gy + h6);               // divided by exactly 1.0 in the reference
          double Fy = (h1 * gx + h4 * gy + h7);
          int Y = (int)Fy - offY, X = (int)Fx - offX;
          double q = (Fy - offY - (double)Y), p = (Fx - offX - (double)X);
          if (Y < -m) Y = -m;
          if (X < -m) X = -m;
          if (Y > m + H - 1) Y = m + H - 1;
          if (X > m + W - 1) X = m + W - 1;
          if (Y + 1 > m + H - 1) Y = m + H - 2;
          if (X + 1 > m + W - 1) X = m + W - 2;
          const int16_t* pa = centre + Y * PP + X;
          double v = (1.0 - q) * ((1.0 - p) * (double)pa[0] + p * (double)pa[1]);
          v += q * ((1.0 - p) * (double)pa[PP] + p * (double)pa[PP + 1]);
          if (v > 255) v = 255;                               // hard-coded 8-bit clip, :969-972
          if (v < 0) v = 0;
          return (int)(int16_t)(v + 0.5);
        };
        int satd = 0;
        if (!use_had) {                                       // SAD (HadamardME = 0): 64 samples per pass
          for (int base = 0; base < W * H; base += 64) {
            const int i = base + lane;
            const bool act = i < W * H;
            const int py = act ? i / W : 0, px = act ? i - py * W : 0;
            int d = (int)sh.org[py * W + px] - warp(px, py);
            satd += hopd_wave_sum(act ? (d < 0 ? -d : d) : 0);
          }
        } else if (had8) {                                    // 64 lanes = one 8x8 block
          const int nblk = (W * H) >> 6, bw = W >> 3;
          for (int blk = 0; blk < nblk; blk++) {
            const int px = (blk % bw) * 8 + (lane & 7), py = (blk / bw) * 8 + (lane >> 3);
            int d = (int)sh.org[py * W + px] - warp(px, py);
            satd += hopd_satd8x8_wave(d, lane);
          }
        } else if (had4) {                                    // four 4x4 blocks per pass: lane = 16*q + 4*row + col
          const int nb4 = (W >> 2) * (H >> 2), bw4 = W >> 2;
          for (int b0 = 0; b0 < nb4; b0 += 4) {
            const int blk = b0 + (lane >> 4);
            const bool act = blk < nb4;
            const int bb = act ? blk : 0;
            const int px = (bb % bw4) * 4 + (lane & 3), py = (bb / bw4) * 4 + ((lane >> 2) & 3);
            int d = (int)sh.org[py * W + px] - warp(px, py);
            int sb = hopd_satd4x4_quad(act ? d : 0, lane);    // SATD of this lane's block
            satd += hopd_wave_sum((act && (lane & 15) == 0) ? sb : 0);
          }
        }
        if (lane == 0) {
          uint32_t dist = (uint32_t)satd >> (pic.bd_y - 8);
          dist += mvc;
          int v[6] = { cx[0] / lastStep, cy[0] / lastStep, (cx[1] - 2 * W + 1) / lastStep, cy[1] / lastStep,
                       (cx[2] - 2 * W + 1) / lastStep, (cy[2] - 2 * H + 1) / lastStep };
          uint32_t bits = 0;
          for (int k = 0; k < 6; k++) bits += hopd_component_bits(v[k]);       // getBitsGT: corners 0..2 (affine)
          dist += (jb.lambda_cost * bits) >> 16;                               // :5346-5358
          sh.cand_cost[ci] = dist;
        }
      }
      __syncthreads();
      // ---- first-best in visit order, strict '<' against the incumbent (:5361) ----
      unsigned long long kbest = ~0ull;
      for (int ci = tid; ci < ncand; ci += 256) {
        unsigned long long key = ((unsigned long long)sh.cand_cost[ci] << 16) | (unsigned long long)ci;
        kbest = key < kbest ? key : kbest;
      }
      kbest = hopd_wave_min_u64(kbest);
      if (lane == 0 && kbest != ~0ull) atomicMin(&sh.best, kbest);
      __syncthreads();
      const unsigned long long kb = sh.best;
      if (kb != ~0ull && (uint32_t)(kb >> 16) < distBest) {
        distBest = (uint32_t)(kb >> 16);
        const int idx = sh.cand_list[(int)(kb & 0xFFFF)];
        int i3 = idx % 5, i2 = (idx / 5) % 5, i1 = (idx / 25) % 5, i0 = idx / 125, ddx, ddy;
        corner_off(i0, s, ddx, ddy); bestCX[0] = curNX[0] + ddx; bestCY[0] = curNY[0] + ddy;
        corner_off(i1, s, ddx, ddy); bestCX[1] = curNX[1] + ddx; bestCY[1] = curNY[1] + ddy;
        corner_off(i2, s, ddx, ddy); bestCX[2] = curNX[2] + ddx; bestCY[2] = curNY[2] + ddy;
        corner_off(i3, s, ddx, ddy); bestCX[3] = curNX[3] + ddx; bestCY[3] = curNY[3] + ddy;
        for (int k = 0; k < 4; k++) { bestNX[k] = bestCX[k]; bestNY[k] = bestCY[k]; }
        bestSSX = Hor; bestSSY = Ver;
      }
      __syncthreads();                                        // everyone has read sh.best before it is reset
    }
  }
  if (tid == 0) {
    bool flag = false;
    for (int k = 0; k < 4; k++) flag = flag || bestCX[k] != 0 || bestCY[k] != 0;      // :5436-5439
    if (flag) {
      rr.gt_flag = 1;
      rr.gt[0] = bestCX[0] / lastStep;               rr.gt[1] = bestCY[0] / lastStep;
      rr.gt[2] = (bestCX[1] - 2 * W + 1) / lastStep; rr.gt[3] = bestCY[1] / lastStep;
      rr.gt[4] = (bestCX[2] - 2 * W + 1) / lastStep; rr.gt[5] = (bestCY[2] - 2 * H + 1) / lastStep;
      rr.gt[6] = bestCX[3] / lastStep;               rr.gt[7] = (bestCY[3] - 2 * H + 1) / lastStep;
      rr.cost = distBest;
      rr.mv_final[0] = bestSSX >> 2; rr.mv_final[1] = bestSSY >> 2;                    // :5455-5457
      rr.half_final[0] = rr.half_final[1] = 0; rr.qter_final[0] = rr.qter_final[1] = 0;
    } else {
      rr.gt_flag = 0;
      for (int k = 0; k < 8; k++) rr.gt[k] = 0;
      rr.cost = rr.frac_cost;
      rr.mv_final[0] = rr.mv_int[0]; rr.mv_final[1] = rr.mv_int[1];
      rr.half_final[0] = rr.half[0]; rr.half_final[1] = rr.half[1];
      rr.qter_final[0] = rr.qter[0]; rr.qter_final[1] = rr.qter[1];
    }
    res[blockIdx.x] = rr;
  }
}

int hop_launch_gt(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res) {
  const int pr = hop_prof_begin(c, HOP_K_GT_SEARCH, (uint64_t)n);
  hipLaunchKernelGGL(k_gt_search, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), d_res);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "gt_search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
