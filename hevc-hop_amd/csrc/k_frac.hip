// k_frac.hip -- half- then quarter-pel refinement of the SS vector.
// Replaces TEncSearch::xPatternSearchFracDIF (TLibEncoder/TEncSearch.cpp:6564-6610),
// xExtDIFUpSamplingH/Q (:7818-8011: the 16 phase planes m_filteredBlock[4][4]), xPatternRefinement
// (:709-761) and the 8-tap DCT-IF of TComInterpolationFilter (TLibCommon/TComInterpolationFilter.cpp:55-61,
// filterCopy :92-152, filter<> :170-245).  Every phase plane of the reference is "horizontal filter into a
// 14-bit intermediate, then vertical filter with final rounding and clip"; the kernel evaluates exactly that
// for the 9 + 9 candidate phases out of an LDS copy of the (W+8)x(H+8) integer window.
//
// One workgroup per PU.  Per stage: 3 horizontally filtered planes (the three x phases of the stage) are
// built by all 256 threads, then (candidate, 8x8 block) units are dealt round-robin to the 4 waves:
// lane = one sample, vertical 8-tap from LDS, Hadamard across the wave, integer atomicAdd per candidate.
#include "hop_dev.h"

__constant__ int16_t c_luma_taps[4][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int8_t c_refine_h[9][2] = { {0,0},{0,-1},{0,1},{-1,0},{1,0},{-1,-1},{1,-1},{-1,1},{1,1} };   // TEncSearch.cpp:46-57
__constant__ int8_t c_refine_q[9][2] = { {0,0},{0,-1},{0,1},{-1,-1},{1,-1},{-1,0},{1,0},{-1,1},{1,1} };   // TEncSearch.cpp:59-70

// MAXD = largest PU side of the class: 16 (one wave per PU, no barriers: such a PU is a few hundred samples and the
// fixed latencies of a 256-thread workgroup -- five barriers, 47 KB of LDS, 3 workgroups per CU -- dominated) or 64
template <int MAXD>
struct FracShared {
  static constexpr int TP = MAXD + 10;            // integer window pitch  (W+8 <= MAXD+8)
  static constexpr int PP = MAXD + 2;             // phase plane pitch
  int16_t win[(MAXD + 8) * TP];                   // rows -4..H+3, cols -4..W+3 of the reference at the integer vector
  int16_t plane[3][(MAXD + 8) * PP];              // horizontal intermediates (14-bit), rows -4..H+3
  int16_t org[MAXD * MAXD];
  int cand[9];
};

template <int NW>
__device__ static inline void fr_sync() {
  if (NW > 1) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // one wave: program order + LDS counters suffice
}

template <int NW, int MAXD>
__global__ __launch_bounds__(NW * 64) void k_frac(const hop_pu_job* __restrict__ jobs, hop_pics pic, hop_pu_result* __restrict__ res,
                                                  const int32_t* __restrict__ index, const unsigned int* __restrict__ count) {
  typedef FracShared<MAXD> SH;
  constexpr int FR_TP = SH::TP, FR_PP = SH::PP, NT = NW * 64;
  __shared__ SH sh;
  if (blockIdx.x >= *count) return;
  const int pu = index[blockIdx.x];
  const hop_pu_job jb = jobs[pu];
  // only the fields the stage reads are loaded and only the ones it changes are stored: a local copy of the 100-byte result
  // lived on the scratch stack (40 B private segment per lane, 2.2 KB of HBM writes per PU, profiles/r01_traffic.json)
  hop_pu_result* __restrict__ rp = res + pu;
  if (rp->not_valid) return;
  const int W = jb.w, H = jb.h, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bd = pic.bd_y, headRoom = 14 - bd, maxVal = (1 << bd) - 1;
  const bool use_had = (jb.flags & HOP_FLAG_HADME) != 0;
  const bool had8 = ((W & 7) == 0) && ((H & 7) == 0);
  const int mvx = rp->mv_int[0], mvy = rp->mv_int[1];
  {
    const int16_t* src = pic.ss_y + (ptrdiff_t)(jb.pu_y + mvy - 4) * pic.stride_y + (jb.pu_x + mvx - 4);
    for (int i = tid; i < (H + 8) * (W + 8); i += NT) {
      int r = i / (W + 8), c = i - r * (W + 8);
      sh.win[r * FR_TP + c] = src[(ptrdiff_t)r * pic.stride_y + c];
    }
    for (int i = tid; i < W * H; i += NT) {
      int r = i / W, c = i - r * W;
      sh.org[i] = pic.org_y[(size_t)(jb.pu_y + r) * pic.pic_w + jb.pu_x + c];
    }
  }
  int half0 = 0, half1 = 0, qter0 = 0, qter1 = 0;
  uint32_t cost_best = 0xFFFFFFFFu;
  for (int stage = 0; stage < 2; stage++) {
    // stage 0: half-pel, candidates (2*hx, 2*hy); stage 1: quarter-pel around the half-pel winner
    const int step = stage == 0 ? 2 : 1;
    const int fxc = stage == 0 ? 0 : 2 * half0, fyc = stage == 0 ? 0 : 2 * half1;
    fr_sync<NW>();
    if (tid < 9) sh.cand[tid] = 0;
    // ---- three horizontal phase planes: fx = fxc + (p-1)*step ----
    for (int i = tid; i < 3 * (H + 8) * W; i += NT) {
      const int p = i / ((H + 8) * W), rem = i - p * (H + 8) * W;
      const int r = rem / W, c = rem - r * W;
      const int fx = fxc + (p - 1) * step;
      const int xi = fx >> 2, xf = fx & 3;
      const int16_t* q = sh.win + r * FR_TP + (c + xi + 4);          // integer sample at column c+xi
      int16_t out;
      if (xf == 0) {                                                  // filterCopy, isFirst && !isLast (:112-127)
        int16_t val = (int16_t)(q[0] << headRoom);
        out = (int16_t)(val - (int16_t)8192);
      } else {                                                        // filter<8,false,true,false>
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += q[k - 3] * c_luma_taps[xf][k];
        const int shift = 6 - headRoom;
        out = (int16_t)((sum + (-8192 * (1 << shift))) >> shift);
      }
      sh.plane[p][r * FR_PP + c] = out;
    }
    fr_sync<NW>();
    // ---- (candidate, block) units ----
    auto sample = [&](int ci, int px, int py) -> int {
      const int8_t* rf = stage == 0 ? c_refine_h[ci] : c_refine_q[ci];
      const int p = rf[0] + 1;
      const int fy = fyc + rf[1] * step;
      const int yi = fy >> 2, yf = fy & 3;
      const int16_t* t = sh.plane[p] + (py + yi + 4) * FR_PP + px;    // intermediate at row py+yi
      int16_t val;
      if (yf == 0) {                                                  // filterCopy, !isFirst && isLast (:130-150)
        const int shift = headRoom;
        const int16_t offset = (int16_t)(8192 + (shift ? (1 << (shift - 1)) : 0));
        val = (int16_t)((t[0] + offset) >> shift);
      } else {                                                        // filter<8,true,false,true>
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += t[(k - 3) * FR_PP] * c_luma_taps[yf][k];
        const int shift = 6 + headRoom;
        val = (int16_t)((sum + (1 << (shift - 1)) + (8192 << 6)) >> shift);
      }
      if (val < 0) val = 0;
      if (val > maxVal) val = (int16_t)maxVal;
      return val;
    };
    if (!use_had) {
      const int per = (W * H + 63) >> 6;                              // 64-sample groups per candidate
      for (int u = wave; u < 9 * per; u += NW) {
        const int ci = u / per, g = u - ci * per;
        const int i = g * 64 + lane;
        const bool act = i < W * H;
        const int py = act ? i / W : 0, px = act ? i - py * W : 0;
        int d = (int)sh.org[py * W + px] - sample(ci, px, py);
        int s = hopd_wave_sum(act ? (d < 0 ? -d : d) : 0);
        if (lane == 0) atomicAdd(&sh.cand[ci], s);
      }
    } else if (had8) {
      const int per = (W * H) >> 6, bw = W >> 3;
      for (int u = wave; u < 9 * per; u += NW) {
        const int ci = u / per, blk = u - ci * per;
        const int px = (blk % bw) * 8 + (lane & 7), py = (blk / bw) * 8 + (lane >> 3);
        int d = (int)sh.org[py * W + px] - sample(ci, px, py);
        int s = hopd_satd8x8_wave(d, lane);
        if (lane == 0) atomicAdd(&sh.cand[ci], s);
      }
    } else {
      const int nb4 = (W >> 2) * (H >> 2), bw4 = W >> 2, per = (nb4 + 3) >> 2;
      for (int u = wave; u < 9 * per; u += NW) {
        const int ci = u / per, g = u - ci * per;
        const int blk = g * 4 + (lane >> 4);
        const bool act = blk < nb4;
        const int bb = act ? blk : 0;
        const int px = (bb % bw4) * 4 + (lane & 3), py = (bb / bw4) * 4 + ((lane >> 2) & 3);
        int d = (int)sh.org[py * W + px] - sample(ci, px, py);
        int sb = hopd_satd4x4_quad(act ? d : 0, lane);
        int s = hopd_wave_sum((act && (lane & 15) == 0) ? sb : 0);
        if (lane == 0) atomicAdd(&sh.cand[ci], s);
      }
    }
    fr_sync<NW>();
    // ---- first-best over the 9 candidates in table order (:723-756), all threads redundantly ----
    uint32_t best = 0xFFFFFFFFu; int bi = 0;
    for (int ci = 0; ci < 9; ci++) {
      const int8_t* rf = stage == 0 ? c_refine_h[ci] : c_refine_q[ci];
      uint32_t d = (uint32_t)sh.cand[ci] >> (bd - 8);
      if (stage == 0) d += hopd_mv_cost(jb.lambda_cost, rf[0] + (mvx << 1), rf[1] + (mvy << 1), 1, jb.pred_x, jb.pred_y);        // cost scale 1, :4615
      else d += hopd_mv_cost(jb.lambda_cost, rf[0] + (((mvx << 1) + half0) << 1), rf[1] + (((mvy << 1) + half1) << 1), 0, jb.pred_x, jb.pred_y);   // :6599-6608
      if (d < best) { best = d; bi = ci; }
    }
    if (stage == 0) { half0 = c_refine_h[bi][0]; half1 = c_refine_h[bi][1]; }
    else { qter0 = c_refine_q[bi][0]; qter1 = c_refine_q[bi][1]; }
    cost_best = best;
  }
  if (tid == 0) {
    rp->half[0] = half0; rp->half[1] = half1; rp->qter[0] = qter0; rp->qter[1] = qter1;
    rp->frac_cost = cost_best; rp->cost = cost_best;
    rp->mv_final[0] = mvx; rp->mv_final[1] = mvy;
    rp->half_final[0] = half0; rp->half_final[1] = half1; rp->qter_final[0] = qter0; rp->qter_final[1] = qter1;
  }
}

int hop_launch_frac(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res) {
  // scratch (after the SS search's use of it on the same stream): 2 counters + 2 index lists, the layout of hop_launch_size_classes
  void* sc; int r = hop_scratch(c, 256 + (size_t)n * 8, &sc); if (r) return r;
  unsigned int* counts = (unsigned int*)sc;
  int32_t* small_list = (int32_t*)((char*)sc + 256);
  int32_t* big_list = small_list + n;
  const int pr = hop_prof_begin(c, HOP_K_FRAC, (uint64_t)n);
  hop_launch_size_classes(c, n, d_jobs, d_res, sc);
  // grids are upper bounds: blocks beyond the class count exit on their first instruction
  hipLaunchKernelGGL((k_frac<4, 64>), dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), d_res, big_list, counts + 1);
  hipLaunchKernelGGL((k_frac<1, 16>), dim3(n), dim3(64), 0, c->stream, d_jobs, hop_make_pics(c), d_res, small_list, counts);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "frac launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
