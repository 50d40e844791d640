// k_gt_search.hip -- GT / HOP 4-corner diamond search.
// Replaces TEncSearch::xPatternSearchGT, active branch IT_GT_SEARCH == 2
// (TLibEncoder/TEncSearch.cpp:4686-4790 set-up, :5093-5467 search) with
// TComPrediction::calcParamProjective (TLibCommon/TComPrediction.cpp:807-832),
// ProjectiveTransform GRID_SIZE-2 / bilinear branch (:904-1030) and xGetHADs / SAD cost.
//
// Mapping to CDNA4 (FP64-VALU bound: ~33 double-rate ops per warped sample, no HBM traffic beyond the patch)
//   * PUs of up to 256 samples (16x16 and below) get ONE WAVE each (64-thread workgroups, no barriers at all:
//     the search of such a PU is a chain of 6-18 short dependent iterations, so barrier and dispatch overhead, not
//     arithmetic, decides); larger PUs get a 4-wave workgroup.  A prep kernel sorts the PUs into the two classes.
//   * The 2Wx2H search patch (the reference's m_filteredBlock[0][0]: the SS reference around the start vector
//     clamped to [0,maxVal]) sits in LDS as horizontally PAIRED samples (P[y][x], P[y][x+1]) so one LDS read
//     feeds one bilinear row; the original block sits beside it.
//   * The corner combinations of an iteration are enumerated in the reference's visit order and filtered with
//     the exact integer form of the reference's double test "h[2]==0 && h[5]==0".  While the four centres form
//     a parallelogram (always, because only affine candidates are ever accepted) the surviving set is the same
//     56 combinations every time: it is built once per PU and reused; the general enumeration stays as the
//     fallback.  One thread per surviving candidate does the homography's IEEE divisions once per iteration and
//     parks h0,h3,h6,h1,h4,h7 + the bit cost in LDS.
//   * work item = (candidate, 8x8 block).  8 lanes own one item, a lane owns one 8-sample ROW of the block:
//     the warp (IEEE double in the reference's operation order, -ffp-contract=off) and the horizontal
//     Hadamard butterflies stay in registers, the vertical butterflies are DPP row operations, the block
//     SATD is added to its candidate with one LDS atomic.  A wave therefore carries 8 items (8 candidates of
//     an 8x8 PU at once); PUs with a 4-multiple side use 4x4 blocks (4 lanes x 4 samples, 16 items per wave).
//   * first-best in visit order = min over cost<<16|order, strict '<' against the incumbent.
#include "hop_dev.h"

#define GT_MAXC 640
#define GT_CHUNK 56               // candidates evaluated together (the parallelogram set has exactly 56)

// MAXD = largest PU side of the class: 16 (one wave) or 64 (four waves)
template <typename PT, int MAXD>
struct GtShared {
  PT       patch[2 * MAXD * (2 * MAXD + 2)];   // 2H rows x (2W + 2) pitch; element = P[y][x] | P[y][x+1] << (4*sizeof(PT))
  int16_t  org[MAXD * MAXD];
  double   ch[GT_CHUNK][6];      // h0, h3, h6, h1, h4, h7 of the candidates of the current chunk
  int      cax[GT_CHUNK][2][5];  // GtAxis {A, B, C, aq, ar} of the x and y coordinate of the candidates (exact integer warp)
  uint32_t cfix[GT_CHUNK];       // mv cost + GT bit cost; 0xFFFFFFFF = degenerate (denominator 0), never evaluated by the reference
  int      csatd[GT_CHUNK];
  uint8_t  alive[GT_CHUNK];      // candidates of the chunk still below the incumbent (early termination in rounds of blocks)
  int      n_alive;
  union { uint32_t cand_cost[GT_MAXC]; uint8_t flag[GT_MAXC]; };   // flag: only inside gt_enumerate, before any cost of the iteration is written
  uint16_t cand_list[GT_MAXC];
  uint16_t fixed_list[GT_CHUNK]; // the 56 combinations with d0 + d2 == d1 + d3 (parallelogram centres)
  int n_cand, n_fixed;
  unsigned long long best;
};

// offsets of one corner in the reference's visit order (y outer +s,0,-s; x inner +s,0,-s; diamond mask)
// index: 0:(0,+s) 1:(+s,0) 2:(0,0) 3:(-s,0) 4:(0,-s)      (dx,dy)
__device__ static inline void corner_off(int idx, int s, int& dx, int& dy) {
  dx = (idx == 1) ? s : (idx == 3) ? -s : 0;
  dy = (idx == 0) ? s : (idx == 4) ? -s : 0;
}

// DPP lane exchange (all lanes active): quad_perm [1,0,3,2] = xor 1, [2,3,0,1] = xor 2, row_half_mirror = i <-> 7-i
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
template <int CTRL>
__device__ static inline int dpp_get(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }

// butterfly across lanes: the lane whose sign is -1 keeps partner - self, the other self + partner.
// Any pairing along three GF(2)-independent lane masks (here 7, 2, 1 within 8 lanes) yields the Walsh-Hadamard
// coefficients up to order and sign, which sum|coef| ignores.
template <int CTRL>
__device__ static inline int lane_bfly(int v, int sgn) { return __mul24(v, sgn) + dpp_get<CTRL>(v); }   // |v| < 2^23: full-rate 24-bit multiply instead of v_mul_lo_u32

template <int NW>
__device__ static inline void wg_sync() {
  if (NW > 1) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // one wave: program order + LDS counters suffice; keep the compiler from reordering LDS traffic
}

// ---------------------------------------------------------------------------------------------------------------
// The warp in exact integer arithmetic (ProjectiveTransform, TComPrediction.cpp:904-1030, for affine candidates).
//
// For an affine candidate on the doubled grid (h[2] = h[5] = +/-0, denominator exactly 1.0)
//     Fx = fl(fl(fl(h0*x) + fl(h3*y)) + h6),  h0 = fl((x1-x0)/Wd), h3 = fl((x3-x0)/Hd), h6 = x0,  Wd = 2W-1, Hd = 2H-1
// is the double nearest (to ~1e-13) to the rational  Rx = Nx/D,  Nx = (x1-x0)*Hd*x + (x3-x0)*Wd*y + x0*D,  D = Wd*Hd,
// and likewise Fy ~ Ny/D.  The reference then takes X = (int)Fx - offX, p = Fx - offX - X, the same for Y / q, clamps
// X, Y, and rounds the bilinear blend  v = (1-q)((1-p)a + p b) + q((1-p)c + p d)  with (Pel)(clip(v) + 0.5).
//   * With X from the reference and  rp = Nx - (X+offX)*D  (so rp/D is the exact value p approximates; same for rq),
//     D^2 * v_exact = (D-rq)((D-rp)a + rp b) + rq((D-rp)c + rp d)  is an integer, and because D is odd v_exact is never
//     closer than 1/(2 D^2) >= 1.9e-9 to a rounding boundary k + 0.5.  The reference's double v differs from v_exact by
//     < 2.5e-10 (coordinate error 1.1e-13 x sample range 1023, plus blend rounding), so both round to the same Pel.
//   * X itself: Rx is at least 1/D from an integer unless it IS one, so (int)Fx = trunc(Rx) except possibly when
//     Nx % D == 0, where the reference may land on K-1 with p ~ 1.  Inside the patch (K-1, p=1) and (K, p=0) read
//     the same sample; they differ only where X is clamped.  Truncation differs from floor only for Fx < 0, which
//     also lies in the clamped region.  So: rows that stay inside the clamp range use the pure integer recurrence;
//     rows that touch it (rare after the first iteration) take X from the reference's own double expression and
//     rp = Nx - (X+offX)*D, which covers both effects without case analysis.
// The blend needs 37 bits, so its last step runs in double -- on integers < 2^53, i.e. exactly.
// Cost per warped sample: ~25 integer + 7 double-rate VALU operations instead of 33 double-rate + 20 integer.
// ---------------------------------------------------------------------------------------------------------------
typedef short gt_s2 __attribute__((ext_vector_type(2)));
struct GtAxis { int A, B, C, aq, ar; };      // N(x,y) = A*x + B*y + C ;  A = aq*D + ar, 0 <= ar < D

__host__ __device__ static inline int floordiv_i(int a, int b) { int q = a / b; return (a % b != 0 && (a < 0) != (b < 0)) ? q - 1 : q; }

// one axis of one row of BS samples: P[k] = clamped integer position relative to the patch centre, R[k] = D * fraction
template <int BS>
__device__ static inline void gt_axis(const int* __restrict__ axp, const double* __restrict__ hv, int gx0, int gy, int off, int lo, int hi,
                                      int D, float rcpD, int (&P)[BS], int (&R)[BS]) {
  GtAxis ax; ax.A = axp[0]; ax.B = axp[1]; ax.C = axp[2]; ax.aq = axp[3]; ax.ar = axp[4];
  const int N0 = __mul24(ax.A, gx0) + __mul24(ax.B, gy) + ax.C;      // |N0| < 2^23 (W, H <= 64)
  int q = (int)floorf((float)N0 * rcpD);                             // floor(N0 / D), off by at most one
  int r = N0 - __mul24(q, D);
  if (r < 0) { q--; r += D; } else if (r >= D) { q++; r -= D; }
  q -= off;
  P[0] = q; R[0] = r;
#pragma unroll
  for (int k = 1; k < BS; k++) {
    r += ax.ar;
    const bool ge = r >= D;
    r -= ge ? D : 0;
    q += ax.aq + (ge ? 1 : 0);
    P[k] = q; R[k] = r;
  }
  if (min(P[0], P[BS - 1]) <= lo || max(P[0], P[BS - 1]) > hi) {     // the row touches the clamped region
    const double hA = hv[0], hBy = hv[1] * gy, hC = hv[2];           // reference operation order, :934-947
#pragma unroll
    for (int k = 0; k < BS; k++) {
      const double F = (hA * (gx0 + k) + hBy + hC);
      const int T = (int)F;
      R[k] = N0 + __mul24(k, ax.A) - __mul24(T, D);
      P[k] = min(max(T - off, lo), hi);                              // :950-961: clamp to [-m, m+size-1], pull back by one if the +1 tap would leave
    }
  }
}

// evaluate `nc` candidates (axes in sh.cax, doubles in sh.ch) over all blocks of the PU; adds block costs into sh.csatd
// BS = 8: 8x8 Hadamard (xCalcHADs8x8, TComRdCost.cpp:1481-1575) ; BS = 4: 4x4 (xCalcHADs4x4, :1387-1479);
// HAD = false: plain SAD of the same samples (HadamardME = 0)
// One ROUND: blocks blk0 .. blk0+bpr-1 of the `na` candidates listed in sh.alive.
template <typename SH, typename PT, int NW, int BS, bool HAD>
__device__ static inline void gt_eval(SH& sh, int na, int blk0, int bpr, int W, int H, int m, int PP, const PT* __restrict__ centre, int wave, int lane) {
  constexpr int IPW = 64 / BS;                                // items per wave
  constexpr int HS = 4 * (int)sizeof(PT);                     // bits per sample inside a pair
  constexpr unsigned HM = (1u << HS) - 1u;
  const int bw = W / BS, items = na * bpr;
  const int sub = lane / BS, row = lane % BS;
  const float inv_bpr = 1.0f / (float)bpr, inv_bw = 1.0f / (float)bw;
  const int offX = W / 2, offY = H / 2;                       // offsetX/Y of the doubled grid, :919-920
  const int D = (2 * W - 1) * (2 * H - 1);
  const float rcpD = 1.0f / (float)D;
  const double Dd = (double)D, invD2 = 1.0 / (Dd * Dd);
  const int sgnA = (BS == 8) ? ((row & 4) ? -1 : 1) : ((row & 2) ? -1 : 1);   // stage over the top lane bit (mirror for BS 8)
  const int sgn2 = (row & 2) ? -1 : 1, sgn1 = (row & 1) ? -1 : 1;
  for (int base = wave * IPW; base < items; base += NW * IPW) {
    const int item = base + sub;
    const bool act = item < items;
    const int it = act ? item : 0;
    // small exact quotients without integer division: (i + 0.5) / n in float is off an integer by >= 0.5/64 >> rounding error (i < 4096)
    const int ai = (int)(((float)it + 0.5f) * inv_bpr), blk = blk0 + it - ai * bpr;
    const int cand = sh.alive[ai];
    const int by = (int)(((float)blk + 0.5f) * inv_bw), bx = blk - by * bw;
    const int py = by * BS + row, px0 = bx * BS;
    int X[BS], rp[BS], Y[BS], rq[BS];
    gt_axis<BS>(&sh.cax[cand][0][0], &sh.ch[cand][0], px0 + offX, py + offY, offX, -m, m + W - 2, D, rcpD, X, rp);
    gt_axis<BS>(&sh.cax[cand][1][0], &sh.ch[cand][3], px0 + offX, py + offY, offY, -m, m + H - 2, D, rcpD, Y, rq);
    int d[BS];                                                // HAD on 8-bit content: the predicted Pel; otherwise org - Pel
    constexpr bool PACKED = HAD && HS == 8;                   // 8-bit samples: every Hadamard intermediate fits 16 bits (<= 64 * 255)
    const int16_t* orow = sh.org + py * W + px0;
#pragma unroll
    for (int k = 0; k < BS; k++) {
      const PT* pa = centre + (__mul24(Y[k], PP) + X[k]);
      const unsigned top = pa[0], bot = pa[PP];
      const int omr = D - rp[k];
      const int t = __mul24((int)(top & HM), omr) + __mul24((int)(top >> HS), rp[k]);     // D * ((1-p) a + p b)
      const int u = __mul24((int)(bot & HM), omr) + __mul24((int)(bot >> HS), rp[k]);     // D * ((1-p) c + p d)
      const double nv = __builtin_fma((double)rq[k], (double)(u - t), (double)t * Dd);    // D^2 * v, exact
      int pel = (int)__builtin_fma(nv, invD2, 0.5);                                       // (Pel)(clip(v) + 0.5), :969-975
      pel = min(max(pel, 0), 255);                                                        // the hard-coded 8-bit clip
      d[k] = PACKED ? pel : (int)orow[k] - pel;
    }
    int s = 0;
    if (PACKED) {
      // two samples per register (v_pk_*_i16): vertical butterflies first (a DPP move + one packed multiply-add per
      // register and stage), then the horizontal ones; the last horizontal stage pairs the two halves of a register
      // and is folded into the magnitude sum: |a+b| + |a-b| = 2 max(|a|,|b|)
      constexpr int NR = BS / 2;
      gt_s2 v[NR];
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const gt_s2 o = *(const gt_s2*)(orow + 2 * i);          // orow is 4-byte aligned (px0 and W are multiples of 4)
        gt_s2 pz; pz.x = (short)d[2 * i]; pz.y = (short)d[2 * i + 1];
        v[i] = o - pz;
      }
      const gt_s2 sA = { (short)sgnA, (short)sgnA }, s2v = { (short)sgn2, (short)sgn2 }, s1v = { (short)sgn1, (short)sgn1 };
#pragma unroll
      for (int i = 0; i < NR; i++) {
        if (BS == 8) v[i] = v[i] * sA + __builtin_bit_cast(gt_s2, dpp_get<DPP_HALF_MIRROR>(__builtin_bit_cast(int, v[i])));
        else         v[i] = v[i] * sA + __builtin_bit_cast(gt_s2, dpp_get<DPP_XOR2>(__builtin_bit_cast(int, v[i])));
        if (BS == 8) v[i] = v[i] * s2v + __builtin_bit_cast(gt_s2, dpp_get<DPP_XOR2>(__builtin_bit_cast(int, v[i])));
        v[i] = v[i] * s1v + __builtin_bit_cast(gt_s2, dpp_get<DPP_XOR1>(__builtin_bit_cast(int, v[i])));
      }
#pragma unroll
      for (int len = NR / 2; len >= 1; len >>= 1)              // registers i and i+len hold samples 2*len apart
#pragma unroll
        for (int i = 0; i < NR; i += 2 * len)
#pragma unroll
          for (int j = i; j < i + len; j++) { const gt_s2 a = v[j], b = v[j + len]; v[j] = a + b; v[j + len] = a - b; }
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const gt_s2 a = __builtin_elementwise_abs(v[i]);
        s += 2 * (int)(a.x > a.y ? a.x : a.y);
      }
    } else if (HAD) {
      // horizontal butterflies in registers
#pragma unroll
      for (int len = 1; len < BS; len <<= 1)
#pragma unroll
        for (int i = 0; i < BS; i += 2 * len)
#pragma unroll
          for (int j = i; j < i + len; j++) { int a = d[j], b = d[j + len]; d[j] = a + b; d[j + len] = a - b; }
      // vertical butterflies across the BS lanes of the item
#pragma unroll
      for (int k = 0; k < BS; k++) {
        int v = d[k];
        if (BS == 8) { v = lane_bfly<DPP_HALF_MIRROR>(v, sgnA); v = lane_bfly<DPP_XOR2>(v, sgn2); v = lane_bfly<DPP_XOR1>(v, sgn1); }
        else { v = lane_bfly<DPP_XOR2>(v, sgnA); v = lane_bfly<DPP_XOR1>(v, sgn1); }
        s += v < 0 ? -v : v;
      }
    } else {
#pragma unroll
      for (int k = 0; k < BS; k++) s += d[k] < 0 ? -d[k] : d[k];
    }
    // sum over the item's lanes
    if (BS == 8) s += dpp_get<DPP_HALF_MIRROR>(s);
    s += dpp_get<DPP_XOR2>(s);
    s += dpp_get<DPP_XOR1>(s);
    if (HAD) s = (BS == 8) ? ((s + 2) >> 2) : ((s + 1) >> 1);
    if (act && row == 0) atomicAdd(&sh.csatd[cand], s);
  }
}

// enumerate + filter the 625 combinations for centres (cx,cy) and step s; ordered compaction into `list`.
// affine test :5323 on calcParamProjective's h[2], h[5]: numerators and denominator are products of small integers
// (exact in double); h == 0.0 <=> numerator == 0 and denominator != 0 (0/0 = NaN, x/0 = inf)
template <typename SH, int NW>
__device__ static inline int gt_enumerate(SH& sh, const int (&cx)[4], const int (&cy)[4], int s, uint16_t* list, int tid, int wave, int lane) {
  for (int idx = tid; idx < 625; idx += NW * 64) {
    int i3 = idx % 5, i2 = (idx / 5) % 5, i1 = (idx / 25) % 5, i0 = idx / 125;
    int dx0, dy0, dx1, dy1, dx2, dy2, dx3, dy3;
    corner_off(i0, s, dx0, dy0); corner_off(i1, s, dx1, dy1); corner_off(i2, s, dx2, dy2); corner_off(i3, s, dx3, dy3);
    bool ok = !(i0 == i1 && i0 == i2 && i0 == i3);                     // not a pure translation, :5289
    int x0 = cx[0] + dx0, x1 = cx[1] + dx1, x2 = cx[2] + dx2, x3 = cx[3] + dx3;
    int y0 = cy[0] + dy0, y1 = cy[1] + dy1, y2 = cy[2] + dy2, y3 = cy[3] + dy3;
    int ddx1 = x1 - x2, ddx2 = x3 - x2, ddx3 = x0 - x1 + x2 - x3;
    int ddy1 = y1 - y2, ddy2 = y3 - y2, ddy3 = y0 - y1 + y2 - y3;
    int num2 = ddx3 * ddy2 - ddx2 * ddy3, num5 = ddx1 * ddy3 - ddx3 * ddy1, den = ddx1 * ddy2 - ddx2 * ddy1;
    ok = ok && (num2 == 0) && (num5 == 0) && (den != 0);
    sh.flag[idx] = ok ? 1 : 0;
  }
  wg_sync<NW>();
  if (wave == 0) {                                            // ordered compaction by one wave
    int base = 0;
    for (int c0 = 0; c0 < 625; c0 += 64) {
      int idx = c0 + lane;
      bool f = idx < 625 && sh.flag[idx];
      unsigned long long mask = __ballot(f);
      int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (f) list[pos] = (uint16_t)idx;
      base += __popcll(mask);
    }
    if (lane == 0) sh.n_cand = base;
  }
  wg_sync<NW>();
  return sh.n_cand;
}

// NW = waves per PU (1 or 4); MAXD = largest PU side handled; the PUs of the class come through `index`
template <typename PT, int NW, int MAXD>
__global__ __launch_bounds__(NW * 64) void k_gt_search(const hop_pu_job* __restrict__ jobs, hop_pics pic, hop_pu_result* __restrict__ res,
                                                        const int32_t* __restrict__ index, const unsigned int* __restrict__ count) {
  typedef GtShared<PT, MAXD> SH;
  __shared__ SH sh;
  constexpr int HS = 4 * (int)sizeof(PT);
  constexpr int NT = NW * 64;
  if (blockIdx.x >= *count) return;
  const int pu = index[blockIdx.x];
  const hop_pu_job* jp = jobs + pu;
  hop_pu_result* rp = res + pu;
  const int W = jp->w, H = jp->h, pu_x = jp->pu_x, pu_y = jp->pu_y, n_amvp = jp->n_amvp;
  const int pred_x = jp->pred_x, pred_y = jp->pred_y;
  const uint32_t lambda_cost = jp->lambda_cost;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int maxVal = (1 << pic.bd_y) - 1;
  const bool use_had = (jp->flags & HOP_FLAG_HADME) != 0;
  const int PP = 2 * W + 2;                                   // patch pitch
  const int mv0x = rp->mv_int[0], mv0y = rp->mv_int[1];
  // original block -> LDS
  for (int i = tid; i < W * H; i += NT) {
    int r = i / W, c = i - r * W;
    sh.org[i] = pic.org_y[(size_t)(pu_y + r) * pic.pic_w + pu_x + c];
  }
  const int nssWindow = (min(H, W) >> 1) * 2;                 // :4756-4759
  int lastStep = nssWindow >> 6; if (lastStep == 0) lastStep = 1;   // :4763-4765 (IT_MAX_NSS_Iteration 6)
  const int m = nssWindow / 2;
  uint32_t distBest = rp->frac_cost;                          // incumbent = cost after the fractional search, :4769
  int bestC[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};             // iBestCornerX/Y
  int bN[2][4], cN[2][4];                                     // iBestNSSCenter / iCurrNSSCenter
  int bestSSX = 0, bestSSY = 0;
  const int rx1 = 2 * W - 1, ry2 = 2 * H - 1;                 // rest corners (0,0) (rx1,0) (rx1,ry2) (0,ry2), :4786-4789
  const int restX[4] = {0, rx1, rx1, 0}, restY[4] = {0, 0, ry2, ry2};
#pragma unroll
  for (int k = 0; k < 4; k++) { bN[0][k] = restX[k]; bN[1][k] = restY[k]; }
  const bool had8 = ((W & 7) == 0) && ((H & 7) == 0);
  // the combination set of parallelogram centres (independent of the step): built once from the rest corners
  {
    const int n = gt_enumerate<SH, NW>(sh, restX, restY, 1, sh.fixed_list, tid, wave, lane);
    if (tid == 0) sh.n_fixed = (n <= GT_CHUNK) ? n : -1;
    wg_sync<NW>();
  }

  for (int b = 0; b < 1 + n_amvp; b++) {                      // start vectors, :5106-5178
    int sx, sy;
    if (b == 0) { sx = mv0x; sy = mv0y; if (sx == 0 && sy == 0) continue; }   // ssBestCand[0] == best integer MV
    else {
      int ax = jp->amvp[2 * (b - 1)], ay = jp->amvp[2 * (b - 1) + 1];
      if (ax == 0 && ay == 0) continue;
      sx = (int)(int16_t)ax >> 2; sy = (int)(int16_t)ay >> 2;
    }
    const int Hor = (int)(int16_t)(sx * 4), Ver = (int)(int16_t)(sy * 4);
    wg_sync<NW>();                                            // previous start's readers are done with the patch
    // patch: rows -H/2 .. 3H/2-1, cols -W/2 .. 3W/2-1 around the displaced PU, clamped to [0,maxVal]
    // (filterCopy twice, TComInterpolationFilter.cpp:92-152 via TEncSearch.cpp:5161-5165,:7832,:7837);
    // element (r,c) = sample(r,c) | sample(r,c+1) << HS  (the last column's partner is never used)
    {
      const int16_t* src = pic.ss_y + (ptrdiff_t)(pu_y + sy - H / 2) * pic.stride_y + (pu_x + sx - W / 2);
      for (int i = tid; i < 4 * W * H; i += NT) {
        int r = i / (2 * W), c = i - r * (2 * W);
        int v0 = src[(ptrdiff_t)r * pic.stride_y + c];
        int v1 = (c + 1 < 2 * W) ? src[(ptrdiff_t)r * pic.stride_y + c + 1] : 0;
        v0 = min(maxVal, max(0, v0)); v1 = min(maxVal, max(0, v1));
        sh.patch[r * PP + c] = (PT)((unsigned)v0 | ((unsigned)v1 << HS));
      }
    }
    wg_sync<NW>();
    const PT* centre = sh.patch + (H / 2) * PP + W / 2;
    const uint32_t mvc = hopd_mv_cost(lambda_cost, Hor, Ver, 0, pred_x, pred_y);    // :5345, cost scale 0
    int iter = 1;
    for (int j0 = nssWindow; (j0 > 1) && (iter <= 6); j0 /= 2) {    // :5181
      iter++;
      if (j0 == nssWindow) {
#pragma unroll
        for (int k = 0; k < 4; k++) { bN[0][k] = restX[k]; bN[1][k] = restY[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) { cN[0][k] = bN[0][k]; cN[1][k] = bN[1][k]; }
      const int s = j0 / 2;
      if (tid == 0) sh.best = ~0ull;
      // candidate set: the fixed parallelogram set, or the general enumeration if the centres are not one
      const bool para = (cN[0][0] - cN[0][1] + cN[0][2] - cN[0][3] == 0) && (cN[1][0] - cN[1][1] + cN[1][2] - cN[1][3] == 0) && sh.n_fixed > 0;
      int ncand;
      const uint16_t* clist;
      if (para) { ncand = sh.n_fixed; clist = sh.fixed_list; wg_sync<NW>(); }
      else { ncand = gt_enumerate<SH, NW>(sh, cN[0], cN[1], s, sh.cand_list, tid, wave, lane); clist = sh.cand_list; }
      for (int c0 = 0; c0 < ncand; c0 += GT_CHUNK) {
        const int nc = min(GT_CHUNK, ncand - c0);
        // ---- per-candidate homography + bit cost: one thread per candidate ----
        if (tid < nc) {
          const int idx = clist[c0 + tid];
          int i3 = idx % 5, i2 = (idx / 5) % 5, i1 = (idx / 25) % 5, i0 = idx / 125, ddx, ddy;
          corner_off(i0, s, ddx, ddy); const int x0 = cN[0][0] + ddx, y0 = cN[1][0] + ddy;
          corner_off(i1, s, ddx, ddy); const int x1 = cN[0][1] + ddx, y1 = cN[1][1] + ddy;
          corner_off(i2, s, ddx, ddy); const int x2 = cN[0][2] + ddx, y2 = cN[1][2] + ddy;
          corner_off(i3, s, ddx, ddy); const int x3 = cN[0][3] + ddx, y3 = cN[1][3] + ddy;
          // calcParamProjective on the doubled grid, TComPrediction.cpp:807-832; h[2] = h[5] = +/-0 here, so
          // "+ h[2]*x[1]" adds +/-0 and the value is the plain quotient
          const double Wd = (double)(2 * W) - 1.0, Hd = (double)(2 * H) - 1.0;
          sh.ch[tid][0] = (double)(x1 - x0) / Wd; sh.ch[tid][1] = (double)(x3 - x0) / Hd; sh.ch[tid][2] = (double)x0;
          sh.ch[tid][3] = (double)(y1 - y0) / Wd; sh.ch[tid][4] = (double)(y3 - y0) / Hd; sh.ch[tid][5] = (double)y0;
          {                                                     // the same maps as exact rationals over D = Wd*Hd (gt_axis)
            const int Wi = 2 * W - 1, Hi = 2 * H - 1, Di = Wi * Hi;
            int aq = floordiv_i(x1 - x0, Wi);
            sh.cax[tid][0][0] = (x1 - x0) * Hi; sh.cax[tid][0][1] = (x3 - x0) * Wi; sh.cax[tid][0][2] = x0 * Di;
            sh.cax[tid][0][3] = aq; sh.cax[tid][0][4] = (x1 - x0) * Hi - aq * Di;
            aq = floordiv_i(y1 - y0, Wi);
            sh.cax[tid][1][0] = (y1 - y0) * Hi; sh.cax[tid][1][1] = (y3 - y0) * Wi; sh.cax[tid][1][2] = y0 * Di;
            sh.cax[tid][1][3] = aq; sh.cax[tid][1][4] = (y1 - y0) * Hi - aq * Di;
          }
          uint32_t bits = hopd_component_bits(x0 / lastStep) + hopd_component_bits(y0 / lastStep)
                        + hopd_component_bits((x1 - 2 * W + 1) / lastStep) + hopd_component_bits(y1 / lastStep)
                        + hopd_component_bits((x2 - 2 * W + 1) / lastStep) + hopd_component_bits((y2 - 2 * H + 1) / lastStep);   // getBitsGT: corners 0..2
          const int den = (x1 - x2) * (y3 - y2) - (x3 - x2) * (y1 - y2);                      // 0 => h[2] is NaN/inf: not affine, :5323
          sh.cfix[tid] = den != 0 ? mvc + ((lambda_cost * bits) >> 16) : 0xFFFFFFFFu;         // :5345-5358
          sh.csatd[tid] = 0;
        }
        wg_sync<NW>();
        // ---- evaluation in rounds of blocks with exact early termination: block costs are non-negative, so a
        //      candidate whose partial cost has reached the incumbent cannot become the first-best (strict '<', :5361) ----
        const int shift_dn = pic.bd_y - 8;
        const int nblk = had8 ? (W >> 3) * (H >> 3) : (W >> 2) * (H >> 2);
        const int R = (nblk % 8 == 0 && nblk >= 16) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1, bpr = nblk / R;
        for (int r = 0; r < R; r++) {
          if (wave == 0) {                                     // (re)build the list of candidates still in the race; order is irrelevant
            const bool keep = lane < nc && sh.cfix[lane] != 0xFFFFFFFFu && ((uint32_t)sh.csatd[lane] >> shift_dn) + sh.cfix[lane] < distBest;
            const unsigned long long mk = __ballot(keep);
            if (keep) sh.alive[__popcll(mk & ((1ull << lane) - 1ull))] = (uint8_t)lane;
            if (lane == 0) sh.n_alive = __popcll(mk);
          }
          wg_sync<NW>();
          const int na = sh.n_alive;
          if (na == 0) break;                                  // uniform
          if (!use_had) {
            if (had8) gt_eval<SH, PT, NW, 8, false>(sh, na, r * bpr, bpr, W, H, m, PP, centre, wave, lane);
            else      gt_eval<SH, PT, NW, 4, false>(sh, na, r * bpr, bpr, W, H, m, PP, centre, wave, lane);
          } else if (had8) gt_eval<SH, PT, NW, 8, true>(sh, na, r * bpr, bpr, W, H, m, PP, centre, wave, lane);
          else             gt_eval<SH, PT, NW, 4, true>(sh, na, r * bpr, bpr, W, H, m, PP, centre, wave, lane);
          wg_sync<NW>();
        }
        // a candidate that left the race keeps a partial cost >= the incumbent: it cannot win, which is all that matters
        if (tid < nc) sh.cand_cost[c0 + tid] = sh.cfix[tid] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((uint32_t)sh.csatd[tid] >> shift_dn) + sh.cfix[tid];
        wg_sync<NW>();
      }
      // ---- first-best in visit order, strict '<' against the incumbent (:5361) ----
      unsigned long long kbest = ~0ull;
      for (int ci = tid; ci < ncand; ci += NT) {
        unsigned long long key = ((unsigned long long)sh.cand_cost[ci] << 16) | (unsigned long long)ci;
        kbest = key < kbest ? key : kbest;
      }
      kbest = hopd_wave_min_u64(kbest);
      if (NW > 1) {
        if (lane == 0 && kbest != ~0ull) atomicMin(&sh.best, kbest);
        __syncthreads();
        kbest = sh.best;
      }
      const unsigned long long kb = kbest;
      if (kb != ~0ull && (uint32_t)(kb >> 16) != 0xFFFFFFFFu && (uint32_t)(kb >> 16) < distBest) {
        distBest = (uint32_t)(kb >> 16);
        const int idx = clist[(int)(kb & 0xFFFF)];
        const int ii[4] = { idx / 125, (idx / 25) % 5, (idx / 5) % 5, idx % 5 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int ddx, ddy; corner_off(ii[k], s, ddx, ddy);
          bestC[0][k] = cN[0][k] + ddx; bestC[1][k] = cN[1][k] + ddy;
          bN[0][k] = bestC[0][k]; bN[1][k] = bestC[1][k];
        }
        bestSSX = Hor; bestSSY = Ver;
      }
      wg_sync<NW>();                                          // everyone has read sh.best / the lists before they are rewritten
    }
  }
  if (tid == 0) {
    bool flag = false;
#pragma unroll
    for (int k = 0; k < 4; k++) flag = flag || bestC[0][k] != 0 || bestC[1][k] != 0;     // :5436-5439
    if (flag) {
      rp->gt_flag = 1;
      rp->gt[0] = bestC[0][0] / lastStep;               rp->gt[1] = bestC[1][0] / lastStep;
      rp->gt[2] = (bestC[0][1] - 2 * W + 1) / lastStep; rp->gt[3] = bestC[1][1] / lastStep;
      rp->gt[4] = (bestC[0][2] - 2 * W + 1) / lastStep; rp->gt[5] = (bestC[1][2] - 2 * H + 1) / lastStep;
      rp->gt[6] = bestC[0][3] / lastStep;               rp->gt[7] = (bestC[1][3] - 2 * H + 1) / lastStep;
      rp->cost = distBest;
      rp->mv_final[0] = bestSSX >> 2; rp->mv_final[1] = bestSSY >> 2;                    // :5455-5457
      rp->half_final[0] = 0; rp->half_final[1] = 0; rp->qter_final[0] = 0; rp->qter_final[1] = 0;
    } else {
      rp->gt_flag = 0;
      for (int k = 0; k < 8; k++) rp->gt[k] = 0;
      rp->cost = rp->frac_cost;
      rp->mv_final[0] = mv0x; rp->mv_final[1] = mv0y;
      rp->half_final[0] = rp->half[0]; rp->half_final[1] = rp->half[1];
      rp->qter_final[0] = rp->qter[0]; rp->qter_final[1] = rp->qter[1];
    }
  }
}

// class lists: valid PUs of at most 256 samples with both sides <= 16 -> one wave; the other valid PUs -> four waves
__global__ void k_gt_prep(const hop_pu_job* __restrict__ jobs, const hop_pu_result* __restrict__ res, int n,
                          unsigned int* __restrict__ counts, int32_t* __restrict__ small_list, int32_t* __restrict__ big_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && !res[i].not_valid;               // bNotValCU: xMotionEstimation returned before the GT search
  const bool small = live && jobs[i].w <= 16 && jobs[i].h <= 16;
  const bool big = live && !small;
  // one atomic per wave and class (the per-thread version serialised 131072 atomics on two addresses)
  const unsigned long long ms = __ballot(small), mb = __ballot(big);
  const int lane = threadIdx.x & 63;
  unsigned int bs = 0, bb = 0;
  if (lane == 0) { if (ms) bs = atomicAdd(counts, (unsigned)__popcll(ms)); if (mb) bb = atomicAdd(counts + 1, (unsigned)__popcll(mb)); }
  bs = __shfl(bs, 0); bb = __shfl(bb, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (small) small_list[bs + __popcll(ms & below)] = i;
  if (big)   big_list[bb + __popcll(mb & below)] = i;
}

// sort the valid PUs into the two size classes (<= 16x16: one wave per PU; larger: four); sc = 256 B of counters + 2 lists of n
void hop_launch_size_classes(hop_ctx* c, int n, const hop_pu_job* d_jobs, const hop_pu_result* d_res, void* sc) {
  unsigned int* counts = (unsigned int*)sc;
  int32_t* small_list = (int32_t*)((char*)sc + 256);
  int32_t* big_list = small_list + n;
  (void)hipMemsetAsync(counts, 0, 8, c->stream);
  hipLaunchKernelGGL(k_gt_prep, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, d_res, n, counts, small_list, big_list);
}

int hop_launch_gt(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res) {
  // scratch (after the SS search's use of it on the same stream): 2 counters + 2 index lists
  void* sc; int r = hop_scratch(c, 256 + (size_t)n * 8, &sc); if (r) return r;
  unsigned int* counts = (unsigned int*)sc;
  int32_t* small_list = (int32_t*)((char*)sc + 256);
  int32_t* big_list = small_list + n;
  hop_pics pic = hop_make_pics(c);
  const int pr = hop_prof_begin(c, HOP_K_GT_SEARCH, (uint64_t)n);
  hop_launch_size_classes(c, n, d_jobs, d_res, sc);
  // grids are upper bounds: blocks beyond the class count exit on their first instruction
  if (c->bd_y == 8) {
    hipLaunchKernelGGL((k_gt_search<uint16_t, 4, 64>), dim3(n), dim3(256), 0, c->stream, d_jobs, pic, d_res, big_list, counts + 1);
    hipLaunchKernelGGL((k_gt_search<uint16_t, 1, 16>), dim3(n), dim3(64), 0, c->stream, d_jobs, pic, d_res, small_list, counts);
  } else {
    hipLaunchKernelGGL((k_gt_search<uint32_t, 4, 64>), dim3(n), dim3(256), 0, c->stream, d_jobs, pic, d_res, big_list, counts + 1);
    hipLaunchKernelGGL((k_gt_search<uint32_t, 1, 16>), dim3(n), dim3(64), 0, c->stream, d_jobs, pic, d_res, small_list, counts);
  }
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "gt_search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
