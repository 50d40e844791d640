// k_deblock.hip -- the deblocking filter over the resident reconstruction picture, gfx950.
//
// replaces: TComLoopFilter::loopFilterPic (TLibCommon/TComLoopFilter.cpp:129-153) with xDeblockCU (:166-227), xSetEdgefilterTU / PU (:254-338), xSetLoopfilterParam
// (:341-393), xGetBoundaryStrengthSingle (:395-519), xEdgeFilterLuma (:522-632), xEdgeFilterChroma (:635-737) and the sample filters (:758-881).
//
// The reference walks CTU by CTU and CU by CU, keeping edge flags and strengths in arrays of one CTU.  Here the picture is flat: an edge segment (4 samples of an edge on
// the 8x8 grid) is a pure function of the two 4x4 units it separates -- the CU depth, partition shape and transform depth of the Q unit say whether an edge lies there
// and whether it is a transform edge, the two units' modes / cbf / vectors give the strength -- and segments of one direction never touch each other's samples (a filter
// reads 4 and writes 3 samples either side, edges are 8 apart).  So:
//   k_dbk_strength   one thread per 4x4 unit: the strengths of its left and upper edge segments into two byte planes       (reads 2 x 44 B of partition data per unit)
//   k_dbk_luma<DIR>  one thread per edge segment: 4 lines x 8 samples in registers, decisions on lines 0 and 3, filtered lines back; consecutive threads take consecutive
//                    segments along a sample row, so a wave's loads are contiguous (vertical edges: 16 B per thread per row; horizontal: 8 B per thread per row)
//   k_dbk_chroma<DIR> one thread per segment and plane on the 8-sample chroma grid, strength 2 only
// all vertical edges of the picture first, then all horizontal ones on the result, as loopFilterPic does.  HBM-bound: the planes are read and written once per direction
// (2 x 2 x 3 B/sample) plus the partition data (44 B per 16 samples); DESIGN.md section 4 has the figures.  The pictures of a stacked context are blockIdx.z.
#include "hop_dev.h"

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hop_set_err((c), HOP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

struct DbkGeo {
  int w, h, wctu, n_ctu;          // one picture
  int pitch_rows;                 // rows between the pictures of a stack in the reconstruction planes
  int w4, h4;                     // the picture in 4x4 units
  int qp, beta_off, tc_off, cb_off, cr_off, bd, disable;
};

__constant__ uint8_t c_dbk_tc[54] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };          // sm_tcTable :59-62
__constant__ uint8_t c_dbk_beta[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };   // sm_betaTable :64-67
__constant__ uint8_t c_dbk_cqp[58] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };  // g_aucChromaScale

__device__ __forceinline__ int dbk_clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
__device__ __forceinline__ int dbk_z(int ux, int uy) {   // z-order index of unit (ux, uy) inside a CTU
  return (ux & 1) | ((uy & 1) << 1) | ((ux & 2) << 1) | ((uy & 2) << 2) | ((ux & 4) << 2) | ((uy & 4) << 3) | ((ux & 8) << 3) | ((uy & 8) << 4);
}
__device__ __forceinline__ const hop_cu_part* dbk_part(const hop_cu_part* parts, const DbkGeo& g, int pic, int x4, int y4) {
  return parts + ((size_t)pic * g.n_ctu + (size_t)(y4 >> 4) * g.wctu + (x4 >> 4)) * 256 + dbk_z(x4 & 15, y4 & 15);
}

// the strength of the edge segment between unit Q and its neighbour P (dx4, dy4 = -1, 0 or 0, -1); xr: Q's offset inside its CU along the direction, in samples
__device__ int dbk_strength(const hop_cu_part* q, const hop_cu_part* p, int xr, int dir, int at_picture_border, int disable) {
  if (disable || q->part_size == 15) return 0;
  const int cu = 64 >> q->depth, tu = cu >> q->tr_idx, ps = q->part_size;
  bool edge = false, preset = false;
  if (xr == 0) { edge = preset = !at_picture_border; }                      // the CU's own edge (:283-284 after :270-271)
  else {
    if ((xr & (tu - 1)) == 0) edge = preset = true;                         // a transform unit's first edge (:254-275)
    if (dir == 0) { if (((ps == 2 || ps == 3) && xr == (cu >> 1)) || (ps == 6 && xr == (cu >> 2)) || (ps == 7 && xr == cu - (cu >> 2))) edge = true; }   // Nx2N, NxN, nLx2N, nRx2N
    else          { if (((ps == 1 || ps == 3) && xr == (cu >> 1)) || (ps == 4 && xr == (cu >> 2)) || (ps == 5 && xr == cu - (cu >> 2))) edge = true; }   // 2NxN, NxN, 2NxnU, 2NxnD
  }
  if (!edge) return 0;
  if (p->pred_mode == 1 || q->pred_mode == 1) return 2;
  if (preset && (((q->cbf[0] >> q->tr_idx) & 1) || ((p->cbf[0] >> p->tr_idx) & 1))) return 1;
  int pmx = p->mv[0], pmy = p->mv[1], qmx = q->mv[0], qmy = q->mv[1];
  if (p->ref_idx < 0) pmx = pmy = 0;
  if (q->ref_idx < 0) qmx = qmy = 0;
  const bool other_ref = (p->ref_idx < 0) != (q->ref_idx < 0) || (p->ref_idx >= 0 && p->ref_idx != q->ref_idx);
  return (other_ref || abs(qmx - pmx) >= 4 || abs(qmy - pmy) >= 4) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_dbk_strength(DbkGeo g, const hop_cu_part* __restrict__ parts, uint8_t* __restrict__ bsv, uint8_t* __restrict__ bsh) {
  const int i = blockIdx.x * 256 + threadIdx.x, pic = blockIdx.z;
  if (i >= g.w4 * g.h4) return;
  const int x4 = i % g.w4, y4 = i / g.w4;
  const hop_cu_part* q = dbk_part(parts, g, pic, x4, y4);
  const int cu = 64 >> q->depth;
  uint8_t v = 0, h = 0;
  if ((x4 & 1) == 0) v = (uint8_t)dbk_strength(q, x4 ? dbk_part(parts, g, pic, x4 - 1, y4) : q, (x4 * 4) & (cu - 1), 0, x4 == 0, g.disable);
  if ((y4 & 1) == 0) h = (uint8_t)dbk_strength(q, y4 ? dbk_part(parts, g, pic, x4, y4 - 1) : q, (y4 * 4) & (cu - 1), 1, y4 == 0, g.disable);
  bsv[(size_t)pic * g.w4 * g.h4 + i] = v; bsh[(size_t)pic * g.w4 * g.h4 + i] = h;
}

// one line across the edge: s[0..3] = p3..p0, s[4..7] = q0..q3 (xPelFilterLuma :758-826)
__device__ __forceinline__ void dbk_luma_line(int* s, int tc, bool strong, int thr_cut, bool second_p, bool second_q, int maxv) {
  const int m0 = s[0], m1 = s[1], m2 = s[2], m3 = s[3], m4 = s[4], m5 = s[5], m6 = s[6], m7 = s[7];
  if (strong) {
    s[3] = dbk_clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[4] = dbk_clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[2] = dbk_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[5] = dbk_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[1] = dbk_clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[6] = dbk_clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < thr_cut) {
      delta = dbk_clip3(-tc, tc, delta);
      s[3] = dbk_clip3(0, maxv, m3 + delta); s[4] = dbk_clip3(0, maxv, m4 - delta);
      const int tc2 = tc >> 1;
      if (second_p) s[2] = dbk_clip3(0, maxv, m2 + dbk_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (second_q) s[5] = dbk_clip3(0, maxv, m5 + dbk_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
}
__device__ __forceinline__ bool dbk_strong(const int* s, int d, int beta, int tc) {          // xUseStrongFiltering :860-870
  return (abs(s[0] - s[3]) + abs(s[7] - s[4])) < (beta >> 3) && d < (beta >> 2) && abs(s[3] - s[4]) < ((tc * 5 + 1) >> 1);
}

// DIR 0: vertical edges at x = 8, 16, ...; a thread = 4 rows of one edge.  DIR 1: horizontal edges at y = 8, 16, ...; a thread = 4 columns of one edge.
template <int DIR>
__global__ __launch_bounds__(256) void k_dbk_luma(DbkGeo g, const uint8_t* __restrict__ bs_plane, int16_t* __restrict__ rec) {
  const int pic = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  const int ne = DIR == 0 ? (g.w4 >> 1) : g.w4, nl = DIR == 0 ? g.h4 : (g.h4 >> 1);          // segments along a row of threads x rows of threads
  if (i >= ne * nl) return;
  const int a = i % ne, b = i / ne;
  const int x4 = DIR == 0 ? 2 * a : a, y4 = DIR == 0 ? b : 2 * b;
  const int bs = bs_plane[(size_t)pic * g.w4 * g.h4 + (size_t)y4 * g.w4 + x4];
  if (!bs) return;
  const int scale = 1 << (g.bd - 8), maxv = (1 << g.bd) - 1;
  const int tc = c_dbk_tc[dbk_clip3(0, 53, g.qp + 2 * (bs - 1) + (g.tc_off << 1))] * scale, beta = c_dbk_beta[dbk_clip3(0, 51, g.qp + (g.beta_off << 1))] * scale;
  int16_t* base = rec + ((size_t)pic * g.pitch_rows + (size_t)y4 * 4) * g.w + x4 * 4;
  int s[4][8];
  if (DIR == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const short4 lo = *(const short4*)(base + (size_t)k * g.w - 4), hi = *(const short4*)(base + (size_t)k * g.w);
      s[k][0] = lo.x; s[k][1] = lo.y; s[k][2] = lo.z; s[k][3] = lo.w; s[k][4] = hi.x; s[k][5] = hi.y; s[k][6] = hi.z; s[k][7] = hi.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) { const short4 v = *(const short4*)(base + (ptrdiff_t)(r - 4) * g.w); s[0][r] = v.x; s[1][r] = v.y; s[2][r] = v.z; s[3][r] = v.w; }
  }
  const int dp0 = abs(s[0][1] - 2 * s[0][2] + s[0][3]), dq0 = abs(s[0][4] - 2 * s[0][5] + s[0][6]), dp3 = abs(s[3][1] - 2 * s[3][2] + s[3][3]), dq3 = abs(s[3][4] - 2 * s[3][5] + s[3][6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const int side = (beta + (beta >> 1)) >> 3;
  const bool strong = dbk_strong(s[0], 2 * d0, beta, tc) && dbk_strong(s[3], 2 * d3, beta, tc);
#pragma unroll
  for (int k = 0; k < 4; k++) dbk_luma_line(s[k], tc, strong, tc * 10, dp < side, dq < side, maxv);
  if (DIR == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      *(short4*)(base + (size_t)k * g.w - 4) = make_short4((short)s[k][0], (short)s[k][1], (short)s[k][2], (short)s[k][3]);
      *(short4*)(base + (size_t)k * g.w) = make_short4((short)s[k][4], (short)s[k][5], (short)s[k][6], (short)s[k][7]);
    }
  } else {
#pragma unroll
    for (int r = 1; r < 7; r++) *(short4*)(base + (ptrdiff_t)(r - 4) * g.w) = make_short4((short)s[0][r], (short)s[1][r], (short)s[2][r], (short)s[3][r]);
  }
}

// chroma edges lie on the 8-sample chroma grid (16 luma samples); a thread = the 2 chroma lines of one luma unit along the edge, one plane (blockIdx.y)
template <int DIR>
__global__ __launch_bounds__(256) void k_dbk_chroma(DbkGeo g, const uint8_t* __restrict__ bs_plane, int16_t* __restrict__ cb, int16_t* __restrict__ cr) {
  const int pic = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x, comp = blockIdx.y;
  const int ne = DIR == 0 ? (g.w4 + 3) >> 2 : g.w4, nl = DIR == 0 ? g.h4 : (g.h4 + 3) >> 2;
  if (i >= ne * nl) return;
  const int a = i % ne, b = i / ne;
  const int x4 = DIR == 0 ? 4 * a : a, y4 = DIR == 0 ? b : 4 * b;
  const int bs = bs_plane[(size_t)pic * g.w4 * g.h4 + (size_t)y4 * g.w4 + x4];
  if (bs < 2) return;
  const int wc = g.w >> 1, scale = 1 << (g.bd - 8), maxv = (1 << g.bd) - 1;
  const int qi = g.qp + (comp == 0 ? g.cb_off : g.cr_off), qpc = qi < 0 ? qi : qi > 57 ? qi - 6 : c_dbk_cqp[qi];                     // QpUV :53
  const int tc = c_dbk_tc[dbk_clip3(0, 53, qpc + 2 * (bs - 1) + (g.tc_off << 1))] * scale;
  int16_t* base = (comp == 0 ? cb : cr) + ((size_t)pic * (g.pitch_rows >> 1) + (size_t)y4 * 2) * wc + x4 * 2;
  const ptrdiff_t o = DIR == 0 ? 1 : wc, step = DIR == 0 ? wc : 1;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int16_t* t = base + k * step;
    const int m2 = t[-2 * o], m3 = t[-o], m4 = t[0], m5 = t[o];
    const int delta = dbk_clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    t[-o] = (int16_t)dbk_clip3(0, maxv, m3 + delta); t[0] = (int16_t)dbk_clip3(0, maxv, m4 - delta);
  }
}

extern "C" int hop_deblock_frame(hop_ctx* c, const hop_deblock_params* p, const hop_cu_part* parts) {
  if (!c || !p || !parts) return hop_set_err(c, HOP_ERR_ARG, "hop_deblock_frame: bad argument");
  if (c->bd_y != c->bd_c) return hop_set_err(c, HOP_ERR_ARG, "hop_deblock_frame: luma and chroma bit depths must be equal");
  if (p->qp < 0 || p->qp > 51 || p->beta_offset_div2 < -6 || p->beta_offset_div2 > 6 || p->tc_offset_div2 < -6 || p->tc_offset_div2 > 6 || p->cb_qp_offset < -12 || p->cb_qp_offset > 12 ||
      p->cr_qp_offset < -12 || p->cr_qp_offset > 12) return hop_set_err(c, HOP_ERR_ARG, "hop_deblock_frame: parameter out of range");
  const int n_pic = c->sub_pitch ? (c->pic_h - c->sub_h) / c->sub_pitch + 1 : 1, h = c->sub_pitch ? c->sub_h : c->pic_h;
  if ((c->pic_w & 7) || (h & 7)) return hop_set_err(c, HOP_ERR_ARG, "hop_deblock_frame: the picture size must be a multiple of the minimum CU size (8)");
  DbkGeo g;
  g.w = c->pic_w; g.h = h; g.wctu = (g.w + 63) >> 6; g.n_ctu = g.wctu * ((h + 63) >> 6); g.pitch_rows = c->sub_pitch; g.w4 = g.w >> 2; g.h4 = h >> 2;
  g.qp = p->qp; g.beta_off = p->beta_offset_div2; g.tc_off = p->tc_offset_div2; g.cb_off = p->cb_qp_offset; g.cr_off = p->cr_qp_offset; g.bd = c->bd_y; g.disable = p->disable ? 1 : 0;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n_parts = (size_t)n_pic * g.n_ctu * 256, n_units = (size_t)n_pic * g.w4 * g.h4;
  hop_cu_part* d_parts = nullptr; uint8_t* d_bs = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_parts, n_parts * sizeof(hop_cu_part)));
  hipError_t e = hipMalloc((void**)&d_bs, 2 * n_units);
  if (e == hipSuccess) e = hipMemcpyAsync(d_parts, parts, n_parts * sizeof(hop_cu_part), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const unsigned units = (unsigned)(g.w4 * g.h4);
    const int rec_k = hop_prof_begin(c, HOP_K_DEBLOCK, (uint64_t)n_pic * g.n_ctu);
    hipLaunchKernelGGL(k_dbk_strength, dim3((units + 255) / 256, 1, n_pic), dim3(256), 0, c->stream, g, d_parts, d_bs, d_bs + n_units);
    hipLaunchKernelGGL(k_dbk_luma<0>, dim3((units / 2 + 255) / 256, 1, n_pic), dim3(256), 0, c->stream, g, d_bs, c->rec[0]);
    hipLaunchKernelGGL(k_dbk_chroma<0>, dim3((((g.w4 + 3) >> 2) * g.h4 + 255) / 256, 2, n_pic), dim3(256), 0, c->stream, g, d_bs, c->rec[1], c->rec[2]);
    hipLaunchKernelGGL(k_dbk_luma<1>, dim3((units / 2 + 255) / 256, 1, n_pic), dim3(256), 0, c->stream, g, d_bs + n_units, c->rec[0]);
    hipLaunchKernelGGL(k_dbk_chroma<1>, dim3((g.w4 * ((g.h4 + 3) >> 2) + 255) / 256, 2, n_pic), dim3(256), 0, c->stream, g, d_bs + n_units, c->rec[1], c->rec[2]);
    hop_prof_end(c, rec_k);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(d_parts); (void)hipFree(d_bs);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_deblock_frame: %s", hipGetErrorString(e));
  return HOP_OK;
}
