/* oracle/hop_oracle_lf.c -- TEST INFRASTRUCTURE.  CPU restatement of the deblocking filter of the reference, the way the reference walks it: CTU by CTU,
 * CU by CU along the coding quadtree, per CU the edge flags and boundary strengths in arrays of the CTU's 256 partitions, then the CU's edges on the 8x8 grid.
 *   TComLoopFilter::loopFilterPic             TLibCommon/TComLoopFilter.cpp:129-153   all vertical edges of the picture, then all horizontal ones
 *   xDeblockCU                                :166-227                                 recursion, edge flags, strengths, luma edges every 8, chroma every 16 samples
 *   xSetEdgefilterTU / PU / xSetLoopfilterParam :254-393                               transform-unit edges, partition edges, picture border
 *   xGetBoundaryStrengthSingle                :395-519                                 2 intra, 1 coded residual at a transform edge, else vectors / reference
 *   xEdgeFilterLuma / xEdgeFilterChroma       :522-756                                 decisions on lines 0 and 3 of a 4-line segment; chroma only at strength 2
 *   xPelFilterLuma / Chroma, xUseStrongFiltering, xCalcDP / DQ  :758-881
 * For the configurations of the path: one slice, one tile, no PCM, no lossless CUs, every CU at the slice QP (MaxDeltaQP 0).  The product's kernels (csrc/k_deblock.hip)
 * are laid out the other way round -- one thread per edge segment of the picture -- and are tested against this file; this file is pinned by the reference encoder itself:
 * with it in place of loopFilterPic the encoder writes the unmodified encoder's bitstream and reconstruction (tests/test_encoder_pic.py), and HOP_PIC_CHECK compares its
 * planes with those of the reference's own loopFilterPic. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hop_oracle.h"

static const uint8_t TC_TABLE[54] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };   /* sm_tcTable :59-62 */
static const uint8_t BETA_TABLE[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };   /* sm_betaTable :64-67 */
static const uint8_t CHROMA_QP[58] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };  /* g_aucChromaScale, TComRom.cpp */

typedef struct {
  int w, h, wctu, qp, bd, beta_off, tc_off, cb_off, cr_off, disable;
  int16_t* pl[3];
  const hop_o_cu_part* parts;
  uint8_t bs[2][256], edge[2][256];
  int left_edge, top_edge, internal_edge;
} Lf;

static int zidx(int ux, int uy) { int z = 0; for (int b = 0; b < 4; b++) z |= (((ux >> b) & 1) << (2 * b)) | (((uy >> b) & 1) << (2 * b + 1)); return z; }
static int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
/* the partition that holds luma sample (x, y), or NULL outside the picture */
static const hop_o_cu_part* part_at(const Lf* f, int x, int y) {
  if (x < 0 || y < 0 || x >= f->w || y >= f->h) return NULL;
  return &f->parts[(size_t)((y >> 6) * f->wctu + (x >> 6)) * 256 + zidx((x & 63) >> 2, (y & 63) >> 2)];
}

/* xSetEdgefilterMultiple: n units of the edge `edge_idx` units into the block at unit (ux, uy) of the CTU; the strength is pre-set only on a block's own first edge */
static void set_edges(Lf* f, int ux, int uy, int dir, int edge_idx, int value, int wu, int hu) {
  const int n = dir == 0 ? hu : wu;
  for (int i = 0; i < n; i++) {
    const int z = dir == 0 ? zidx(ux + edge_idx, uy + i) : zidx(ux + i, uy + edge_idx);
    f->edge[dir][z] = (uint8_t)value;
    if (edge_idx == 0) f->bs[dir][z] = (uint8_t)value;
  }
}
static void set_edges_tu(Lf* f, const hop_o_cu_part* ctu, int ux, int uy, int size_u, int depth) {
  const hop_o_cu_part* p = &ctu[zidx(ux, uy)];
  if (p->tr_idx + p->depth > depth) { const int h = size_u >> 1; for (int q = 0; q < 4; q++) set_edges_tu(f, ctu, ux + (q & 1) * h, uy + (q >> 1) * h, h, depth + 1); return; }
  const int tu = (64 >> p->depth >> p->tr_idx) >> 2;
  set_edges(f, ux, uy, 0, 0, f->internal_edge, tu, tu); set_edges(f, ux, uy, 1, 0, f->internal_edge, tu, tu);
}
static void boundary_strength(Lf* f, int cx, int cy, int dir, int z, int ux, int uy) {
  const int x = cx + ux * 4, y = cy + uy * 4;
  const hop_o_cu_part* q = part_at(f, x, y); const hop_o_cu_part* p = dir == 0 ? part_at(f, x - 4, y) : part_at(f, x, y - 4);
  int bs = 0;
  if (p->pred_mode == 1 || q->pred_mode == 1) bs = 2;
  else if (f->bs[dir][z] && (((q->cbf[0] >> q->tr_idx) & 1) || ((p->cbf[0] >> p->tr_idx) & 1))) bs = 1;
  else {                                                                   /* one list, one picture in it: the P-slice branch (:497-514) */
    int pmx = p->mv[0], pmy = p->mv[1], qmx = q->mv[0], qmy = q->mv[1];
    if (p->ref_idx < 0) pmx = pmy = 0;
    if (q->ref_idx < 0) qmx = qmy = 0;
    bs = ((p->ref_idx < 0) != (q->ref_idx < 0) || (p->ref_idx >= 0 && p->ref_idx != q->ref_idx) || abs(qmx - pmx) >= 4 || abs(qmy - pmy) >= 4) ? 1 : 0;
  }
  f->bs[dir][z] = (uint8_t)bs;
}
static void luma_line(int16_t* s, int o, int tc, int strong, int thr_cut, int second_p, int second_q, int maxv) {
  const int m0 = s[-4 * o], m1 = s[-3 * o], m2 = s[-2 * o], m3 = s[-o], m4 = s[0], m5 = s[o], m6 = s[2 * o], m7 = s[3 * o];
  if (strong) {
    s[-o] = (int16_t)clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[0] = (int16_t)clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[-2 * o] = (int16_t)clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[o] = (int16_t)clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[-3 * o] = (int16_t)clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[2 * o] = (int16_t)clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < thr_cut) {
      delta = clip3(-tc, tc, delta);
      s[-o] = (int16_t)clip3(0, maxv, m3 + delta); s[0] = (int16_t)clip3(0, maxv, m4 - delta);
      const int tc2 = tc >> 1;
      if (second_p) s[-2 * o] = (int16_t)clip3(0, maxv, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (second_q) s[o] = (int16_t)clip3(0, maxv, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
}
static int strong_line(const int16_t* s, int o, int d, int beta, int tc) {
  const int ds = abs(s[-4 * o] - s[-o]) + abs(s[3 * o] - s[0]);
  return ds < (beta >> 3) && d < (beta >> 2) && abs(s[-o] - s[0]) < ((tc * 5 + 1) >> 1);
}
static void edge_luma(Lf* f, int cx, int cy, int ux, int uy, int size_u, int dir, int e) {
  const int stride = f->w, o = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1, maxv = (1 << f->bd) - 1, scale = 1 << (f->bd - 8);
  for (int i = 0; i < size_u; i++) {
    const int eux = dir == 0 ? ux + e : ux + i, euy = dir == 0 ? uy + i : uy + e;
    const int bs = f->bs[dir][zidx(eux, euy)];
    if (!bs) continue;
    int16_t* s = f->pl[0] + (size_t)(cy + euy * 4) * stride + cx + eux * 4;
    const int qp = f->qp;                                                  /* (QP_P + QP_Q + 1) >> 1 of two CUs at the slice QP */
    const int tc = TC_TABLE[clip3(0, 53, qp + 2 * (bs - 1) + (f->tc_off << 1))] * scale, beta = BETA_TABLE[clip3(0, 51, qp + (f->beta_off << 1))] * scale;
    const int side = (beta + (beta >> 1)) >> 3, thr_cut = tc * 10;
    const int16_t* l0 = s; const int16_t* l3 = s + 3 * step;
    const int dp0 = abs(l0[-3 * o] - 2 * l0[-2 * o] + l0[-o]), dq0 = abs(l0[0] - 2 * l0[o] + l0[2 * o]), dp3 = abs(l3[-3 * o] - 2 * l3[-2 * o] + l3[-o]), dq3 = abs(l3[0] - 2 * l3[o] + l3[2 * o]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
    if (d < beta) {
      const int strong = strong_line(l0, o, 2 * d0, beta, tc) && strong_line(l3, o, 2 * d3, beta, tc);
      for (int k = 0; k < 4; k++) luma_line(s + k * step, o, tc, strong, thr_cut, dp < side, dq < side, maxv);
    }
  }
}
static void edge_chroma(Lf* f, int cx, int cy, int ux, int uy, int size_u, int dir, int e) {
  if ((dir == 0 && ((ux + e) & 3)) || (dir == 1 && ((uy + e) & 3))) return;     /* chroma edges lie on the 8-sample chroma grid of the CTU (:664-667) */
  const int stride = f->w >> 1, o = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1, maxv = (1 << f->bd) - 1, scale = 1 << (f->bd - 8);
  for (int i = 0; i < size_u; i++) {
    const int eux = dir == 0 ? ux + e : ux + i, euy = dir == 0 ? uy + i : uy + e;
    const int bs = f->bs[dir][zidx(eux, euy)];
    if (bs <= 1) continue;
    for (int c = 1; c <= 2; c++) {
      const int qpc = CHROMA_QP[clip3(0, 57, f->qp + (c == 1 ? f->cb_off : f->cr_off))];
      const int tc = TC_TABLE[clip3(0, 53, qpc + 2 * (bs - 1) + (f->tc_off << 1))] * scale;
      int16_t* s = f->pl[c] + (size_t)((cy >> 1) + euy * 2) * stride + (cx >> 1) + eux * 2;
      for (int k = 0; k < 2; k++) {
        int16_t* t = s + k * step;
        const int m2 = t[-2 * o], m3 = t[-o], m4 = t[0], m5 = t[o];
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        t[-o] = (int16_t)clip3(0, maxv, m3 + delta); t[0] = (int16_t)clip3(0, maxv, m4 - delta);
      }
    }
  }
}
static void deblock_cu(Lf* f, int ctu, int ux, int uy, int size_u, int depth, int dir) {
  const hop_o_cu_part* cp = &f->parts[(size_t)ctu * 256];
  const hop_o_cu_part* p = &cp[zidx(ux, uy)];
  const int cx = (ctu % f->wctu) * 64, cy = (ctu / f->wctu) * 64;
  if (p->part_size == 15) return;                                          /* SIZE_NONE: nothing coded here (:168-171) */
  if (p->depth > depth) {
    const int h = size_u >> 1;
    for (int q = 0; q < 4; q++) { const int sx = ux + (q & 1) * h, sy = uy + (q >> 1) * h; if (cx + sx * 4 < f->w && cy + sy * 4 < f->h) deblock_cu(f, ctu, sx, sy, h, depth + 1, dir); }
    return;
  }
  /* xSetLoopfilterParam: one slice, one tile -- the neighbour exists wherever the picture does */
  f->internal_edge = !f->disable; f->left_edge = !(cx + ux * 4 == 0 || f->disable); f->top_edge = !(cy + uy * 4 == 0 || f->disable);
  set_edges_tu(f, cp, ux, uy, size_u, depth);
  set_edges(f, ux, uy, 0, 0, f->left_edge, size_u, size_u); set_edges(f, ux, uy, 1, 0, f->top_edge, size_u, size_u);
  const int hu = size_u >> 1, qu = size_u >> 2;
  switch (p->part_size) {                                                  /* xSetEdgefilterPU :278-338 */
    case 1: set_edges(f, ux, uy, 1, hu, f->internal_edge, size_u, size_u); break;
    case 2: set_edges(f, ux, uy, 0, hu, f->internal_edge, size_u, size_u); break;
    case 3: set_edges(f, ux, uy, 0, hu, f->internal_edge, size_u, size_u); set_edges(f, ux, uy, 1, hu, f->internal_edge, size_u, size_u); break;
    case 4: set_edges(f, ux, uy, 1, qu, f->internal_edge, size_u, size_u); break;
    case 5: set_edges(f, ux, uy, 1, size_u - qu, f->internal_edge, size_u, size_u); break;
    case 6: set_edges(f, ux, uy, 0, qu, f->internal_edge, size_u, size_u); break;
    case 7: set_edges(f, ux, uy, 0, size_u - qu, f->internal_edge, size_u, size_u); break;
    default: break;
  }
  for (int y = 0; y < size_u; y++) for (int x = 0; x < size_u; x++) {     /* the strength of every flagged unit on the 8-sample grid of this direction (:199-215) */
    const int z = zidx(ux + x, uy + y);
    const int on_grid = dir == 0 ? ((ux + x) & 1) == 0 : ((uy + y) & 1) == 0;
    if (f->edge[dir][z] && on_grid) boundary_strength(f, cx, cy, dir, z, ux + x, uy + y);
  }
  for (int e = 0; e < size_u; e += 2) {
    edge_luma(f, cx, cy, ux, uy, size_u, dir, e);
    if ((e & 3) == 0) edge_chroma(f, cx, cy, ux, uy, size_u, dir, e);
  }
}

/* y / cb / cr: the reconstruction, pitch w (w / 2), filtered in place.  parts: 256 per CTU in z-order, CTUs in raster order.  Returns 0. */
int hop_o_deblock_frame(int w, int h, int bit_depth, int qp, int beta_offset_div2, int tc_offset_div2, int cb_qp_offset, int cr_qp_offset, int disable,
                        const hop_o_cu_part* parts, int16_t* y, int16_t* cb, int16_t* cr) {
  Lf* f = (Lf*)calloc(1, sizeof(Lf));
  f->w = w; f->h = h; f->wctu = (w + 63) / 64; f->qp = qp; f->bd = bit_depth; f->beta_off = beta_offset_div2; f->tc_off = tc_offset_div2; f->cb_off = cb_qp_offset; f->cr_off = cr_qp_offset;
  f->disable = disable; f->pl[0] = y; f->pl[1] = cb; f->pl[2] = cr; f->parts = parts;
  const int n = f->wctu * ((h + 63) / 64);
  for (int dir = 0; dir < 2; dir++)
    for (int a = 0; a < n; a++) { memset(f->bs[dir], 0, 256); memset(f->edge[dir], 0, 256); deblock_cu(f, a, 0, 0, 16, 0, dir); }
  free(f);
  return 0;
}
