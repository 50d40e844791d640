/* oracle/hop_oracle_sao.c -- TEST INFRASTRUCTURE.  CPU restatement of the two picture-wide passes of the reference's SAO encoder:
 *   TEncSampleAdaptiveOffset::getStatistics / getBlkStats   TLibEncoder/TEncSampleAdaptiveOffset.cpp:305-352, :862-1383   per CTU, component and type: count and sum of
 *                                                           (original - deblocked) per edge class / band
 *   TComSampleAdaptiveOffset::offsetCTU / offsetBlock       TLibCommon/TComSampleAdaptiveOffset.cpp:365-707               the decided offsets applied, CTU by CTU
 * for the configuration of the path (SAOLcuBoundary 0: statistics leave out the 5 (3) columns and 4 (2) rows a CTU's right and lower neighbour's deblocking would still
 * change; one slice, one tile: a neighbour sample is unavailable only outside the picture).  The reference walks lines with running sign buffers; restated here per
 * sample: a sample of an edge type counts / is offset when both its neighbours along the type's direction lie inside the picture.  The decision between the two passes is
 * host logic of the product (hevc-hop_amd/host/hop_sao.cpp), compiled as it is into the CPU spine library; everything is pinned by the reference encoder itself
 * (tests/test_encoder_pic.py: HOP_PIC_SAO; with HOP_PIC_CHECK statistics, parameters and planes are compared with the reference's own, CTU by CTU). */
#include <stdint.h>
#include <string.h>
#include "hop_oracle.h"

static const int EO_DX[4] = { 1, 0, 1, -1 }, EO_DY[4] = { 0, 1, 1, 1 };   /* second neighbour of EO_0, EO_90, EO_135, EO_45; the first one is its mirror image */
static int sgn(int v) { return (v > 0) - (v < 0); }

/* stats: [ctu][comp][type 0..4][class 0..31][count, diff] int32 */
int hop_o_sao_stats(int w, int h, int bit_depth, const int16_t* const src[3], const int16_t* const org[3], int32_t* stats) {
  const int wctu = (w + 63) / 64, hctu = (h + 63) / 64;
  memset(stats, 0, (size_t)wctu * hctu * 3 * 5 * 32 * 2 * sizeof(int32_t));
  for (int ctu = 0; ctu < wctu * hctu; ctu++) for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, cs = 64 >> sh;
    const int x0 = (ctu % wctu) * cs, y0 = (ctu / wctu) * cs;
    const int bw = x0 + cs > pw ? pw - x0 : cs, bh = y0 + cs > ph ? ph - y0 : cs;
    const int right = x0 + cs < pw, below = y0 + cs < ph;                  /* (:323-327) */
    const int end_x = right ? bw - (c ? 3 : 5) : bw, end_y = below ? bh - (c ? 2 : 4) : bh;
    int32_t* st = stats + (size_t)(ctu * 3 + c) * 5 * 32 * 2;
    for (int y = 0; y < end_y; y++) for (int x = 0; x < end_x; x++) {
      const int X = x0 + x, Y = y0 + y, s = src[c][(size_t)Y * pw + X], d = org[c][(size_t)Y * pw + X] - s;
      for (int t = 0; t < 4; t++) {
        const int ax = X - EO_DX[t], ay = Y - EO_DY[t], bx = X + EO_DX[t], by = Y + EO_DY[t];
        if (ax < 0 || ax >= pw || ay < 0 || bx < 0 || bx >= pw || by >= ph) continue;
        const int cls = 2 + sgn(s - src[c][(size_t)ay * pw + ax]) + sgn(s - src[c][(size_t)by * pw + bx]);
        st[(t * 32 + cls) * 2]++; st[(t * 32 + cls) * 2 + 1] += d;
      }
      const int band = s >> (bit_depth - 5);
      st[(4 * 32 + band) * 2]++; st[(4 * 32 + band) * 2 + 1] += d;
    }
  }
  return 0;
}

/* params: [ctu][comp] reconstructed parameters (mode 0 = off; type 0..3 edge, 4 band; offset per class / band).  dst may not alias src. */
int hop_o_sao_apply(int w, int h, int bit_depth, const int16_t* const src[3], const hop_o_sao_param* params, int16_t* const dst[3]) {
  const int wctu = (w + 63) / 64, hctu = (h + 63) / 64, maxv = (1 << bit_depth) - 1;
  for (int c = 0; c < 3; c++) memcpy(dst[c], src[c], (size_t)(w >> (c ? 1 : 0)) * (h >> (c ? 1 : 0)) * 2);
  for (int ctu = 0; ctu < wctu * hctu; ctu++) for (int c = 0; c < 3; c++) {
    const hop_o_sao_param* p = &params[ctu * 3 + c];
    if (p->mode == 0) continue;
    const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, cs = 64 >> sh;
    const int x0 = (ctu % wctu) * cs, y0 = (ctu / wctu) * cs;
    const int bw = x0 + cs > pw ? pw - x0 : cs, bh = y0 + cs > ph ? ph - y0 : cs, t = p->type;
    for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
      const int X = x0 + x, Y = y0 + y, s = src[c][(size_t)Y * pw + X];
      int o;
      if (t < 4) {
        const int ax = X - EO_DX[t], ay = Y - EO_DY[t], bx = X + EO_DX[t], by = Y + EO_DY[t];
        if (ax < 0 || ax >= pw || ay < 0 || bx < 0 || bx >= pw || by >= ph) continue;
        o = p->offset[2 + sgn(s - src[c][(size_t)ay * pw + ax]) + sgn(s - src[c][(size_t)by * pw + bx])];
      } else o = p->offset[s >> (bit_depth - 5)];
      const int v = s + o;
      dst[c][(size_t)Y * pw + X] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
    }
  }
  return 0;
}
