// hop_sao.cpp -- the parameter decision of the SAO encoder: host logic between the two picture-wide device passes (statistics, offsetting).
//
// replaces: TEncSampleAdaptiveOffset::decideBlkParams (TLibEncoder/TEncSampleAdaptiveOffset.cpp:754-860) with deriveModeNewRDO (:565-704), deriveModeMergeRDO (:706-752),
// deriveOffsets (:460-562), estIterOffset (:427-457), getDistortion / estSaoDist (:381-425), TComSampleAdaptiveOffset::getMergeList / reconstructBlkSAOParam /
// invertQuantOffsets (TLibCommon/TComSampleAdaptiveOffset.cpp:224-339) and the syntax the rate is counted on: TEncSbac::codeSAOBlkParam / codeSAOOffsetParam /
// codeSaoMaxUvlc / codeSaoMerge / codeSaoTypeIdx (TLibEncoder/TEncSbac.cpp:2097-2170, :2369-2470) on the counting coder (TEncBinCoderCABACCounter.cpp:72-108).
// The CTUs are decided one after the other: a CTU's rate is counted on the two SAO contexts as the CTUs before it left them, its merge candidates are the reconstructed
// parameters of its left and upper neighbour; the fraction of a bit the coder carries comes in from the RD search (hop_rd_fraction_download) and runs on.
// Serial, a few hundred operations per CTU: it stays on the host.  Pinned inside the reference encoder (oracle/enc_shim_pic.cpp, HOP_PIC_SAO): same parameters CTU by
// CTU, same bitstream.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/hophip.h"

namespace {
enum { SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, TYPE_BO = 4, N_TYPES = 5, EO_CLASSES = 5, BO_CLASSES = 32, CLS_PLAIN = 2 };
const double MAX_COST = 1.7e+308;

struct Coder { uint8_t merge, type; uint64_t frac; };                    // the two SAO context models + TEncBinCABAC::m_fracBits
inline void reset_bits(Coder& k) { k.frac &= 32767; }                     // TEncSbac::resetBits -> TEncBinCABAC::resetBits (TEncBinCoderCABAC.cpp:163-170): the fraction stays
inline uint32_t written_bits(const Coder& k) { return (uint32_t)(k.frac >> 15); }
inline void bin(Coder& k, uint8_t& state, int b) { k.frac += hop_cabac_bin_bits(&state, b); }
inline void ep(Coder& k, int n) { k.frac += (uint64_t)32768 * n; }

struct Stat { const int32_t* p; int64_t count(int c) const { return p[2 * c]; } int64_t diff(int c) const { return p[2 * c + 1]; } };
struct Ctx {
  int bit_depth, max_offset, step_log2, enabled[3];
  double lambda[3];
  const int32_t* stats;
  Stat stat(int ctu, int comp, int type) const { Stat s = { stats + ((size_t)(ctu * 3 + comp) * N_TYPES + type) * 32 * 2 }; return s; }
};

uint8_t init_state(int init_value, int qp) {                              // ContextModel::init, ContextModel.cpp:56-65
  qp = qp < 0 ? 0 : qp > 51 ? 51 : qp;
  const int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset;
  st = st < 1 ? 1 : st > 126 ? 126 : st;
  const unsigned mp = st >= 64;
  return (uint8_t)(((mp ? (st - 64) : (63 - st)) << 1) + mp);
}

void code_max_uvlc(Coder& k, unsigned code, unsigned max_symbol) {       // codeSaoMaxUvlc :2102-2128: truncated unary, bypass
  if (max_symbol == 0) return;
  if (code == 0) { ep(k, 1); return; }
  ep(k, (int)code);
  if (max_symbol > code) ep(k, 1);
}
void code_type_idx(Coder& k, unsigned v) { if (v == 0) bin(k, k.type, 0); else { bin(k, k.type, 1); ep(k, 1); } }   // :2157-2168
void code_offset_param(const Ctx& c, Coder& k, int comp, const hop_sao_param& p) {                                   // codeSAOOffsetParam :2369-2436
  if (!c.enabled[comp]) return;
  if (comp < 2) code_type_idx(k, p.mode == SAO_OFF ? 0 : p.type == TYPE_BO ? 1 : 2);
  if (p.mode != SAO_NEW) return;
  int off[4], n = 0;
  if (p.type == TYPE_BO) for (int i = 0; i < 4; i++) off[n++] = p.offset[(p.aux + i) % BO_CLASSES];
  else for (int i = 0; i < EO_CLASSES; i++) if (i != CLS_PLAIN) off[n++] = p.offset[i];
  for (int i = 0; i < 4; i++) code_max_uvlc(k, (unsigned)abs(off[i]), (unsigned)c.max_offset);
  if (p.type == TYPE_BO) { for (int i = 0; i < 4; i++) if (off[i] != 0) ep(k, 1); ep(k, 5); }      // signs, sao_band_position
  else if (comp < 2) ep(k, 2);                                                                      // sao_eo_class
}
void code_blk_param(const Ctx& c, Coder& k, const hop_sao_param blk[3], bool left_avail, bool above_avail, bool only_merge) {   // codeSAOBlkParam :2439-2470
  bool left = false, above = false;
  if (left_avail) { left = blk[0].mode == SAO_MERGE && blk[0].type == 0; bin(k, k.merge, left ? 1 : 0); }
  if (above_avail && !left) { above = blk[0].mode == SAO_MERGE && blk[0].type == 1; bin(k, k.merge, above ? 1 : 0); }
  if (only_merge) return;
  if (!left && !above) for (int comp = 0; comp < 3; comp++) code_offset_param(c, k, comp, blk[comp]);
}

inline int64_t est_dist(int64_t count, int64_t offset, int64_t diff_sum, int shift) { return (count * offset * offset - diff_sum * offset * 2) >> shift; }
int64_t distortion(const Ctx& c, int type, int aux, const int* inv_offset, const Stat& s) {        // getDistortion :381-420
  const int shift = 2 * (c.bit_depth - 8);                                                         // 2 * DISTORTION_PRECISION_ADJUSTMENT(bit depth - 8), FULL_NBIT 0 (TypeDef.h:162-167)
  int64_t d = 0;
  if (type == TYPE_BO) for (int i = aux; i < aux + 4; i++) { const int b = i % BO_CLASSES; d += est_dist(s.count(b), inv_offset[b], s.diff(b), shift); }
  else for (int i = 0; i < EO_CLASSES; i++) d += est_dist(s.count(i), inv_offset[i], s.diff(i), shift);
  return d;
}
int iter_offset(int type, double lambda, int offset_in, int64_t count, int64_t diff_sum, int shift, int bit_increase, int64_t& best_dist, double& best_cost, int offset_th) {   // estIterOffset :427-457
  int it = offset_in, out = 0;
  double min_cost = lambda;
  while (it != 0) {
    int64_t rate = type == TYPE_BO ? abs(it) + 2 : abs(it) + 1;
    if (abs(it) == offset_th) rate--;
    const int64_t dist = est_dist(count, (int64_t)(it << bit_increase), diff_sum, shift);
    const double cost = (double)dist + lambda * (double)rate;
    if (cost < min_cost) { min_cost = cost; out = it; best_dist = dist; best_cost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
double round_ibdi(int bit_depth, double x) {                                                      // xRoundIbdi / xRoundIbdi2 :51-59
  if (bit_depth > 8) return x > 0 ? (double)(int)(((int)x + (1 << (bit_depth - 8 - 1))) / (1 << (bit_depth - 8))) : (double)(int)(((int)x - (1 << (bit_depth - 8 - 1))) / (1 << (bit_depth - 8)));
  return x >= 0 ? (double)(int)(x + 0.5) : (double)(int)(x - 0.5);
}
void derive_offsets(const Ctx& c, int comp, int type, const Stat& s, int* q, int& aux) {           // deriveOffsets :460-562
  const int shift = 2 * (c.bit_depth - 8), th = c.max_offset;
  memset(q, 0, sizeof(int) * 32);
  const int n = type == TYPE_BO ? BO_CLASSES : EO_CLASSES;
  for (int k = 0; k < n; k++) {
    if (type != TYPE_BO && k == CLS_PLAIN) continue;
    if (s.count(k) == 0) continue;
    int v = (int)round_ibdi(c.bit_depth, (double)(s.diff(k) << (c.bit_depth - 8)) / (double)(s.count(k) << c.step_log2));
    q[k] = v < -th ? -th : v > th ? th : v;
  }
  if (type != TYPE_BO) {
    int64_t d; double cost;
    for (int k = 0; k < EO_CLASSES; k++) {
      if (k < CLS_PLAIN && q[k] < 0) q[k] = 0;                                                     // valleys take offsets >= 0, peaks <= 0
      if (k > CLS_PLAIN && q[k] > 0) q[k] = 0;
      if (q[k] != 0) q[k] = iter_offset(type, c.lambda[comp], q[k], s.count(k), s.diff(k), shift, c.step_log2, d, cost, th);
    }
    aux = 0;
    return;
  }
  int64_t dist[BO_CLASSES]; double cost[BO_CLASSES];
  memset(dist, 0, sizeof(dist));
  for (int k = 0; k < BO_CLASSES; k++) {
    cost[k] = c.lambda[comp];
    if (q[k] != 0) q[k] = iter_offset(type, c.lambda[comp], q[k], s.count(k), s.diff(k), shift, c.step_log2, dist[k], cost[k], th);
  }
  double min_cost = MAX_COST;
  for (int band = 0; band < BO_CLASSES - 4 + 1; band++) {
    double v = cost[band]; v += cost[band + 1]; v += cost[band + 2]; v += cost[band + 3];
    if (v < min_cost) { min_cost = v; aux = band; }
  }
  int keep[BO_CLASSES]; memset(keep, 0, sizeof(keep));
  for (int i = 0; i < 4; i++) { const int b = (aux + i) % BO_CLASSES; keep[b] = q[b]; }
  memcpy(q, keep, sizeof(keep));
}
void invert_quant(const Ctx& c, int type, int aux, int* dst, const int* src) {                     // invertQuantOffsets (TComSampleAdaptiveOffset.cpp:224-247)
  int coded[32]; memcpy(coded, src, sizeof(coded)); memset(dst, 0, sizeof(int) * 32);
  if (type == TYPE_BO) for (int i = 0; i < 4; i++) { const int b = (aux + i) % BO_CLASSES; dst[b] = coded[b] * (1 << c.step_log2); }
  else for (int i = 0; i < EO_CLASSES; i++) dst[i] = coded[i] * (1 << c.step_log2);
}
void set_param(hop_sao_param& p, int mode, int type, int aux, const int* off) {
  memset(&p, 0, sizeof(p)); p.mode = (int8_t)mode; p.type = (int8_t)type; p.aux = (int8_t)aux;
  if (off) for (int i = 0; i < 32; i++) p.offset[i] = (int8_t)off[i];
}

// coder labels of the reference: CUR = the CTU's entry state, MID / TEMP inside a mode, NEXT = the winner's exit state
void mode_new(const Ctx& c, int ctu, const hop_sao_param* merge[2], hop_sao_param out[3], double& norm_cost, Coder& goon, const Coder& cur, Coder& temp) {   // deriveModeNewRDO :565-704
  int64_t dist[3], mode_dist[3] = { 0, 0, 0 };
  hop_sao_param test[3];
  int inv[32], q[32], aux = 0;
  Coder mid;
  set_param(out[0], SAO_OFF, 0, 0, NULL);
  goon = cur; code_blk_param(c, goon, out, merge[0] != NULL, merge[1] != NULL, true); mid = goon;
  // luma
  set_param(out[0], SAO_OFF, 0, 0, NULL);
  reset_bits(goon); code_offset_param(c, goon, 0, out[0]);
  double min_cost = c.lambda[0] * (double)written_bits(goon);
  temp = goon;
  if (c.enabled[0]) for (int type = 0; type < N_TYPES; type++) {
    derive_offsets(c, 0, type, c.stat(ctu, 0, type), q, aux);
    set_param(test[0], SAO_NEW, type, aux, q);
    invert_quant(c, type, aux, inv, q);
    dist[0] = distortion(c, type, aux, inv, c.stat(ctu, 0, type));
    goon = mid; reset_bits(goon); code_offset_param(c, goon, 0, test[0]);
    const double cost = (double)dist[0] + c.lambda[0] * (double)(int)written_bits(goon);
    if (cost < min_cost) { min_cost = cost; mode_dist[0] = dist[0]; out[0] = test[0]; temp = goon; }
  }
  goon = temp; mid = goon;
  // chroma, both planes with one type
  double cost = 0; uint32_t prev = 0;
  reset_bits(goon);
  for (int comp = 1; comp < 3; comp++) {
    set_param(out[comp], SAO_OFF, 0, 0, NULL); mode_dist[comp] = 0;
    code_offset_param(c, goon, comp, out[comp]);
    const uint32_t now = written_bits(goon);
    cost += c.lambda[comp] * (now - prev); prev = now;
  }
  min_cost = cost;
  for (int type = 0; type < N_TYPES; type++) {
    goon = mid; reset_bits(goon); prev = 0; cost = 0;
    for (int comp = 1; comp < 3; comp++) {
      if (!c.enabled[comp]) { set_param(test[comp], SAO_OFF, 0, 0, NULL); dist[comp] = 0; continue; }
      derive_offsets(c, comp, type, c.stat(ctu, comp, type), q, aux);
      set_param(test[comp], SAO_NEW, type, aux, q);
      invert_quant(c, type, aux, inv, q);
      dist[comp] = distortion(c, type, aux, inv, c.stat(ctu, comp, type));
      code_offset_param(c, goon, comp, test[comp]);
      const uint32_t now = written_bits(goon);
      cost += dist[comp] + (c.lambda[comp] * (now - prev)); prev = now;
    }
    if (cost < min_cost) { min_cost = cost; for (int comp = 1; comp < 3; comp++) { mode_dist[comp] = dist[comp]; out[comp] = test[comp]; } }
  }
  norm_cost = 0;
  for (int comp = 0; comp < 3; comp++) norm_cost += (double)mode_dist[comp] / c.lambda[comp];
  goon = cur; reset_bits(goon); code_blk_param(c, goon, out, merge[0] != NULL, merge[1] != NULL, false);
  norm_cost += (double)written_bits(goon);
}
void mode_merge(const Ctx& c, int ctu, const hop_sao_param* merge[2], hop_sao_param out[3], double& norm_cost, Coder& goon, const Coder& cur, Coder& temp) {   // deriveModeMergeRDO :706-752
  norm_cost = MAX_COST;
  for (int mt = 0; mt < 2; mt++) {
    if (!merge[mt]) continue;
    hop_sao_param test[3];
    double norm_dist = 0;
    for (int comp = 0; comp < 3; comp++) {
      const hop_sao_param& m = merge[mt][comp];
      test[comp] = m; test[comp].mode = SAO_MERGE; test[comp].type = (int8_t)mt;
      if (m.mode != SAO_OFF) { int off[32]; for (int i = 0; i < 32; i++) off[i] = m.offset[i]; norm_dist += (double)distortion(c, m.type, m.aux, off, c.stat(ctu, comp, m.type)) / c.lambda[comp]; }
    }
    goon = cur; reset_bits(goon); code_blk_param(c, goon, test, merge[0] != NULL, merge[1] != NULL, false);
    const double cost = norm_dist + (double)(int)written_bits(goon);
    if (cost < norm_cost) { norm_cost = cost; for (int comp = 0; comp < 3; comp++) out[comp] = test[comp]; temp = goon; }
  }
  goon = temp;
}
}  // namespace

extern "C" int hop_sao_decide(int n_ctu, int ctus_per_row, int bit_depth, const int32_t* stats, const hop_sao_params* p, hop_sao_param* coded, hop_sao_param* recon) {
  if (n_ctu <= 0 || ctus_per_row <= 0 || !stats || !p || !coded || !recon || bit_depth < 8 || bit_depth > 12 || p->slice_type < 0 || p->slice_type > 4) return HOP_ERR_ARG;
  for (int k = 0; k < 3; k++) if (!(p->lambda[k] > 0)) return HOP_ERR_ARG;
  static const uint8_t init_merge[5] = { 153, 153, 153, 153, 153 }, init_type[5] = { 160, 185, 200, 185, 185 };   // INIT_SAO_MERGE_FLAG / INIT_SAO_TYPE_IDX (ContextTables.h:488-519), rows B, P, I, ISS, PSS
  Ctx c;
  c.bit_depth = bit_depth; c.max_offset = (1 << ((bit_depth < 10 ? bit_depth : 10) - 5)) - 1; c.step_log2 = bit_depth > 10 ? bit_depth - 10 : 0; c.stats = stats;
  for (int k = 0; k < 3; k++) { c.enabled[k] = p->enabled[k] ? 1 : 0; c.lambda[k] = p->lambda[k]; }
  memset(coded, 0, sizeof(hop_sao_param) * 3 * (size_t)n_ctu); memset(recon, 0, sizeof(hop_sao_param) * 3 * (size_t)n_ctu);
  if (!c.enabled[0] && !c.enabled[1] && !c.enabled[2]) return HOP_OK;
  Coder goon; goon.merge = init_state(init_merge[p->slice_type], p->qp); goon.type = init_state(init_type[p->slice_type], p->qp); goon.frac = p->rd_fraction & 32767;   // initRDOCabacCoder :239-247
  for (int ctu = 0; ctu < n_ctu; ctu++) {
    const Coder cur = goon;
    Coder next = goon, temp = goon;
    const hop_sao_param* merge[2] = { ctu % ctus_per_row ? &recon[(size_t)(ctu - 1) * 3] : NULL, ctu >= ctus_per_row ? &recon[(size_t)(ctu - ctus_per_row) * 3] : NULL };   // left, above
    double min_cost = MAX_COST, cost;
    hop_sao_param mode[3];
    mode_new(c, ctu, merge, mode, cost, goon, cur, temp);
    if (cost < min_cost) { min_cost = cost; memcpy(&coded[(size_t)ctu * 3], mode, sizeof(mode)); next = goon; }
    temp = goon;                                                            // (BLK_TEMP keeps what the last mode left in it when no merge candidate exists)
    mode_merge(c, ctu, merge, mode, cost, goon, cur, temp);
    if (cost < min_cost) { min_cost = cost; memcpy(&coded[(size_t)ctu * 3], mode, sizeof(mode)); next = goon; }
    goon = next;
    // reconstructBlkSAOParam (TComSampleAdaptiveOffset.cpp:305-339)
    for (int comp = 0; comp < 3; comp++) {
      hop_sao_param& r = recon[(size_t)ctu * 3 + comp]; r = coded[(size_t)ctu * 3 + comp];
      if (r.mode == SAO_NEW) { int src[32], dst[32]; for (int i = 0; i < 32; i++) src[i] = r.offset[i]; invert_quant(c, r.type, r.aux, dst, src); for (int i = 0; i < 32; i++) r.offset[i] = (int8_t)dst[i]; }
      else if (r.mode == SAO_MERGE) r = merge[r.type][comp];
    }
  }
  return HOP_OK;
}
