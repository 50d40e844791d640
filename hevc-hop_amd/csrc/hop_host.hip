// hop_host.hip -- CTU-level host logic and the profiling facility of libhophip (no kernels here).
#include <cstdlib>
#include <cstring>
#include "hop_dev.h"

// ---------------------------------------------------------------------------------------------
// profiling: hipEvent pairs on the context stream around each kernel launch
// ---------------------------------------------------------------------------------------------
int hop_prof_begin(hop_ctx* c, int kernel, uint64_t units) {
  if (!c->prof_on) return -1;
  if (c->prof_n == c->prof_cap) {
    int ncap = c->prof_cap ? c->prof_cap * 2 : 256;
    hop_prof_rec* nr = (hop_prof_rec*)realloc(c->prof_recs, (size_t)ncap * sizeof(hop_prof_rec));
    if (!nr) return -1;
    for (int i = c->prof_cap; i < ncap; i++) { nr[i].a = nullptr; nr[i].b = nullptr; }
    c->prof_recs = nr; c->prof_cap = ncap;
  }
  hop_prof_rec& r = c->prof_recs[c->prof_n];
  if (!r.a && (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)) return -1;
  r.kernel = kernel; r.units = units;
  (void)hipEventRecord(r.a, c->stream);
  return c->prof_n++;
}
void hop_prof_end(hop_ctx* c, int rec) {
  if (rec >= 0) (void)hipEventRecord(c->prof_recs[rec].b, c->stream);
}
static int prof_fold(hop_ctx* c) {
  if (hipStreamSynchronize(c->stream) != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "profile: stream sync failed");
  for (int i = 0; i < c->prof_n; i++) {
    hop_prof_rec& r = c->prof_recs[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_ms[r.kernel] += ms; c->prof_launches[r.kernel] += 1; c->prof_units[r.kernel] += r.units;
    }
  }
  c->prof_n = 0;
  return HOP_OK;
}

extern "C" {

int hop_profile_enable(hop_ctx* c, int on) {
  if (!c) return HOP_ERR_ARG;
  int r = prof_fold(c); if (r) return r;
  c->prof_on = on != 0;
  return HOP_OK;
}
int hop_profile_reset(hop_ctx* c) {
  if (!c) return HOP_ERR_ARG;
  int r = prof_fold(c); if (r) return r;
  memset(c->prof_ms, 0, sizeof(c->prof_ms)); memset(c->prof_launches, 0, sizeof(c->prof_launches)); memset(c->prof_units, 0, sizeof(c->prof_units));
  return HOP_OK;
}
int hop_profile_read(hop_ctx* c, int kernel, uint64_t* launches, double* total_ms, uint64_t* units) {
  if (!c || kernel < 0 || kernel >= HOP_K_COUNT) return hop_set_err(c, HOP_ERR_ARG, "hop_profile_read: bad kernel id");
  int r = prof_fold(c); if (r) return r;
  if (launches) *launches = c->prof_launches[kernel];
  if (total_ms) *total_ms = c->prof_ms[kernel];
  if (units) *units = c->prof_units[kernel];
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------
// PU enumeration of one CTU (see include/hophip.h)
// ---------------------------------------------------------------------------------------------
struct EnumState {
  int pic_w, pic_h, ctu_addr, wctu, search_range, n_amvp, flags, with_amp;
  int pred[2], amvp[4];
  uint32_t lambda_cost;
  hop_pu_job* out; int32_t* cu_out; int max_out, n;
};

static void emit(EnumState& s, int cu_x, int cu_y, int cu_s, int px, int py, int w, int h, int off_x, int off_y, int cu_tag) {
  if (s.n >= s.max_out) { s.n++; return; }
  hop_pu_job& j = s.out[s.n];
  memset(&j, 0, sizeof(j));
  j.pu_x = cu_x + px; j.pu_y = cu_y + py; j.w = w; j.h = h;
  int r[6];
  // isFirstRow / isFirstCol of getPartOffset: the CU touches the picture's top / left edge (TComDataCU.cpp:2258-2262)
  hop_set_search_range(s.pic_w, s.pic_h, cu_x, cu_y, cu_s, s.ctu_addr, s.wctu, s.pred[0], s.pred[1], s.search_range,
                       off_x, off_y, cu_y == 0, cu_x == 0, r);
  j.rng_left = r[0]; j.rng_right = r[1]; j.rng_top = r[2]; j.rng_bottom = r[3]; j.off_x = r[4]; j.off_y = r[5];
  j.pred_x = s.pred[0]; j.pred_y = s.pred[1]; j.lambda_cost = s.lambda_cost; j.n_amvp = s.n_amvp;
  for (int k = 0; k < 4; k++) j.amvp[k] = s.amvp[k];
  j.flags = s.flags;
  if (s.cu_out) s.cu_out[s.n] = cu_tag;
  s.n++;
}

static void enum_cu(EnumState& s, int x, int y, int size, int depth, int zidx) {
  const bool boundary = (x + size > s.pic_w) || (y + size > s.pic_h);     // TEncCu.cpp:407-409
  if (!boundary) {
    const int S = size, tag = (depth << 16) | zidx;
    emit(s, x, y, S, 0, 0, S, S, 0, 0, tag);                              // SIZE_2Nx2N   :466
    // SIZE_NxN only at max depth for CUs larger than 8x8 (:497-504): never with CTU 64 / depth 4
    emit(s, x, y, S, 0, 0, S / 2, S, 0, 0, tag);                          // SIZE_Nx2N    :509
    emit(s, x, y, S, S / 2, 0, S / 2, S, S / 2, 0, tag);
    emit(s, x, y, S, 0, 0, S, S / 2, 0, 0, tag);                          // SIZE_2NxN    :518
    emit(s, x, y, S, 0, S / 2, S, S / 2, 0, S / 2, tag);
    if (s.with_amp && S >= 16) {                                          // getAMPAcc(depth), :528
      emit(s, x, y, S, 0, 0, S, S / 4, 0, 0, tag);                        // SIZE_2NxnU   :546
      emit(s, x, y, S, 0, S / 4, S, 3 * S / 4, 0, S / 4, tag);
      emit(s, x, y, S, 0, 0, S, 3 * S / 4, 0, 0, tag);                    // SIZE_2NxnD   :555
      emit(s, x, y, S, 0, 3 * S / 4, S, S / 4, 0, S / 4 + S / 2, tag);
      emit(s, x, y, S, 0, 0, S / 4, S, 0, S, tag);                        // SIZE_nLx2N   :592 (offY = height: TComDataCU.cpp:2283-2286)
      emit(s, x, y, S, S / 4, 0, 3 * S / 4, S, S / 4, S, tag);
      emit(s, x, y, S, 0, 0, 3 * S / 4, S, 0, 0, tag);                    // SIZE_nRx2N   :601
      emit(s, x, y, S, 3 * S / 4, 0, S / 4, S, S / 4 + S / 2, 0, tag);
    }
  }
  if (size > 8) {                                                         // further split, :755-865
    const int h = size / 2;
    for (int q = 0; q < 4; q++) {
      const int sx = x + (q & 1) * h, sy = y + (q >> 1) * h;
      if (sx < s.pic_w && sy < s.pic_h) enum_cu(s, sx, sy, h, depth + 1, zidx * 4 + q);
    }
  }
}

int hop_enumerate_ctu_jobs(int pic_w, int pic_h, int ctu_addr, int search_range, const int pred_qpel[2],
                           int n_amvp, const int amvp_qpel[4], uint32_t lambda_cost, int flags, int with_amp,
                           hop_pu_job* out, int32_t* cu_index_out, int max_out) {
  if (pic_w <= 0 || pic_h <= 0 || (pic_w & 7) || (pic_h & 7) || !pred_qpel || n_amvp < 0 || n_amvp > 2 || (n_amvp && !amvp_qpel) || (max_out && !out)) return HOP_ERR_ARG;
  EnumState s;
  s.pic_w = pic_w; s.pic_h = pic_h; s.ctu_addr = ctu_addr; s.wctu = (pic_w + 63) / 64;
  const int hctu = (pic_h + 63) / 64;
  if (ctu_addr < 0 || ctu_addr >= s.wctu * hctu) return HOP_ERR_ARG;
  s.search_range = search_range; s.n_amvp = n_amvp; s.flags = flags; s.with_amp = with_amp;
  s.pred[0] = pred_qpel[0]; s.pred[1] = pred_qpel[1];
  for (int k = 0; k < 4; k++) s.amvp[k] = (amvp_qpel && k < 2 * n_amvp) ? amvp_qpel[k] : 0;
  s.lambda_cost = lambda_cost; s.out = out; s.cu_out = cu_index_out; s.max_out = max_out; s.n = 0;
  enum_cu(s, (ctu_addr % s.wctu) * 64, (ctu_addr / s.wctu) * 64, 64, 0, 0);
  return s.n;
}

} // extern "C"
