// hop_search.cpp -- see hop_search.h.  Forwards to the C ABI; holds no algorithm of its own beyond the
// partition geometry tables of TComDataCU (cited inline).
#include "hop_search.h"

namespace hop {

int numPartitions(PartSize ps) { return ps == SIZE_2Nx2N ? 1 : ps == SIZE_NxN ? 4 : 2; }

void getPartGeometry(PartSize ps, int S, int idx, int& x, int& y, int& w, int& h, int& offX, int& offY) {
  x = y = offX = offY = 0; w = h = S;
  switch (ps) {                                   // sizes: TComDataCU::getPartIndexAndSize; offsets: getPartOffset, TComDataCU.cpp:2263-2295
    case SIZE_2NxN:  h = S / 2; if (idx) { y = S / 2; offY = S / 2; } break;
    case SIZE_Nx2N:  w = S / 2; if (idx) { x = S / 2; offX = S / 2; } break;
    case SIZE_NxN:   w = h = S / 2; if (idx & 1) { x = S / 2; offX = S / 2; } if (idx & 2) { y = S / 2; offY = S / 2; } break;
    case SIZE_2NxnU: if (!idx) h = S / 4; else { y = S / 4; h = 3 * S / 4; offY = S / 4; } break;
    case SIZE_2NxnD: if (!idx) h = 3 * S / 4; else { y = 3 * S / 4; h = S / 4; offY = S / 4 + S / 2; } break;
    case SIZE_nLx2N: if (!idx) w = S / 4; else { x = S / 4; w = 3 * S / 4; offX = S / 4; } offY = S; break;   // riOffsetY = getHeight(0), :2285
    case SIZE_nRx2N: if (!idx) w = 3 * S / 4; else { x = 3 * S / 4; w = S / 4; offX = S / 4 + S / 2; } break;
    default: break;
  }
}

Search::Search(int picW, int picH, int bitDepth, int searchRange, bool useFastEnc, bool useHadME, int device)
    : m_picW(picW), m_picH(picH), m_searchRange(searchRange), m_flags((useFastEnc ? HOP_FLAG_FEN : 0) | (useHadME ? HOP_FLAG_HADME : 0)) {
  if (hop_ctx_create(&m_ctx, picW, picH, bitDepth, bitDepth, device) != HOP_OK) { m_err = hop_last_error(nullptr); m_ctx = nullptr; }
}
Search::~Search() { hop_ctx_destroy(m_ctx); }

#define HOP_TRY(call) do { if ((call) != HOP_OK) { m_err = hop_last_error(m_ctx); return false; } } while (0)

bool Search::setOriginal(const int16_t* y, int sy, const int16_t* cb, const int16_t* cr, int sc) { HOP_TRY(hop_upload_orig(m_ctx, y, sy, cb, cr, sc)); return true; }
bool Search::resetSSRef() { HOP_TRY(hop_ssref_reset(m_ctx)); HOP_TRY(hop_sync(m_ctx)); return true; }
bool Search::copyYuv2SSRef(int x, int y, int size, const int16_t* recY, const int16_t* recCb, const int16_t* recCr) {
  const int32_t r[4] = { x, y, size, 0 };
  HOP_TRY(hop_ssref_commit_cus(m_ctx, 1, r, recY, recCb, recCr));
  return true;
}

bool Search::motionEstimation(const CuPos& cu, PartSize ps, const Mv* mvPred, const Mv (*amvp)[2], const int* nAmvp,
                              bool useGT, uint32_t bitsIn, std::vector<MotionResult>& out) {
  const int n = numPartitions(ps), wctu = (m_picW + 63) / 64;
  std::vector<hop_pu_job> jobs(n);
  std::vector<hop_pu_result> res(n);
  for (int i = 0; i < n; i++) {
    int x, y, w, h, offX, offY; getPartGeometry(ps, cu.cuSize, i, x, y, w, h, offX, offY);
    hop_pu_job& j = jobs[i];
    j.pu_x = cu.cuX + x; j.pu_y = cu.cuY + y; j.w = w; j.h = h;
    int r[6];                                     // xSetSearchRange x2, TEncSearch.cpp:4555,4577
    hop_set_search_range(m_picW, m_picH, cu.cuX, cu.cuY, cu.cuSize, cu.ctuAddr, wctu, mvPred[i].hor, mvPred[i].ver, m_searchRange,
                         offX, offY, cu.cuY == 0, cu.cuX == 0, r);
    j.rng_left = r[0]; j.rng_right = r[1]; j.rng_top = r[2]; j.rng_bottom = r[3]; j.off_x = r[4]; j.off_y = r[5];
    j.pred_x = mvPred[i].hor; j.pred_y = mvPred[i].ver; j.lambda_cost = m_rd.lambdaMotionSAD();
    j.n_amvp = nAmvp[i];
    for (int k = 0; k < 2; k++) { j.amvp[2 * k] = k < nAmvp[i] ? amvp[i][k].hor : 0; j.amvp[2 * k + 1] = k < nAmvp[i] ? amvp[i][k].ver : 0; }
    j.flags = m_flags;
  }
  const int stage = useGT ? HOP_STAGE_GT : HOP_STAGE_FRAC;
  HOP_TRY(hop_me_search(m_ctx, n, jobs.data(), res.data(), stage));
  out.assign(n, MotionResult());
  for (int i = 0; i < n; i++) {
    MotionResult& m = out[i];
    m.notValCU = res[i].not_valid != 0;           // caller sets getTotalCost() = MAX_DOUBLE, TEncCu.cpp:1418-1422
    if (m.notValCU) continue;
    int mvq[2]; uint32_t bits, cost;
    hop_me_finish(&jobs[i], &res[i], stage, bitsIn, mvq, &bits, &cost);
    m.mv = Mv(mvq[0], mvq[1]); m.bits = bits; m.cost = cost; m.gtFlag = res[i].gt_flag != 0;
    for (int k = 0; k < 4; k++) m.gt[k] = Mv(res[i].gt[2 * k], res[i].gt[2 * k + 1]);
  }
  return true;
}

bool Search::motionCompensation(const CuPos& cu, PartSize ps, const MotionResult* res, bool useGT,
                                std::vector<int16_t>& predY, std::vector<int16_t>& predCb, std::vector<int16_t>& predCr) {
  const int n = numPartitions(ps);
  std::vector<hop_pred_job> jobs(n);
  size_t tot = 0;
  for (int i = 0; i < n; i++) {
    int x, y, w, h, offX, offY; getPartGeometry(ps, cu.cuSize, i, x, y, w, h, offX, offY);
    hop_pred_job& j = jobs[i];
    j.pu_x = cu.cuX + x; j.pu_y = cu.cuY + y; j.w = w; j.h = h; j.mv_x = res[i].mv.hor; j.mv_y = res[i].mv.ver; j.use_gt = useGT ? 1 : 0;
    for (int k = 0; k < 4; k++) { j.gt[2 * k] = res[i].gt[k].hor; j.gt[2 * k + 1] = res[i].gt[k].ver; }
    tot += (size_t)w * h;
  }
  predY.resize(tot); predCb.resize(tot / 4); predCr.resize(tot / 4);
  HOP_TRY(hop_pred_inter(m_ctx, n, jobs.data(), predY.data(), predCb.data(), predCr.data()));
  return true;
}

}  // namespace hop
