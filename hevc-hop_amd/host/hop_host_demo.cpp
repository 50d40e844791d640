// hop_host_demo.cpp -- smallest C++ caller of the host mirror (needs an MI355X to run): one CU through
// motionEstimation -> motionCompensation -> copyYuv2SSRef, the sequence TEncCu::xCheckRDCostInter /
// xCompressCU performs (TEncCu.cpp:1399-1448, :869-880).  Also the link check for libhophip.so from C++.
#include <cstdio>
#include <vector>
#include "hop_search.h"
int main() {
  const int W = 256, H = 128;
  hop::Search s(W, H, 8, 128, true, true);
  if (!s.ok()) { fprintf(stderr, "no context: %s\n", s.error().c_str()); return 2; }
  std::vector<int16_t> y(W * H), c(W * H / 4, 128);
  for (int i = 0; i < W * H; i++) y[i] = (int16_t)((((i % W) % 15) * 13 + ((i / W) % 15) * 7) & 255);   // 15-px periodic "lenslet"
  s.setOriginal(y.data(), W, c.data(), c.data(), W / 2);
  s.resetSSRef();
  s.rdCost().setLambda(57.908);
  std::vector<int16_t> rec(64 * 64), recc(32 * 32, 128);
  for (int cu = 0; cu < 3; cu++) {                      // code the first CTU row's first three CTUs as "already reconstructed"
    for (int r = 0; r < 64; r++) for (int q = 0; q < 64; q++) rec[r * 64 + q] = y[r * W + cu * 64 + q];
    s.copyYuv2SSRef(cu * 64, 0, 64, rec.data(), recc.data(), recc.data());
  }
  hop::CuPos cu = { 192, 0, 32, 3 };
  hop::Mv pred[2] = { hop::Mv(-60, 0), hop::Mv(-60, 0) };
  hop::Mv amvp[2][2] = { { hop::Mv(-60, 0), hop::Mv(0, 0) }, { hop::Mv(-60, 0), hop::Mv(0, 0) } };
  int nA[2] = { 1, 1 };
  std::vector<hop::MotionResult> res;
  if (!s.motionEstimation(cu, hop::SIZE_Nx2N, pred, amvp, nA, true, 0, res)) { fprintf(stderr, "%s\n", s.error().c_str()); return 1; }
  for (size_t i = 0; i < res.size(); i++)
    printf("PU %zu: notVal=%d mv=(%d,%d) bits=%u cost=%u gt=%d\n", i, res[i].notValCU, res[i].mv.hor, res[i].mv.ver, res[i].bits, res[i].cost, res[i].gtFlag);
  std::vector<int16_t> py, pcb, pcr;
  if (!s.motionCompensation(cu, hop::SIZE_Nx2N, res.data(), true, py, pcb, pcr)) { fprintf(stderr, "%s\n", s.error().c_str()); return 1; }
  printf("pred[0..3] = %d %d %d %d\n", py[0], py[1], py[2], py[3]);
  return 0;
}
