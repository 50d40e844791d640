// hop_search.h -- host-side C++ mirror of the reference's call surface for the HOP hot path.
//
// The reference (HM-15.0 fork) drives this path through TEncSearch / TComPrediction / TComRdCost members
// (SURVEY.md section 8(b)).  These classes keep the reference's names, argument meaning and "error" behaviour
// (sentinels, not exceptions) for that path so that a maintainer can forward the reference's members to them
// one to one and so that tests read like the reference's call sites.  They own no compute:
// everything forwards to the C ABI of include/hophip.h (libhophip.so, HIP kernels).  C++11, no dependencies.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/hophip.h"

namespace hop {

// TComMv (TLibCommon/TComMv.h): two Shorts
struct Mv { int16_t hor, ver; Mv(int h = 0, int v = 0) : hor((int16_t)h), ver((int16_t)v) {} };

// PartSize (TLibCommon/TypeDef.h) restricted to the inter shapes the ISS slice tests
enum PartSize { SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN, SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N };

// The CU-dependent scalars the reference reads from TComDataCU on this path
struct CuPos { int cuX, cuY, cuSize, ctuAddr; };

// TComDataCU::getPartIndexAndSize + getPartOffset (TLibCommon/TComDataCU.cpp:2251-2296): PU rectangle inside the
// CU and the (riOffsetX, riOffsetY) pair handed to xSetSearchRange, including the SIZE_nLx2N offY quirk.
void getPartGeometry(PartSize ps, int cuSize, int partIdx, int& x, int& y, int& w, int& h, int& offX, int& offY);
int  numPartitions(PartSize ps);

// TComRdCost, the members the path uses (TLibCommon/TComRdCost.h:145-215, TComRdCost.cpp:167-173)
class RdCost {
 public:
  void     setLambda(double lambda) { m_dLambda = lambda; m_sqrtLambda = std::sqrt(lambda); m_uiLambdaMotionSAD = (uint32_t)std::floor(65536.0 * m_sqrtLambda); }
  double   getLambda() const { return m_dLambda; }
  uint32_t lambdaMotionSAD() const { return m_uiLambdaMotionSAD; }
  static uint32_t xGetComponentBits(int v) { return hop_component_bits(v); }
  static uint32_t getBitsGT(const int gt[8]) { return hop_bits_gt(gt); }
  uint32_t getCost(uint32_t bits) const { return (m_uiLambdaMotionSAD * bits) >> 16; }
 private:
  double m_dLambda = 0, m_sqrtLambda = 0; uint32_t m_uiLambdaMotionSAD = 0;
};

// What TEncSearch::xMotionEstimation returns through its reference parameters (TEncSearch.cpp:4479-4500)
struct MotionResult {
  Mv       mv;            // rcMv, quarter-pel
  uint32_t bits = 0;      // ruiBits
  uint32_t cost = 0;      // ruiCost
  bool     notValCU = false;   // bNotValCU (TEncSearch.cpp:4603-4611)
  bool     gtFlag = false;
  Mv       gt[4];         // rcGT0..rcGT3
};

// TEncSearch for the ISS motion path + TComPrediction::motionCompensation + TEncCu::xCopyYuv2SSRef.
// One instance per encoder, strictly single-threaded like the reference; every call is synchronous.
class Search {
 public:
  // TEncTop::create/init wiring (TEncTop.cpp:89-101, 299-310): picture size, bit depth, SearchRange, FEN, HadamardME
  Search(int picW, int picH, int bitDepth, int searchRange, bool useFastEnc, bool useHadME, int device = 0);
  ~Search();
  bool ok() const { return m_ctx != nullptr; }
  const std::string& error() const { return m_err; }
  RdCost& rdCost() { return m_rd; }

  // TEncTop::encode's copy of the original into the picture (TEncTop.cpp:363-368)
  bool setOriginal(const int16_t* y, int strideY, const int16_t* cb, const int16_t* cr, int strideC);
  // TComSlice::setRefPicList for an ISS slice: SS reference := sentinel (TComSlice.cpp:366-378)
  bool resetSSRef();
  // TEncCu::xCopyYuv2SSRef (TEncCu.cpp:1677-1715) for one finalised CU (rec* contiguous size^2 / (size/2)^2)
  bool copyYuv2SSRef(int x, int y, int size, const int16_t* recY, const int16_t* recCb, const int16_t* recCr);

  // TEncSearch::xMotionEstimation for all PUs of one partitioning of one CU (the loop of predInterSearch,
  // TEncSearch.cpp:3199-3420): mvPred[i] / amvp per PU as xEstimateMvPredAMVP delivers them.
  // bitsIn = ruiBits on entry.  Returns false on a device/argument error (error() has the text).
  bool motionEstimation(const CuPos& cu, PartSize ps, const Mv* mvPred, const Mv (*amvp)[2], const int* nAmvp,
                        bool useGT, uint32_t bitsIn, std::vector<MotionResult>& out);

  // TComPrediction::motionCompensation (TComPrediction.cpp:419-528) for the PUs of one CU: predictions are
  // returned packed PU after PU (luma w*h, chroma (w/2)*(h/2) each)
  bool motionCompensation(const CuPos& cu, PartSize ps, const MotionResult* res, bool useGT,
                          std::vector<int16_t>& predY, std::vector<int16_t>& predCb, std::vector<int16_t>& predCr);
 private:
  hop_ctx* m_ctx = nullptr;
  int m_picW, m_picH, m_searchRange, m_flags;
  RdCost m_rd;
  std::string m_err;
};

}  // namespace hop
