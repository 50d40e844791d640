// k_intra.hip -- the distortion half of the 35-mode intra rough search.
// Replaces, for one luma block of TEncSearch::estIntraPredQT (TLibEncoder/TEncSearch.cpp:2430-2461):
//   TComPattern::initAdiPattern  (TLibCommon/TComPattern.cpp:179-313): fillReferenceSamples (:374-558) + the
//     [1 2 1] / strong-32 reference smoothing (:237-299),
//   TComPattern::getPredictorPtr (:583-607, filter decision table :49-56),
//   TComPrediction::predIntraLumaAng (TLibCommon/TComPrediction.cpp:340-372): xPredIntraPlanar (:1468-1505),
//     predIntraGetPredValDC (:130-167) + xDCPredFiltering (:1521-1541), xPredIntraAng (:192-338),
//   TComRdCost::calcHAD (TLibCommon/TComRdCost.cpp:391-425)
// for all 35 modes; the caller adds xModeBitsIntra * sqrt(lambda) (:2460-2461) on the host.
// Neighbour samples come from the context's reconstruction picture; availability is given per 4-sample unit in
// the reference's bNeighborFlags order (it depends on the CU structure and coding order, which the caller owns).
//
// One workgroup per block.  The reference line (4N+1 samples) and its smoothed copy live in LDS; every predicted
// sample is a closed form of the line (planar and DC included), so a work item is (mode, 8x8 block) = one wave:
// lane = sample, Hadamard across the wave, one LDS atomic per (mode, block).  No prediction buffer, no barriers
// between modes.
#include "hop_dev.h"
#include "k_intra_dev.inl"

__global__ __launch_bounds__(256) void k_intra_rough(const hop_intra_job* __restrict__ jobs, hop_pics pic, const int16_t* __restrict__ rec_y,
                                                     uint32_t* __restrict__ satd_out) {
  __shared__ IntraShared sh;
  intra_rough_body(sh, jobs + blockIdx.x, pic, rec_y, satd_out + (size_t)blockIdx.x * 35);
}

// the prediction of ONE mode per block, written into the context's prediction picture (the first step of
// TEncSearch::xIntraCodingLumaBlk, TLibEncoder/TEncSearch.cpp:1046-1049: initAdiPattern + predIntraLumaAng)
__global__ __launch_bounds__(256) void k_intra_pred(const hop_intra_job* __restrict__ jobs, const int32_t* __restrict__ modes, hop_pics pic,
                                                    const int16_t* __restrict__ rec_y) {
  __shared__ IntraShared sh;
  intra_pred_body(sh, jobs + blockIdx.x, modes[blockIdx.x], pic, rec_y);
}

// the chroma prediction of xIntraCodingChromaBlk (TLibEncoder/TEncSearch.cpp:1200-1215: initAdiPatternChroma + predIntraChromaAng) for both planes: block i of
// size jobs[i].size at the chroma position (x / 2, y / 2), availability per 2-sample unit, mode modes[i] (the caller resolves DM_CHROMA_IDX to the luma mode)
__global__ __launch_bounds__(256) void k_intra_pred_chroma(const hop_intra_job* __restrict__ jobs, const int32_t* __restrict__ modes, hop_pics pic,
                                                           const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr) {
  __shared__ IntraShared sh;
  const int bi = blockIdx.x >> 1;
  intra_pred_chroma_body(sh, jobs + bi, modes[bi], 1 + (blockIdx.x & 1), pic, rec_cb, rec_cr);
}

int hop_launch_intra_pred_chroma(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_pred_chroma, dim3(2 * n), dim3(256), 0, c->stream, d_jobs, d_modes, hop_make_pics(c), c->rec[1], c->rec[2]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_pred_chroma launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

int hop_launch_intra_pred(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_pred, dim3(n), dim3(256), 0, c->stream, d_jobs, d_modes, hop_make_pics(c), c->rec[0]);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_pred launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

int hop_launch_intra(hop_ctx* c, int n, const hop_intra_job* d_jobs, uint32_t* d_satd) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, (uint64_t)n);
  hipLaunchKernelGGL(k_intra_rough, dim3(n), dim3(256), 0, c->stream, d_jobs, hop_make_pics(c), c->rec[0], d_satd);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_rough launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
