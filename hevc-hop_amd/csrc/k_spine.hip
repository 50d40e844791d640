// k_spine.hip -- what the RD spine (host/hop_spine.cpp) needs on the device besides the candidate kernels, and the spine's product backend.
//   hop_valid_pattern      TComRdCost::isValidPattern (TLibCommon/TComRdCost.cpp:430-443) on the resident SS reference, as TEncCu::xCheckRDCostMerge2Nx2N (:1319),
//                          TEncSearch::xMergeEstimation (TEncSearch.cpp:3076) and xGetTemplateCost (:4447) call it
//   hop_recon_stash        the swap of m_ppcRecoYuvBest / m_ppcRecoYuvTemp in TEncCu::xCheckBestMode (TEncCu.cpp:1572-1575) and xCopyYuv2Pic (:1623-1662): a candidate's
//                          reconstruction put aside / brought back, block copies inside HBM
//   hop_recon_put_device   TComYuv::copyToPicYuv of an intra candidate's pcRecoYuv (TEncCu.cpp:1476 and the picture copy of xCopyYuv2Pic)
//   hop_ssref_commit_recon TEncCu::xCopyYuv2SSRef (:1677-1715) from the resident reconstruction picture
//   hop_encode_frame       TEncSlice::compressSlice's CTU loop (TEncSlice.cpp:1000-1196) = the spine over these kernels
// All byte moving: one 2-byte element per thread-iteration, rows contiguous.
#include <string.h>
#include <chrono>
#include <mutex>
#include <thread>
#include <functional>
#include <string>
#include <vector>
#include "hop_dev.h"
#include "../host/hop_spine.h"

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hop_set_err((c), HOP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define STASH_SAMPLES 6144          // 64 x 64 luma + 2 x 32 x 32 chroma
#define STASH_SLOTS (16 * hopspine::SPINE_LANES)   // 16 per CTU row in flight

__global__ void k_valid_pattern(const int32_t* __restrict__ q, int n, const int16_t* __restrict__ y00, int stride, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t* p = q + 6 * i;
  const int16_t* lb = y00 + (ptrdiff_t)(p[1] + (p[5] >> 2) + p[3] + 4) * stride + (p[0] + (p[4] >> 2));
  out[i] = (lb[0] != HOP_NOT_VALID && lb[p[2] + 4] != HOP_NOT_VALID) ? 1 : 0;
}

// grid (n, 3): block i of plane blockIdx.y between the reconstruction picture and its stash slot; rect4 = x, y, size, slot
template <bool RESTORE>
__global__ __launch_bounds__(256) void k_recon_stash(const int32_t* __restrict__ rect4, int16_t* __restrict__ ry, int16_t* __restrict__ rcb, int16_t* __restrict__ rcr, int pic_w,
                                                     int16_t* __restrict__ stash) {
  const int i = blockIdx.x, comp = blockIdx.y;
  const int sh = comp ? 1 : 0, x = rect4[4 * i] >> sh, y = rect4[4 * i + 1] >> sh, s = rect4[4 * i + 2] >> sh, pitch = pic_w >> sh;
  int16_t* pic = comp == 0 ? ry : comp == 1 ? rcb : rcr;
  int16_t* st = stash + (size_t)rect4[4 * i + 3] * STASH_SAMPLES + (comp == 0 ? 0 : comp == 1 ? 4096 : 5120);
  for (int k = threadIdx.x; k < s * s; k += blockDim.x) {
    const int r = k / s, c = k - r * s;
    if (RESTORE) pic[(size_t)(y + r) * pitch + x + c] = st[k]; else st[k] = pic[(size_t)(y + r) * pitch + x + c];
  }
}

// ---- the levels of the chosen CUs (what TEncSearch / TEncCu leave in TComDataCU::m_pcTrCoeff*) ----
#define COEF_PER_CTU 6144           // 64 x 64 luma + 2 x 32 x 32 chroma TCoeff
__device__ static inline int spine_zidx(int x4, int y4) { int z = 0; for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1)); return z; }
// where the levels of the CU at (x, y) start in the image of its slot: Y; Cb at + 4096 - 12 * abs ... computed by the callers from abs
__device__ static inline int32_t* spine_coef_at(int32_t* coefpic, int x, int y, int pic_w, int pic_h, int& abs_idx) {
  const int slot = y / pic_h, yr = y - slot * pic_h, wctu = (pic_w + 63) >> 6, hctu = (pic_h + 63) >> 6;
  abs_idx = spine_zidx((x & 63) >> 2, (yr & 63) >> 2);
  return coefpic + ((size_t)slot * wctu * hctu + (size_t)(yr >> 6) * wctu + (x >> 6)) * COEF_PER_CTU;
}
// grid (n): the levels of candidate i (cu^2 luma, cu^2 / 4 Cb, cu^2 / 4 Cr TCoeff at i * 1.5 cu^2; coef == NULL: a candidate without residual) into the image of its slot
__global__ __launch_bounds__(256) void k_coef_put(const hop_rqt_job* __restrict__ jobs, const int32_t* __restrict__ coef, int32_t* __restrict__ coefpic, int pic_w, int pic_h) {
  const int i = blockIdx.x, cu2 = 1 << (2 * jobs[i].log2_cu);
  int abs_idx; int32_t* ctu = spine_coef_at(coefpic, jobs[i].x, jobs[i].y, pic_w, pic_h, abs_idx);
  const int32_t* src = coef ? coef + (size_t)i * (cu2 + (cu2 >> 1)) : nullptr;
  for (int k = threadIdx.x; k < cu2; k += blockDim.x) ctu[16 * abs_idx + k] = src ? src[k] : 0;
  for (int k = threadIdx.x; k < (cu2 >> 2); k += blockDim.x) { ctu[4096 + 4 * abs_idx + k] = src ? src[cu2 + k] : 0; ctu[5120 + 4 * abs_idx + k] = src ? src[cu2 + (cu2 >> 2) + k] : 0; }
}
// grid (n): block i between the image of the slot its y names and its stash slot; rect4 = x, y, size, slot
template <bool RESTORE>
__global__ __launch_bounds__(256) void k_coef_stash(const int32_t* __restrict__ rect4, int32_t* __restrict__ coefpic, int pic_w, int pic_h, int32_t* __restrict__ stash) {
  const int i = blockIdx.x, cu2 = rect4[4 * i + 2] * rect4[4 * i + 2];
  int abs_idx; int32_t* ctu = spine_coef_at(coefpic, rect4[4 * i], rect4[4 * i + 1], pic_w, pic_h, abs_idx);
  int32_t* st = stash + (size_t)rect4[4 * i + 3] * COEF_PER_CTU;
  for (int k = threadIdx.x; k < cu2; k += blockDim.x) { if (RESTORE) ctu[16 * abs_idx + k] = st[k]; else st[k] = ctu[16 * abs_idx + k]; }
  for (int k = threadIdx.x; k < (cu2 >> 2); k += blockDim.x) {
    if (RESTORE) { ctu[4096 + 4 * abs_idx + k] = st[4096 + k]; ctu[5120 + 4 * abs_idx + k] = st[5120 + k]; }
    else { st[4096 + k] = ctu[4096 + 4 * abs_idx + k]; st[5120 + k] = ctu[5120 + 4 * abs_idx + k]; }
  }
}

// grid (n, 3): the packed reconstruction of CU i (size^2 luma samples at i * size^2; size^2 / 2 chroma samples, Cb then Cr, at i * size^2 / 2) into the picture
__global__ __launch_bounds__(256) void k_recon_put(const hop_rqt_job* __restrict__ jobs, const int16_t* __restrict__ reco_y, const int16_t* __restrict__ reco_c,
                                                   int16_t* __restrict__ ry, int16_t* __restrict__ rcb, int16_t* __restrict__ rcr, int pic_w) {
  const int i = blockIdx.x, comp = blockIdx.y;
  const int S = 1 << jobs[i].log2_cu, sh = comp ? 1 : 0, x = jobs[i].x >> sh, y = jobs[i].y >> sh, s = S >> sh, pitch = pic_w >> sh;
  const int16_t* src = comp == 0 ? reco_y + (size_t)i * S * S : reco_c + (size_t)i * (S * S / 2) + (comp == 2 ? s * s : 0);
  int16_t* pic = comp == 0 ? ry : comp == 1 ? rcb : rcr;
  for (int k = threadIdx.x; k < s * s; k += blockDim.x) { const int r = k / s, c = k - r * s; pic[(size_t)(y + r) * pitch + x + c] = src[k]; }
}

// grid (n, 3): CU i lies in a candidate slot (its y names copy k = y / pic_h of the pictures, hop_ctx_set_slots): the row above it and the column to its left -- what
// intra prediction inside the CU can read from outside it, 2 * size + 1 / 2 * size samples -- from the reconstruction picture into that copy
__global__ __launch_bounds__(128) void k_slot_prepare(const hop_rqt_job* __restrict__ jobs, int16_t* __restrict__ ry, int16_t* __restrict__ rcb, int16_t* __restrict__ rcr, int pic_w, int pic_h) {
  const int i = blockIdx.x, comp = blockIdx.y, sh = comp ? 1 : 0;
  const int slot = jobs[i].y / pic_h;
  if (slot == 0) return;
  const int w = pic_w >> sh, h = pic_h >> sh, n = (1 << jobs[i].log2_cu) >> sh, px = jobs[i].x >> sh, py = (jobs[i].y % pic_h) >> sh;
  int16_t* pic = comp == 0 ? ry : comp == 1 ? rcb : rcr;
  int16_t* cpy = pic + (size_t)slot * h * w;
  if (py > 0) for (int k = threadIdx.x; k <= 2 * n; k += blockDim.x) { const int xx = px - 1 + k; if (xx >= 0 && xx < w) cpy[(size_t)(py - 1) * w + xx] = pic[(size_t)(py - 1) * w + xx]; }
  if (px > 0) for (int k = threadIdx.x; k < 2 * n; k += blockDim.x) { const int yy = py + k; if (yy < h) cpy[(size_t)yy * w + px - 1] = pic[(size_t)yy * w + px - 1]; }
}

static int spine_stage(hop_ctx* c, size_t bytes, void** out) {      // a small staging area of its own (the host-array entries keep theirs)
  return hop_scratch(c, bytes, out);
}

extern "C" {

int hop_valid_pattern(hop_ctx* c, int n, const int32_t* xywh_mv, uint8_t* out) {
  if (!c || n < 0 || (n && (!xywh_mv || !out))) return hop_set_err(c, HOP_ERR_ARG, "hop_valid_pattern: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {                                      // the two probes must stay inside the padded plane (+ guard rows)
    const int32_t* p = xywh_mv + 6 * i;
    const int px = p[0] + (p[4] >> 2), py = p[1] + (p[5] >> 2) + p[3] + 4;
    if (px < -HOP_MARGIN_Y || px + p[2] + 4 >= c->pic_w + HOP_MARGIN_Y || py < -HOP_MARGIN_Y - HOP_GUARD_ROWS || py >= c->pic_h + HOP_MARGIN_Y + HOP_GUARD_ROWS)
      return hop_set_err(c, HOP_ERR_ARG, "hop_valid_pattern: query %d leaves the padded picture", i);
  }
  void* st; int r = spine_stage(c, (size_t)n * 24 + 256 + n, &st); if (r) return r;
  char* b = (char*)st; const size_t o = ((size_t)n * 24 + 255) & ~(size_t)255;
  HIPCHK(c, hipMemcpyAsync(b, xywh_mv, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_valid_pattern, dim3((n + 63) / 64), dim3(64), 0, c->stream, (const int32_t*)b, n, c->ss00[0], c->stride_y, (uint8_t*)(b + o));
  HIPCHK(c, hipMemcpyAsync(out, b + o, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_recon_stash(hop_ctx* c, int n, const int32_t* rect4, int restore) {
  if (!c || n < 0 || (n && !rect4)) return hop_set_err(c, HOP_ERR_ARG, "hop_recon_stash: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const int x = rect4[4 * i], y = rect4[4 * i + 1], s = rect4[4 * i + 2], slot = rect4[4 * i + 3];
    if ((s != 8 && s != 16 && s != 32 && s != 64) || x < 0 || y < 0 || (x % s) || ((y % c->pic_h) % s) || x + s > c->pic_w || y % c->pic_h + s > c->pic_h || y / c->pic_h > c->slots || slot < 0 || slot >= STASH_SLOTS)   // y may name a candidate slot's copy (hop_ctx_set_slots)
      return hop_set_err(c, HOP_ERR_ARG, "hop_recon_stash: block %d", i);
  }
  if (!c->stash) { HIPCHK(c, hipMalloc((void**)&c->stash, (size_t)STASH_SLOTS * STASH_SAMPLES * 2)); c->stash_slots = STASH_SLOTS; }
  void* st; int r = spine_stage(c, (size_t)n * 16, &st); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(st, rect4, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  if (restore) hipLaunchKernelGGL(k_recon_stash<true>, dim3(n, 3), dim3(256), 0, c->stream, (const int32_t*)st, c->rec[0], c->rec[1], c->rec[2], c->pic_w, c->stash);
  else hipLaunchKernelGGL(k_recon_stash<false>, dim3(n, 3), dim3(256), 0, c->stream, (const int32_t*)st, c->rec[0], c->rec[1], c->rec[2], c->pic_w, c->stash);
  if (c->coefpic) {                                                    // the candidate's levels travel with its reconstruction
    if (restore) hipLaunchKernelGGL(k_coef_stash<true>, dim3(n), dim3(256), 0, c->stream, (const int32_t*)st, c->coefpic, c->pic_w, c->pic_h, c->coef_stash);
    else hipLaunchKernelGGL(k_coef_stash<false>, dim3(n), dim3(256), 0, c->stream, (const int32_t*)st, c->coefpic, c->pic_w, c->pic_h, c->coef_stash);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));                        // the staging area is the context's scratch: the next call may reuse it
  return HOP_OK;
}

int hop_recon_put_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const int16_t* d_reco_y, const int16_t* d_reco_c) {
  if (!c || n < 0 || (n && (!d_jobs || !d_reco_y || !d_reco_c))) return hop_set_err(c, HOP_ERR_ARG, "hop_recon_put_device: bad argument");
  if (n == 0) return HOP_OK;
  hipLaunchKernelGGL(k_recon_put, dim3(n, 3), dim3(256), 0, c->stream, d_jobs, d_reco_y, d_reco_c, c->rec[0], c->rec[1], c->rec[2], c->pic_w);
  HIPCHK(c, hipGetLastError());
  return HOP_OK;
}

int hop_coef_put_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const int32_t* d_coef) {
  if (!c || n < 0 || (n && !d_jobs)) return hop_set_err(c, HOP_ERR_ARG, "hop_coef_put_device: bad argument");
  if (n == 0 || !c->coefpic) return HOP_OK;
  hipLaunchKernelGGL(k_coef_put, dim3(n), dim3(256), 0, c->stream, d_jobs, d_coef, c->coefpic, c->pic_w, c->pic_h);
  HIPCHK(c, hipGetLastError());
  return HOP_OK;
}

// the levels hop_encode_frame left: per CTU (the pictures of a stacked context one after the other, each in raster order) 4096 luma + 1024 Cb + 1024 Cr TCoeff in the
// reference's per-CTU layout (TComDataCU::m_pcTrCoeffY / Cb / Cr: a CU's block at 16 x / 4 x its z-order partition index, a TU's coefficients in raster order inside it)
int hop_rd_fraction_download(hop_ctx* c, uint16_t* out) {
  if (!c || !out) return hop_set_err(c, HOP_ERR_ARG, "hop_rd_fraction_download: bad argument");
  if (!c->rd_fraction) return hop_set_err(c, HOP_ERR_STATE, "hop_rd_fraction_download: hop_encode_frame has not run on this context");
  memcpy(out, c->rd_fraction, (size_t)c->rd_fraction_n * sizeof(uint16_t));
  return HOP_OK;
}
int hop_levels_download(hop_ctx* c, int32_t* out) {
  if (!c || !out) return hop_set_err(c, HOP_ERR_ARG, "hop_levels_download: bad argument");
  if (!c->coefpic) return hop_set_err(c, HOP_ERR_STATE, "hop_levels_download: hop_encode_frame has not run on this context");
  const int wctu = (c->pic_w + 63) >> 6, n_pic = c->sub_pitch ? (c->pic_h - c->sub_h) / c->sub_pitch + 1 : 1, ph = c->sub_pitch ? c->sub_h : c->pic_h, hctu = (ph + 63) >> 6;
  for (int k = 0; k < n_pic; k++) {
    const size_t first = (size_t)(k * (c->sub_pitch >> 6)) * wctu;      // the picture's first CTU in the stack's CTU grid (pitch: a multiple of 64)
    HIPCHK(c, hipMemcpyAsync(out + (size_t)k * wctu * hctu * COEF_PER_CTU, c->coefpic + first * COEF_PER_CTU, (size_t)wctu * hctu * COEF_PER_CTU * 4, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_ssref_commit_recon(hop_ctx* c, int n, const int32_t* rect4) {
  if (!c || n < 0 || (n && !rect4)) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_commit_recon: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const int x = rect4[4 * i], y = rect4[4 * i + 1], s = rect4[4 * i + 2];
    if ((s != 8 && s != 16 && s != 32 && s != 64) || x < 0 || y < 0 || (x % s) || (y % s) || x + s > c->pic_w || y + s > c->pic_h) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_commit_recon: block %d", i);
  }
  void* st; int r = spine_stage(c, (size_t)n * 16, &st); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(st, rect4, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_ssref_commit(c, n, (const int32_t*)st, c->rec[0], c->rec[1], c->rec[2], 0); if (r) return r;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// the spine's product backend: every request goes to the kernels of this library; device-resident arenas, small transfers only
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
using namespace hopspine;

struct Bail { int code; };
// wall time and calls per kind of request of the last hop_encode_frame (hop_encode_stats): 0 me_search, 1 pred_inter, 2 distortion, 3 valid_pattern, 4 inter_cu with
// residual, 5 inter_cu without, 6 intra_cu, 7 recon stash, 8 commit, 9 the wait for the evaluation chains issued on streams of their own (then 4 - 6 hold the issue time only)
double g_stat_ms[16]; double g_stat_calls[16];
std::mutex g_stat_lock;
struct Tick { int k; std::chrono::steady_clock::time_point t0; explicit Tick(int k) : k(k), t0(std::chrono::steady_clock::now()) {}
              ~Tick() { const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                        std::lock_guard<std::mutex> g(g_stat_lock); g_stat_ms[k] += d; g_stat_calls[k] += 1; } };
#define BK(call) do { int r_ = (call); if (r_ != HOP_OK) throw Bail{ r_ }; } while (0)
#define BH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { hop_set_err(c, HOP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); throw Bail{ HOP_ERR_DEVICE }; } } while (0)

class HipBackend : public BatchInner {
 public:
  enum { MAXN = 2048, MAXP = 32768 };   // candidates of one class / predictor jobs in one batch (one per CTU row in flight, all pictures together)
  explicit HipBackend(hop_ctx* ctx, bool sub = false) : c(ctx), arena(nullptr), is_sub_(sub), rr_(0) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    o_jobs = take(MAXN * sizeof(hop_rqt_job)); o_syn = take(MAXN * sizeof(hop_cu_syntax)); o_isyn = take(MAXN * sizeof(hop_intra_cu_syntax)); o_isyn_out = take(MAXN * sizeof(hop_intra_cu_syntax));
    o_opts = take(MAXN * sizeof(hop_intra_rqt_opt)); o_sjobs = take(MAXN * sizeof(hop_intra_search_job)); o_sres = take(MAXN * sizeof(hop_intra_search_result));
    o_res = take(MAXN * sizeof(hop_rqt_result)); o_cres = take(MAXN * sizeof(hop_intra_chroma_result)); o_coef = take((size_t)MAXN * 6144 * 4);
    o_reco_y = take((size_t)MAXN * 4096 * 2); o_reco_c = take((size_t)MAXN * 2048 * 2);
    o_ctx_in = take(MAXN * sizeof(hop_cabac_ctx)); o_cu_in = take(MAXN * sizeof(hop_cabac_cu_ctx)); o_ctx_after = take(MAXN * sizeof(hop_cabac_ctx));
    o_ctx_out = take(MAXN * sizeof(hop_cabac_ctx)); o_cu_out = take(MAXN * sizeof(hop_cabac_cu_ctx)); o_fin = take(MAXN * sizeof(hop_cu_final));
    o_bits = take(MAXN * 4); o_skipped = take(MAXN * 4); o_cost = take(MAXN * 8); o_dist = take(MAXN * 4);
    o_pjobs = take(MAXP * sizeof(hop_pred_job)); o_djobs = take(MAXP * sizeof(hop_dist_job)); o_pout = take(MAXP * 4);
    o_mjobs = take(MAXP * sizeof(hop_pu_job)); o_mres = take(MAXP * sizeof(hop_pu_result));
    // what a candidate batch sends and gets back travels as ONE copy each way between pinned host memory and these two regions, packed for the batch's n
    io_bytes = (size_t)MAXN * (sizeof(hop_rqt_job) + sizeof(hop_intra_cu_syntax) + sizeof(hop_intra_rqt_opt) + sizeof(hop_intra_search_job) + sizeof(hop_rqt_result) +
                               sizeof(hop_intra_search_result) + sizeof(hop_intra_chroma_result) + 2 * sizeof(hop_cabac_ctx) + 2 * sizeof(hop_cabac_cu_ctx) + sizeof(hop_cu_final) + 4 * sizeof(hop_pred_job) + 64) + 16 * 256;
    o_in = take(io_bytes); o_out = take(io_bytes);
    bytes = o;
    hin = hout = nullptr;
    if (hipMalloc((void**)&arena, bytes) != hipSuccess || hipHostMalloc((void**)&hin, io_bytes) != hipSuccess || hipHostMalloc((void**)&hout, io_bytes) != hipSuccess) {
      if (arena) (void)hipFree(arena);
      arena = nullptr; hop_set_err(c, HOP_ERR_DEVICE, "spine arena allocation failed");
    }
  }
  ~HipBackend() {
    for (auto b : subs_) delete b;
    for (auto v : views_) hop_ctx_destroy(v);
    if (arena) (void)hipFree(arena); if (hin) (void)hipHostFree(hin); if (hout) (void)hipHostFree(hout);
  }
  // k further streams (views of the context, each with a backend of its own): the candidate evaluations of one round -- SS/GT candidates with and without residual,
  // intra 2Nx2N, intra NxN: chains that touch different candidate slots -- are issued one per stream and collected in end_round()
  bool add_streams(int k) {
    for (int i = 0; i < k; i++) {
      hop_ctx* v = nullptr;
      if (hop_ctx_create_view(c, &v) != HOP_OK) return false;
      HipBackend* b = new HipBackend(v, true);
      if (!b->ok()) { delete b; hop_ctx_destroy(v); return false; }
      views_.push_back(v); subs_.push_back(b);
    }
    return true;
  }
  bool can_defer() { return !subs_.empty(); }
  void end_round() {
    bool any = false; for (auto b : subs_) any |= (bool)b->collect_;
    if (!any) return;
    Tick t(9);                                                            // the wait for the evaluation chains of the round
    for (auto b : subs_) if (b->collect_) { std::function<void()> f; f.swap(b->collect_); f(); if (b->c->err[0] && !c->err[0]) strncpy(c->err, b->c->err, sizeof(c->err) - 1); }
  }
  bool ok() const { return arena != nullptr; }

  void on_device() { if (hipSetDevice(c->device) != hipSuccess) { hop_set_err(c, HOP_ERR_DEVICE, "hipSetDevice(%d) failed", c->device); throw Bail{ HOP_ERR_DEVICE }; } }   // batches are served by whichever worker thread arrives last: the current device is per-thread state
  void begin_frame() { on_device(); BK(hop_ssref_reset(c)); BK(hop_sync(c)); }
  // the short requests travel through the backend's pinned buffers: jobs the kernels read once are read by the device straight from pinned host memory, results the
  // kernels write once are written straight into it -- no staging copies, one synchronisation per request
  void me_search(int, int n, const hop_pu_job* jobs, hop_pu_result* res) {
    on_device(); Tick t(0);
    if (n > MAXP) { BK(hop_me_search(c, n, jobs, res, HOP_STAGE_GT)); return; }
    BK(hop_check_pu_jobs(c, n, jobs));
    memcpy(hin, jobs, (size_t)n * sizeof(hop_pu_job));
    hipStream_t st = c->stream;
    BH(hipMemcpyAsync(arena + o_mjobs, hin, (size_t)n * sizeof(hop_pu_job), hipMemcpyHostToDevice, st));   // (the search kernels read a job many times: a device copy)
    BK(hop_me_search_device(c, n, (const hop_pu_job*)(arena + o_mjobs), (hop_pu_result*)(arena + o_mres), HOP_STAGE_GT));
    BH(hipMemcpyAsync(hout, arena + o_mres, (size_t)n * sizeof(hop_pu_result), hipMemcpyDeviceToHost, st));
    BH(hipStreamSynchronize(st));
    memcpy(res, hout, (size_t)n * sizeof(hop_pu_result));
  }
  void pred_inter(int, int n, const hop_pred_job* jobs) {
    on_device(); Tick t(1);
    if ((size_t)n * sizeof(hop_pred_job) > io_bytes) { BK(hop_pred_inter(c, n, jobs, nullptr, nullptr, nullptr)); return; }
    BK(hop_check_pred_jobs(c, n, jobs));
    memcpy(hin, jobs, (size_t)n * sizeof(hop_pred_job));
    BK(hop_pred_inter_device(c, n, (const hop_pred_job*)hin));
    BH(hipStreamSynchronize(c->stream));
  }
  void distortion(int, int n, const hop_dist_job* jobs, uint32_t* out) { on_device(); Tick t(2); BK(hop_distortion(c, n, jobs, out)); }
  void valid_pattern(int, int n, const int32_t* q, uint8_t* out) { on_device(); Tick t(3); BK(hop_valid_pattern(c, n, q, out)); }
  void recon_save(int lane, int slot, int x, int y, int size) { on_device(); Tick t(7); const int32_t r[4] = { x, y, size, lane * 16 + slot }; BK(hop_recon_stash(c, 1, r, 0)); }
  void recon_restore(int lane, int slot, int x, int y, int size) { on_device(); Tick t(7); const int32_t r[4] = { x, y, size, lane * 16 + slot }; BK(hop_recon_stash(c, 1, r, 1)); }
  void commit(int, int x, int y, int size) { on_device(); Tick t(8); const int32_t r[4] = { x, y, size, 0 }; BK(hop_ssref_commit_recon(c, 1, r)); }
  // a picture coded by several ranks (hop_encode_set_shard): a CTU's block of the reconstruction picture out to the host, and a block another rank coded in -- into the
  // reconstruction picture and, as the commits of its CUs would have, into the SS reference
  void export_block(int x, int y, int w, int h, int16_t* py, int16_t* pcb, int16_t* pcr) {
    on_device();
    const size_t P = (size_t)c->pic_w, PC = P / 2;
    BH(hipMemcpy2DAsync(py, (size_t)w * 2, c->rec[0] + (size_t)y * P + x, P * 2, (size_t)w * 2, h, hipMemcpyDeviceToHost, c->stream));
    BH(hipMemcpy2DAsync(pcb, (size_t)w, c->rec[1] + (size_t)(y / 2) * PC + x / 2, PC * 2, (size_t)w, h / 2, hipMemcpyDeviceToHost, c->stream));
    BH(hipMemcpy2DAsync(pcr, (size_t)w, c->rec[2] + (size_t)(y / 2) * PC + x / 2, PC * 2, (size_t)w, h / 2, hipMemcpyDeviceToHost, c->stream));
    BH(hipStreamSynchronize(c->stream));
  }
  void import_block(int x, int y, int w, int h, const int16_t* py, const int16_t* pcb, const int16_t* pcr) {
    on_device();
    const size_t P = (size_t)c->pic_w, PC = P / 2;
    BH(hipMemcpy2DAsync(c->rec[0] + (size_t)y * P + x, P * 2, py, (size_t)w * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, c->stream));
    BH(hipMemcpy2DAsync(c->rec[1] + (size_t)(y / 2) * PC + x / 2, PC * 2, pcb, (size_t)w, (size_t)w, h / 2, hipMemcpyHostToDevice, c->stream));
    BH(hipMemcpy2DAsync(c->rec[2] + (size_t)(y / 2) * PC + x / 2, PC * 2, pcr, (size_t)w, (size_t)w, h / 2, hipMemcpyHostToDevice, c->stream));
    std::vector<int32_t> r;
    if (w == 64 && h == 64) { const int32_t q[4] = { x, y, 64, 0 }; r.assign(q, q + 4); }
    else for (int yy = y; yy < y + h; yy += 8) for (int xx = x; xx < x + w; xx += 8) { const int32_t q[4] = { xx, yy, 8, 0 }; r.insert(r.end(), q, q + 4); }
    BK(hop_ssref_commit_recon(c, (int)(r.size() / 4), r.data()));
    BH(hipStreamSynchronize(c->stream));
  }

  void pred_cost(int, int n, const hop_pred_job* jobs, int kind, uint32_t* out) { pred_cost_n(1, &n, jobs, &kind, out); }
  // every sequence walked in order by a workgroup of ONE launch (k_pred_cost); one synchronisation at the end
  void pred_cost_n(int m, const int* len, const hop_pred_job* jobs, const int* kinds, uint32_t* out) {
    on_device();
    int total = 0, maxlen = 0; for (int s = 0; s < m; s++) { total += len[s]; if (len[s] > maxlen) maxlen = len[s]; }
    if (total == 0) return;
    if (total > MAXP || (size_t)total * (sizeof(hop_pred_job) + sizeof(hop_dist_job)) + 512 > io_bytes) {   // larger batches in parts (whole sequences)
      if (m == 1) throw Bail{ HOP_ERR_ARG };
      int h = m / 2, at = 0; for (int s = 0; s < h; s++) at += len[s];
      pred_cost_n(h, len, jobs, kinds, out); pred_cost_n(m - h, len + h, jobs + at, kinds + h, out + at);
      return;
    }
    Tick t(1);
    // the sequences as they are, one after the other; ONE launch rates them all, a workgroup per candidate (k_pred_cost)
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t h_first = al((size_t)total * sizeof(hop_pred_job)), h_kinds = h_first + al((size_t)(m + 1) * 4), h_seq = h_kinds + al((size_t)m * 4), h_bytes = h_seq + al((size_t)total * 4);
    if (h_bytes > io_bytes) throw Bail{ HOP_ERR_ARG };
    memcpy(hin, jobs, (size_t)total * sizeof(hop_pred_job));
    int32_t* hf = (int32_t*)(hin + h_first); int32_t* hk = (int32_t*)(hin + h_kinds); int32_t* hs = (int32_t*)(hin + h_seq);
    for (int s = 0, at = 0; s < m; at += len[s], s++) { hf[s] = at; hk[s] = kinds[s]; for (int q = 0; q < len[s]; q++) hs[at + q] = s; }
    hf[m] = total;
    BK(hop_check_pred_jobs(c, total, jobs));
    hipStream_t st = c->stream;
    uint32_t* o = (uint32_t*)hout;
    BK(hop_launch_pred_cost(c, total, (const int32_t*)(hin + h_seq), (const int32_t*)(hin + h_first), (const hop_pred_job*)hin, (const int32_t*)(hin + h_kinds), o));   // jobs read from, costs written to pinned host memory
    BH(hipStreamSynchronize(st));
    memcpy(out, o, (size_t)total * 4);
  }
  void stash_n(int n, const int32_t* rect4, int restore) { on_device(); Tick t(7); BK(hop_recon_stash(c, n, rect4, restore)); }
  void commit_n(int n, const int32_t* rect4) { on_device(); Tick t(8); BK(hop_ssref_commit_recon(c, n, rect4)); }
  void inter_cu(int, const InterEval& e, const Coder& in, EvalResult& out) { const InterEval* pe = &e; const Coder* pi = &in; EvalResult* po = &out; inter_n(1, &pe, &pi, &po); }
  void intra_cu(int, const IntraEval& e, const Coder& in, EvalResult& out) { const IntraEval* pe = &e; const Coder* pi = &in; EvalResult* po = &out; intra_n(1, &pe, &pi, &po); }

  // n candidates of ONE class (CU size; with or without residual)
  void inter_n(int n, const InterEval* const* e, const Coder* const* in, EvalResult* const* out) {
    if (n > MAXN) { for (int o = 0; o < n; o += MAXN) inter_n(n - o < MAXN ? n - o : MAXN, e + o, in + o, out + o); return; }   // larger batches in parts
    if (!subs_.empty()) { HipBackend* b = subs_[rr_++ % subs_.size()]; if (b->collect_) { std::function<void()> f; f.swap(b->collect_); f(); } b->inter_n(n, e, in, out); return; }
    on_device();
    Tick t(e[0]->skip_res ? 5 : 4);
    hipStream_t s = c->stream;
    const bool skip = e[0]->skip_res != 0;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t i_jobs = take(n * sizeof(hop_rqt_job)), i_syn = take(n * sizeof(hop_cu_syntax)), i_cx = take(n * sizeof(hop_cabac_ctx)), i_cu = take(n * sizeof(hop_cabac_cu_ctx)), in_bytes = o;
    int n_pred = 0; for (int i = 0; i < n; i++) n_pred += e[i]->n_pred;
    const size_t i_pred = take((size_t)n_pred * sizeof(hop_pred_job));    // (read by the device straight from the pinned buffer: not part of the copy)
    o = 0;
    const size_t r_fin = take(n * sizeof(hop_cu_final)), r_bits = take(n * 4), r_skipped = take(n * 4), r_cx = take(n * sizeof(hop_cabac_ctx)), r_cu = take(n * sizeof(hop_cabac_cu_ctx)),
                 r_res = take(skip ? 0 : n * sizeof(hop_rqt_result)), out_bytes = o;
    hop_rqt_job* hj = (hop_rqt_job*)(hin + i_jobs); hop_cu_syntax* hs = (hop_cu_syntax*)(hin + i_syn); hop_cabac_ctx* hx = (hop_cabac_ctx*)(hin + i_cx); hop_cabac_cu_ctx* hu = (hop_cabac_cu_ctx*)(hin + i_cu);
    for (int i = 0; i < n; i++) { hj[i] = e[i]->job; hj[i].ctx_index = i; hs[i] = e[i]->syn; hx[i] = in[i]->r; hu[i] = in[i]->c; }
    char* din = arena + o_in; char* dout = arena + o_out;
    BH(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, s));
    if (n_pred > 0) {                                                     // the candidates' final motion compensation (InterEval::pred) first, on this stream
      hop_pred_job* hp = (hop_pred_job*)(hin + i_pred);
      for (int i = 0, k = 0; i < n; i++) for (int q = 0; q < e[i]->n_pred; q++) hp[k++] = e[i]->pred[q];
      BK(hop_check_pred_jobs(c, n_pred, hp));
      BK(hop_pred_inter_device(c, n_pred, hp));
    }
    const hop_rqt_job* d_jobs = (const hop_rqt_job*)(din + i_jobs); const hop_cu_syntax* d_syn = (const hop_cu_syntax*)(din + i_syn);
    const hop_cabac_ctx* d_cx = (const hop_cabac_ctx*)(din + i_cx); const hop_cabac_cu_ctx* d_cu = (const hop_cabac_cu_ctx*)(din + i_cu);
    if (skip) {
      BK(hop_inter_cu_skip_device(c, n, d_jobs, d_syn, d_cx, d_cu, (hop_cu_final*)(dout + r_fin), (uint32_t*)(dout + r_bits), (double*)(arena + o_cost),
                                  (hop_cabac_ctx*)(dout + r_cx), (hop_cabac_cu_ctx*)(dout + r_cu)));
    } else {
      hop_inter_class k; memset(&k, 0, sizeof(k));
      k.n = n; k.cls = hj[0]; k.cls.x = 0; k.cls.y = 0; k.cls.ctx_index = 0;
      k.d_jobs = d_jobs; k.d_syntax = d_syn; k.d_results = (hop_rqt_result*)(dout + r_res);
      k.d_coef = (int32_t*)(arena + o_coef); k.d_ctx_after = (hop_cabac_ctx*)(arena + o_ctx_after); k.d_finals = (hop_cu_final*)(dout + r_fin);
      k.d_bits = (uint32_t*)(dout + r_bits); k.d_skipped = (uint32_t*)(dout + r_skipped); k.d_cost = (double*)(arena + o_cost);
      k.d_ctx_out = (hop_cabac_ctx*)(dout + r_cx); k.d_cu_ctx_out = (hop_cabac_cu_ctx*)(dout + r_cu);
      BK(hop_inter_cu_device_classes(c, 1, &k, d_cx, d_cu));
    }
    BK(hop_coef_put_device(c, n, d_jobs, skip ? nullptr : (const int32_t*)(arena + o_coef)));   // the candidate's levels into the image of its slot
    BH(hipMemcpyAsync(hout, dout, out_bytes, hipMemcpyDeviceToHost, s));
    std::vector<const Coder*> inv(in, in + n); std::vector<EvalResult*> outv(out, out + n);
    auto collect = [this, s, n, skip, r_fin, r_bits, r_skipped, r_cx, r_cu, r_res, inv, outv]() {
      BH(hipStreamSynchronize(s));
      const hop_cu_final* fin = (const hop_cu_final*)(hout + r_fin); const uint32_t* bits = (const uint32_t*)(hout + r_bits); const uint32_t* skipped = (const uint32_t*)(hout + r_skipped);
      const hop_cabac_ctx* cx = (const hop_cabac_ctx*)(hout + r_cx); const hop_cabac_cu_ctx* cu = (const hop_cabac_cu_ctx*)(hout + r_cu); const hop_rqt_result* res = (const hop_rqt_result*)(hout + r_res);
      for (int i = 0; i < n; i++) {
        EvalResult& r = *outv[i];
        r.bits = bits[i]; r.dist = fin[i].dist[0] + fin[i].dist[1] + fin[i].dist[2]; r.cost = 0; r.skipped = skip ? 1 : (int)skipped[i]; r.root_cbf = (int)fin[i].root_cbf;
        if (skip) { memset(r.tr_idx, 0, sizeof(r.tr_idx)); memset(r.cbf, 0, sizeof(r.cbf)); memset(r.tskip, 0, sizeof(r.tskip)); }
        else { memcpy(r.tr_idx, res[i].tr_idx, 256); memcpy(r.cbf, res[i].cbf, 768); memcpy(r.tskip, res[i].tskip, 768); }
        r.after = *inv[i]; r.after.r = cx[i]; r.after.c = cu[i];
      }
    };
    if (is_sub_) collect_ = collect; else collect();                      // a sub-backend's batch is collected in the owner's end_round()
  }
  void intra_n(int n, const IntraEval* const* e, const Coder* const* in, EvalResult* const* out) {
    if (n > MAXN) { for (int o = 0; o < n; o += MAXN) intra_n(n - o < MAXN ? n - o : MAXN, e + o, in + o, out + o); return; }
    if (!subs_.empty()) { HipBackend* b = subs_[rr_++ % subs_.size()]; if (b->collect_) { std::function<void()> f; f.swap(b->collect_); f(); } b->intra_n(n, e, in, out); return; }
    on_device();
    Tick t(6);
    hipStream_t s = c->stream;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t i_jobs = take(n * sizeof(hop_rqt_job)), i_syn = take(n * sizeof(hop_intra_cu_syntax)), i_opt = take(n * sizeof(hop_intra_rqt_opt)), i_sj = take(n * sizeof(hop_intra_search_job)),
                 i_cx = take(n * sizeof(hop_cabac_ctx)), i_cu = take(n * sizeof(hop_cabac_cu_ctx)), in_bytes = o;
    o = 0;
    const size_t r_res = take(n * sizeof(hop_rqt_result)), r_sres = take(n * sizeof(hop_intra_search_result)), r_cres = take(n * sizeof(hop_intra_chroma_result)), r_bits = take(n * 4),
                 r_dist = take(n * 4), r_cx = take(n * sizeof(hop_cabac_ctx)), r_cu = take(n * sizeof(hop_cabac_cu_ctx)), out_bytes = o;
    hop_rqt_job* hj = (hop_rqt_job*)(hin + i_jobs); hop_intra_cu_syntax* hs = (hop_intra_cu_syntax*)(hin + i_syn); hop_intra_rqt_opt* ho = (hop_intra_rqt_opt*)(hin + i_opt);
    hop_intra_search_job* hq = (hop_intra_search_job*)(hin + i_sj); hop_cabac_ctx* hx = (hop_cabac_ctx*)(hin + i_cx); hop_cabac_cu_ctx* hu = (hop_cabac_cu_ctx*)(hin + i_cu);
    for (int i = 0; i < n; i++) { hj[i] = e[i]->job; hj[i].ctx_index = i; hs[i] = e[i]->syn; ho[i] = e[i]->opt; hq[i] = e[i]->sjob; hx[i] = in[i]->r; hu[i] = in[i]->c; }
    char* din = arena + o_in; char* dout = arena + o_out;
    BH(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, s));
    hop_intra_class k; memset(&k, 0, sizeof(k));
    k.n = n; k.part_nxn = e[0]->part_nxn; k.num_full_rd = e[0]->sjob.num_full_rd; k.cls = hj[0]; k.cls.x = 0; k.cls.y = 0; k.cls.ctx_index = 0;
    k.d_jobs = (const hop_rqt_job*)(din + i_jobs); k.d_syntax = (const hop_intra_cu_syntax*)(din + i_syn); k.d_opts = (const hop_intra_rqt_opt*)(din + i_opt);
    k.d_sjobs = (const hop_intra_search_job*)(din + i_sj); k.d_sresults = (hop_intra_search_result*)(dout + r_sres); k.d_results = (hop_rqt_result*)(dout + r_res);
    k.d_cresults = (hop_intra_chroma_result*)(dout + r_cres); k.d_coef = (int32_t*)(arena + o_coef); k.d_reco_y = (int16_t*)(arena + o_reco_y); k.d_reco_c = (int16_t*)(arena + o_reco_c);
    k.d_syntax_out = (hop_intra_cu_syntax*)(arena + o_isyn_out); k.d_dist = (uint32_t*)(dout + r_dist); k.d_bits = (uint32_t*)(dout + r_bits); k.d_cost = (double*)(arena + o_cost);
    k.d_ctx_out = (hop_cabac_ctx*)(dout + r_cx); k.d_cu_ctx_out = (hop_cabac_cu_ctx*)(dout + r_cu);
    if (c->slots > 0) {                                                   // candidates in slots: their neighbouring row and column first
      hipLaunchKernelGGL(k_slot_prepare, dim3(n, 3), dim3(128), 0, s, k.d_jobs, c->rec[0], c->rec[1], c->rec[2], c->pic_w, c->pic_h);
      BH(hipGetLastError());
    }
    BK(hop_intra_cu_device_classes(c, 1, &k, (const hop_cabac_ctx*)(din + i_cx), (const hop_cabac_cu_ctx*)(din + i_cu)));
    BK(hop_recon_put_device(c, n, k.d_jobs, k.d_reco_y, k.d_reco_c));
    BK(hop_coef_put_device(c, n, k.d_jobs, k.d_coef));
    BH(hipMemcpyAsync(hout, dout, out_bytes, hipMemcpyDeviceToHost, s));
    std::vector<const Coder*> inv(in, in + n); std::vector<EvalResult*> outv(out, out + n);
    auto collect = [this, s, n, r_res, r_sres, r_cres, r_bits, r_dist, r_cx, r_cu, inv, outv]() {
      BH(hipStreamSynchronize(s));
      const hop_rqt_result* res = (const hop_rqt_result*)(hout + r_res); const hop_intra_search_result* sres = (const hop_intra_search_result*)(hout + r_sres);
      const hop_intra_chroma_result* cres = (const hop_intra_chroma_result*)(hout + r_cres); const uint32_t* bits = (const uint32_t*)(hout + r_bits); const uint32_t* dist = (const uint32_t*)(hout + r_dist);
      const hop_cabac_ctx* cx = (const hop_cabac_ctx*)(hout + r_cx); const hop_cabac_cu_ctx* cu = (const hop_cabac_cu_ctx*)(hout + r_cu);
      for (int i = 0; i < n; i++) {
        EvalResult& r = *outv[i];
        r.bits = bits[i]; r.dist = dist[i]; r.cost = 0; r.skipped = 0; r.root_cbf = 1;
        memcpy(r.tr_idx, res[i].tr_idx, 256); memcpy(r.cbf, res[i].cbf, 768); memcpy(r.tskip, res[i].tskip, 768);
        for (int p = 0; p < 4; p++) r.luma_dir[p] = sres[i].best_dir[p];
        r.chroma_dir = cres[i].best_mode;
        r.after = *inv[i]; r.after.r = cx[i]; r.after.c = cu[i];
      }
    };
    if (is_sub_) collect_ = collect; else collect();
  }
 private:
  hop_ctx* c; char* arena; size_t bytes; char* hin; char* hout; size_t io_bytes, o_in, o_out;
  bool is_sub_; int rr_; std::vector<hop_ctx*> views_; std::vector<HipBackend*> subs_; std::function<void()> collect_;   // collect_: the second half of a batch issued on this (sub) backend
  size_t o_jobs, o_syn, o_isyn, o_isyn_out, o_opts, o_sjobs, o_sres, o_res, o_cres, o_coef, o_reco_y, o_reco_c, o_ctx_in, o_cu_in, o_ctx_after, o_ctx_out, o_cu_out, o_fin, o_bits, o_skipped,
         o_cost, o_dist, o_pjobs, o_djobs, o_pout, o_mjobs, o_mres;
};

}  // namespace

extern "C" {

int hop_sizeof_cu_part(void) { return (int)sizeof(hopspine::Part); }
int hop_encode_progress(hop_ctx* c, int64_t* ctus_retired) {
  if (!c || !ctus_retired) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_progress: bad argument");
  *ctus_retired = (int64_t)c->enc_progress.load();
  return HOP_OK;
}
int hop_encode_cancel(hop_ctx* c) { if (!c) return HOP_ERR_ARG; c->enc_cancel.store(1); return HOP_OK; }
int hop_encode_set_shard(hop_ctx* c, int rank, int world, hop_allgather_fn fn, void* user) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_set_shard: rank %d of %d", rank, world);
  c->shard_rank = rank; c->shard_world = world; c->shard_fn = fn; c->shard_user = user;
  return HOP_OK;
}
void hop_encode_stats(double ms[16], double calls[16]) { memcpy(ms, g_stat_ms, sizeof(g_stat_ms)); memcpy(calls, g_stat_calls, sizeof(g_stat_calls)); }

// One picture through the RD spine on the device: the original must be resident (hop_upload_orig).  Afterwards the reconstruction picture (hop_recon_download) holds the
// reconstruction before the loop filters and the SS reference equals it.  Outputs may be NULL.
int hop_encode_frame(hop_ctx* c, const hop_enc_params* p, double* ctu_cost, uint32_t* ctu_bits, uint32_t* ctu_dist, hop_cu_part* parts, uint64_t* n_candidates) {
  if (!c || !p) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_encode_frame: hop_upload_orig has not been called");
  if (!p->plain_intra && (c->bd_y != 8 || c->bd_c != 8)) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: the HOP configuration is 8-bit (the GT warp clips to 255)");
  if (c->bd_y != c->bd_c) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: luma and chroma bit depths must be equal");
  if (p->qp < 0 || p->qp > 51 || p->mi_size <= 0) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: qp / micro-image size");
  // a stacked context (hop_ctx_set_stack): its pictures are coded side by side, every one as a picture of its own
  const int n_pic = c->sub_pitch ? (c->pic_h - c->sub_h) / c->sub_pitch + 1 : 1, pic_h = c->sub_pitch ? c->sub_h : c->pic_h;
  if (n_pic > 1 && p->wavefront_lag <= 0) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: the pictures of a stacked context are coded as wavefronts (wavefront_lag > 0)");
  hopspine::EncConfig cfg;
  if (p->plain_intra) hopspine::default_plain_config(cfg, c->pic_w, pic_h, p->qp, c->bd_y); else hopspine::default_hop_config(cfg, c->pic_w, pic_h, p->qp, p->mi_size);
  cfg.wpp = (p->wpp || p->wavefront_lag > 0) ? 1 : 0;
  c->enc_progress.store(0); c->enc_cancel.store(0);
  cfg.progress = &c->enc_progress; cfg.cancel = &c->enc_cancel;
  if (c->slots > 0 && !p->plain_intra) { cfg.spec_slots = c->slots; cfg.slot_pitch = c->pic_h; }   // hop_ctx_set_slots: the SS/GT candidates of a CU side by side
  struct FnComm : hopspine::ShardComm { hop_ctx* c; void allgather(const void* s, void* r, size_t b) { if (c->shard_fn(c->shard_user, s, r, b) != 0) { hop_set_err(c, HOP_ERR_STATE, "hop_encode_frame: the all-gather callback failed"); throw Bail{ HOP_ERR_STATE }; } } } comm;
  comm.c = c;
  if (c->shard_world > 1) {
    if (p->wavefront_lag <= 0 || n_pic != 1 || p->streams > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: a sharded picture (hop_encode_set_shard) is ONE picture coded as a wavefront with batched requests");
    cfg.shard_rank = c->shard_rank; cfg.shard_world = c->shard_world; cfg.shard = &comm;
  }
  if (p->wavefront_lag > 0 && p->first_ctus > 0) return hop_set_err(c, HOP_ERR_ARG, "hop_encode_frame: first_ctus applies to the raster-order mode");
  {                                                                      // the images of the levels (one per candidate slot) and their part of the stash
    const size_t ctus = (size_t)((c->pic_w + 63) >> 6) * ((c->pic_h + 63) >> 6), want = ctus * (c->slots + 1) * COEF_PER_CTU * 4;
    if (c->coefpic) { (void)hipFree(c->coefpic); c->coefpic = nullptr; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc((void**)&c->coefpic, want));
    HIPCHK(c, hipMemsetAsync(c->coefpic, 0, want, c->stream));
    if (!c->coef_stash) HIPCHK(c, hipMalloc((void**)&c->coef_stash, (size_t)STASH_SLOTS * COEF_PER_CTU * 4));
  }
  hopspine::posted_requests_allowed = false;                             // (see hop_spine.h: no posted predictions on the device; stash / restore / commit are posted)
  HipBackend be(c);
  if (!be.ok()) return HOP_ERR_DEVICE;
  if (c->slots > 0 && p->wavefront_lag > 0 && !p->plain_intra) {         // candidates side by side: their evaluation chains of one round on streams of their own
    int k = 4; if (const char* e = getenv("HOP_SPINE_STREAMS")) k = atoi(e);
    if (c->prof_on) k = 1;                                               // profiling (hop_profile_*) records on the context's own stream
    if (k > 1 && !be.add_streams(k > 8 ? 8 : k)) return hop_set_err(c, HOP_ERR_DEVICE, "hop_encode_frame: could not create the evaluation streams");
  }
  // streams > 1: one view of the context (own stream, own work areas) per CTU row in flight; their launch chains overlap on the device
  std::vector<hop_ctx*> views; std::vector<HipBackend*> vbe; std::vector<Backend*> lanes;
  if (p->wavefront_lag > 0 && p->streams > 1 && n_pic == 1) {
    if (!c->stash) { HIPCHK(c, hipMalloc((void**)&c->stash, (size_t)STASH_SLOTS * STASH_SAMPLES * 2)); c->stash_slots = STASH_SLOTS; }
    const int nv = p->streams > 64 ? 64 : p->streams;
    for (int i = 0; i < nv; i++) {
      hop_ctx* v = nullptr;
      if (hop_ctx_create_view(c, &v) != HOP_OK) break;
      HipBackend* b = new HipBackend(v);
      if (!b->ok()) { delete b; hop_ctx_destroy(v); break; }
      views.push_back(v); vbe.push_back(b); lanes.push_back(b);
    }
    if ((int)lanes.size() != nv) { for (auto b : vbe) delete b; for (auto v : views) hop_ctx_destroy(v); return hop_set_err(c, HOP_ERR_DEVICE, "hop_encode_frame: could not create %d views", nv); }
  }
  memset(g_stat_ms, 0, sizeof(g_stat_ms)); memset(g_stat_calls, 0, sizeof(g_stat_calls));
  hopspine::LogBackend* lg = (lanes.empty() && getenv("HOP_SPINE_LOG")) ? new hopspine::LogBackend(&be, getenv("HOP_SPINE_LOG")) : nullptr;   // debugging aid: every request and its answer
  hopspine::BatchInner* use = lg ? (hopspine::BatchInner*)lg : (hopspine::BatchInner*)&be;
  // a stacked context in groups (HOP_SPINE_GROUPS, default 1): the pictures are dealt to G groups, each with a backend on a view of the context (streams, pinned buffers and
  // work areas of its own) and a rendezvous of its own on its own worker threads, all running side by side -- one group packs and unpacks while another one's launches
  // run.  The pictures stay independent of each other, so the results do not depend on G.
  int G = 1;
  if (n_pic > 1 && !lg && !c->prof_on) { if (const char* e = getenv("HOP_SPINE_GROUPS")) G = atoi(e); if (G > 8) G = 8; if (G > n_pic) G = n_pic; if (G < 1) G = 1; }
  std::vector<hop_ctx*> gviews; std::vector<HipBackend*> gbe(1, &be);
  if (G > 1) {
    if (!c->stash) { HIPCHK(c, hipMalloc((void**)&c->stash, (size_t)STASH_SLOTS * STASH_SAMPLES * 2)); c->stash_slots = STASH_SLOTS; }
    int ks = 4; if (const char* e = getenv("HOP_SPINE_STREAMS")) ks = atoi(e);
    bool ok = true;
    for (int g = 1; g < G && ok; g++) {
      hop_ctx* v = nullptr;
      if (hop_ctx_create_view(c, &v) != HOP_OK) { ok = false; break; }
      gviews.push_back(v);
      HipBackend* b = new HipBackend(v); gbe.push_back(b);
      ok = b->ok() && (c->slots <= 0 || p->plain_intra || ks <= 1 || b->add_streams(ks > 8 ? 8 : ks));
    }
    if (!ok) { for (size_t g = 1; g < gbe.size(); g++) delete gbe[g]; for (auto v : gviews) hop_ctx_destroy(v); return hop_set_err(c, HOP_ERR_DEVICE, "hop_encode_frame: could not create the %d picture groups", G); }
  }
  std::vector<int> g_off(G + 1, 0); for (int g = 0; g < G; g++) g_off[g + 1] = g_off[g] + n_pic / G + (g < n_pic % G ? 1 : 0);
  std::vector<hopspine::Encoder*> encs;
  for (int k = 0, g = 0; k < n_pic; k++) { while (k >= g_off[g + 1]) g++; cfg.y_origin = k * c->sub_pitch; encs.push_back(new hopspine::Encoder(cfg, G > 1 ? (hopspine::Backend*)gbe[g] : (hopspine::Backend*)use)); }
  hopspine::Encoder& enc = *encs[0];
  std::vector<FILE*> tfs;                                                  // candidate traces: trace_path, for the pictures of a stack trace_path.<k>
  if (p->trace_path && p->trace_path[0]) for (int k = 0; k < n_pic; k++) {
    std::string name = p->trace_path; if (n_pic > 1) name += "." + std::to_string(k);
    FILE* f = fopen(name.c_str(), "w"); tfs.push_back(f); encs[k]->trace = f;
  }
  int rc = HOP_OK;
  try {
    if (G > 1) {
      const int cols = enc.n_ctu() / ((pic_h + 63) / 64), rows = (pic_h + 63) / 64, lag = p->wavefront_lag > cols ? cols : p->wavefront_lag, rif = (cols + lag - 1) / lag + 1, lanes_per_pic = rif < rows ? rif : rows;
      int hw = (int)std::thread::hardware_concurrency(); if (hw > 16) hw = 16; if (hw < 1) hw = 1;
      const int threads = hw / G > 2 ? hw / G : 2;
      be.begin_frame();                                                    // once for all groups: the SS reference back to the sentinel
      std::vector<int> grc(G, HOP_OK); std::vector<std::thread> th;
      for (int g = 0; g < G; g++) th.emplace_back([&, g]() {
        (void)hipSetDevice(c->device);
        try { hopspine::Encoder::encode_pictures_wavefront(encs.data() + g_off[g], g_off[g + 1] - g_off[g], gbe[g], p->wavefront_lag, g_off[g] * lanes_per_pic, false, threads); }
        catch (const Bail& b) { grc[g] = b.code; } catch (...) { grc[g] = HOP_ERR_STATE; }
      });
      for (auto& t : th) t.join();
      for (int g = 0; g < G; g++) {
        const hopspine::Encoder& e0 = *encs[g_off[g]];
        g_stat_calls[14] += (double)e0.batch_rounds; g_stat_calls[15] += (double)e0.batch_requests; g_stat_ms[14] += e0.batch_serve_s * 1e3; if (e0.batch_run_s * 1e3 > g_stat_ms[13]) g_stat_ms[13] = e0.batch_run_s * 1e3;
        if (grc[g] != HOP_OK && rc == HOP_OK) { rc = grc[g]; if (g > 0 && gviews[g - 1]->err[0] && !c->err[0]) strncpy(c->err, gviews[g - 1]->err, sizeof(c->err) - 1); }
      }
      if (rc != HOP_OK && !c->err[0]) rc = HOP_ERR_STATE;
    }
    else if (n_pic > 1) { hopspine::Encoder::encode_pictures_wavefront(encs.data(), n_pic, use, p->wavefront_lag); g_stat_calls[14] = (double)enc.batch_rounds; g_stat_calls[15] = (double)enc.batch_requests; g_stat_ms[14] = enc.batch_serve_s * 1e3; g_stat_ms[13] = enc.batch_run_s * 1e3; }
    else if (!lanes.empty()) enc.encode_frame_wavefront_direct(lanes.data(), (int)lanes.size(), p->wavefront_lag);
    else if (p->wavefront_lag > 0) { enc.encode_frame_wavefront(use, p->wavefront_lag); g_stat_calls[14] = (double)enc.batch_rounds; g_stat_calls[15] = (double)enc.batch_requests; g_stat_ms[14] = enc.batch_serve_s * 1e3; g_stat_ms[13] = enc.batch_run_s * 1e3; }
    else enc.encode_frame(p->first_ctus);
  } catch (const Bail& b) { rc = b.code; } catch (...) { rc = c->err[0] ? HOP_ERR_DEVICE : HOP_ERR_STATE; }
  delete lg;
  for (FILE* f : tfs) if (f) fclose(f);
  for (size_t g = 1; g < gbe.size(); g++) delete gbe[g];
  for (auto v : gviews) hop_ctx_destroy(v);
  for (auto b : vbe) delete b;
  for (auto v : views) { if (rc != HOP_OK && v->err[0] && !c->err[0]) strncpy(c->err, v->err, sizeof(c->err) - 1); hop_ctx_destroy(v); }
  for (auto e : encs) { g_stat_calls[10] += (double)e->reach_below.load(); g_stat_calls[11] += (double)e->reach_above.load(); }   // the visibility check of the wavefront (hop_spine.h)
  g_stat_ms[10] = (double)enc.first_below.load(); g_stat_ms[11] = (double)enc.first_above.load();
  if (rc == HOP_OK) {
    const int n = enc.n_ctu();
    if (n_candidates) *n_candidates = 0;
    free(c->rd_fraction); c->rd_fraction = (uint16_t*)malloc((size_t)n_pic * n * sizeof(uint16_t)); c->rd_fraction_n = c->rd_fraction ? n_pic * n : 0;
    for (int k = 0; k < n_pic; k++) {                                     // picture k's results at [k * n, (k + 1) * n)
      const hopspine::Encoder& e = *encs[k];
      if (ctu_cost) memcpy(ctu_cost + (size_t)k * n, e.ctu_cost.data(), n * sizeof(double));
      if (ctu_bits) memcpy(ctu_bits + (size_t)k * n, e.ctu_bits.data(), n * 4);
      if (ctu_dist) memcpy(ctu_dist + (size_t)k * n, e.ctu_dist.data(), n * 4);
      if (parts) memcpy(parts + (size_t)k * n * 256, e.pic.data(), e.pic.size() * sizeof(hopspine::Part));
      if (c->rd_fraction) memcpy(c->rd_fraction + (size_t)k * n, e.ctu_rd_fraction.data(), n * sizeof(uint16_t));
      if (n_candidates) *n_candidates += e.n_candidates;
    }
  }
  for (auto e : encs) delete e;
  return rc;
}

}  // extern "C"
