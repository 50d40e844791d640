// k_sao.hip -- the two picture-wide passes of the SAO encoder, gfx950.
//
// replaces: TEncSampleAdaptiveOffset::getStatistics / getBlkStats (TLibEncoder/TEncSampleAdaptiveOffset.cpp:305-352, :862-1383) and TComSampleAdaptiveOffset::offsetCTU /
// offsetBlock (TLibCommon/TComSampleAdaptiveOffset.cpp:365-707); the decision between them is host logic (host/hop_sao.cpp).
//
// The reference walks a CTU's lines with running sign buffers; per sample the rule is: a sample belongs to an edge type when both its neighbours along the type's direction
// lie inside the picture (one slice, one tile), its class is sgn(s - a) + sgn(s - b); the statistics leave out the columns / rows the right / lower neighbour CTU's
// deblocking could still change (5 / 4 luma, 3 / 2 chroma: SAOLcuBoundary 0).
//   k_sao_stats   a workgroup per CTU and component: 256 threads stride over the CTU's samples (rows contiguous: coalesced 2-byte loads of the deblocked and the original
//                 plane, neighbours from the same rows +-1 out of L2), 5 x 32 (count, sum) pairs accumulated in lane-private LDS columns, written once per workgroup.
//   k_sao_apply   a thread per sample: reads the untouched copy, writes the picture; the CTU's parameters (36 B per component) through the scalar cache.
// Both are HBM-bound: 4 B per sample in (statistics), 2 + 2 B per sample (offsetting) plus the copy; DESIGN.md section 4.
#include "hop_dev.h"
#include <math.h>
#include <vector>

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hop_set_err((c), HOP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

struct SaoGeo { int w, h, wctu, n_ctu, pitch_rows, bd; };

__device__ __forceinline__ int sao_sgn(int v) { return (v > 0) - (v < 0); }

// Every group of 4 lanes owns a private column of the histogram in LDS -- 20 edge bins (4 types x 5 classes) + 32 bands, each a packed (count << 20) + sum word: a column
// sees at most 64 samples, so the sum stays inside +-2^19 at 8 and 10 bit -- laid out bin-major ([bin][column], rows 17 words apart), so the five ds_add per sample of a
// wave spread over 16 addresses per bin instead of one (the first version kept one histogram per workgroup and spent its time in LDS atomics on a handful of hot bins:
// 206 GB/s; a column per lane removed the contention but made clearing and reducing 54 KB per workgroup the larger part of the work).  At the end each
// (bin, wave) row is unpacked and summed by one thread.
#define SAO_BINS 52
#define SAO_COLS 16      /* columns per wave: 4 lanes share one */
#define SAO_ROW 17
__global__ __launch_bounds__(256) void k_sao_stats(SaoGeo g, const int16_t* __restrict__ rec_y, const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr,
                                                   const int16_t* __restrict__ org_y, const int16_t* __restrict__ org_cb, const int16_t* __restrict__ org_cr, int32_t* __restrict__ stats) {
  __shared__ int32_t priv[4][SAO_BINS][SAO_ROW];
  __shared__ int32_t sums[4][SAO_BINS][2];
  const int ctu = blockIdx.x, comp = blockIdx.y, pic = blockIdx.z, sh = comp ? 1 : 0;
  const int pw = g.w >> sh, ph = g.h >> sh, cs = 64 >> sh;
  const int x0 = (ctu % g.wctu) * cs, y0 = (ctu / g.wctu) * cs;
  const int bw = min(cs, pw - x0), bh = min(cs, ph - y0);
  const int end_x = (x0 + cs < pw) ? bw - (comp ? 3 : 5) : bw, end_y = (y0 + cs < ph) ? bh - (comp ? 2 : 4) : bh;
  const size_t plane_off = (size_t)pic * (g.pitch_rows >> sh) * pw;
  const int16_t* src = (comp == 0 ? rec_y : comp == 1 ? rec_cb : rec_cr) + plane_off;
  const int16_t* org = (comp == 0 ? org_y : comp == 1 ? org_cb : org_cr) + plane_off;
  const int wave = threadIdx.x >> 6, lane = (threadIdx.x & 63) >> 2;
  for (int i = threadIdx.x; i < 4 * SAO_BINS * SAO_ROW; i += 256) (&priv[0][0][0])[i] = 0;
  __syncthreads();
  int32_t (*mine)[SAO_ROW] = priv[wave];
  for (int i = threadIdx.x; i < cs * end_y; i += 256) {
    const int x = i & (cs - 1), y = i / cs;
    if (x >= end_x) continue;
    const int X = x0 + x, Y = y0 + y;
    const int16_t* p = src + (size_t)Y * pw + X;
    const int s = p[0], d = org[(size_t)Y * pw + X] - s, add = (1 << 20) + d;
    const bool l = X > 0, r = X + 1 < pw, u = Y > 0, b = Y + 1 < ph;
    if (l && r) atomicAdd(&mine[0 * 5 + 2 + sao_sgn(s - p[-1]) + sao_sgn(s - p[1])][lane], add);
    if (u && b) atomicAdd(&mine[1 * 5 + 2 + sao_sgn(s - p[-pw]) + sao_sgn(s - p[pw])][lane], add);
    if (l && r && u && b) {
      atomicAdd(&mine[2 * 5 + 2 + sao_sgn(s - p[-pw - 1]) + sao_sgn(s - p[pw + 1])][lane], add);
      atomicAdd(&mine[3 * 5 + 2 + sao_sgn(s - p[-pw + 1]) + sao_sgn(s - p[pw - 1])][lane], add);
    }
    atomicAdd(&mine[20 + (s >> (g.bd - 5))][lane], add);                    // (ds_add without return: nothing waits for it)
  }
  __syncthreads();
  if (threadIdx.x < 4 * SAO_BINS) {
    const int wv = threadIdx.x / SAO_BINS, bin = threadIdx.x % SAO_BINS;
    int cnt = 0, dif = 0;
    for (int k = 0; k < SAO_COLS; k++) { const int v = priv[wv][bin][k]; const int c = (v + (1 << 19)) >> 20; cnt += c; dif += v - (c << 20); }
    sums[wv][bin][0] = cnt; sums[wv][bin][1] = dif;
  }
  __syncthreads();
  int32_t* out = stats + (((size_t)pic * g.n_ctu + ctu) * 3 + comp) * 5 * 32 * 2;
  for (int i = threadIdx.x; i < 5 * 32 * 2; i += 256) {
    const int t = i >> 6, cls = (i >> 1) & 31, which = i & 1;
    const int bin = t < 4 ? (cls < 5 ? t * 5 + cls : -1) : 20 + cls;
    out[i] = bin < 0 ? 0 : sums[0][bin][which] + sums[1][bin][which] + sums[2][bin][which] + sums[3][bin][which];
  }
}

__global__ __launch_bounds__(256) void k_sao_apply(SaoGeo g, const hop_sao_param* __restrict__ params, const int16_t* __restrict__ src_y, const int16_t* __restrict__ src_cb,
                                                   const int16_t* __restrict__ src_cr, int16_t* __restrict__ dst_y, int16_t* __restrict__ dst_cb, int16_t* __restrict__ dst_cr) {
  const int comp = blockIdx.y, pic = blockIdx.z, sh = comp ? 1 : 0;
  const int pw = g.w >> sh, ph = g.h >> sh, cs = 64 >> sh;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= pw * ph) return;
  const int X = i % pw, Y = i / pw;
  const hop_sao_param* p = params + (((size_t)pic * g.n_ctu + (size_t)(Y / cs) * g.wctu + X / cs) * 3 + comp);
  if (p->mode == 0) return;
  const size_t plane_off = (size_t)pic * (g.pitch_rows >> sh) * pw;
  const int16_t* s0 = (comp == 0 ? src_y : comp == 1 ? src_cb : src_cr) + (size_t)pic * ph * pw + (size_t)Y * pw + X;      // the copy is packed picture after picture
  int16_t* d0 = (comp == 0 ? dst_y : comp == 1 ? dst_cb : dst_cr) + plane_off + (size_t)Y * pw + X;
  const int s = s0[0], t = p->type;
  int o;
  if (t < 4) {
    const int dx = t == 1 ? 0 : t == 3 ? -1 : 1, dy = t == 0 ? 0 : 1;
    const int ax = X - dx, ay = Y - dy, bx = X + dx, by = Y + dy;
    if (ax < 0 || ax >= pw || ay < 0 || bx < 0 || bx >= pw || by >= ph) return;
    o = p->offset[2 + sao_sgn(s - s0[-dy * pw - dx]) + sao_sgn(s - s0[dy * pw + dx])];
  } else o = p->offset[s >> (g.bd - 5)];
  const int v = s + o, maxv = (1 << g.bd) - 1;
  d0[0] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
}

static int sao_geo(hop_ctx* c, SaoGeo& g, int& n_pic, const char* who) {
  if (c->bd_y != c->bd_c) return hop_set_err(c, HOP_ERR_ARG, "%s: luma and chroma bit depths must be equal", who);
  n_pic = c->sub_pitch ? (c->pic_h - c->sub_h) / c->sub_pitch + 1 : 1;
  const int h = c->sub_pitch ? c->sub_h : c->pic_h;
  if ((c->pic_w & 7) || (h & 7)) return hop_set_err(c, HOP_ERR_ARG, "%s: the picture size must be a multiple of the minimum CU size (8)", who);
  g.w = c->pic_w; g.h = h; g.wctu = (g.w + 63) >> 6; g.n_ctu = g.wctu * ((h + 63) >> 6); g.pitch_rows = c->sub_pitch; g.bd = c->bd_y;
  return HOP_OK;
}

extern "C" int hop_sao_stats(hop_ctx* c, int32_t* stats) {
  if (!c || !stats) return hop_set_err(c, HOP_ERR_ARG, "hop_sao_stats: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_sao_stats: hop_upload_orig has not been called");
  SaoGeo g; int n_pic; const int rc = sao_geo(c, g, n_pic, "hop_sao_stats"); if (rc != HOP_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)n_pic * g.n_ctu * 3 * 5 * 32 * 2;
  int32_t* d = nullptr;
  HIPCHK(c, hipMalloc((void**)&d, n * sizeof(int32_t)));
  const int rk = hop_prof_begin(c, HOP_K_SAO, (uint64_t)n_pic * g.n_ctu);
  hipLaunchKernelGGL(k_sao_stats, dim3(g.n_ctu, 3, n_pic), dim3(256), 0, c->stream, g, c->rec[0], c->rec[1], c->rec[2], c->org_y, c->org_cb, c->org_cr, d);
  hop_prof_end(c, rk);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(stats, d, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_sao_stats: %s", hipGetErrorString(e));
  return HOP_OK;
}

extern "C" int hop_sao_apply(hop_ctx* c, const hop_sao_param* recon) {
  if (!c || !recon) return hop_set_err(c, HOP_ERR_ARG, "hop_sao_apply: bad argument");
  SaoGeo g; int n_pic; const int rc = sao_geo(c, g, n_pic, "hop_sao_apply"); if (rc != HOP_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t ny = (size_t)g.w * g.h, ncc = ny >> 2, n_par = (size_t)n_pic * g.n_ctu * 3;
  int16_t* copy = nullptr; hop_sao_param* d_par = nullptr;
  HIPCHK(c, hipMalloc((void**)&copy, (ny + 2 * ncc) * n_pic * sizeof(int16_t)));
  hipError_t e = hipMalloc((void**)&d_par, n_par * sizeof(hop_sao_param));
  int16_t* cy = copy; int16_t* ccb = copy + ny * n_pic; int16_t* ccr = ccb + ncc * n_pic;
  for (int k = 0; k < n_pic && e == hipSuccess; k++) {                    // the untouched picture(s), packed (the reference's m_tempPicYuv)
    e = hipMemcpyAsync(cy + ny * k, c->rec[0] + (size_t)k * g.pitch_rows * g.w, ny * 2, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ccb + ncc * k, c->rec[1] + (size_t)k * (g.pitch_rows >> 1) * (g.w >> 1), ncc * 2, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ccr + ncc * k, c->rec[2] + (size_t)k * (g.pitch_rows >> 1) * (g.w >> 1), ncc * 2, hipMemcpyDeviceToDevice, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(d_par, recon, n_par * sizeof(hop_sao_param), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const int rk = hop_prof_begin(c, HOP_K_SAO, (uint64_t)n_pic * g.n_ctu);
    hipLaunchKernelGGL(k_sao_apply, dim3((unsigned)((ny + 255) / 256), 3, n_pic), dim3(256), 0, c->stream, g, d_par, cy, ccb, ccr, c->rec[0], c->rec[1], c->rec[2]);
    hop_prof_end(c, rk);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(copy); (void)hipFree(d_par);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_sao_apply: %s", hipGetErrorString(e));
  return HOP_OK;
}

extern "C" int hop_sao_frame(hop_ctx* c, const hop_sao_params* p, hop_sao_param* coded) {
  if (!c || !p || !coded) return hop_set_err(c, HOP_ERR_ARG, "hop_sao_frame: bad argument");
  SaoGeo g; int n_pic; int rc = sao_geo(c, g, n_pic, "hop_sao_frame"); if (rc != HOP_OK) return rc;
  const size_t per_pic = (size_t)g.n_ctu * 3;
  int32_t* stats = (int32_t*)malloc((size_t)n_pic * per_pic * 5 * 32 * 2 * sizeof(int32_t));
  hop_sao_param* recon = (hop_sao_param*)malloc((size_t)n_pic * per_pic * sizeof(hop_sao_param));
  if (!stats || !recon) { free(stats); free(recon); return hop_set_err(c, HOP_ERR_DEVICE, "hop_sao_frame: out of host memory"); }
  rc = hop_sao_stats(c, stats);
  for (int k = 0; k < n_pic && rc == HOP_OK; k++) {                       // every picture of a stack is decided as a picture of its own
    hop_sao_params pk = *p;                                               // a stack coded by hop_encode_frame on this context: every picture starts from the fraction ITS last CTU left
    if (n_pic > 1 && c->rd_fraction && c->rd_fraction_n == n_pic * g.n_ctu) pk.rd_fraction = c->rd_fraction[(size_t)k * g.n_ctu + g.n_ctu - 1];
    rc = hop_sao_decide(g.n_ctu, g.wctu, g.bd, stats + (size_t)k * per_pic * 5 * 32 * 2, &pk, coded + (size_t)k * per_pic, recon + (size_t)k * per_pic);
    if (rc != HOP_OK) hop_set_err(c, rc, "hop_sao_frame: hop_sao_decide refused its arguments");
  }
  if (rc == HOP_OK) rc = hop_sao_apply(c, recon);
  free(stats); free(recon);
  return rc;
}

// ---- PSNR: the sums of squared differences between the resident original and the reconstruction ----
// replaces: the three loops of TEncGOP::xCalculateAddPSNR (TLibEncoder/TEncGOP.cpp:2383-2456).  One pass over both pictures (4 B per sample), a wave-level reduction and
// one 64-bit atomic per workgroup and plane.
__global__ __launch_bounds__(256) void k_ssd(SaoGeo g, const int16_t* __restrict__ rec_y, const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr,
                                             const int16_t* __restrict__ org_y, const int16_t* __restrict__ org_cb, const int16_t* __restrict__ org_cr, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[4];
  const int comp = blockIdx.y, pic = blockIdx.z, sh = comp ? 1 : 0;
  const int pw = g.w >> sh, ph = g.h >> sh;
  const size_t off = (size_t)pic * (g.pitch_rows >> sh) * pw, n = (size_t)pw * ph;
  const int16_t* a = (comp == 0 ? rec_y : comp == 1 ? rec_cb : rec_cr) + off;
  const int16_t* b = (comp == 0 ? org_y : comp == 1 ? org_cb : org_cr) + off;
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const int d = (int)b[i] - (int)a[i]; s += (unsigned long long)(d * d); }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[pic * 3 + comp], part[0] + part[1] + part[2] + part[3]);
}

extern "C" int hop_psnr(hop_ctx* c, uint64_t* ssd, double* psnr) {
  if (!c || (!ssd && !psnr)) return hop_set_err(c, HOP_ERR_ARG, "hop_psnr: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_psnr: hop_upload_orig has not been called");
  SaoGeo g; int n_pic; const int rc = sao_geo(c, g, n_pic, "hop_psnr"); if (rc != HOP_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIPCHK(c, hipMalloc((void**)&d, (size_t)n_pic * 3 * 8));
  hipError_t e = hipMemsetAsync(d, 0, (size_t)n_pic * 3 * 8, c->stream);
  std::vector<unsigned long long> h((size_t)n_pic * 3);
  if (e == hipSuccess) {
    const size_t n = (size_t)g.w * g.h; const unsigned blocks = (unsigned)((n / 256 / 8 < 1 ? 1 : n / 256 / 8) > 4096 ? 4096 : (n / 256 / 8 < 1 ? 1 : n / 256 / 8));
    hipLaunchKernelGGL(k_ssd, dim3(blocks, 3, n_pic), dim3(256), 0, c->stream, g, c->rec[0], c->rec[1], c->rec[2], c->org_y, c->org_cb, c->org_cr, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_psnr: %s", hipGetErrorString(e));
  const double maxv = (double)(255 << (g.bd - 8)), size = (double)g.w * g.h;
  for (int k = 0; k < n_pic * 3; k++) {
    if (ssd) ssd[k] = h[k];
    if (psnr) { const double ref = maxv * maxv * size / (k % 3 ? 4.0 : 1.0); psnr[k] = h[k] ? 10.0 * log10(ref / (double)h[k]) : 99.99; }   // :2449-2456
  }
  return HOP_OK;
}
