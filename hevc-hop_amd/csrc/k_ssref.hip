// k_ssref.hip -- SS-reference upkeep in HBM.
//   reset : TComSlice::xGetRefPic sentinel fill -> TComPicYuv::setPicPel(NOT_VALID)
//           (TLibCommon/TComSlice.cpp:241-255, TComPicYuv.cpp:199-207)
//   commit: TEncCu::xCopyYuv2SSRef leaf (TLibEncoder/TEncCu.cpp:1677-1697) = copy the finalised CU's
//           reconstruction, then TComPicYuv::extendPicBorder (TComPicYuv.cpp:236-275).  The reference
//           re-extends every border of the picture per CU; a margin sample only ever mirrors ONE picture
//           sample (the nearest edge sample), so the same state is reached by rewriting just the margin
//           samples whose source lies in the committed block -- an incremental halo update.
// HBM-bound byte moving: coalesced 2-byte-element rows, no LDS needed (each byte is touched once).
#include "hop_dev.h"

__global__ void k_fill_sentinel(int16_t* __restrict__ p, size_t n) {
  // 0xFFFF = -1 in every Pel; 16-byte stores, grid-stride
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t n8 = n >> 3;
  uint4* p4 = (uint4*)p;
  uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (size_t k = i; k < n8; k += stride) p4[k] = v;
  for (size_t k = (n8 << 3) + i; k < n; k += stride) p[k] = (int16_t)-1;
}

int hop_launch_ssref_reset(hop_ctx* c) {
  // the guard rows around each plane are (re)filled too
  size_t ny = (size_t)c->stride_y * (c->pic_h + 2 * HOP_MARGIN_Y + 2 * HOP_GUARD_ROWS);
  size_t nc = (size_t)c->stride_c * ((c->pic_h >> 1) + 2 * HOP_MARGIN_C + 2 * HOP_GUARD_ROWS);
  hipLaunchKernelGGL(k_fill_sentinel, dim3(2048), dim3(256), 0, c->stream, c->ss_alloc[0], ny);
  hipLaunchKernelGGL(k_fill_sentinel, dim3(1024), dim3(256), 0, c->stream, c->ss_alloc[1], nc);
  hipLaunchKernelGGL(k_fill_sentinel, dim3(1024), dim3(256), 0, c->stream, c->ss_alloc[2], nc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "ssref reset launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// One plane of one CU.  p00 = sample (0,0) of the padded plane; (x0,y0,s) the block in this plane's
// units; pw,ph the plane size; m its margin.  src(r,c) = value of block sample (r,c).
// Work item = one block sample; the samples on a picture edge also write their margin replicas:
//   left/right edge sample  -> m replicas on its row
//   top/bottom edge sample  -> m replicas on its column
//   corner sample           -> the m x m corner area
// A stacked context (hop_ctx_set_stack) holds independent pictures of `ph` rows whose origins lie `sub_pitch` rows apart: the block's own picture
// is the one its first row falls into, and that picture's top and bottom edges are the ones that are extended (sub_pitch 0: one picture).
template <bool PACKED>
__device__ static void commit_plane(int16_t* __restrict__ p00, int stride, int pw, int ph, int sub_pitch, int m,
                                    int x0, int y0, int s, const int16_t* __restrict__ src, int src_pitch) {
  const int n = s * s;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int r = i / s, cidx = i - r * s;
    int16_t v = PACKED ? src[i] : src[(size_t)(y0 + r) * src_pitch + x0 + cidx];
    p00[(size_t)(y0 + r) * stride + x0 + cidx] = v;
  }
  __syncthreads();   // (not needed for correctness of the replicas below, which re-read src; keeps write order tidy)
  const int x1 = x0 + s - 1, y1 = y0 + s - 1;
  const int top = sub_pitch ? (y0 / sub_pitch) * sub_pitch : 0;           // first row of the block's picture
  const bool L = (x0 == 0), R = (x1 == pw - 1), T = (y0 == top), B = (y1 == top + ph - 1);
  int16_t* pt = p00 + (size_t)top * stride;                               // sample (0,0) of that picture: the replicas above and below it are written relative to it
  auto val = [&](int r, int cidx) -> int16_t { return PACKED ? src[r * s + cidx] : src[(size_t)(y0 + r) * src_pitch + x0 + cidx]; };
  if (L || R) {
    for (int i = threadIdx.x; i < s * m; i += blockDim.x) {
      int r = i / m, k = i - r * m;
      if (L) p00[(size_t)(y0 + r) * stride - m + k] = val(r, 0);
      if (R) p00[(size_t)(y0 + r) * stride + pw + k] = val(r, s - 1);
    }
  }
  if (T || B) {
    for (int i = threadIdx.x; i < s * m; i += blockDim.x) {
      int k = i / s, cidx = i - k * s;
      if (T) pt[-(ptrdiff_t)(k + 1) * stride + x0 + cidx] = val(0, cidx);
      if (B) pt[(size_t)(ph + k) * stride + x0 + cidx] = val(s - 1, cidx);
    }
  }
  if ((L || R) && (T || B)) {
    for (int i = threadIdx.x; i < m * m; i += blockDim.x) {
      int k = i / m, q = i - k * m;
      if (T && L) pt[-(ptrdiff_t)(k + 1) * stride - m + q] = val(0, 0);
      if (T && R) pt[-(ptrdiff_t)(k + 1) * stride + pw + q] = val(0, s - 1);
      if (B && L) pt[(size_t)(ph + k) * stride - m + q] = val(s - 1, 0);
      if (B && R) pt[(size_t)(ph + k) * stride + pw + q] = val(s - 1, s - 1);
    }
  }
}

// grid = (n CUs, 3 planes).  PACKED: rec_* hold the CU blocks back to back (rect[3] = luma offset);
// otherwise rec_* are whole reconstruction pictures with pitch pic_w / pic_w/2.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_ssref_commit(const int32_t* __restrict__ rect4, const int16_t* __restrict__ rec_y,
                                                      const int16_t* __restrict__ rec_cb, const int16_t* __restrict__ rec_cr,
                                                      int16_t* __restrict__ y00, int16_t* __restrict__ cb00, int16_t* __restrict__ cr00,
                                                      int pic_w, int pic_h, int sub_pitch, int stride_y, int stride_c) {
  const int cu = blockIdx.x, comp = blockIdx.y;
  const int x = rect4[4 * cu], y = rect4[4 * cu + 1], s = rect4[4 * cu + 2], off = rect4[4 * cu + 3];
  if (comp == 0) {
    commit_plane<PACKED>(y00, stride_y, pic_w, pic_h, sub_pitch, HOP_MARGIN_Y, x, y, s, PACKED ? rec_y + off : rec_y, pic_w);
  } else {
    const int16_t* src = comp == 1 ? rec_cb : rec_cr;
    commit_plane<PACKED>(comp == 1 ? cb00 : cr00, stride_c, pic_w >> 1, pic_h >> 1, sub_pitch >> 1, HOP_MARGIN_C, x >> 1, y >> 1, s >> 1,
                         PACKED ? src + (off >> 2) : src, pic_w >> 1);
  }
}

int hop_launch_ssref_commit(hop_ctx* c, int n, const int32_t* d_rect4, const int16_t* d_y, const int16_t* d_cb, const int16_t* d_cr, int packed) {
  dim3 grid(n, 3), block(256);
  const int pr = hop_prof_begin(c, HOP_K_COMMIT, (uint64_t)n);
  if (packed)
    hipLaunchKernelGGL(k_ssref_commit<true>, grid, block, 0, c->stream, d_rect4, d_y, d_cb, d_cr, c->ss00[0], c->ss00[1], c->ss00[2], c->pic_w, c->sub_pitch ? c->sub_h : c->pic_h, c->sub_pitch, c->stride_y, c->stride_c);
  else
    hipLaunchKernelGGL(k_ssref_commit<false>, grid, block, 0, c->stream, d_rect4, d_y, d_cb, d_cr, c->ss00[0], c->ss00[1], c->ss00[2], c->pic_w, c->sub_pitch ? c->sub_h : c->pic_h, c->sub_pitch, c->stride_y, c->stride_c);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "ssref commit launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
