// k_ss_search.hip -- SS integer full search.
// Replaces TEncSearch::xPatternSearch (TLibEncoder/TEncSearch.cpp:6262-6371) with the SAD family
// (TLibCommon/TComRdCost.cpp:513-1011), isValidPattern (:444-458) and getCost (TComRdCost.h:185-192).
//
// Mapping to CDNA4
//   * one workgroup (4 waves) = one tile of 128 x 32 displacements of one PU;
//     one wave = a strip of 128 x NP(8) displacements; lane l owns displacements x = 2l, 2l+1.
//   * the reference window of the tile ((128+W) x (32+H-1) samples) is staged once in LDS, biased by
//     +1 so that the -1 sentinel becomes 0 and every sample is an unsigned 16-bit value: |a-b| is
//     unchanged and v_sad_u16 (2 abs-diff-accumulate per lane per instruction) applies directly.
//   * the original block is wave-uniform: it is fetched with scalar loads (s_load) straight from the
//     resident original picture and fed to v_sad_u16 as an SGPR operand -- no LDS, no VGPR traffic.
//   * each LDS row a lane reads is reused for NP displacements (NP/2 with FEN row subsampling) out
//     of a rotating register window, so the kernel is VALU-bound (v_sad_u16), not LDS-bound.
//   * argmin in the reference's scan order (y outer, x inner, strict '<') = minimum of the 64-bit key
//     cost<<32 | dy<<16 | dx: per-lane, then per-wave (DPP shuffles), then one atomicMin per wave.
#include "hop_dev.h"

#define SS_NP 8                 // displacement rows per wave
#define SS_TW 128               // displacement columns per tile
#define SS_TH (4 * SS_NP)       // displacement rows per tile (4 waves)
#define SS_MAXW 64
#define SS_LS (SS_TW + SS_MAXW + 8)              // LDS row pitch in samples (even)
#define SS_ROWS (SS_TH + SS_MAXW - 1)            // LDS rows
#define SS_MAX_TILES_X 3                         // window <= 257 wide
#define SS_MAX_TILES_Y 9                         // window <= 257 tall  -> 9 tiles of 32 (8.03)
#define SS_MAX_TILES (SS_MAX_TILES_X * SS_MAX_TILES_Y)

__device__ static inline uint32_t sad_u16x2(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_u16(a, b, acc);    // |a.lo-b.lo| + |a.hi-b.hi| + acc
}

typedef unsigned short hop_us2 __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t bias_pk(uint32_t v) {   // two independent 16-bit +1 (0xFFFF wraps to 0 inside its half)
  hop_us2 a = __builtin_bit_cast(hop_us2, v);
  a += hop_us2{1, 1};
  return __builtin_bit_cast(uint32_t, a);
}

template <int STEP>
__device__ static inline void ss_strip(const uint16_t* __restrict__ tile, const int16_t* __restrict__ org, int org_stride,
                                       int W, int H, int wave, int lane, uint32_t (&acc_e)[SS_NP], uint32_t (&acc_o)[SS_NP]) {
  // tile row 0 = first displacement row of the workgroup; this wave starts at row wave*NP
  const uint16_t* base = tile + (size_t)(wave * SS_NP) * SS_LS + 2 * lane;
  const int npairs = W >> 1;
  for (int cp = 0; cp < npairs; cp++) {
    uint32_t e[SS_NP], o[SS_NP];
    const uint16_t* col = base + 2 * cp;
#pragma unroll
    for (int j = 0; j < SS_NP; j++) {            // rows 0..NP-1 of the window
      uint32_t w0 = *(const uint32_t*)(col + (size_t)j * SS_LS);
      uint32_t w1 = *(const uint32_t*)(col + (size_t)j * SS_LS + 2);
      e[j] = w0; o[j] = __builtin_amdgcn_alignbit(w1, w0, 16);
    }
    const uint32_t* orow = (const uint32_t*)(org + 2 * cp);   // wave-uniform address -> scalar loads
    const int org_stride_dw = org_stride >> 1;
    for (int rb = 0; rb < H; rb += SS_NP) {
#pragma unroll
      for (int k = 0; k < SS_NP / STEP; k++) {
        const int r = rb + k * STEP;             // original row (uniform)
        if (r < H) {
          uint32_t ov = orow[(size_t)r * org_stride_dw] + 0x00010001u;   // bias +1 (no carry: samples <= 4095)
#pragma unroll
          for (int j = 0; j < SS_NP; j++) {
            const int slot = (k * STEP + j) % SS_NP;
            acc_e[j] = sad_u16x2(e[slot], ov, acc_e[j]);
            acc_o[j] = sad_u16x2(o[slot], ov, acc_o[j]);
          }
        }
#pragma unroll
        for (int s = 0; s < STEP; s++) {         // slide: rows r+NP+s replace rows r+s
          const int row = r + SS_NP + s;
          const int slot = (k * STEP + s) % SS_NP;
          if (row < H + SS_NP - 1) {
            uint32_t w0 = *(const uint32_t*)(col + (size_t)row * SS_LS);
            uint32_t w1 = *(const uint32_t*)(col + (size_t)row * SS_LS + 2);
            e[slot] = w0; o[slot] = __builtin_amdgcn_alignbit(w1, w0, 16);
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_ss_search(const hop_pu_job* __restrict__ jobs, hop_pics pic, unsigned long long* __restrict__ best_key) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[SS_ROWS * SS_LS];
  const int jidx = blockIdx.x / SS_MAX_TILES, t = blockIdx.x % SS_MAX_TILES;
  const hop_pu_job jb = jobs[jidx];
  const int W = jb.w, H = jb.h;
  const int win_w = jb.rng_right - jb.rng_left + 1, win_h = jb.rng_bottom - jb.rng_top + 1;
  if (win_w <= 0 || win_h <= 0) return;
  // tiles start on an even absolute column so that the staging loads are 4-byte aligned
  const int xa = (jb.pu_x + jb.rng_left) & ~1;                    // absolute column of tile column 0 (pu_x is a multiple of 4)
  const int x_first = xa - jb.pu_x;                               // displacement of tile column 0 (<= rng_left)
  const int tiles_x = (jb.rng_right - x_first + SS_TW) / SS_TW, tiles_y = (win_h + SS_TH - 1) / SS_TH;
  if (t >= tiles_x * tiles_y) return;
  const int tx = t % tiles_x, ty = t / tiles_x;
  const int dx0 = x_first + tx * SS_TW, dy0 = jb.rng_top + ty * SS_TH;   // displacement of tile origin
  // ---- stage the reference window, biased by +1 ----
  const int rows = min(SS_TH, jb.rng_bottom - dy0 + 1) + H - 1;
  const int cols = (min(SS_TW, jb.rng_right - dx0 + 1) + W + 3) & ~1;   // even; lanes beyond the window read stale LDS and are discarded
  const int16_t* src = pic.ss_y + (ptrdiff_t)(jb.pu_y + dy0) * pic.stride_y + (jb.pu_x + dx0);
  const int cw = cols >> 1;
  for (int i = threadIdx.x; i < rows * cw; i += 256) {
    int r = i / cw, cdw = i - r * cw;
    uint32_t v = *(const uint32_t*)(src + (ptrdiff_t)r * pic.stride_y + 2 * cdw);
    *(uint32_t*)(tile + (size_t)r * SS_LS + 2 * cdw) = bias_pk(v);        // per-half +1 (v_pk_add_u16): -1 -> 0 without a carry into the neighbour
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dy0 + wave * SS_NP > jb.rng_bottom) return;                  // whole strip outside the window (uniform per wave)
  uint32_t acc_e[SS_NP], acc_o[SS_NP];
#pragma unroll
  for (int j = 0; j < SS_NP; j++) { acc_e[j] = 0; acc_o[j] = 0; }
  const int16_t* org = pic.org_y + (size_t)jb.pu_y * pic.pic_w + jb.pu_x;
  const bool sub = (jb.flags & HOP_FLAG_FEN) && H > 8;             // TEncSearch.cpp:6303-6309
  if (sub) ss_strip<2>(tile, org, pic.pic_w, W, H, wave, lane, acc_e, acc_o);
  else     ss_strip<1>(tile, org, pic.pic_w, W, H, wave, lane, acc_e, acc_o);
  // ---- cost, validity, first-best ----
  const int shift_up = sub ? 1 : 0, shift_dn = pic.bd_y - 8;
  unsigned long long best = ~0ull;
#pragma unroll
  for (int j = 0; j < SS_NP; j++) {
    const int dy = dy0 + wave * SS_NP + j;
#pragma unroll
    for (int hlf = 0; hlf < 2; hlf++) {
      const int dx = dx0 + 2 * lane + hlf;
      bool ok = dx >= jb.rng_left && dx <= jb.rng_right && dy <= jb.rng_bottom;
      ok = ok && !((dx >= jb.off_x) && (dy > jb.off_y));           // :6328
      if (ok) {                                                    // isValidPattern, TComRdCost.cpp:444-458
        const int16_t* plb = pic.ss_y + (ptrdiff_t)(jb.pu_y + dy + H + 4) * pic.stride_y + (jb.pu_x + dx);
        ok = (plb[0] != HOP_NOT_VALID) && (plb[W + 4] != HOP_NOT_VALID);
      }
      if (ok) {
        uint32_t sad = ((hlf ? acc_o[j] : acc_e[j]) << shift_up) >> shift_dn;
        sad += hopd_mv_cost(jb.lambda_cost, dx, dy, 2, jb.pred_x, jb.pred_y);   // cost scale 2, :4560
        unsigned long long key = ((unsigned long long)sad << 32) | ((unsigned long long)(uint32_t)(dy - jb.rng_top) << 16) | (uint32_t)(dx - jb.rng_left);
        best = key < best ? key : best;
      }
    }
  }
  best = hopd_wave_min_u64(best);
  if (lane == 0 && best != ~0ull) atomicMin(best_key + jidx, best);
}

__global__ void k_ss_init(unsigned long long* best_key, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) best_key[i] = ~0ull;
}

__global__ void k_ss_finalize(const hop_pu_job* __restrict__ jobs, const unsigned long long* __restrict__ best_key, hop_pics pic,
                              const int16_t* __restrict__ ss_buf0, hop_pu_result* __restrict__ res, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const hop_pu_job jb = jobs[i];
  hop_pu_result r;
  unsigned long long key = best_key[i];
  for (int k = 0; k < 8; k++) r.gt[k] = 0;
  r.half[0] = r.half[1] = r.qter[0] = r.qter[1] = 0; r.frac_cost = 0; r.gt_flag = 0;
  r.half_final[0] = r.half_final[1] = r.qter_final[0] = r.qter_final[1] = 0;
  if (key == ~0ull) {                                              // no valid candidate, :6356-6360
    r.mv_int[0] = r.mv_int[1] = 0; r.sad = 0xFFFFFFFFu; r.not_valid = 1;
  } else {
    int dx = (int)(key & 0xFFFF) + jb.rng_left, dy = (int)((key >> 16) & 0xFFFF) + jb.rng_top;
    uint32_t cost = (uint32_t)(key >> 32);
    r.mv_int[0] = dx; r.mv_int[1] = dy;
    r.sad = cost - hopd_mv_cost(jb.lambda_cost, dx, dy, 2, jb.pred_x, jb.pred_y);   // :6365
    // :4603-4606: zero vector or first sample of the padded buffer still the sentinel
    r.not_valid = ((dx == 0 && dy == 0) || ss_buf0[0] == HOP_NOT_VALID) ? 1 : 0;
  }
  r.cost = r.sad;
  r.mv_final[0] = r.mv_int[0]; r.mv_final[1] = r.mv_int[1];
  res[i] = r;
}

int hop_launch_ss_search(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_res) {
  void* sc; int r = hop_scratch(c, (size_t)n * 8, &sc); if (r) return r;
  unsigned long long* keys = (unsigned long long*)sc;
  hop_pics pic = hop_make_pics(c);
  const int pr = hop_prof_begin(c, HOP_K_SS_SEARCH, (uint64_t)n);
  hipLaunchKernelGGL(k_ss_init, dim3((n + 255) / 256), dim3(256), 0, c->stream, keys, n);
  hipLaunchKernelGGL(k_ss_search, dim3((unsigned)n * SS_MAX_TILES), dim3(256), 0, c->stream, d_jobs, pic, keys);
  hipLaunchKernelGGL(k_ss_finalize, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, keys, pic, c->ss_buf[0], d_res, n);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "ss_search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
