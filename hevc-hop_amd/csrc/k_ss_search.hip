// k_ss_search.hip -- SS integer full search.
// Replaces TEncSearch::xPatternSearch (TLibEncoder/TEncSearch.cpp:6262-6371) with the SAD family
// (TLibCommon/TComRdCost.cpp:513-1011), isValidPattern (:444-458) and getCost (TComRdCost.h:185-192).
//
// Mapping to CDNA4 (integer-VALU bound on v_sad_u16; HBM sees each window once, via L2)
//   * one workgroup (4 waves) = one tile of displacements of one PU.
//     MAIN tiles: 128 x 32 displacements; a wave = a strip of 128 x NP(8); lane l owns x = 2l, 2l+1.
//     EDGE tiles: the +-128 window is 257 (or 258) wide, so after two main tile columns 1-2 columns remain;
//     they are searched with the transposed mapping (lane = displacement ROW, 64 rows per wave, both
//     columns per lane) so that they cost 1/16 of a main strip instead of a whole one.
//   * the reference window of the tile is staged once in LDS, biased by +1 so that the -1 sentinel becomes 0
//     and every sample is an unsigned 16-bit value: |a-b| is unchanged and v_sad_u16 (2 abs-diff-accumulate
//     per lane per instruction) applies directly.  The row pitch is an odd number of dwords, so both the
//     row-wise (main) and the column-wise (edge) accesses are bank-conflict free.
//   * the original block is staged too, transposed to [column pair][used row] and biased, so the inner loop
//     fetches 4 rows of one column pair with one broadcast ds_read_b128 (wave-uniform address).
//   * main strips: each LDS row a lane reads is reused for NP displacements (NP/2 with FEN row subsampling)
//     out of a rotating register window.
//   * strips that lie completely in the region the reference rejects by rule (x >= offX' && y > offY',
//     :6328) or outside the window are skipped: the reference computes their SAD and throws it away.
//   * argmin in the reference's scan order (y outer, x inner, strict '<') = minimum of the 64-bit key
//     cost<<32 | dy<<16 | dx: per-lane, then per-wave, then one atomicMin per wave.
#include <algorithm>
#include "hop_dev.h"

#define SS_NP 8                 // displacement rows per wave (main)
#define SS_TW 128               // displacement columns per main tile
#define SS_TH_MIN (4 * SS_NP)   // displacement rows per main tile for H = 64; smaller PUs get taller tiles (ss_tile_h)
#define SS_MAXW 64
#define SS_LS 202               // LDS row pitch in samples: even, and 101 dwords (odd) -> conflict-free both ways
#define SS_ROWS (SS_TH_MIN + SS_MAXW - 1)        // LDS rows of a main tile (95)
// displacement rows of a main tile: as many strips of NP rows as the 95 LDS rows hold next to the block height, so that
// small PUs (whose strips are short) amortise the staging of a tile over more strips; a wave loops over strips w, w+4, ...
__device__ static inline int ss_tile_h(int H) { return ((SS_ROWS - (H - 1)) / SS_NP) * SS_NP; }
#define SS_EDGE_ROWS 128        // displacement rows per edge tile (2 active waves x 64 lanes)
#define SS_EDGE_LS 70           // pitch of an edge tile: W + 2 + 2 samples, 35 dwords (odd)
#define SS_PROBE 5              // extra staged rows: the validity probes sit at row dy + H + 4 (isValidPattern)
#define SS_TILE_ELEMS ((SS_ROWS + SS_PROBE) * SS_LS)   // 20200 >= (128 + 63 + 5) * 70 = 13720
#define SS_MAX_TILES 27         // 3 x 9 main tiles (H = 64), or 2 x 9 main + 3 edge

typedef unsigned short hop_us2 __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t bias_pk(uint32_t v) {   // two independent 16-bit +1 (0xFFFF wraps to 0 inside its half)
  hop_us2 a = __builtin_bit_cast(hop_us2, v);
  a += hop_us2{1, 1};
  return __builtin_bit_cast(uint32_t, a);
}
__device__ static inline uint32_t sad_u16x2(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_u16(a, b, acc);    // |a.lo-b.lo| + |a.hi-b.hi| + acc
}

// ---- main strip: lane = 2 columns, NP rows, rotating window ----
template <int STEP>
__device__ static inline void ss_strip(const uint16_t* __restrict__ tile, const uint32_t* __restrict__ orgT, int HS,
                                       int W, int H, int strip, int lane, uint32_t (&acc_e)[SS_NP], uint32_t (&acc_o)[SS_NP]) {
  // tile row 0 = first displacement row of the workgroup; this strip starts at row strip*NP
  const uint16_t* base = tile + (size_t)(strip * SS_NP) * SS_LS + 2 * lane;
  const int npairs = W >> 1;
  constexpr int PER = SS_NP / STEP;               // original rows per period of the unrolled loop (8 or 4)
  for (int cp = 0; cp < npairs; cp++) {
    uint32_t e[SS_NP], o[SS_NP];
    const uint16_t* col = base + 2 * cp;
#pragma unroll
    for (int j = 0; j < SS_NP; j++) {            // rows 0..NP-1 of the window
      uint32_t w0 = *(const uint32_t*)(col + (size_t)j * SS_LS);
      uint32_t w1 = *(const uint32_t*)(col + (size_t)j * SS_LS + 2);
      e[j] = w0; o[j] = __builtin_amdgcn_alignbit(w1, w0, 16);
    }
    const uint32_t* ocol = orgT + cp * HS;       // used rows of this column pair, contiguous, wave-uniform
    for (int rb = 0; rb < H; rb += SS_NP) {
      uint32_t ov[PER];
#pragma unroll
      for (int q = 0; q < PER; q += 4) {         // broadcast ds_read_b128 (HS is padded to a multiple of 8)
        uint4 t = *(const uint4*)(ocol + rb / STEP + q);
        ov[q] = t.x; ov[q + 1] = t.y; ov[q + 2] = t.z; ov[q + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int r = rb + k * STEP;             // original row (uniform)
        if (r < H) {
#pragma unroll
          for (int j = 0; j < SS_NP; j++) {
            const int slot = (k * STEP + j) % SS_NP;
            acc_e[j] = sad_u16x2(e[slot], ov[k], acc_e[j]);
            acc_o[j] = sad_u16x2(o[slot], ov[k], acc_o[j]);
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

// ---- edge strip: lane = one displacement row, columns dx0 and dx0+1 ----
__device__ static inline void ss_edge(const uint16_t* __restrict__ tile, const uint32_t* __restrict__ orgT, int HS, int step,
                                      int W, int H, int wave, int lane, uint32_t& acc_e, uint32_t& acc_o) {
  const uint16_t* base = tile + (size_t)(wave * 64 + lane) * SS_EDGE_LS;
  const int npairs = W >> 1, hs = H / step;
  for (int cp = 0; cp < npairs; cp++) {
    const uint16_t* col = base + 2 * cp;
    const uint32_t* ocol = orgT + cp * HS;
    for (int rr = 0; rr < hs; rr++) {
      const uint16_t* p = col + (size_t)(rr * step) * SS_EDGE_LS;
      uint32_t w0 = *(const uint32_t*)p, w1 = *(const uint32_t*)(p + 2);
      uint32_t ov = ocol[rr];
      acc_e = sad_u16x2(w0, ov, acc_e);
      acc_o = sad_u16x2(__builtin_amdgcn_alignbit(w1, w0, 16), ov, acc_o);
    }
  }
}

// cost + validity + key of one displacement.  bits_x / bits_y are the exp-Golomb lengths of the two MV components
// (cost scale 2, :4560), hoisted by the caller; probe points at the staged sample (dy+H+4, dx) of the tile when
// the tile holds sentinels (biased sentinel == 0), else NULL: isValidPattern, TComRdCost.cpp:444-458.
__device__ static inline unsigned long long ss_key(const hop_pu_job& jb, int W, int dx, int dy, uint32_t acc, int shift_up, int shift_dn,
                                                   uint32_t bits_x, uint32_t bits_y, const uint16_t* probe) {
  bool ok = dx >= jb.rng_left && dx <= jb.rng_right && dy <= jb.rng_bottom;
  ok = ok && !((dx >= jb.off_x) && (dy > jb.off_y));           // :6328
  if (ok && probe) ok = (probe[0] != 0) && (probe[W + 4] != 0);
  if (!ok) return ~0ull;
  uint32_t sad = (acc << shift_up) >> shift_dn;
  sad += (jb.lambda_cost * (bits_x + bits_y)) >> 16;           // getCost(x,y), TComRdCost.h:185-192
  return ((unsigned long long)sad << 32) | ((unsigned long long)(uint32_t)(dy - jb.rng_top) << 16) | (uint32_t)(dx - jb.rng_left);
}

// tile grid of one PU's search window
struct SsGeom { int x_first, tiles_x, tiles_y, n_main, n_edge, TH; };
__device__ static inline bool ss_geom(const hop_pu_job& jb, SsGeom& g) {
  const int win_w = jb.rng_right - jb.rng_left + 1, win_h = jb.rng_bottom - jb.rng_top + 1;
  if (win_w <= 0 || win_h <= 0) return false;
  // tiles start on an even absolute column so that the staging loads are 4-byte aligned
  const int xa = (jb.pu_x + jb.rng_left) & ~1;                    // absolute column of tile column 0 (pu_x is a multiple of 4)
  g.x_first = xa - jb.pu_x;                                       // displacement of tile column 0 (<= rng_left)
  const int span = jb.rng_right - g.x_first + 1;                  // columns to cover, 1..258
  g.tiles_x = (span + SS_TW - 1) / SS_TW;
  const int last_w = span - (g.tiles_x - 1) * SS_TW;              // width of the last tile column
  const bool has_edge = last_w <= 2;                              // searched by edge tiles instead
  if (has_edge) g.tiles_x -= 1;
  g.TH = ss_tile_h(jb.h);
  g.tiles_y = (win_h + g.TH - 1) / g.TH;
  g.n_main = g.tiles_x * g.tiles_y;
  g.n_edge = has_edge ? (win_h + SS_EDGE_ROWS - 1) / SS_EDGE_ROWS : 0;
  return true;
}
__device__ static inline void ss_tile_origin(const hop_pu_job& jb, const SsGeom& g, int t, int& dx0, int& dy0) {
  if (t < g.n_main) { dx0 = g.x_first + (t % g.tiles_x) * SS_TW; dy0 = jb.rng_top + (t / g.tiles_x) * g.TH; }
  else { dx0 = g.x_first + g.tiles_x * SS_TW; dy0 = jb.rng_top + (t - g.n_main) * SS_EDGE_ROWS; }
}

// work list: one entry (job << 5 | tile) per tile that is not rejected as a whole by the rule of :6328.
// Slots are reserved with one atomicAdd per PU; the order of the list is irrelevant (the argmin is order-free).
__global__ void k_ss_prep(const hop_pu_job* __restrict__ jobs, int n, unsigned int* __restrict__ counter, uint32_t* __restrict__ list,
                          unsigned long long* __restrict__ best_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  best_key[i] = ~0ull;
  const hop_pu_job jb = jobs[i];
  SsGeom g;
  if (!ss_geom(jb, g)) return;
  uint32_t keep = 0;
  const int nt = g.n_main + g.n_edge;
  for (int t = 0; t < nt; t++) {
    int dx0, dy0; ss_tile_origin(jb, g, t, dx0, dy0);
    if (!(dx0 >= jb.off_x && dy0 > jb.off_y)) keep |= 1u << t;     // every displacement of a skipped tile has dx >= offX' and dy > offY'
  }
  const int cnt = __popc(keep);
  if (!cnt) return;
  unsigned int base = atomicAdd(counter, (unsigned int)cnt);
  for (int t = 0; t < nt; t++) if (keep & (1u << t)) list[base++] = ((uint32_t)i << 5) | (uint32_t)t;
}

// persistent workgroups: each walks the work list with a grid stride
__global__ __launch_bounds__(256) void k_ss_search(const hop_pu_job* __restrict__ jobs, hop_pics pic, const unsigned int* __restrict__ counter,
                                                   const uint32_t* __restrict__ list, unsigned long long* __restrict__ best_key) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[SS_TILE_ELEMS];
  __shared__ __attribute__((aligned(16))) uint32_t orgT[32 * 64];
  __shared__ int has_sentinel;
  const unsigned int total = *counter;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned int wi = blockIdx.x; wi < total; wi += gridDim.x) {
  const uint32_t ent = list[wi];
  const int jidx = (int)(ent >> 5), t = (int)(ent & 31);
  const hop_pu_job jb = jobs[jidx];
  const int W = jb.w, H = jb.h;
  SsGeom g;
  ss_geom(jb, g);
  const int tiles_x = g.tiles_x, n_main = g.n_main, TH = g.TH;
  (void)tiles_x;
  const bool edge = t >= n_main;
  const bool sub = (jb.flags & HOP_FLAG_FEN) && H > 8;             // TEncSearch.cpp:6303-6309
  const int step = sub ? 2 : 1;
  const int hs = H / step, HS = (hs + 7) & ~7;                    // used rows of the original, padded pitch
  int dx0, dy0, rows, cols, pitch;
  ss_tile_origin(jb, g, t, dx0, dy0);                              // displacement of the tile origin
  if (!edge) {
    rows = min(TH, jb.rng_bottom - dy0 + 1) + H - 1 + SS_PROBE;   // + the rows of the validity probes (dy + H + 4)
    cols = (min(SS_TW, jb.rng_right - dx0 + 1) + W + 6) & ~1;     // even; + the probe columns (dx + W + 4); lanes beyond the window read stale LDS and are discarded
    pitch = SS_LS;
  } else {
    rows = min(SS_EDGE_ROWS, jb.rng_bottom - dy0 + 1) + H - 1 + SS_PROBE;
    cols = W + 6;
    pitch = SS_EDGE_LS;
  }
  // ---- stage the reference window (biased +1) and the transposed original; note whether any sentinel was seen ----
  __syncthreads();                                                 // the previous tile's readers are done with LDS
  if (threadIdx.x == 0) has_sentinel = 0;
  __syncthreads();
  {
    const int16_t* src = pic.ss_y + (ptrdiff_t)(jb.pu_y + dy0) * pic.stride_y + (jb.pu_x + dx0);
    const int cw = cols >> 1;
    bool zero = false;
    for (int r = wave; r < rows; r += 4) {
      const uint32_t* srow = (const uint32_t*)(src + (ptrdiff_t)r * pic.stride_y);
      uint32_t* trow = (uint32_t*)(tile + (size_t)r * pitch);
      for (int cdw = lane; cdw < cw; cdw += 64) {
        uint32_t v = bias_pk(srow[cdw]);                                   // per-half +1 (v_pk_add_u16): -1 -> 0 without a carry into the neighbour
        zero = zero || ((v & 0xFFFFu) == 0) || ((v >> 16) == 0);
        trow[cdw] = v;
      }
    }
    if (zero) has_sentinel = 1;
    const int16_t* org = pic.org_y + (size_t)jb.pu_y * pic.pic_w + jb.pu_x;
    const int np = W >> 1;
    for (int i = threadIdx.x; i < np * hs; i += 256) {
      int rr = i / np, cp = i - rr * np;                                   // coalesced along the row
      uint32_t v = *(const uint32_t*)(org + (size_t)(rr * step) * pic.pic_w + 2 * cp);
      orgT[cp * HS + rr] = v + 0x00010001u;                                // original samples are >= 0: no carry
    }
  }
  __syncthreads();
  const bool probe_on = has_sentinel != 0;                          // no sentinel staged -> every probe of this tile is valid
  const int shift_up = sub ? 1 : 0, shift_dn = pic.bd_y - 8;
  unsigned long long best = ~0ull;
  if (!edge) {
    const int dxe = dx0 + 2 * lane;
    const uint32_t bx_e = hopd_component_bits(dxe * 4 - jb.pred_x), bx_o = hopd_component_bits((dxe + 1) * 4 - jb.pred_x);
    for (int strip = wave; strip * SS_NP < TH; strip += 4) {
      const int wy0 = dy0 + strip * SS_NP;
      if (wy0 > jb.rng_bottom) break;                              // strip outside the window (uniform per wave)
      if (dx0 >= jb.off_x && wy0 > jb.off_y) break;                // this and all later strips are rejected by rule
      uint32_t acc_e[SS_NP], acc_o[SS_NP];
#pragma unroll
      for (int j = 0; j < SS_NP; j++) { acc_e[j] = 0; acc_o[j] = 0; }
      if (sub) ss_strip<2>(tile, orgT, HS, W, H, strip, lane, acc_e, acc_o);
      else     ss_strip<1>(tile, orgT, HS, W, H, strip, lane, acc_e, acc_o);
#pragma unroll
      for (int j = 0; j < SS_NP; j++) {
        const int dy = wy0 + j;
        const uint32_t by = hopd_component_bits(dy * 4 - jb.pred_y);
        const uint16_t* pr = probe_on ? tile + (size_t)(strip * SS_NP + j + H + 4) * SS_LS + 2 * lane : nullptr;
        unsigned long long k0 = ss_key(jb, W, dxe, dy, acc_e[j], shift_up, shift_dn, bx_e, by, pr);
        unsigned long long k1 = ss_key(jb, W, dxe + 1, dy, acc_o[j], shift_up, shift_dn, bx_o, by, pr ? pr + 1 : nullptr);
        best = k0 < best ? k0 : best;
        best = k1 < best ? k1 : best;
      }
    }
  } else if (wave < 2 && dy0 + wave * 64 <= jb.rng_bottom && !(dx0 >= jb.off_x && dy0 + wave * 64 > jb.off_y)) {
    const int wy0 = dy0 + wave * 64;
    uint32_t acc_e = 0, acc_o = 0;
    ss_edge(tile, orgT, HS, step, W, H, wave, lane, acc_e, acc_o);
    const int dy = wy0 + lane;
    const uint32_t by = hopd_component_bits(dy * 4 - jb.pred_y);
    const uint16_t* pr = probe_on ? tile + (size_t)(wave * 64 + lane + H + 4) * SS_EDGE_LS : nullptr;
    unsigned long long k0 = ss_key(jb, W, dx0, dy, acc_e, shift_up, shift_dn, hopd_component_bits(dx0 * 4 - jb.pred_x), by, pr);
    unsigned long long k1 = ss_key(jb, W, dx0 + 1, dy, acc_o, shift_up, shift_dn, hopd_component_bits((dx0 + 1) * 4 - jb.pred_x), by, pr ? pr + 1 : nullptr);
    best = k0 < k1 ? k0 : k1;
  }
  best = hopd_wave_min_u64(best);
  if (lane == 0 && best != ~0ull) atomicMin(best_key + jidx, best);
  }   // work-list loop
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
  // scratch: best keys (8 B / PU), the tile counter, the tile work list (<= 27 entries / PU)
  const size_t o_cnt = (size_t)n * 8, o_list = o_cnt + 256;
  void* sc; int r = hop_scratch(c, o_list + (size_t)n * SS_MAX_TILES * 4, &sc); if (r) return r;
  unsigned long long* keys = (unsigned long long*)sc;
  unsigned int* counter = (unsigned int*)((char*)sc + o_cnt);
  uint32_t* list = (uint32_t*)((char*)sc + o_list);
  hop_pics pic = hop_make_pics(c);
  const int pr = hop_prof_begin(c, HOP_K_SS_SEARCH, (uint64_t)n);
  (void)hipMemsetAsync(counter, 0, 4, c->stream);
  hipLaunchKernelGGL(k_ss_prep, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, counter, list, keys);
  // persistent grid: 3 workgroups per CU fit by LDS (48.6 KB each); a few more rounds of them smooth the tail
  const unsigned grid = (unsigned)std::min<size_t>((size_t)n * SS_MAX_TILES, (size_t)256 * 3 * 4);
  hipLaunchKernelGGL(k_ss_search, dim3(grid), dim3(256), 0, c->stream, d_jobs, pic, counter, list, keys);
  hipLaunchKernelGGL(k_ss_finalize, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, keys, pic, c->ss_buf[0], d_res, n);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "ss_search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
