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

// tile grid of a search window [l,r] x [top,bottom] of displacements of a block at column bx, height bh
struct SsGeom { int x_first, tiles_x, tiles_y, n_main, n_edge, TH; };
__device__ static inline bool ss_geom_w(int bx, int bh, int l, int r, int top, int bottom, SsGeom& g) {
  const int win_w = r - l + 1, win_h = bottom - top + 1;
  if (win_w <= 0 || win_h <= 0) return false;
  // tiles start on an even absolute column so that the staging loads are 4-byte aligned
  const int xa = (bx + l) & ~1;                                   // absolute column of tile column 0 (bx is a multiple of 4)
  g.x_first = xa - bx;                                            // displacement of tile column 0 (<= l)
  const int span = r - g.x_first + 1;                             // columns to cover
  g.tiles_x = (span + SS_TW - 1) / SS_TW;
  const int last_w = span - (g.tiles_x - 1) * SS_TW;              // width of the last tile column
  const bool has_edge = last_w <= 2;                              // searched by edge tiles instead
  if (has_edge) g.tiles_x -= 1;
  g.TH = ss_tile_h(bh);
  g.tiles_y = (win_h + g.TH - 1) / g.TH;
  g.n_main = g.tiles_x * g.tiles_y;
  g.n_edge = has_edge ? (win_h + SS_EDGE_ROWS - 1) / SS_EDGE_ROWS : 0;
  return true;
}
__device__ static inline void ss_origin_w(const SsGeom& g, int top, int t, int& dx0, int& dy0) {
  if (t < g.n_main) { dx0 = g.x_first + (t % g.tiles_x) * SS_TW; dy0 = top + (t / g.tiles_x) * g.TH; }
  else { dx0 = g.x_first + g.tiles_x * SS_TW; dy0 = top + (t - g.n_main) * SS_EDGE_ROWS; }
}
__device__ static inline bool ss_geom(const hop_pu_job& jb, SsGeom& g) { return ss_geom_w(jb.pu_x, jb.h, jb.rng_left, jb.rng_right, jb.rng_top, jb.rng_bottom, g); }
__device__ static inline void ss_tile_origin(const hop_pu_job& jb, const SsGeom& g, int t, int& dx0, int& dy0) { ss_origin_w(g, jb.rng_top, t, dx0, dy0); }

// stage the reference window of a tile (biased +1); note whether a sentinel was seen.  (x0,y0) = picture position of the
// tile's first sample.  Rows are clamped into the allocation (margin + guard rows): a clamped row only feeds displacements
// that lie outside every window a checked job can have.
__device__ static inline void ss_stage_ref(uint16_t* __restrict__ tile, const hop_pics& pic, int x0, int y0, int rows, int cols, int pitch,
                                           int wave, int lane, int* has_sentinel) {
  const int ylo = -(HOP_MARGIN_Y + HOP_GUARD_ROWS) + 1, yhi = pic.pic_h + HOP_MARGIN_Y + HOP_GUARD_ROWS - 2;
  const int cw = cols >> 1;
  bool zero = false;
  for (int r = wave; r < rows; r += 4) {
    const int y = min(max(y0 + r, ylo), yhi);
    const uint32_t* srow = (const uint32_t*)(pic.ss_y + (ptrdiff_t)y * pic.stride_y + x0);
    uint32_t* trow = (uint32_t*)(tile + (size_t)r * pitch);
    for (int cdw = lane; cdw < cw; cdw += 64) {
      uint32_t v = bias_pk(srow[cdw]);                                   // per-half +1 (v_pk_add_u16): -1 -> 0 without a carry into the neighbour
      zero = zero || ((v & 0xFFFFu) == 0) || ((v >> 16) == 0);
      trow[cdw] = v;
    }
  }
  if (zero) *has_sentinel = 1;
}
// stage one original block transposed to [column pair][used row] (every `step`-th row, hs of them), biased +1
__device__ static inline void ss_stage_org(uint32_t* __restrict__ orgT, const hop_pics& pic, int bx, int by, int W, int hs, int step, int HS, int wave, int lane) {
  const int16_t* org = pic.org_y + (size_t)by * pic.pic_w + bx;
  const int np = W >> 1;
  for (int i = wave * 64 + lane; i < np * hs; i += 256) {
    int rr = i / np, cp = i - rr * np;                                   // coalesced along the row
    uint32_t v = *(const uint32_t*)(org + (size_t)(rr * step) * pic.pic_w + 2 * cp);
    orgT[cp * HS + rr] = v + 0x00010001u;                                // original samples are >= 0: no carry
  }
}
__device__ static inline void ss_stage(uint16_t* __restrict__ tile, uint32_t* __restrict__ orgT, const hop_pics& pic, int bx, int by, int dx0, int dy0,
                                       int rows, int cols, int pitch, int W, int hs, int step, int HS, int wave, int lane, int* has_sentinel) {
  ss_stage_ref(tile, pic, bx + dx0, by + dy0, rows, cols, pitch, wave, lane, has_sentinel);
  ss_stage_org(orgT, pic, bx, by, W, hs, step, HS, wave, lane);
}

// work lists.  Single PUs: one entry (job << 5 | tile) per tile that is not rejected as a whole by the rule of :6328.
// CU families (see below) are collected into cells of the picture; the families of a cell share the staged window and
// get one entry (cell << 8 | tile) per tile of the cell's union window that some member can use; their PUs get no
// entries of their own.  The order of the lists is irrelevant (the argmin is order-free).
__device__ static inline bool ss_family_head(const hop_pu_job* __restrict__ jobs, int n, int i);
__device__ static inline int ss_cell_of(int N, int x, int y, int pic_w, int pic_h, int& cap);
#define SS_CELL_MAX 16          // families per cell
#define SS_CELL_MAX_TILES 255    // tiles of a cell's union window (11 when the members share a predictor)
struct SsCellRec { int N, gx, gy, l, r, top, bottom, nt, flags, cnt; SsGeom g; };     // 16 ints

// pass 1: every family head takes a slot in its cell
__global__ void k_ss_prep1(const hop_pu_job* __restrict__ jobs, int n, int pic_w, int pic_h, int families, unsigned int* __restrict__ cell_count,
                           int32_t* __restrict__ cell_members, int32_t* __restrict__ slot_of, unsigned long long* __restrict__ best_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  best_key[i] = ~0ull;
  int slot = -1;
  if (families && ss_family_head(jobs, n, i)) {
    int cap; const int cell = ss_cell_of(jobs[i].w, jobs[i].pu_x, jobs[i].pu_y, pic_w, pic_h, cap);
    slot = (int)atomicAdd(cell_count + cell, 1u);
    if (slot < cap) cell_members[cell * SS_CELL_MAX + slot] = i;     // a cell holds each CU position once; duplicates in a batch overflow
  }
  slot_of[i] = slot;
}

// pass 3 (after the cells are built): PUs that are not searched with a cell get their own tiles
__global__ void k_ss_prep3_singles(const hop_pu_job* __restrict__ jobs, int n, int pic_w, int pic_h, const int32_t* __restrict__ slot_of,
                           const SsCellRec* __restrict__ cell_rec, unsigned int* __restrict__ counters, uint32_t* __restrict__ list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 5; k++) {
    const int h = i - k;
    if (h < 0 || slot_of[h] < 0) continue;                         // slot_of[h] >= 0: h heads a family of jobs h..h+4
    int cap; const int cell = ss_cell_of(jobs[h].w, jobs[h].pu_x, jobs[h].pu_y, pic_w, pic_h, cap);
    if (slot_of[h] < cap && cell_rec[cell].cnt > 0) return;        // searched with its cell
    break;
  }
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
  unsigned int base = atomicAdd(counters, (unsigned int)cnt);
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
  ss_stage(tile, orgT, pic, jb.pu_x, jb.pu_y, dx0, dy0, rows, cols, pitch, W, hs, step, HS, wave, lane, &has_sentinel);
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

// =====================================================================================================================
// CU families: the five symmetric PUs of one CU (2Nx2N, Nx2N left/right, 2NxN top/bottom) searched in ONE pass.
//
// The five PUs cover the same samples, and a displacement d moves all of them by the same amount, so
//   SAD_PU(d) = sum of the SADs of the CU's four quadrants the PU consists of, at the same d.
// One pass accumulates the quadrant SADs (with FEN row subsampling: over the even CU rows, which are the even rows
// of every member because the members start on even rows; for N = 16 the 2NxN members have H = 8 and are not
// subsampled, so the odd rows of the two halves are accumulated as well) and derives all five costs from them:
// a third (N = 16: a half) of the v_sad_u16 work of five separate searches, one staging of the window instead of five.
// Each member keeps its own window, rule offsets (:6328), MV predictor, validity probes and first-best key, so the
// result of every member is the one the single-PU search gives.
// Per displacement the five cost evaluations would now outweigh the SADs; they are pruned with a bound that cannot
// change the result: a displacement whose SAD alone (<= its cost) exceeds the smallest cost already seen for that
// member is not the minimum, nor tied with it.
// =====================================================================================================================
#define SS_FAM 5
#define SS_FAM_MAX_TILES 255
#define SS_FAM_CHUNK 4
#define FAM_NP 4                 // displacement rows per strip: 4 quadrants x 2 columns x FAM_NP accumulators per lane (+ 2 x 2 x FAM_NP odd-row ones for N = 16)

// ox, oy, w, h of member m inside a CU of size S (units of S/2)
__device__ static inline void fam_member_rect(int m, int S, int& ox, int& oy, int& w, int& h) {
  const int hf = S >> 1;
  ox = (m == 2) ? hf : 0; oy = (m == 4) ? hf : 0;
  w = (m == 1 || m == 2) ? hf : S; h = (m == 3 || m == 4) ? hf : S;
}

// jobs[i..i+4] are the five symmetric PUs of one CU, in the order of hop_enumerate_ctu_jobs
__device__ static inline bool ss_family_head(const hop_pu_job* __restrict__ jobs, int n, int i) {
  if (i < 0 || i + SS_FAM > n) return false;
  const hop_pu_job* a = jobs + i;
  const int S = a->w;
  if (a->h != S || (S != 8 && S != 16 && S != 32 && S != 64) || (a->pu_x & (S - 1)) || (a->pu_y & (S - 1))) return false;
  for (int m = 1; m < SS_FAM; m++) {
    int ox, oy, w, h; fam_member_rect(m, S, ox, oy, w, h);
    const hop_pu_job* b = a + m;
    if (b->pu_x != a->pu_x + ox || b->pu_y != a->pu_y + oy || b->w != w || b->h != h || b->flags != a->flags) return false;
  }
  return true;
}

// cells: 64 columns x CH rows of the picture, CH = max(N, 16); the families (CUs of size N) inside one cell share a staged window
__device__ __host__ static inline int ss_cell_h(int N) { return N < 16 ? 16 : N; }
__device__ __host__ static inline int ss_cells_total(int pic_w, int pic_h) {
  const int cx = (pic_w + 63) / 64;
  return cx * (2 * ((pic_h + 15) / 16) + (pic_h + 31) / 32 + (pic_h + 63) / 64);
}
__device__ static inline int ss_cell_of(int N, int x, int y, int pic_w, int pic_h, int& cap) {
  const int cx = (pic_w + 63) / 64, c16 = (pic_h + 15) / 16, c32 = (pic_h + 31) / 32;
  const int CH = ss_cell_h(N);
  const int base = (N == 8) ? 0 : (N == 16) ? cx * c16 : (N == 32) ? 2 * cx * c16 : cx * (2 * c16 + c32);
  cap = (64 / N) * (CH / N);
  return base + (y / CH) * cx + (x >> 6);
}

// pass 2: one thread per cell: union window of all members, tile grid, the tiles some member can use
__global__ void k_ss_prep2_cells(const hop_pu_job* __restrict__ jobs, int ncells, int pic_w, int pic_h, const unsigned int* __restrict__ cell_count,
                           const int32_t* __restrict__ cell_members, SsCellRec* __restrict__ cell_rec, unsigned int* __restrict__ counters,
                           uint32_t* __restrict__ grp_list) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells || cell_count[c] == 0) return;
  const int32_t* mem = cell_members + c * SS_CELL_MAX;
  const hop_pu_job* h0 = jobs + mem[0];
  const int N = h0->w, CH = ss_cell_h(N);
  const int cap = (64 / N) * (CH / N), cnt = min((int)cell_count[c], cap);
  SsCellRec rec;
  rec.N = N; rec.gx = h0->pu_x & ~63; rec.gy = (h0->pu_y / CH) * CH; rec.flags = h0->flags; rec.cnt = cnt; rec.nt = 0;
  int l = 1 << 30, r = -(1 << 30), t = 1 << 30, b = -(1 << 30);
  for (int f = 0; f < cnt; f++)
    for (int m = 0; m < SS_FAM; m++) {
      const hop_pu_job* j = jobs + mem[f] + m;
      if (j->rng_right < j->rng_left || j->rng_bottom < j->rng_top) continue;
      l = min(l, j->rng_left); r = max(r, j->rng_right); t = min(t, j->rng_top); b = max(b, j->rng_bottom);
    }
  rec.l = l; rec.r = r; rec.top = t; rec.bottom = b;
  if (l <= r) {
    ss_geom_w(rec.gx, CH, l, r, t, b, rec.g);
    rec.nt = rec.g.n_main + rec.g.n_edge;
  }
  if (rec.nt > SS_CELL_MAX_TILES) { rec.cnt = 0; rec.nt = 0; }      // predictors far apart: the members are searched one by one (k_ss_prep2)
  cell_rec[c] = rec;
  if (!rec.nt) return;
  // the tiles some member can use: count, reserve, emit
  unsigned int base = 0;
  for (int pass = 0; pass < 2; pass++) {
    int used_cnt = 0;
    for (int tt = 0; tt < rec.nt; tt++) {
      int dx0, dy0; ss_origin_w(rec.g, rec.top, tt, dx0, dy0);
      const bool edge = tt >= rec.g.n_main;
      const int tw = edge ? 2 : SS_TW, th = edge ? SS_EDGE_ROWS : rec.g.TH;
      bool u = false;
      for (int f = 0; f < cnt && !u; f++)
        for (int m = 0; m < SS_FAM && !u; m++) {
          const hop_pu_job* j = jobs + mem[f] + m;
          if (j->rng_right < j->rng_left || j->rng_bottom < j->rng_top) continue;
          if (dx0 > j->rng_right || dx0 + tw <= j->rng_left || dy0 > j->rng_bottom || dy0 + th <= j->rng_top) continue;
          // the smallest displacement of the tile inside the member's window decides the rule for the whole intersection
          if (max(dx0, j->rng_left) >= j->off_x && max(dy0, j->rng_top) > j->off_y) continue;
          u = true;
        }
      if (u) { if (pass) grp_list[base + used_cnt] = ((uint32_t)c << 8) | (uint32_t)tt; used_cnt++; }
    }
    if (!used_cnt) return;
    if (!pass) base = atomicAdd(counters + 1, (unsigned int)used_cnt);
  }
}

// minimum over the wave, wave-uniform result: four DPP steps inside each row of 16 lanes, then the four row minima through SGPRs
__device__ static inline uint32_t hopd_wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

// MODE 0: all rows (no FEN, or N = 8); 1: even rows (FEN, N >= 32); 2: all rows, odd ones apart (FEN, N = 16)
// q[qy][qx][j][parity of dx] : quadrant SADs of displacement row j ; od[qy][j][parity] : odd rows of the upper / lower half
template <int MODE> struct FamAcc { uint32_t q[2][2][FAM_NP][2]; uint32_t od[2][FAM_NP][2]; };

// one period (FAM_NP window rows) of one column pair: rows rb .. rb+FAM_NP-1 of the CU against the rotating window
template <int MODE>
__device__ __forceinline__ void fam_period(const uint16_t* __restrict__ col, const uint32_t* __restrict__ ocol, int rb, int nrows,
                                           uint32_t (&e)[FAM_NP], uint32_t (&o)[FAM_NP], uint32_t (&qa)[FAM_NP][2], uint32_t (&od)[FAM_NP][2]) {
  constexpr int STEP = (MODE == 1) ? 2 : 1, PER = FAM_NP / STEP;
  uint32_t ov[PER];
  if (PER == 4) { uint4 t = *(const uint4*)(ocol + rb / STEP); ov[0] = t.x; ov[1] = t.y; ov[PER - 2] = t.z; ov[PER - 1] = t.w; }   // broadcast ds_read_b128
  else { uint2 t = *(const uint2*)(ocol + rb / STEP); ov[0] = t.x; ov[PER - 1] = t.y; }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int r = rb + k * STEP;                   // CU row (uniform)
#pragma unroll
    for (int j = 0; j < FAM_NP; j++) {
      const int slot = (k * STEP + j) % FAM_NP;
      if (MODE == 2 && (k & 1)) { od[j][0] = sad_u16x2(e[slot], ov[k], od[j][0]); od[j][1] = sad_u16x2(o[slot], ov[k], od[j][1]); }
      else { qa[j][0] = sad_u16x2(e[slot], ov[k], qa[j][0]); qa[j][1] = sad_u16x2(o[slot], ov[k], qa[j][1]); }
    }
#pragma unroll
    for (int s = 0; s < STEP; s++) {               // slide: rows r+NP+s replace rows r+s
      const int row = r + FAM_NP + s;
      const int slot = (k * STEP + s) % FAM_NP;
      // unconditional: the last FAM_NP-1 slides of a block read staged rows that no displacement of the strip uses
      // (tile row <= TH + N - 1 < SS_ROWS + SS_PROBE), and a branch here would serialise every LDS read
      uint32_t w0 = *(const uint32_t*)(col + (size_t)row * SS_LS);
      uint32_t w1 = *(const uint32_t*)(col + (size_t)row * SS_LS + 2);
      e[slot] = w0; o[slot] = __builtin_amdgcn_alignbit(w1, w0, 16);
    }
  }
}

template <int MODE>
__device__ __forceinline__ void fam_column(const uint16_t* __restrict__ col, const uint32_t* __restrict__ ocol, int N,
                                           uint32_t (&qt)[FAM_NP][2], uint32_t (&qb)[FAM_NP][2], uint32_t (&ot)[FAM_NP][2], uint32_t (&ob)[FAM_NP][2]) {
  uint32_t e[FAM_NP], o[FAM_NP];
#pragma unroll
  for (int j = 0; j < FAM_NP; j++) {
    uint32_t w0 = *(const uint32_t*)(col + (size_t)j * SS_LS);
    uint32_t w1 = *(const uint32_t*)(col + (size_t)j * SS_LS + 2);
    e[j] = w0; o[j] = __builtin_amdgcn_alignbit(w1, w0, 16);
  }
#pragma unroll 2
  for (int rb = 0; rb < N / 2; rb += FAM_NP) fam_period<MODE>(col, ocol, rb, N, e, o, qt, ot);     // upper quadrant (N/2 is a multiple of FAM_NP)
#pragma unroll 2
  for (int rb = N / 2; rb < N; rb += FAM_NP) fam_period<MODE>(col, ocol, rb, N, e, o, qb, ob);     // lower quadrant, the window keeps rotating
}

template <int MODE>
__device__ __forceinline__ void fam_strip(const uint16_t* __restrict__ tile, const uint32_t* __restrict__ orgT, int HS, int N, int strip, int lane, FamAcc<MODE>& A) {
  const uint16_t* base = tile + (size_t)(strip * FAM_NP) * SS_LS + 2 * lane;
  const int nh = N >> 2;                           // column pairs per CU half
  for (int cp = 0; cp < nh; cp++)      fam_column<MODE>(base + 2 * cp, orgT + cp * HS, N, A.q[0][0], A.q[1][0], A.od[0], A.od[1]);
  for (int cp = nh; cp < 2 * nh; cp++) fam_column<MODE>(base + 2 * cp, orgT + cp * HS, N, A.q[0][1], A.q[1][1], A.od[0], A.od[1]);
}

// SAD of member m from the quadrant sums, already scaled like the reference's uiSad (<< 1 when subsampled, >> (bitDepth-8))
template <int MODE>
__device__ __forceinline__ uint32_t fam_member_sad(int m, uint32_t q00, uint32_t q01, uint32_t q10, uint32_t q11, uint32_t ot, uint32_t ob, int shift_dn) {
  uint32_t s;
  switch (m) {
    case 0: s = q00 + q01 + q10 + q11; break;
    case 1: s = q00 + q10; break;
    case 2: s = q01 + q11; break;
    case 3: s = q00 + q01; if (MODE == 2) s += ot; break;
    default: s = q10 + q11; if (MODE == 2) s += ob; break;
  }
  const int up = (MODE == 1) ? 1 : (MODE == 2 && m < 3) ? 1 : 0;
  return (s << up) >> shift_dn;
}

// exact evaluation of one displacement for one member: window, rule of :6328, validity probes, cost, strict '<'
__device__ __forceinline__ void fam_eval(const hop_pu_job* __restrict__ jm, int dx, int dy, uint32_t sadv, bool probe_ok, uint32_t& bc, uint32_t& bp) {
  bool ok = dx >= jm->rng_left && dx <= jm->rng_right && dy >= jm->rng_top && dy <= jm->rng_bottom;
  ok = ok && !((dx >= jm->off_x) && (dy > jm->off_y)) && probe_ok;
  const uint32_t cost = sadv + ((jm->lambda_cost * (hopd_component_bits(dx * 4 - jm->pred_x) + hopd_component_bits(dy * 4 - jm->pred_y))) >> 16);
  const bool upd = ok && cost < bc;
  bc = upd ? cost : bc;
  bp = upd ? (((uint32_t)(dy - jm->rng_top) << 16) | (uint32_t)(dx - jm->rng_left)) : bp;
}

#define FAM_TAB_ROWS 24           // displacement rows one wave handles per family and tile: (TH / FAM_NP / 4 strips, rounded up) * FAM_NP <= 20
template <int MODE>
__device__ __forceinline__ void fam_main(const hop_pu_job* __restrict__ head, const uint16_t* __restrict__ tile, const uint32_t* __restrict__ orgT,
                                         uint2* __restrict__ rowtab, int HS, int N, int TH, int dx0, int dy0, int left_u, int right_u, int bottom_u,
                                         bool probe_on, int shift_dn, int wave, int lane,
                                         uint32_t (&bc)[SS_FAM], uint32_t (&bp)[SS_FAM], uint32_t (&wbest)[SS_FAM]) {
  const int dxe = dx0 + 2 * lane;
  const bool in_e = dxe >= left_u && dxe <= right_u, in_o = dxe + 1 >= left_u && dxe + 1 <= right_u;   // lanes beyond the window hold stale LDS
  const int nstrips = TH / FAM_NP, nst = (nstrips - wave + 3) >> 2;       // strips wave, wave+4, ... of this wave
  // lambda * bits of the horizontal MV component, per member and column:
  // getCost = (lambda * (bx + by)) >> 16 in 32-bit arithmetic = (lambda*bx + lambda*by) >> 16
  uint32_t lbx[SS_FAM][2];
#pragma unroll
  for (int m = 0; m < SS_FAM; m++) {
    const hop_pu_job* jm = head + m;
    lbx[m][0] = jm->lambda_cost * hopd_component_bits(dxe * 4 - jm->pred_x);
    lbx[m][1] = jm->lambda_cost * hopd_component_bits((dxe + 1) * 4 - jm->pred_x);
  }
  // per (member, row of this wave): lambda * bits of the vertical component and a penalty for rows outside the member's
  // window -- computed once, one entry per lane, instead of by every lane for every row (the values are wave-uniform
  // but come from memory, so the compiler would run them on the vector unit)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // the previous family's readers of this wave's table are done (in-order LDS)
  for (int e = lane; e < SS_FAM * nst * FAM_NP; e += 64) {
    const int m = e / (nst * FAM_NP), k = e - m * (nst * FAM_NP);
    const int dy = dy0 + (wave + 4 * (k / FAM_NP)) * FAM_NP + (k % FAM_NP);
    const hop_pu_job* jm = head + m;
    uint2 t;
    t.x = jm->lambda_cost * hopd_component_bits(dy * 4 - jm->pred_y);
    t.y = (dy < jm->rng_top || dy > jm->rng_bottom) ? 0x40000000u : 0u;
    rowtab[m * FAM_TAB_ROWS + k] = t;
  }
  // strips of this wave in which some member has an acceptable displacement: one (member, strip) pair per lane
  unsigned needmask = 0;
  {
    bool need = false;
    if (lane < SS_FAM * nst) {
      const int m = lane / nst, k = lane - m * nst;
      const hop_pu_job* jm = head + m;
      const int wy0 = dy0 + (wave + 4 * k) * FAM_NP;
      need = dx0 <= jm->rng_right && wy0 <= jm->rng_bottom && wy0 + FAM_NP > jm->rng_top &&
             !(max(dx0, jm->rng_left) >= jm->off_x && max(wy0, jm->rng_top) > jm->off_y);
    }
    const unsigned long long bal = __ballot(need);
#pragma unroll
    for (int m = 0; m < SS_FAM; m++) needmask |= (unsigned)(bal >> (m * nst)) & ((1u << nst) - 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int strip = wave, k = 0; strip < nstrips; strip += 4, k++) {
    if (!(needmask & (1u << k))) continue;                         // the reference computes these SADs and throws them away (:6328)
    const int wy0 = dy0 + strip * FAM_NP;
    FamAcc<MODE> A;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int j = 0; j < FAM_NP; j++) {
        A.q[a][0][j][0] = A.q[a][0][j][1] = A.q[a][1][j][0] = A.q[a][1][j][1] = 0; A.od[a][j][0] = A.od[a][j][1] = 0;
      }
    fam_strip<MODE>(tile, orgT, HS, N, strip, lane, A);
#pragma unroll
    for (int j = 0; j < FAM_NP; j++) {
      const int dy = wy0 + j;
      uint32_t sv[SS_FAM][2];
      unsigned tm = 0;                                             // per lane: members for which one of its two displacements can still be the first-best
#pragma unroll
      for (int m = 0; m < SS_FAM; m++) {
        const uint2 rt = rowtab[m * FAM_TAB_ROWS + k * FAM_NP + j];      // broadcast read: lambda*by, row penalty
        sv[m][0] = fam_member_sad<MODE>(m, A.q[0][0][j][0], A.q[0][1][j][0], A.q[1][0][j][0], A.q[1][1][j][0], A.od[0][j][0], A.od[1][j][0], shift_dn);
        sv[m][1] = fam_member_sad<MODE>(m, A.q[0][0][j][1], A.q[0][1][j][1], A.q[1][0][j][1], A.q[1][1][j][1], A.od[0][j][1], A.od[1][j][1], shift_dn);
        const uint32_t c0 = sv[m][0] + ((lbx[m][0] + rt.x) >> 16) + rt.y, c1 = sv[m][1] + ((lbx[m][1] + rt.x) >> 16) + rt.y;
        // only a cost at or below the smallest one seen so far can be (or tie with) the first-best
        tm |= ((in_e && c0 <= wbest[m]) || (in_o && c1 <= wbest[m])) ? (1u << m) : 0u;
      }
      if (!__any(tm != 0)) continue;                               // the common case after the first rows: one branch per row
#pragma unroll
      for (int m = 0; m < SS_FAM; m++) {
        if (!__any((tm >> m) & 1u)) continue;
        const hop_pu_job* jm = head + m;
        bool p0 = true, p1 = true;
        if (probe_on) {                                            // isValidPattern, TComRdCost.cpp:444-458, at the member's own corner samples
          int ox, oy, w, h; fam_member_rect(m, N, ox, oy, w, h);
          const uint16_t* pr = tile + (size_t)(strip * FAM_NP + j + oy + h + 4) * SS_LS + 2 * lane + ox;
          p0 = (pr[0] != 0) && (pr[w + 4] != 0); p1 = (pr[1] != 0) && (pr[w + 5] != 0);
        }
        fam_eval(jm, dxe, dy, sv[m][0], p0, bc[m], bp[m]);
        fam_eval(jm, dxe + 1, dy, sv[m][1], p1, bc[m], bp[m]);
        wbest[m] = min(wbest[m], hopd_wave_min_u32(bc[m]));
      }
    }
  }
}

// edge tile of a family: lane = displacement row, columns dx0 and dx0+1
template <int MODE>
__device__ __forceinline__ void fam_edge_quad(const uint16_t* __restrict__ base, const uint32_t* __restrict__ orgT, int HS, int cp0, int cp1, int rr0, int rr1,
                                              uint32_t (&q)[2], uint32_t (&od)[2]) {
  constexpr int STEP = (MODE == 1) ? 2 : 1;
  for (int cp = cp0; cp < cp1; cp++) {
    const uint16_t* col = base + 2 * cp;
    const uint32_t* ocol = orgT + cp * HS;
    for (int rr = rr0; rr < rr1; rr += (MODE == 2 ? 2 : 1)) {
      const uint16_t* p = col + (size_t)(rr * STEP) * SS_EDGE_LS;
      uint32_t w0 = *(const uint32_t*)p, w1 = *(const uint32_t*)(p + 2);
      uint32_t ov = ocol[rr];
      q[0] = sad_u16x2(w0, ov, q[0]);
      q[1] = sad_u16x2(__builtin_amdgcn_alignbit(w1, w0, 16), ov, q[1]);
      if (MODE == 2) {                             // the odd row that follows
        p += SS_EDGE_LS; w0 = *(const uint32_t*)p; w1 = *(const uint32_t*)(p + 2); ov = ocol[rr + 1];
        od[0] = sad_u16x2(w0, ov, od[0]);
        od[1] = sad_u16x2(__builtin_amdgcn_alignbit(w1, w0, 16), ov, od[1]);
      }
    }
  }
}

template <int MODE>
__device__ __forceinline__ void fam_edge(const hop_pu_job* __restrict__ head, const uint16_t* __restrict__ tile, const uint32_t* __restrict__ orgT,
                                         int HS, int N, int dx0, int wy0, bool probe_on, int shift_dn, int wave, int lane,
                                         uint32_t (&bc)[SS_FAM], uint32_t (&bp)[SS_FAM]) {
  constexpr int STEP = (MODE == 1) ? 2 : 1;
  const uint16_t* base = tile + (size_t)(wave * 64 + lane) * SS_EDGE_LS;             // `wave` = which 64-row half of the edge tile
  const int nh = N >> 2, hs = N / STEP, hh = hs >> 1;
  uint32_t q[2][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}}, od[2][2] = {{0, 0}, {0, 0}};
  fam_edge_quad<MODE>(base, orgT, HS, 0, nh, 0, hh, q[0][0], od[0]);
  fam_edge_quad<MODE>(base, orgT, HS, nh, 2 * nh, 0, hh, q[0][1], od[0]);
  fam_edge_quad<MODE>(base, orgT, HS, 0, nh, hh, hs, q[1][0], od[1]);
  fam_edge_quad<MODE>(base, orgT, HS, nh, 2 * nh, hh, hs, q[1][1], od[1]);
  const int dy = wy0 + lane;
#pragma unroll
  for (int m = 0; m < SS_FAM; m++) {
    const hop_pu_job* jm = head + m;
    bool p0 = true, p1 = true;
    if (probe_on) {
      int ox, oy, w, h; fam_member_rect(m, N, ox, oy, w, h);
      const uint16_t* pr = tile + (size_t)(wave * 64 + lane + oy + h + 4) * SS_EDGE_LS + ox;
      p0 = (pr[0] != 0) && (pr[w + 4] != 0); p1 = (pr[1] != 0) && (pr[w + 5] != 0);
    }
    fam_eval(jm, dx0, dy, fam_member_sad<MODE>(m, q[0][0][0], q[0][1][0], q[1][0][0], q[1][1][0], od[0][0], od[1][0], shift_dn), p0, bc[m], bp[m]);
    fam_eval(jm, dx0 + 1, dy, fam_member_sad<MODE>(m, q[0][0][1], q[0][1][1], q[1][0][1], q[1][1][1], od[0][1], od[1][1], shift_dn), p1, bc[m], bp[m]);
  }
}

// fold a wave's best of one member into the global key; skipped when the wave has nothing that could win
__device__ __forceinline__ void fam_flush(unsigned long long* __restrict__ best_key, uint32_t bc, uint32_t bp, int lane) {
  const uint32_t wm = hopd_wave_min_u32(bc);
  if (wm == 0xFFFFFFFFu) return;                                   // uniform
  if (wm > (uint32_t)(__hip_atomic_load(best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32)) return;
  unsigned long long key = bc == 0xFFFFFFFFu ? ~0ull : (((unsigned long long)bc << 32) | bp);
  key = hopd_wave_min_u64(key);
  if (lane == 0) atomicMin(best_key, key);
}

// persistent workgroups over the cell work list: entry = cell << 8 | tile.  One staging of the reference window
// (128 + 64 columns, TH + CH - 1 rows) serves every family of the cell; the waves then walk (family, strip) pairs
// without further barriers.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ss_family(const hop_pu_job* __restrict__ jobs, hop_pics pic, const unsigned int* __restrict__ counter,
                                                   const uint32_t* __restrict__ list, const SsCellRec* __restrict__ cell_rec,
                                                   const int32_t* __restrict__ cell_members, unsigned long long* __restrict__ best_key) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[SS_TILE_ELEMS];
  __shared__ __attribute__((aligned(16))) uint32_t orgT[32 * 64];
  __shared__ int has_sentinel;
  __shared__ uint2 rowtab_all[4][SS_FAM * FAM_TAB_ROWS];          // per wave: (lambda*by, row penalty) of the rows it handles for the current family
  const unsigned int total = *counter;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned int wi = blockIdx.x; wi < total; wi += gridDim.x) {
    const uint32_t ent = list[wi];
    const int cell = (int)(ent >> 8), t = (int)(ent & 255);
    const SsCellRec* cr = cell_rec + cell;
    const int32_t* mem = cell_members + cell * SS_CELL_MAX;
    const int N = cr->N, cnt = cr->cnt, gx = cr->gx, gy = cr->gy, CH = ss_cell_h(N);
    SsGeom g = cr->g;
    const int l_u = cr->l, r_u = cr->r, top_u = cr->top, bottom_u = cr->bottom;
    const bool edge = t >= g.n_main;
    const bool fen = (cr->flags & HOP_FLAG_FEN) != 0;
    const int mode = (!fen || N == 8) ? 0 : (N == 16) ? 2 : 1;
    const int step = (mode == 1) ? 2 : 1;
    const int hs = N / step, HS = (hs + 7) & ~7, fstride = (N >> 1) * HS;      // dwords of one family's transposed original
    int dx0, dy0, rows, cols, pitch;
    ss_origin_w(g, top_u, t, dx0, dy0);
    if (!edge) {
      rows = min(g.TH, bottom_u - dy0 + 1) + CH - 1 + SS_PROBE;
      cols = (min(SS_TW, r_u - dx0 + 1) + 64 + 6) & ~1;
      pitch = SS_LS;
    } else {
      rows = min(SS_EDGE_ROWS, bottom_u - dy0 + 1) + CH - 1 + SS_PROBE;
      cols = 64 + 6;
      pitch = SS_EDGE_LS;
    }
    __syncthreads();                                               // the previous tile's readers are done with LDS
    if (threadIdx.x == 0) has_sentinel = 0;
    __syncthreads();
    ss_stage_ref(tile, pic, gx + dx0, gy + dy0, rows, cols, pitch, wave, lane, &has_sentinel);
    for (int f = 0; f < cnt; f++) {
      const hop_pu_job* head = jobs + mem[f];
      ss_stage_org(orgT + f * fstride, pic, head->pu_x, head->pu_y, N, hs, step, HS, wave, lane);
    }
    __syncthreads();
    const bool probe_on = has_sentinel != 0;
    const int shift_dn = pic.bd_y - 8;
    if (!edge) {
      for (int f = 0; f < cnt; f++) {
        const int hidx = mem[f];
        const hop_pu_job* head = jobs + hidx;
        const uint16_t* tf = tile + (size_t)(head->pu_y - gy) * SS_LS + (head->pu_x - gx);
        const uint32_t* of = orgT + f * fstride;
        uint32_t bc[SS_FAM], bp[SS_FAM], wbest[SS_FAM];
#pragma unroll
        for (int m = 0; m < SS_FAM; m++) {
          bc[m] = 0xFFFFFFFFu; bp[m] = 0;
          wbest[m] = (uint32_t)(__hip_atomic_load(best_key + hidx + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);   // costs other tiles have already reached
        }
        if (mode == 0)      fam_main<0>(head, tf, of, rowtab_all[wave], HS, N, g.TH, dx0, dy0, l_u, r_u, bottom_u, probe_on, shift_dn, wave, lane, bc, bp, wbest);
        else if (mode == 1) fam_main<1>(head, tf, of, rowtab_all[wave], HS, N, g.TH, dx0, dy0, l_u, r_u, bottom_u, probe_on, shift_dn, wave, lane, bc, bp, wbest);
        else                fam_main<2>(head, tf, of, rowtab_all[wave], HS, N, g.TH, dx0, dy0, l_u, r_u, bottom_u, probe_on, shift_dn, wave, lane, bc, bp, wbest);
#pragma unroll
        for (int m = 0; m < SS_FAM; m++) fam_flush(best_key + hidx + m, bc[m], bp[m], lane);
      }
    } else {
      for (int item = wave; item < 2 * cnt; item += 4) {           // (family, 64-row half of the edge tile)
        const int f = item >> 1, half = item & 1;
        if (dy0 + half * 64 > bottom_u) continue;
        const int hidx = mem[f];
        const hop_pu_job* head = jobs + hidx;
        const uint16_t* tf = tile + (size_t)(head->pu_y - gy) * SS_EDGE_LS + (head->pu_x - gx);
        const uint32_t* of = orgT + f * fstride;
        uint32_t bc[SS_FAM], bp[SS_FAM];
#pragma unroll
        for (int m = 0; m < SS_FAM; m++) { bc[m] = 0xFFFFFFFFu; bp[m] = 0; }
        const int wy0 = dy0 + half * 64;
        if (mode == 0)      fam_edge<0>(head, tf, of, HS, N, dx0, wy0, probe_on, shift_dn, half, lane, bc, bp);
        else if (mode == 1) fam_edge<1>(head, tf, of, HS, N, dx0, wy0, probe_on, shift_dn, half, lane, bc, bp);
        else                fam_edge<2>(head, tf, of, HS, N, dx0, wy0, probe_on, shift_dn, half, lane, bc, bp);
#pragma unroll
        for (int m = 0; m < SS_FAM; m++) fam_flush(best_key + hidx + m, bc[m], bp[m], lane);
      }
    }
  }
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
  // scratch: best keys (8 B / PU), counters, the single-PU tile list (<= 27 entries / PU), slot of every family head,
  // the cells (count, members, record) and the cell tile list
  const int ncells = ss_cells_total(c->pic_w, c->pic_h);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_cnt = al((size_t)n * 8), o_list = o_cnt + 256, o_slot = al(o_list + (size_t)n * SS_MAX_TILES * 4), o_ccnt = al(o_slot + (size_t)n * 4);
  const size_t o_cmem = al(o_ccnt + (size_t)ncells * 4), o_crec = al(o_cmem + (size_t)ncells * SS_CELL_MAX * 4), o_glist = al(o_crec + (size_t)ncells * sizeof(SsCellRec));
  const size_t glist_cap = (size_t)std::min<size_t>((size_t)n / SS_FAM + 1, (size_t)ncells) * SS_CELL_MAX_TILES;
  void* sc; int r = hop_scratch(c, o_glist + glist_cap * 4, &sc); if (r) return r;
  unsigned long long* keys = (unsigned long long*)sc;
  unsigned int* counters = (unsigned int*)((char*)sc + o_cnt);
  uint32_t* list = (uint32_t*)((char*)sc + o_list);
  int32_t* slot_of = (int32_t*)((char*)sc + o_slot);
  unsigned int* cell_count = (unsigned int*)((char*)sc + o_ccnt);
  int32_t* cell_members = (int32_t*)((char*)sc + o_cmem);
  SsCellRec* cell_rec = (SsCellRec*)((char*)sc + o_crec);
  uint32_t* grp_list = (uint32_t*)((char*)sc + o_glist);
  hop_pics pic = hop_make_pics(c);
  const int fam = c->ss_families ? 1 : 0;
  const int pr = hop_prof_begin(c, HOP_K_SS_SEARCH, (uint64_t)n);
  (void)hipMemsetAsync(counters, 0, 8, c->stream);
  if (fam) (void)hipMemsetAsync(cell_count, 0, (size_t)ncells * 4, c->stream);
  hipLaunchKernelGGL(k_ss_prep1, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, c->pic_w, c->pic_h, fam, cell_count, cell_members, slot_of, keys);
  if (fam) hipLaunchKernelGGL(k_ss_prep2_cells, dim3((ncells + 255) / 256), dim3(256), 0, c->stream, d_jobs, ncells, c->pic_w, c->pic_h, cell_count, cell_members, cell_rec, counters, grp_list);
  hipLaunchKernelGGL(k_ss_prep3_singles, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, c->pic_w, c->pic_h, slot_of, cell_rec, counters, list);
  // persistent grids: 3 workgroups per CU fit by LDS (48.6 KB each); a few more rounds of them smooth the tail
  const unsigned grid = (unsigned)std::min<size_t>((size_t)n * SS_MAX_TILES, (size_t)256 * 3 * 4);
  if (fam) hipLaunchKernelGGL(k_ss_family, dim3(grid), dim3(256), 0, c->stream, d_jobs, pic, counters + 1, grp_list, cell_rec, cell_members, keys);
  hipLaunchKernelGGL(k_ss_search, dim3(grid), dim3(256), 0, c->stream, d_jobs, pic, counters, list, keys);
  hipLaunchKernelGGL(k_ss_finalize, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, keys, pic, c->ss_buf[0], d_res, n);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "ss_search launch: %s", hipGetErrorString(e));
  return HOP_OK;
}
