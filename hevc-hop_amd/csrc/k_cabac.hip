// k_cabac.hip -- CABAC bit estimator for residual coding: the first building block of SURVEY 8(a) row a0 / 8(f)-1
// (it supplies the rate terms of rows a8, a8b and the tables of a11).
// Replaces: ContextModel::init (TLibCommon/ContextModel.cpp:56-65) with the fork's initialisation tables
// (TLibCommon/ContextTables.h:340-546, five slice types), the state transition / fractional bit tables (:67-128,
// FAST_BIT_EST), TEncSbac::estBit (TLibEncoder/TEncSbac.cpp:2175-2370) and -- on the device, for batches of TUs --
// TEncSbac::codeCoeffNxN (:1829-2092) + codeLastSignificantXY (:1772-1827) + codeTransformSkipFlags (:1608-1628) +
// xWriteCoefRemainExGolomb (:381-402) driven through the counting bin coder (TEncBinCoderCABACCounter.cpp:72-108).
//
// Mapping: the coder is a serial walk with adaptive context states, so ONE LANE codes one TU (64 TUs per wave); the
// 150 context states of each lane live in LDS ([state][lane] bytes), the 64-bit coefficient-group map in a register
// pair.  Every TU starts from the context snapshot its job names -- the RD search of the reference loads a snapshot
// before every test (TEncSbac::load, m_pppcRDSbacCoder) -- and can hand back the updated states.
#include <string.h>
#include "hop_dev.h"

__constant__ uint8_t c_next_mps[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
__constant__ uint8_t c_next_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
#define HOP_ENTROPY_BITS { \
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9, \
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95, \
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00, \
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3, \
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0, \
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577, \
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057, \
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb }
__constant__ int32_t c_entropy_bits[128] = HOP_ENTROPY_BITS;
static const int32_t h_entropy_bits[128] = HOP_ENTROPY_BITS;
__constant__ uint8_t c_cb_group_idx[32] = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };
__constant__ uint8_t c_cb_ctx_ind_map[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

// offsets of the context sets inside hop_cabac_ctx.state (the reference's set order, TEncSbac.cpp:76-88)
#define CX_QT_CBF 0
#define CX_TRANS_SUBDIV 8
#define CX_ROOT_CBF 11
#define CX_SIG_CG 12
#define CX_SIG 16
#define CX_LAST_X 58
#define CX_LAST_Y 88
#define CX_ONE 118
#define CX_ABS 142
#define CX_TS 148
#define CX_COUNT 150

// (context initialisation and TEncSbac::estBit are host logic: host/hop_hostlogic.cpp)

// ---- device: counted bits of codeCoeffNxN, one lane per TU ----
struct CabacLds { uint8_t st[172][64]; uint16_t absCoeff[16][64]; };      // rows 0..151: hop_cabac_ctx; 152..171: the CU-level sets of hop_cabac_cu_ctx (k_rqt.inl)

#define CBIN(idx, b) do { const int i_ = (idx); const uint8_t s_ = sh.st[i_][lane]; const int b_ = (b); frac += (unsigned long long)c_entropy_bits[s_ ^ b_]; \
                          sh.st[i_][lane] = ((s_ & 1) == b_) ? c_next_mps[s_] : c_next_lps[s_]; } while (0)

// counted bits of one TU on the lane's context states: the coded_block_flag if asked for (encodeQtCbf), then codeCoeffNxN
struct CabacLds1 { uint8_t st[172][1]; uint16_t absCoeff[16][1]; };         // the same for ONE lane (a TU or CU with a wave of its own): 204 bytes instead of 13 KB

template <class LDS>
__device__ static unsigned long long cb_code_tu_at(LDS& sh, const int lane, const int32_t* __restrict__ coef, const int log2, const int chroma, const int scan_idx,
                                                   const int sign_hide, const int use_ts, const int ts_flag, const int cbf_ctx_plus1,
                                                   const uint16_t* __restrict__ scan, const uint16_t* __restrict__ scanCG);
template <class LDS>
__device__ static unsigned long long cb_code_tu(LDS& sh, const int lane, const int32_t* __restrict__ coef, const int log2, const int chroma, const int scan_idx,
                                                const int sign_hide, const int use_ts, const int ts_flag, const int cbf_ctx_plus1, const uint16_t* __restrict__ scans) {
  const int so = (log2 == 2) ? 0 : (log2 == 3) ? 16 : (log2 == 4) ? 80 : 336, co = (log2 == 2) ? 0 : (log2 == 3) ? 1 : (log2 == 4) ? 5 : 21;
  return cb_code_tu_at(sh, lane, coef, log2, chroma, scan_idx, sign_hide, use_ts, ts_flag, cbf_ctx_plus1, scans + scan_idx * 1360 + so, scans + 4080 + scan_idx * 85 + co);
}
// the same with the TU's scan and coefficient-group scan given (a caller that holds them in LDS)
template <class LDS>
__device__ static unsigned long long cb_code_tu_at(LDS& sh, const int lane, const int32_t* __restrict__ coef, const int log2, const int chroma, const int scan_idx,
                                                   const int sign_hide, const int use_ts, const int ts_flag, const int cbf_ctx_plus1,
                                                   const uint16_t* __restrict__ scan, const uint16_t* __restrict__ scanCG) {
  unsigned long long frac = 0;
  const int width = 1 << log2, nco = width * width;
  int numSig = 0;
  for (int i = 0; i < nco; i++) numSig += coef[i] != 0;
  if (cbf_ctx_plus1) CBIN(CX_QT_CBF + cbf_ctx_plus1 - 1, numSig != 0 ? 1 : 0);      // encodeQtCbf (TEncSbac.cpp:1596-1600)
  if (numSig != 0) {
    if (use_ts && width == 4) CBIN(CX_TS + chroma, ts_flag ? 1 : 0);
    unsigned long long cgFlag = 0;
    const int numBlkSide = width >> 2;
    int scanPosLast = -1, posLast;
    do {
      posLast = scan[++scanPosLast];
      const int py = posLast >> log2, px = posLast - (py << log2);
      const int c = coef[posLast];
      if (c) cgFlag |= 1ull << (numBlkSide * (py >> 2) + (px >> 2));
      numSig -= (c != 0);
    } while (numSig > 0);
    {                                                            // codeLastSignificantXY
      int posY = posLast >> log2, posX = posLast - (posY << log2);
      if (scan_idx == 2) { const int t = posX; posX = posY; posY = t; }
      const int gX = c_cb_group_idx[posX], gY = c_cb_group_idx[posY], gMax = c_cb_group_idx[width - 1];
      const int cb = log2 - 2;
      const int off = chroma ? 0 : (cb * 3 + ((cb + 1) >> 2)), shf = chroma ? cb : ((cb + 3) >> 2);
      const int bx = CX_LAST_X + 15 * chroma + off, by = CX_LAST_Y + 15 * chroma + off;
      int k;
      for (k = 0; k < gX; k++) CBIN(bx + (k >> shf), 1);
      if (gX < gMax) CBIN(bx + (k >> shf), 0);
      for (k = 0; k < gY; k++) CBIN(by + (k >> shf), 1);
      if (gY < gMax) CBIN(by + (k >> shf), 0);
      if (gX > 3) frac += 32768ull * (unsigned long long)((gX - 2) >> 1);
      if (gY > 3) frac += 32768ull * (unsigned long long)((gY - 2) >> 1);
    }
    const int baseCG = CX_SIG_CG + 2 * chroma, baseSig = CX_SIG + (chroma ? 27 : 0);
    const int lastScanSet = scanPosLast >> 4;
    unsigned c1 = 1, goRice = 0;
    int scanPosSig = scanPosLast;
    for (int subSet = lastScanSet; subSet >= 0; subSet--) {
      int numNonZero = 0;
      const int subPos = subSet << 4;
      goRice = 0;
      unsigned coeffSigns = 0;
      int lastNZ = -1, firstNZ = 16;
      if (scanPosSig == scanPosLast) {
        const int c = coef[posLast];
        sh.absCoeff[0][lane] = (uint16_t)(c < 0 ? -c : c); coeffSigns = (c < 0); numNonZero = 1;
        lastNZ = scanPosSig; firstNZ = scanPosSig; scanPosSig--;
      }
      const int cgBlkPos = scanCG[subSet], cgPosY = cgBlkPos / numBlkSide, cgPosX = cgBlkPos - cgPosY * numBlkSide;
      unsigned r = 0, l = 0;
      if (cgPosX < numBlkSide - 1) r = (unsigned)((cgFlag >> (cgPosY * numBlkSide + cgPosX + 1)) & 1ull);
      if (cgPosY < numBlkSide - 1) l = (unsigned)((cgFlag >> ((cgPosY + 1) * numBlkSide + cgPosX)) & 1ull);
      if (subSet == lastScanSet || subSet == 0) cgFlag |= 1ull << cgBlkPos;
      else CBIN(baseCG + ((r || l) ? 1 : 0), (int)((cgFlag >> cgBlkPos) & 1ull));
      if ((cgFlag >> cgBlkPos) & 1ull) {
        const int patternSigCtx = (width == 4) ? -1 : (int)(r + (l << 1));
        for (; scanPosSig >= subPos; scanPosSig--) {
          const int blkPos = scan[scanPosSig], posY = blkPos >> log2, posX = blkPos - (posY << log2);
          const int c = coef[blkPos];
          const int sig = (c != 0);
          if (scanPosSig > subPos || subSet == 0 || numNonZero) {
            int ctxSig;                                            // getSigCtxInc, TComTrQuant.cpp:2038-2092
            if (posX + posY == 0) ctxSig = 0;
            else if (log2 == 2) ctxSig = c_cb_ctx_ind_map[4 * posY + posX];
            else {
              const int offset = log2 == 3 ? (scan_idx == 0 ? 9 : 15) : (!chroma ? 21 : 12);
              const int xs = posX & 3, ys = posY & 3;
              int cnt;
              if (patternSigCtx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
              else if (patternSigCtx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
              else if (patternSigCtx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
              else cnt = 2;
              ctxSig = ((!chroma && ((posX >> 2) + (posY >> 2)) > 0) ? 3 : 0) + offset + cnt;
            }
            CBIN(baseSig + ctxSig, sig);
          }
          if (sig) {
            const int a = c < 0 ? -c : c;
            sh.absCoeff[numNonZero][lane] = (uint16_t)a;                 // |TCoeff| <= 32768
            coeffSigns = 2 * coeffSigns + (c < 0);
            numNonZero++;
            if (lastNZ == -1) lastNZ = scanPosSig;
            firstNZ = scanPosSig;
          }
        }
      } else {
        scanPosSig = subPos - 1;
      }
      if (numNonZero > 0) {
        const int signHidden = (lastNZ - firstNZ >= 4);
        unsigned ctxSet = (subSet > 0 && !chroma) ? 2 : 0;
        if (c1 == 0) ctxSet++;
        c1 = 1;
        const int baseOne = CX_ONE + (chroma ? 16 : 0) + 4 * ctxSet;
        const int numC1 = numNonZero < 8 ? numNonZero : 8;
        int firstC2 = -1;
        for (int idx = 0; idx < numC1; idx++) {
          const int sym = sh.absCoeff[idx][lane] > 1;
          CBIN(baseOne + c1, sym);
          if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = idx; }
          else if ((c1 < 3) && (c1 > 0)) c1++;
        }
        if (c1 == 0 && firstC2 != -1) CBIN(CX_ABS + (chroma ? 4 : 0) + ctxSet, sh.absCoeff[firstC2][lane] > 2);
        if (sign_hide && signHidden) frac += 32768ull * (unsigned long long)(numNonZero - 1);
        else frac += 32768ull * (unsigned long long)numNonZero;
        int firstCoeff2 = 1;
        if (c1 == 0 || numNonZero > 8) {
          for (int idx = 0; idx < numNonZero; idx++) {
            const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
            const int a = sh.absCoeff[idx][lane];
            if (a >= baseLevel) {                                // xWriteCoefRemainExGolomb: only the number of bypass bins counts
              int codeNumber = a - baseLevel; unsigned length;
              if (codeNumber < (3 << goRice)) { length = (unsigned)codeNumber >> goRice; frac += 32768ull * (unsigned long long)(length + 1 + goRice); }
              else {
                length = goRice; codeNumber -= (3 << goRice);
                while (codeNumber >= (1 << length)) codeNumber -= (1 << (length++));
                frac += 32768ull * (unsigned long long)(3 + length + 1 - goRice + length);
              }
              if (a > 3 * (1 << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
            }
            if (a >= 2) firstCoeff2 = 0;
          }
        }
      }
    }
  }
  return frac;
}

__global__ __launch_bounds__(64) void k_coeff_bits(const hop_coeff_bits_job* __restrict__ jobs, int n, const hop_cabac_ctx* __restrict__ ctx_in,
                                                   const uint16_t* __restrict__ scans, const int32_t* __restrict__ coef_all,
                                                   unsigned long long* __restrict__ bits_out, hop_cabac_ctx* __restrict__ ctx_out) {
  __shared__ CabacLds sh;
  const int lane = threadIdx.x, j = blockIdx.x * 64 + lane;
  const bool inb = j < n;
  const hop_coeff_bits_job jb = jobs[inb ? j : 0];
  const bool live = inb && jb.log2_size >= 2;                         // < 2: an empty slot of a job table (k_rqt.inl)
  {
    const uint8_t* src = ctx_in[jb.ctx_index].state;
    for (int i = 0; i < 152; i++) sh.st[i][lane] = src[i];
  }
  const unsigned long long frac = live ? cb_code_tu(sh, lane, coef_all + jb.coeff_offset, jb.log2_size, jb.comp != 0, jb.scan_idx, jb.sign_hide, jb.use_ts, jb.ts_flag,
                                                   jb.cbf_ctx_plus1, scans) : 0ull;
  if (live) {
    bits_out[j] = frac;
    if (ctx_out) {
      uint8_t* dst = ctx_out[j].state;
      for (int i = 0; i < 150; i++) dst[i] = sh.st[i][lane];
      const unsigned left = ((unsigned)sh.st[150][lane] | ((unsigned)sh.st[151][lane] << 8)) + (unsigned)(frac & 32767ull);   // what resetBits will keep
      dst[150] = (uint8_t)(left & 0xFF); dst[151] = (uint8_t)((left >> 8) & 0x7F);
    }
  }
}

const int32_t* hop_entropy_bits_host(void) { return h_entropy_bits; }

int hop_launch_coeff_bits(hop_ctx* c, int n, const hop_coeff_bits_job* d_jobs, const hop_cabac_ctx* d_ctx, const int32_t* d_coef,
                          unsigned long long* d_bits, hop_cabac_ctx* d_ctx_out) {
  const int pr = hop_prof_begin(c, HOP_K_CABAC, (uint64_t)n);
  hipLaunchKernelGGL(k_coeff_bits, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_jobs, n, d_ctx, c->rdoq_scans, d_coef, d_bits, d_ctx_out);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "coeff_bits launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

// ---- the mode-decision half of the intra rough search (rest of row a7): TEncSearch::estIntraPredQT :2440-2493 ----
// per block: for the 35 modes bits of the luma direction from the CI_CURR_BEST state (xModeBitsIntra :7734 = codeIntraDirLumaAng TEncSbac.cpp:770-831),
// cost = SATD + bits * sqrt(lambda) in double (:2461), the sorted candidate list (xUpdateCandList :7747-7767), the MPMs appended if missing (:2466-2488).
__device__ static inline void intra_modes_body(const int i, const hop_intra_modes_job* jobs, const uint32_t* satd, hop_intra_modes_result* res) {
  const hop_intra_modes_job jb = jobs[i];
  const uint32_t* sd = satd + (size_t)i * 35;
  uint32_t modes[11]; double costs[8];
  const int nf = jb.num_full_rd;
  for (int q = 0; q < 11; q++) modes[q] = 0;
  for (int q = 0; q < 8; q++) costs[q] = 1.7e+308;
  const unsigned left = (unsigned)jb.frac_left & 32767u;
  for (int mode = 0; mode < 35; mode++) {
    int idx = -1;
    for (int q = 0; q < jb.pred_num; q++) if (mode == jb.preds[q]) idx = q;
    const unsigned long long frac = left + (unsigned long long)c_entropy_bits[jb.ctx_state ^ (idx != -1 ? 1 : 0)] + 32768ull * (unsigned long long)(idx == -1 ? 5 : (idx ? 2 : 1));
    const double cost = (double)sd[mode] + (double)(uint32_t)(frac >> 15) * jb.sqrt_lambda;
    int shift = 0;
    while (shift < nf && cost < costs[nf - 1 - shift]) shift++;
    if (shift) {
      for (int q = 1; q < shift; q++) { modes[nf - q] = modes[nf - 1 - q]; costs[nf - q] = costs[nf - 1 - q]; }
      modes[nf - shift] = (uint32_t)mode; costs[nf - shift] = cost;
    }
  }
  int cnt = nf;
  for (int j = 0; j < jb.mpm_cand; j++) {
    bool inc = false;
    for (int q = 0; q < cnt; q++) inc |= (jb.preds[j] == (int)modes[q]);
    if (!inc) modes[cnt++] = (uint32_t)jb.preds[j];
  }
  hop_intra_modes_result r;
  r.n = (uint32_t)cnt;
  for (int q = 0; q < 11; q++) r.modes[q] = modes[q];
  for (int q = 0; q < 8; q++) r.costs[q] = costs[q];
  res[i] = r;
}
__global__ void k_intra_modes(const hop_intra_modes_job* __restrict__ jobs, int n, const uint32_t* __restrict__ satd, hop_intra_modes_result* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) intra_modes_body(i, jobs, satd, res);
}
int hop_launch_intra_modes(hop_ctx* c, int n, const hop_intra_modes_job* d_jobs, const uint32_t* d_satd, hop_intra_modes_result* d_res) {
  const int pr = hop_prof_begin(c, HOP_K_INTRA, 0);
  hipLaunchKernelGGL(k_intra_modes, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_jobs, n, d_satd, d_res);
  hop_prof_end(c, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "intra_modes launch: %s", hipGetErrorString(e));
  return HOP_OK;
}

#include "k_intra_dev.inl"
#include "k_leaf_fused.inl"
#include "k_rqt.inl"
#include "k_walk.inl"
