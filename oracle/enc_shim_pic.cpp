// oracle/enc_shim_pic.cpp -- the PICTURE-level reference-side binding of include/hophip.h, compiled against the reference's own headers.
//
// One member of the reference encoder is re-defined:
//   TEncCu::compressCU   TLibEncoder/TEncCu.cpp:246-264 (xCompressCU :371-892 and everything below it)
// At the first CTU of a picture the whole picture goes through hop_encode_frame (original uploaded, RD search of every CTU on the device, decisions by the library's host
// spine); each compressCU call then only fills the CTU's TComDataCU -- the per-partition arrays of TComDataCU.h:96-170 from hop_cu_part, the levels m_pcTrCoeffY / Cb / Cr
// from hop_levels_download, getTotalCost / Bits / Distortion from the per-CTU outputs -- and copies the CTU's reconstruction into the picture (what xCopyYuv2Pic leaves
// there, TEncCu.cpp:1623-1662).  Everything after it is the reference's own object code: the counting pass of TEncSlice::compressSlice, TEncSlice::encodeSlice, the loop
// filters, SAO, the picture hash, the bitstream writer.
// oracle/Makefile.ref links this file with the reference's objects (compressCU weakened with objcopy, nothing of the reference edited or copied) and with libhophip.so
// into oracle/_ref/TAppEncoderPic.  That program runs on the GPU box (tests/test_gpu_encoder_pic.py): the bitstream and the reconstruction it writes must be, byte for
// byte, those of the unmodified reference encoder (tests/golden/encoder_hop_qp32.json) -- every decision, level and reconstructed sample the entropy coder and the loop
// filters consume came out of hop_encode_frame.
// With -DHOP_PIC_CPU the six hop_* entries the binding calls are adapters inside this file over the CPU spine (oracle/libhop_spine_cpu.so, named by HOP_PIC_SPINE):
// oracle/_ref/TAppEncoderPicCpu runs in the build container (tests/test_encoder_pic.py), so the marshalling below is checked without a GPU, and the CPU spine's levels and
// partition data are pinned to the reference's bitstream as well.
// Further switches (all off by default; documented where they are implemented below): HOP_PIC_DEBLOCK / HOP_PIC_SAO replace TComLoopFilter::loopFilterPic and
// TEncSampleAdaptiveOffset::SAOProcess by hop_deblock_frame / hop_sao_frame (+ hop_psnr); HOP_PIC_CHECK lets the reference's own member run on the same input and compares;
// HOP_PIC_LF_FUZZ / HOP_PIC_SAO_FUZZ feed both with random pictures and HOP_PIC_LF_DUMP / HOP_PIC_SAO_DUMP write the reference's answers out as fixtures.  Besides the HOP
// configuration (ISS slices, 8 bit) the binding carries the plain intra configurations (I slices, 8 and 10 bit) through hop_encode_frame's plain_intra mode.
// This file never touches the CPU restatement except through those adapters.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cassert>
#include <vector>
#include <list>
#include <map>
#include <string>
#include <sstream>
#include <iostream>
#include <fstream>
#include <algorithm>
#include <stdint.h>
#define private public
#define protected public
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComPic.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncCu.h"
#include "TLibEncoder/TEncSbac.h"
#include "TLibEncoder/TEncBinCoderCABAC.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibEncoder/TEncSampleAdaptiveOffset.h"
#undef private
#undef protected
#include "../include/hophip.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
// HOP_PIC_CHECK runs the reference's own RD search, whose GT search reads past the end of its reference picture buffer (TEncSearch::xExtDIFUpSamplingH under xPatternSearchGT,
// TEncSearch.cpp:4686-): when that lands on an unmapped page the process dies -- the unmodified encoder does too, now and then.  Say so and exit with a code of its own.
static void reference_faulted(int) { static const char m[] = "hop pic check: the reference's own code faulted\n"; if (write(2, m, sizeof(m) - 1)) {} void* bt[40]; backtrace_symbols_fd(bt, backtrace(bt, 40), 2); _exit(77); }

#ifdef HOP_PIC_CPU
#include <dlfcn.h>
// the six entries of the library the binding uses, over the CPU spine (same argument meaning; the context is a record of the picture)
struct hop_ctx { int w, h, bd; std::vector<int16_t> org[3], rec[3]; };
namespace {
typedef long (*enc_fn)(int, int, int, int, int, const int16_t*, const int16_t*, const int16_t*, const char*, double*, uint32_t*, uint32_t*, void*, int16_t*, int16_t*, int16_t*, void*);
typedef long (*wpp_fn)(int, int, int, int, int, const int16_t*, const int16_t*, const int16_t*, const char*, double*, uint32_t*, uint32_t*, void*, int16_t*, int16_t*, int16_t*, double*);
typedef long (*lev_fn)(int32_t*, long); typedef long (*frac_fn)(uint16_t*, long);
typedef long (*plain_fn)(int, int, int, int, const int16_t*, const int16_t*, const int16_t*, const char*, double*, uint32_t*, uint32_t*, void*, int16_t*, int16_t*, int16_t*);
typedef int (*dbk_fn)(int, int, int, int, int, int, int, int, int, const void*, int16_t*, int16_t*, int16_t*);
typedef int (*sst_fn)(int, int, int, const int16_t* const*, const int16_t* const*, int32_t*); typedef int (*sap_fn)(int, int, int, const int16_t* const*, const void*, int16_t* const*);
typedef int (*sdc_fn)(int, int, int, const int32_t*, const hop_sao_params*, hop_sao_param*, hop_sao_param*);
enc_fn g_enc = NULL; wpp_fn g_wpp = NULL; lev_fn g_lev = NULL; frac_fn g_frac = NULL; dbk_fn g_dbk = NULL; sst_fn g_sst = NULL; sap_fn g_sap = NULL; sdc_fn g_sdc = NULL; plain_fn g_plain = NULL; std::string g_err = "no error";
}
extern "C" {
int hop_ctx_create(hop_ctx** out, int w, int h, int bdy, int bdc, int) {
  const char* so = getenv("HOP_PIC_SPINE");
  void* lib = so ? dlopen(so, RTLD_NOW | RTLD_LOCAL) : NULL;
  if (!lib) { g_err = so ? dlerror() : "HOP_PIC_SPINE names the CPU spine library"; return HOP_ERR_DEVICE; }
  g_enc = (enc_fn)dlsym(lib, "hop_spine_cpu_encode"); g_wpp = (wpp_fn)dlsym(lib, "hop_spine_cpu_encode_wpp"); g_lev = (lev_fn)dlsym(lib, "hop_spine_cpu_last_levels"); g_frac = (frac_fn)dlsym(lib, "hop_spine_cpu_last_rd_fraction"); g_dbk = (dbk_fn)dlsym(lib, "hop_o_deblock_frame"); g_sst = (sst_fn)dlsym(lib, "hop_o_sao_stats"); g_sap = (sap_fn)dlsym(lib, "hop_o_sao_apply"); g_sdc = (sdc_fn)dlsym(lib, "hop_sao_decide"); g_plain = (plain_fn)dlsym(lib, "hop_spine_cpu_encode_plain");
  int (*szp)(void) = (int (*)(void))dlsym(lib, "hop_spine_sizeof_part");
  if (!g_enc || !g_wpp || !g_lev || !g_frac || !g_dbk || !g_sst || !g_sap || !g_sdc || !g_plain || !szp || szp() != (int)sizeof(hop_cu_part) || bdy != bdc || (bdy != 8 && bdy != 10)) { g_err = "not the spine library this binding was written for"; return HOP_ERR_DEVICE; }
  *out = new hop_ctx(); (*out)->w = w; (*out)->h = h; (*out)->bd = bdy; return HOP_OK;
}
void hop_ctx_destroy(hop_ctx* c) { delete c; }
const char* hop_last_error(const hop_ctx*) { return g_err.c_str(); }
int hop_ctx_set_slots(hop_ctx*, int) { return HOP_OK; }
int hop_upload_orig(hop_ctx* c, const int16_t* y, int sy, const int16_t* cb, const int16_t* cr, int sc) {
  const int16_t* src[3] = { y, cb, cr };
  for (int k = 0; k < 3; k++) { const int w = k ? c->w / 2 : c->w, h = k ? c->h / 2 : c->h, s = k ? sc : sy; c->org[k].resize((size_t)w * h); c->rec[k].assign((size_t)w * h, 0);
                                for (int r = 0; r < h; r++) memcpy(&c->org[k][(size_t)r * w], src[k] + (size_t)r * s, w * 2); }
  return HOP_OK;
}
int hop_encode_frame(hop_ctx* c, const hop_enc_params* p, double* cost, uint32_t* bits, uint32_t* dist, hop_cu_part* parts, uint64_t* nc) {
  if (p->first_ctus || (p->plain_intra && (p->wpp || p->wavefront_lag)) || (!p->plain_intra && c->bd != 8)) { g_err = "the adapter covers whole pictures: the HOP configuration (8 bit), the plain intra ones in raster order"; return HOP_ERR_ARG; }
  const long n = p->plain_intra ? g_plain(c->w, c->h, p->qp, c->bd, &c->org[0][0], &c->org[1][0], &c->org[2][0], p->trace_path, cost, bits, dist, parts, &c->rec[0][0], &c->rec[1][0], &c->rec[2][0]) :
                 (p->wpp || p->wavefront_lag) ? g_wpp(c->w, c->h, p->qp, p->mi_size, p->wavefront_lag, &c->org[0][0], &c->org[1][0], &c->org[2][0], p->trace_path, cost, bits, dist, parts,
                                                      &c->rec[0][0], &c->rec[1][0], &c->rec[2][0], NULL)
                                               : g_enc(c->w, c->h, p->qp, p->mi_size, 0, &c->org[0][0], &c->org[1][0], &c->org[2][0], p->trace_path, cost, bits, dist, parts,
                                                      &c->rec[0][0], &c->rec[1][0], &c->rec[2][0], NULL);
  if (n <= 0) { g_err = "the spine failed"; return HOP_ERR_DEVICE; }
  if (nc) *nc = (uint64_t)n;
  return HOP_OK;
}
int hop_levels_download(hop_ctx* c, int32_t* out) { const long n = (long)((c->w + 63) / 64) * ((c->h + 63) / 64) * 6144; return g_lev(out, n) == n ? HOP_OK : HOP_ERR_DEVICE; }
int hop_rd_fraction_download(hop_ctx* c, uint16_t* out) { const long n = (long)((c->w + 63) / 64) * ((c->h + 63) / 64); return g_frac(out, n) == n ? HOP_OK : HOP_ERR_DEVICE; }
int hop_deblock_frame(hop_ctx* c, const hop_deblock_params* p, const hop_cu_part* parts) {
  return g_dbk(c->w, c->h, c->bd, p->qp, p->beta_offset_div2, p->tc_offset_div2, p->cb_qp_offset, p->cr_qp_offset, p->disable, parts, &c->rec[0][0], &c->rec[1][0], &c->rec[2][0]) == 0 ? HOP_OK : HOP_ERR_DEVICE;
}
int hop_sao_stats(hop_ctx* c, int32_t* stats) { const int16_t* s[3] = { &c->rec[0][0], &c->rec[1][0], &c->rec[2][0] }; const int16_t* o[3] = { &c->org[0][0], &c->org[1][0], &c->org[2][0] }; return g_sst(c->w, c->h, c->bd, s, o, stats) == 0 ? HOP_OK : HOP_ERR_DEVICE; }
int hop_sao_frame(hop_ctx* c, const hop_sao_params* p, hop_sao_param* coded) {      // the product's three steps: statistics and offsetting by the restatement, the decision by the product's host logic
  const int wctu = (c->w + 63) / 64, n = wctu * ((c->h + 63) / 64);
  std::vector<int32_t> st((size_t)n * 3 * 5 * 32 * 2); std::vector<hop_sao_param> recon((size_t)n * 3);
  if (hop_sao_stats(c, &st[0]) != HOP_OK || g_sdc(n, wctu, c->bd, &st[0], p, coded, &recon[0]) != HOP_OK) return HOP_ERR_DEVICE;
  std::vector<int16_t> out[3]; const int16_t* s[3]; int16_t* d[3];
  for (int k = 0; k < 3; k++) { out[k].resize(c->rec[k].size()); s[k] = &c->rec[k][0]; d[k] = &out[k][0]; }
  if (g_sap(c->w, c->h, c->bd, s, &recon[0], d) != 0) return HOP_ERR_DEVICE;
  for (int k = 0; k < 3; k++) c->rec[k] = out[k];
  return HOP_OK;
}
int hop_psnr(hop_ctx* c, uint64_t*, double* psnr) {                        // (plain loops: the adapter has no restatement to call for three sums)
  for (int k = 0; k < 3; k++) { unsigned long long s = 0; for (size_t i = 0; i < c->rec[k].size(); i++) { const int d = c->org[k][i] - c->rec[k][i]; s += (unsigned long long)(d * d); }
                                const double mv = (double)(255 << (c->bd - 8)), ref = mv * mv * c->w * c->h / (k ? 4.0 : 1.0); psnr[k] = s ? 10.0 * log10(ref / (double)s) : 99.99; }
  return HOP_OK;
}
int hop_recon_upload(hop_ctx* c, int comp, const int16_t* src) { memcpy(&c->rec[comp][0], src, c->rec[comp].size() * 2); return HOP_OK; }
int hop_recon_download(hop_ctx* c, int comp, int16_t* dst) { memcpy(dst, &c->rec[comp][0], c->rec[comp].size() * 2); return HOP_OK; }
}
#endif

namespace {
struct Binding {
  hop_ctx* ctx; const TComPic* pic; int w, h, wctu, n;
  std::vector<double> cost; std::vector<uint32_t> bits, dist; std::vector<hop_cu_part> parts; std::vector<int32_t> levels; std::vector<uint16_t> fraction; std::vector<int16_t> rec[3];
  unsigned long pictures, ctus, deblocked, sao; unsigned long long candidates;
  Binding() : ctx(NULL), pic(NULL), w(0), h(0), wctu(0), n(0), pictures(0), ctus(0), deblocked(0), sao(0), candidates(0) {}
  ~Binding() { if (getenv("HOP_SHIM_REPORT")) fprintf(stderr, "hop pic binding: pictures %lu ctus %lu candidates %llu deblocked %lu sao %lu\n", pictures, ctus, candidates, deblocked, sao); if (ctx) hop_ctx_destroy(ctx); }
  void fail(const char* what) { fprintf(stderr, "hop pic binding: %s failed: %s\n", what, hop_last_error(ctx)); exit(1); }
  static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v && *v ? atoi(v) : dflt; }

  // one picture through the library: TEncGOP::compressGOP has set the slice up (QP, lambda, ISS type); TEncSlice::compressSlice is about to loop over its CTUs
  void code_picture(TEncCu* enc, TComDataCU* cu) {
    TComSlice* sl = cu->getSlice(); TEncCfg* cfg = enc->m_pcEncCfg;
    // the HOP configuration (ISS slices, 8 bit) or the plain intra configurations (I slices, 8 or 10 bit: cfg/encoder_intra_main.cfg, encoder_intra_main10.cfg)
    const bool iss = sl->isIntraSS();
    if ((!iss && !sl->isIntra()) || g_bitDepthY != g_bitDepthC || (iss && g_bitDepthY != 8) || (g_bitDepthY != 8 && g_bitDepthY != 10) || g_uiMaxCUWidth != 64 || g_uiMaxCUDepth != 4) { fprintf(stderr, "hop pic binding: bound for ISS pictures at 8 bit and I pictures at 8 / 10 bit, 64x64 CTUs\n"); exit(1); }
    if (!ctx) {
      w = sl->getSPS()->getPicWidthInLumaSamples(); h = sl->getSPS()->getPicHeightInLumaSamples(); wctu = (w + 63) / 64; n = wctu * ((h + 63) / 64);
      if (hop_ctx_create(&ctx, w, h, g_bitDepthY, g_bitDepthC, 0) != HOP_OK) fail("hop_ctx_create");
      if (iss && hop_ctx_set_slots(ctx, env_int("HOP_PIC_SLOTS", 16)) != HOP_OK) fail("hop_ctx_set_slots");              // the candidates of a CU side by side (no effect on the results)
      cost.resize(n); bits.resize(n); dist.resize(n); parts.resize((size_t)n * 256); levels.resize((size_t)n * 6144); fraction.resize(n);
      rec[0].resize((size_t)w * h); rec[1].resize((size_t)w * h / 4); rec[2].resize((size_t)w * h / 4);
    }
    pic = cu->getPic();
    TComPicYuv* org = cu->getPic()->getPicYuvOrg();
    if (hop_upload_orig(ctx, org->getLumaAddr(), org->getStride(), org->getCbAddr(), org->getCrAddr(), org->getCStride()) != HOP_OK) fail("hop_upload_orig");
    hop_enc_params p; memset(&p, 0, sizeof(p));
    p.qp = sl->getSliceQp(); p.mi_size = iss ? sl->getMicroImSize() : 16; p.plain_intra = iss ? 0 : 1;
    p.wpp = cfg->getWaveFrontsynchro() ? 1 : 0;                                                                    // TEncSlice.cpp:1027-1051: the rows' coders synchronised
    p.wavefront_lag = p.wpp ? env_int("HOP_PIC_LAG", 5) : 0;                                                       // rows in flight together only where the reference's rows are independent
    uint64_t nc = 0;
    if (hop_encode_frame(ctx, &p, &cost[0], &bits[0], &dist[0], &parts[0], &nc) != HOP_OK) fail("hop_encode_frame");
    if (hop_levels_download(ctx, &levels[0]) != HOP_OK) fail("hop_levels_download");
    if (hop_rd_fraction_download(ctx, &fraction[0]) != HOP_OK) fail("hop_rd_fraction_download");
    for (int k = 0; k < 3; k++) if (hop_recon_download(ctx, k, &rec[k][0]) != HOP_OK) fail("hop_recon_download");
    pictures++; candidates += nc;
  }

  void fill(TComDataCU* c) {
    const int addr = (int)c->getAddr(), qp = c->getSlice()->getSliceQp();
    const hop_cu_part* pp = &parts[(size_t)addr * 256];
    for (UInt i = 0; i < c->getTotalNumPart(); i++) {
      const hop_cu_part& p = pp[i];
      c->m_puhDepth[i] = p.depth; c->m_puhWidth[i] = c->m_puhHeight[i] = (UChar)(64 >> p.depth);
      if (p.pred_mode == 15) continue;                                   // MODE_NONE: outside the picture -- initSubCU's values at the depth the quadtree left it (TEncCu.cpp:727-775)
      c->m_pePartSize[i] = (Char)p.part_size; c->m_pePredMode[i] = (Char)p.pred_mode; c->m_skipFlag[i] = p.skip != 0;
      c->m_pbMergeFlag[i] = p.merge_flag != 0; c->m_puhMergeIndex[i] = p.merge_idx; c->m_puhInterDir[i] = p.inter_dir;
      c->m_acCUMvField[0].m_pcMv[i].set(p.mv[0], p.mv[1]); c->m_acCUMvField[0].m_pcMvd[i].set(p.mvd[0], p.mvd[1]); c->m_acCUMvField[0].m_piRefIdx[i] = p.ref_idx;
      c->m_apiMVPIdx[0][i] = p.mvp_idx; c->m_apiMVPNum[0][i] = p.mvp_num;
      c->m_gtFlag[i] = p.gt_flag != 0;
      c->m_acCUGT0Field[0].m_pcMv[i].set(p.gt[0], p.gt[1]); c->m_acCUGT1Field[0].m_pcMv[i].set(p.gt[2], p.gt[3]);
      c->m_acCUGT2Field[0].m_pcMv[i].set(p.gt[4], p.gt[5]); c->m_acCUGT3Field[0].m_pcMv[i].set(p.gt[6], p.gt[7]);
      c->m_puhLumaIntraDir[i] = p.luma_dir; c->m_puhChromaIntraDir[i] = p.chroma_dir; c->m_puhTrIdx[i] = p.tr_idx;
      for (int k = 0; k < 3; k++) { c->m_puhCbf[k][i] = p.cbf[k]; c->m_puhTransformSkip[k][i] = p.tskip[k]; }
      c->m_phQP[i] = (Char)qp;
    }
    const int32_t* lv = &levels[(size_t)addr * 6144];
    for (int k = 0; k < 4096; k++) c->m_pcTrCoeffY[k] = lv[k];
    for (int k = 0; k < 1024; k++) { c->m_pcTrCoeffCb[k] = lv[4096 + k]; c->m_pcTrCoeffCr[k] = lv[5120 + k]; }
    c->m_dTotalCost = cost[addr]; c->m_uiTotalBits = bits[addr]; c->m_uiTotalDistortion = dist[addr];
    TComPicYuv* r = c->getPic()->getPicYuvRec();
    const int x0 = (addr % wctu) * 64, y0 = (addr / wctu) * 64, bw = std::min(64, w - x0), bh = std::min(64, h - y0);
    for (int y = 0; y < bh; y++) memcpy(r->getLumaAddr() + (size_t)(y0 + y) * r->getStride() + x0, &rec[0][(size_t)(y0 + y) * w + x0], bw * sizeof(Pel));
    for (int y = 0; y < bh / 2; y++) { memcpy(r->getCbAddr() + (size_t)(y0 / 2 + y) * r->getCStride() + x0 / 2, &rec[1][(size_t)(y0 / 2 + y) * (w / 2) + x0 / 2], (bw / 2) * sizeof(Pel));
                                       memcpy(r->getCrAddr() + (size_t)(y0 / 2 + y) * r->getCStride() + x0 / 2, &rec[2][(size_t)(y0 / 2 + y) * (w / 2) + x0 / 2], (bw / 2) * sizeof(Pel)); }
    ctus++;
  }
} g_b;
}

extern "C" void hop_ref_orig_compress_cu(TEncCu*, TComDataCU*&);   // the reference's own definition (Makefile.ref), reached only by the diagnostic below
namespace {
// HOP_PIC_CHECK=1 (diagnostic): the reference's own compressCU codes the CTU first, and every field the binding is about to fill is compared with what it left
void compare_with_reference(TEncCu* enc, TComDataCU*& cu) {
  hop_ref_orig_compress_cu(enc, cu);
  const int addr = (int)cu->getAddr(); const hop_cu_part* pp = &g_b.parts[(size_t)addr * 256]; const int32_t* lv = &g_b.levels[(size_t)addr * 6144];
  int bad = 0;
#define CMP(name, ref, ours) do { if ((long)(ref) != (long)(ours) && bad++ < 40) fprintf(stderr, "hop pic check: CTU %d part %u %s reference %ld binding %ld\n", addr, i, name, (long)(ref), (long)(ours)); } while (0)
  for (UInt i = 0; i < cu->getTotalNumPart(); i++) {
    const hop_cu_part& p = pp[i];
    CMP("depth", cu->m_puhDepth[i], p.depth); CMP("width", cu->m_puhWidth[i], 64 >> p.depth);
    if (p.pred_mode == 15) { CMP("pred_mode(outside)", cu->m_pePredMode[i], (Char)MODE_NONE); continue; }
    CMP("part_size", cu->m_pePartSize[i], p.part_size); CMP("pred_mode", cu->m_pePredMode[i], p.pred_mode); CMP("skip", cu->m_skipFlag[i], p.skip);
    CMP("merge_flag", cu->m_pbMergeFlag[i], p.merge_flag); CMP("merge_idx", cu->m_puhMergeIndex[i], p.merge_idx); CMP("inter_dir", cu->m_puhInterDir[i], p.inter_dir);
    CMP("mv_x", cu->m_acCUMvField[0].m_pcMv[i].getHor(), p.mv[0]); CMP("mv_y", cu->m_acCUMvField[0].m_pcMv[i].getVer(), p.mv[1]);
    CMP("mvd_x", cu->m_acCUMvField[0].m_pcMvd[i].getHor(), p.mvd[0]); CMP("mvd_y", cu->m_acCUMvField[0].m_pcMvd[i].getVer(), p.mvd[1]);
    CMP("ref_idx", cu->m_acCUMvField[0].m_piRefIdx[i], p.ref_idx); CMP("mvp_idx", cu->m_apiMVPIdx[0][i], p.mvp_idx); CMP("mvp_num", cu->m_apiMVPNum[0][i], p.mvp_num);
    CMP("gt_flag", cu->m_gtFlag[i], p.gt_flag); CMP("gt0x", cu->m_acCUGT0Field[0].m_pcMv[i].getHor(), p.gt[0]); CMP("gt3y", cu->m_acCUGT3Field[0].m_pcMv[i].getVer(), p.gt[7]);
    CMP("luma_dir", cu->m_puhLumaIntraDir[i], p.luma_dir); CMP("chroma_dir", cu->m_puhChromaIntraDir[i], p.chroma_dir); CMP("tr_idx", cu->m_puhTrIdx[i], p.tr_idx);
    for (int k = 0; k < 3; k++) { CMP("cbf", cu->m_puhCbf[k][i], p.cbf[k]); CMP("tskip", cu->m_puhTransformSkip[k][i], p.tskip[k]); }
    for (int k = 0; k < 16; k++) CMP("level_y", cu->m_pcTrCoeffY[i * 16 + k], lv[i * 16 + k]);
    for (int k = 0; k < 4; k++) { CMP("level_cb", cu->m_pcTrCoeffCb[i * 4 + k], lv[4096 + i * 4 + k]); CMP("level_cr", cu->m_pcTrCoeffCr[i * 4 + k], lv[5120 + i * 4 + k]); }
  }
  { UInt i = 0; CMP("bits", cu->m_uiTotalBits, g_b.bits[addr]); CMP("distortion", cu->m_uiTotalDistortion, g_b.dist[addr]); CMP("cost", (long)cu->m_dTotalCost, (long)g_b.cost[addr]);
    CMP("rd coder fraction", ((TEncBinCABAC*)enc->m_pcRDGoOnSbacCoder->m_pcBinIf)->m_fracBits & 32767, g_b.fraction[addr]); }
#undef CMP
  fprintf(stderr, "hop pic check: CTU %d: %d differences\n", addr, bad);
}
}

Void TEncCu::compressCU(TComDataCU*& rpcCU)
{
  static const bool check = getenv("HOP_PIC_CHECK") != NULL;
  static bool once = false; if (check && !once) { once = true; signal(SIGSEGV, reference_faulted); }
  if (g_b.pic != rpcCU->getPic() || rpcCU->getAddr() == 0) g_b.code_picture(this, rpcCU);
  if (check) compare_with_reference(this, rpcCU);
  g_b.fill(rpcCU);
  // the state the RD search leaves in the encoder besides the CTU's data: the counting coder's carried fraction, which the SAO parameter decision after the CTU loop
  // inherits (include/hophip.h: hop_rd_fraction_download)
  ((TEncBinCABAC*)m_pcRDGoOnSbacCoder->m_pcBinIf)->m_fracBits = g_b.fraction[rpcCU->getAddr()];
}

// HOP_PIC_DEBLOCK=1: TComLoopFilter::loopFilterPic (TLibCommon/TComLoopFilter.cpp:129-153) replaced as well -- hop_deblock_frame filters the reconstruction the context
// still holds from hop_encode_frame, with the partition data that call returned; the picture's planes are then the library's.  SAO, the picture hash and the bitstream
// follow in the reference's own code, so the md5 of rec.yuv and of the bitstream (SAO parameters, hash SEI) pin the filter.  With HOP_PIC_CHECK the reference's own
// loopFilterPic runs too and the planes are compared sample by sample.
extern "C" void hop_ref_orig_loop_filter_pic(TComLoopFilter*, TComPic*);
namespace {
// HOP_PIC_LF_FUZZ="seed:qp:beta_offset_div2:tc_offset_div2:cb_qp_offset:cr_qp_offset": before the filters run, the picture's partition data and reconstruction are
// replaced by random ones -- a random coding quadtree per CTU (forced splits at the picture border), intra / inter CUs with every partition shape the size allows, random
// transform trees with random luma cbf, small vectors around the 4-quarter-sample threshold, now and then no reference index; planes of flat 8x8 blocks with small steps
// between them plus noise, so that every branch of the decisions (no filter / weak / strong, second sample on either side) is taken somewhere -- written into the
// reference's TComDataCU / TComPicYuv and into the library's context alike.  Both filters then run on the same input (HOP_PIC_CHECK compares).  HOP_PIC_LF_DUMP=<file>:
// the input and the REFERENCE's output are written out: the fixtures of tests/golden/deblock_ref.npz (oracle/make_golden22.py) for the GPU test.
struct Fuzz {
  uint64_t s; int w, h, wctu; bool all_intra;   // all_intra: the carrier picture is an I slice (no reference picture: SS / GT CUs cannot occur in it)
  uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
  int below(int n) { return (int)(next() % (uint32_t)n); }
  static int z(int ux, int uy) { int v = 0; for (int b = 0; b < 4; b++) v |= (((ux >> b) & 1) << (2 * b)) | (((uy >> b) & 1) << (2 * b + 1)); return v; }
  void tu_tree(hop_cu_part* ctu, int ux, int uy, int su, int t, int min_t, int max_t) {
    if (t < min_t || (t < max_t && su > 1 && below(100) < 45)) { const int h = su >> 1; for (int q = 0; q < 4; q++) tu_tree(ctu, ux + (q & 1) * h, uy + (q >> 1) * h, h, t + 1, min_t, max_t); return; }
    const int coded = below(100) < 50;
    for (int y = 0; y < su; y++) for (int x = 0; x < su; x++) { hop_cu_part& p = ctu[z(ux + x, uy + y)]; p.tr_idx = (uint8_t)t; p.cbf[0] = (uint8_t)(coded ? ((1 << (t + 1)) - 1) : below(1 << t)); p.cbf[1] = p.cbf[2] = (uint8_t)below(2); }
  }
  void cu_tree(hop_cu_part* ctu, int cx, int cy, int ux, int uy, int su, int d) {
    const int x = cx + ux * 4, y = cy + uy * 4, size = su * 4;
    if (x >= w || y >= h) { for (int yy = 0; yy < su; yy++) for (int xx = 0; xx < su; xx++) { hop_cu_part& p = ctu[z(ux + xx, uy + yy)]; memset(&p, 0, sizeof(p)); p.depth = (uint8_t)d; p.pred_mode = 15; p.part_size = 15; p.ref_idx = -1; } return; }
    const bool crosses = x + size > w || y + size > h;
    if (crosses || (d < 3 && below(100) < 55)) { const int hh = su >> 1; for (int q = 0; q < 4; q++) cu_tree(ctu, cx, cy, ux + (q & 1) * hh, uy + (q >> 1) * hh, hh, d + 1); return; }
    const bool intra = all_intra ? true : below(100) < 35;
    int ps = 0;
    if (intra) ps = (d == 3 && below(2)) ? 3 : 0; else { ps = below(size >= 16 ? 8 : 3); if (ps == 3) ps = 0; }
    for (int yy = 0; yy < su; yy++) for (int xx = 0; xx < su; xx++) {
      hop_cu_part& p = ctu[z(ux + xx, uy + yy)]; memset(&p, 0, sizeof(p));
      p.depth = (uint8_t)d; p.pred_mode = intra ? 1 : 0; p.part_size = (uint8_t)ps; p.ref_idx = -1; p.luma_dir = 1; p.chroma_dir = 36; p.mvp_idx = p.mvp_num = -1;
    }
    if (!intra) {                                                          // one vector per PU
      const int npu = ps == 0 ? 1 : 2;
      int16_t mv[2][2]; int8_t ref[2];
      for (int k = 0; k < npu; k++) { mv[k][0] = (int16_t)(below(13) - 6); mv[k][1] = (int16_t)(below(13) - 6); ref[k] = below(100) < 6 ? -1 : 0; }
      for (int yy = 0; yy < su; yy++) for (int xx = 0; xx < su; xx++) {
        int k = 0;
        switch (ps) { case 1: k = yy >= su / 2; break; case 2: k = xx >= su / 2; break; case 4: k = yy >= su / 4; break; case 5: k = yy >= su - su / 4; break; case 6: k = xx >= su / 4; break; case 7: k = xx >= su - su / 4; break; default: break; }
        hop_cu_part& p = ctu[z(ux + xx, uy + yy)]; p.mv[0] = mv[k][0]; p.mv[1] = mv[k][1]; p.ref_idx = ref[k]; p.inter_dir = 1;
      }
    }
    int max_t = 0; while (max_t < 3 && (size >> (max_t + 1)) >= 4) max_t++;
    tu_tree(ctu, ux, uy, su, 0, ((intra && ps == 3) || size == 64) ? 1 : 0, max_t < 2 ? max_t : 2);
  }
};
void fuzz_picture(TComPic* pic, const char* spec) {
  long v[6] = { 1, 32, 0, 0, 0, 0 }; { const char* q = spec; for (int k = 0; k < 6 && q && *q; k++) { v[k] = strtol(q, (char**)&q, 10); if (*q == ':') q++; } }
  Fuzz f; f.s = 0x9E3779B97F4A7C15ull ^ (uint64_t)v[0] * 0x100000001B3ull; f.w = g_b.w; f.h = g_b.h; f.wctu = g_b.wctu; f.all_intra = !pic->getSlice(0)->isIntraSS();
  for (int k = 0; k < 8; k++) f.next();
  TComSlice* sl = pic->getSlice(0);
  sl->setSliceQp((Int)v[1]); sl->setDeblockingFilterBetaOffsetDiv2((Int)v[2]); sl->setDeblockingFilterTcOffsetDiv2((Int)v[3]); sl->getPPS()->setChromaCbQpOffset((Int)v[4]); sl->getPPS()->setChromaCrQpOffset((Int)v[5]);
  for (int a = 0; a < g_b.n; a++) f.cu_tree(&g_b.parts[(size_t)a * 256], (a % g_b.wctu) * 64, (a / g_b.wctu) * 64, 0, 0, 16, 0);
  for (int c = 0; c < 3; c++) {
    const int pw = c ? g_b.w / 2 : g_b.w, ph = c ? g_b.h / 2 : g_b.h, bs = c ? 4 : 8, bw = (pw + bs - 1) / bs;
    std::vector<int> base((size_t)bw * ((ph + bs - 1) / bs));
    for (size_t i = 0; i < base.size(); i++) { const int left = (i % bw) ? base[i - 1] : 60 + f.below(130); const int step = f.below(100) < 55 ? f.below(9) - 4 : f.below(41) - 20; base[i] = std::min(250, std::max(5, left + step)); }
    const int noise = 1 + f.below(3);
    const int up = g_bitDepthY - 8;                                          // a 10-bit picture: the same picture four times as large, with its own low bits
    for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) { const int v8 = std::min(255, std::max(0, base[(size_t)(y / bs) * bw + x / bs] + f.below(2 * noise + 1) - noise)); g_b.rec[c][(size_t)y * pw + x] = (int16_t)(up ? std::min((1 << g_bitDepthY) - 1, (v8 << up) + f.below(1 << up)) : v8); }
    if (hop_recon_upload(g_b.ctx, c, &g_b.rec[c][0]) != HOP_OK) g_b.fail("hop_recon_upload");
  }
  std::fill(g_b.levels.begin(), g_b.levels.end(), 0);
  for (int a = 0; a < g_b.n; a++) { TComDataCU* cu = pic->getCU(a); cu->initCU(pic, a); g_b.fill(cu); }
}
void dump_fixture(const char* path, const hop_deblock_params& p, const std::vector<int16_t> before[3], TComPicYuv* after) {
  FILE* f = fopen(path, "ab"); if (!f) return;
  const int32_t hd[10] = { g_b.w, g_b.h, p.qp, p.beta_offset_div2, p.tc_offset_div2, p.cb_qp_offset, p.cr_qp_offset, g_b.n, (int32_t)sizeof(hop_cu_part), g_bitDepthY };
  fwrite(hd, 4, 10, f); fwrite(&g_b.parts[0], sizeof(hop_cu_part), g_b.parts.size(), f);
  for (int c = 0; c < 3; c++) fwrite(&before[c][0], 2, before[c].size(), f);
  for (int y = 0; y < g_b.h; y++) fwrite(after->getLumaAddr() + (size_t)y * after->getStride(), 2, g_b.w, f);
  for (int y = 0; y < g_b.h / 2; y++) fwrite(after->getCbAddr() + (size_t)y * after->getCStride(), 2, g_b.w / 2, f);
  for (int y = 0; y < g_b.h / 2; y++) fwrite(after->getCrAddr() + (size_t)y * after->getCStride(), 2, g_b.w / 2, f);
  fclose(f);
}
}
Void TComLoopFilter::loopFilterPic(TComPic* pcPic)
{
  static const bool on = getenv("HOP_PIC_DEBLOCK") != NULL, check = getenv("HOP_PIC_CHECK") != NULL;
  if (!on || g_b.pic != pcPic || !g_b.ctx) { hop_ref_orig_loop_filter_pic(this, pcPic); return; }
  if (const char* fz = getenv("HOP_PIC_LF_FUZZ")) fuzz_picture(pcPic, fz);
  std::vector<int16_t> before[3]; if (getenv("HOP_PIC_LF_DUMP") || getenv("HOP_PIC_LF_FUZZ")) for (int k = 0; k < 3; k++) { before[k].resize(g_b.rec[k].size()); if (hop_recon_download(g_b.ctx, k, &before[k][0]) != HOP_OK) g_b.fail("hop_recon_download"); }
  TComSlice* sl = pcPic->getSlice(0);
  hop_deblock_params p; memset(&p, 0, sizeof(p));
  p.qp = sl->getSliceQp(); p.beta_offset_div2 = sl->getDeblockingFilterBetaOffsetDiv2(); p.tc_offset_div2 = sl->getDeblockingFilterTcOffsetDiv2();
  p.cb_qp_offset = sl->getPPS()->getChromaCbQpOffset(); p.cr_qp_offset = sl->getPPS()->getChromaCrQpOffset(); p.disable = sl->getDeblockingFilterDisable() ? 1 : 0;
  if (hop_deblock_frame(g_b.ctx, &p, &g_b.parts[0]) != HOP_OK) g_b.fail("hop_deblock_frame");
  for (int k = 0; k < 3; k++) if (hop_recon_download(g_b.ctx, k, &g_b.rec[k][0]) != HOP_OK) g_b.fail("hop_recon_download");
  TComPicYuv* r = pcPic->getPicYuvRec();
  const int w = g_b.w, h = g_b.h;
  if (check) {
    hop_ref_orig_loop_filter_pic(this, pcPic);
    if (const char* dp = getenv("HOP_PIC_LF_DUMP")) dump_fixture(dp, p, before, r);
    long bad = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) if (r->getLumaAddr()[(size_t)y * r->getStride() + x] != g_b.rec[0][(size_t)y * w + x] && bad++ < 10) fprintf(stderr, "hop pic check: deblocked luma (%d, %d) reference %d library %d\n", x, y, r->getLumaAddr()[(size_t)y * r->getStride() + x], g_b.rec[0][(size_t)y * w + x]);
    for (int y = 0; y < h / 2; y++) for (int x = 0; x < w / 2; x++) {
      if (r->getCbAddr()[(size_t)y * r->getCStride() + x] != g_b.rec[1][(size_t)y * (w / 2) + x] && bad++ < 10) fprintf(stderr, "hop pic check: deblocked Cb (%d, %d) differs\n", x, y);
      if (r->getCrAddr()[(size_t)y * r->getCStride() + x] != g_b.rec[2][(size_t)y * (w / 2) + x] && bad++ < 10) fprintf(stderr, "hop pic check: deblocked Cr (%d, %d) differs\n", x, y);
    }
    fprintf(stderr, "hop pic check: deblocked picture: %ld differences\n", bad);
    if (getenv("HOP_PIC_LF_FUZZ")) {                                       // the random picture is not an encodable one: stop here
      long changed = 0; if (!before[0].empty()) for (int k = 0; k < 3; k++) for (size_t i = 0; i < before[k].size(); i++) changed += before[k][i] != g_b.rec[k][i];
      fprintf(stderr, "hop pic check: fuzz: the filter changed %ld samples\n", changed); fflush(stderr); _exit(bad ? 1 : 0);
    }
  }
  for (int y = 0; y < h; y++) memcpy(r->getLumaAddr() + (size_t)y * r->getStride(), &g_b.rec[0][(size_t)y * w], w * sizeof(Pel));
  for (int y = 0; y < h / 2; y++) { memcpy(r->getCbAddr() + (size_t)y * r->getCStride(), &g_b.rec[1][(size_t)y * (w / 2)], (w / 2) * sizeof(Pel));
                                    memcpy(r->getCrAddr() + (size_t)y * r->getCStride(), &g_b.rec[2][(size_t)y * (w / 2)], (w / 2) * sizeof(Pel)); }
  g_b.deblocked++;
}

// HOP_PIC_SAO=1: TEncSampleAdaptiveOffset::SAOProcess (TLibEncoder/TEncSampleAdaptiveOffset.cpp:251-283) replaced as well: the deblocked picture goes to the context,
// hop_sao_frame gathers the statistics, decides every CTU's parameters (starting from the fraction the RD coder carries) and applies the offsets; the parameters go into
// the picture's SAOBlkParam array, from which the reference's encodeSlice writes them, the planes into the reconstruction.  With HOP_PIC_CHECK the reference's own
// SAOProcess runs on the same input afterwards and statistics, parameters and planes are compared.
extern "C" void hop_ref_orig_sao_process(TEncSampleAdaptiveOffset*, TComPic*, Bool*, const Double*, Bool);
namespace {
// HOP_PIC_SAO_FUZZ="seed:lambda_percent": the picture's original and deblocked planes are replaced by random ones made to spread the decisions -- per CTU one of: no
// error, band-dependent shifts, ringing along one of the four edge directions, noise; often the same kind as the CTU to the left (merges) -- and the lambdas scaled; both SAO
// encoders then run on the same input (HOP_PIC_CHECK compares).  HOP_PIC_SAO_DUMP=<file>: input and the REFERENCE's statistics, parameters and output planes are written
// out (tests/golden/sao_ref.npz, oracle/make_golden23.py).
void sao_fuzz(TComPic* pic, const char* spec, double scale_out[1]) {
  long v[2] = { 1, 100 }; { const char* q = spec; for (int k = 0; k < 2 && q && *q; k++) { v[k] = strtol(q, (char**)&q, 10); if (*q == ':') q++; } }
  scale_out[0] = (double)v[1] / 100.0;
  Fuzz f; f.s = 0xD1B54A32D192ED03ull ^ (uint64_t)v[0] * 0x100000001B3ull; f.w = g_b.w; f.h = g_b.h; f.wctu = g_b.wctu; f.all_intra = false; for (int k = 0; k < 8; k++) f.next();
  std::vector<int> kind(g_b.n);
  for (int a = 0; a < g_b.n; a++) kind[a] = (a % g_b.wctu && f.below(100) < 40) ? kind[a - 1] : f.below(7);
  std::vector<int16_t> org[3];
  for (int c = 0; c < 3; c++) {
    const int pw = c ? g_b.w / 2 : g_b.w, ph = c ? g_b.h / 2 : g_b.h, cs = c ? 32 : 64;
    org[c].resize((size_t)pw * ph);
    const int fx = 3 + f.below(9), fy = 3 + f.below(9), amp = 20 + f.below(60), mid = 60 + f.below(130);
    for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
      const int tri = abs(((x * 16 / fx) + (y * 16 / fy)) % 64 - 32) - 16;                 // a slanted triangle wave + texture
      org[c][(size_t)y * pw + x] = (int16_t)std::min(255, std::max(0, mid + tri * amp / 16 + f.below(7) - 3));                 // (8-bit values; scaled below)
    }
    for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
      const int k = kind[(y / cs) * g_b.wctu + x / cs], o = org[c][(size_t)y * pw + x];
      const int xl = std::max(0, x - 1), xr = std::min(pw - 1, x + 1), yu = std::max(0, y - 1), yd = std::min(ph - 1, y + 1);
      int r = o;
      switch (k) {
        case 1: r = o + ((o >> 3) % 5 == 1 ? 3 : (o >> 3) % 5 == 2 ? -2 : (o >> 3) % 5 == 3 ? 1 : 0); break;                       // band-dependent shifts
        case 2: case 3: case 4: case 5: {                                                                                             // ringing along one of the four edge directions
          const int na = k == 2 ? org[c][(size_t)y * pw + xl] : k == 3 ? org[c][(size_t)yu * pw + x] : k == 4 ? org[c][(size_t)yu * pw + xl] : org[c][(size_t)yu * pw + xr];
          const int nb = k == 2 ? org[c][(size_t)y * pw + xr] : k == 3 ? org[c][(size_t)yd * pw + x] : k == 4 ? org[c][(size_t)yd * pw + xr] : org[c][(size_t)yd * pw + xl];
          r = o + ((o > na && o > nb) ? 4 : (o < na && o < nb) ? -4 : (o > na || o > nb) ? 1 : 0);
        } break;
        case 6: r = o + f.below(5) - 2; break;
        default: break;
      }
      g_b.rec[c][(size_t)y * pw + x] = (int16_t)std::min(255, std::max(0, r));
    }
    if (const int up = g_bitDepthY - 8) for (size_t i = 0; i < org[c].size(); i++) {   // a 10-bit picture: both planes four times as large, the same low bits in both plus a little noise in the reconstruction
      const int lo = f.below(1 << up); org[c][i] = (int16_t)((org[c][i] << up) + lo); g_b.rec[c][i] = (int16_t)std::min((1 << g_bitDepthY) - 1, std::max(0, (g_b.rec[c][i] << up) + lo + f.below(3) - 1));
    }
  }
  TComPicYuv* po = pic->getPicYuvOrg(); TComPicYuv* pr = pic->getPicYuvRec();
  const int w = g_b.w, h = g_b.h;
  for (int y = 0; y < h; y++) { memcpy(po->getLumaAddr() + (size_t)y * po->getStride(), &org[0][(size_t)y * w], w * 2); memcpy(pr->getLumaAddr() + (size_t)y * pr->getStride(), &g_b.rec[0][(size_t)y * w], w * 2); }
  for (int y = 0; y < h / 2; y++) {
    memcpy(po->getCbAddr() + (size_t)y * po->getCStride(), &org[1][(size_t)y * (w / 2)], w); memcpy(po->getCrAddr() + (size_t)y * po->getCStride(), &org[2][(size_t)y * (w / 2)], w);
    memcpy(pr->getCbAddr() + (size_t)y * pr->getCStride(), &g_b.rec[1][(size_t)y * (w / 2)], w); memcpy(pr->getCrAddr() + (size_t)y * pr->getCStride(), &g_b.rec[2][(size_t)y * (w / 2)], w);
  }
  if (hop_upload_orig(g_b.ctx, &org[0][0], w, &org[1][0], &org[2][0], w / 2) != HOP_OK) g_b.fail("hop_upload_orig");
}
}
Void TEncSampleAdaptiveOffset::SAOProcess(TComPic* pPic, Bool* sliceEnabled, const Double* lambdas, Bool isPreDBFSamplesUsed)
{
  static const bool on = getenv("HOP_PIC_SAO") != NULL, check = getenv("HOP_PIC_CHECK") != NULL;
  if (!on || g_b.pic != pPic || !g_b.ctx || isPreDBFSamplesUsed) { hop_ref_orig_sao_process(this, pPic, sliceEnabled, lambdas, isPreDBFSamplesUsed); return; }
  TComPicYuv* r = pPic->getPicYuvRec(); TComSlice* sl = pPic->getSlice(0);
  const int w = g_b.w, h = g_b.h, n = g_b.n;
  double lam[3] = { lambdas[0], lambdas[1], lambdas[2] };
  if (const char* fz = getenv("HOP_PIC_SAO_FUZZ")) { double sc[1]; sao_fuzz(pPic, fz, sc); for (int k = 0; k < 3; k++) lam[k] *= sc[0]; lambdas = lam; }
  for (int y = 0; y < h; y++) memcpy(&g_b.rec[0][(size_t)y * w], r->getLumaAddr() + (size_t)y * r->getStride(), w * sizeof(Pel));
  for (int y = 0; y < h / 2; y++) { memcpy(&g_b.rec[1][(size_t)y * (w / 2)], r->getCbAddr() + (size_t)y * r->getCStride(), (w / 2) * sizeof(Pel));
                                    memcpy(&g_b.rec[2][(size_t)y * (w / 2)], r->getCrAddr() + (size_t)y * r->getCStride(), (w / 2) * sizeof(Pel)); }
  for (int k = 0; k < 3; k++) if (hop_recon_upload(g_b.ctx, k, &g_b.rec[k][0]) != HOP_OK) g_b.fail("hop_recon_upload");      // (a no-op's worth when hop_deblock_frame left it there)
  decidePicParams(sliceEnabled, sl->getDepth());
  hop_sao_params p; memset(&p, 0, sizeof(p));
  for (int k = 0; k < 3; k++) { p.lambda[k] = lambdas[k]; p.enabled[k] = sliceEnabled[k] ? 1 : 0; }
  p.slice_type = (int)sl->getSliceType(); p.qp = sl->getSliceQp();
  p.rd_fraction = (uint32_t)(((TEncBinCABAC*)m_pcRDGoOnSbacCoder->m_pcBinIf)->m_fracBits & 32767);
  std::vector<hop_sao_param> coded((size_t)n * 3);
  std::vector<int32_t> stats;
  if (check) { stats.resize((size_t)n * 3 * 5 * 32 * 2); if (hop_sao_stats(g_b.ctx, &stats[0]) != HOP_OK) g_b.fail("hop_sao_stats"); }
  std::vector<int16_t> before[3]; for (int k = 0; k < 3; k++) before[k] = g_b.rec[k];
  if (hop_sao_frame(g_b.ctx, &p, &coded[0]) != HOP_OK) g_b.fail("hop_sao_frame");
  for (int k = 0; k < 3; k++) if (hop_recon_download(g_b.ctx, k, &g_b.rec[k][0]) != HOP_OK) g_b.fail("hop_recon_download");
  SAOBlkParam* dst = pPic->getPicSym()->getSAOBlkParam();
  if (check) {
    Bool en[3]; hop_ref_orig_sao_process(this, pPic, en, lambdas, isPreDBFSamplesUsed);
    long bad = 0;
    for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) {
      for (int t = 0; t < 5; t++) for (int k = 0; k < 32; k++) {
        const int32_t* s = &stats[((((size_t)a * 3 + c) * 5 + t) * 32 + k) * 2];
        if ((s[0] != m_statData[a][c][t].count[k] || s[1] != m_statData[a][c][t].diff[k]) && bad++ < 10) fprintf(stderr, "hop pic check: SAO statistics CTU %d comp %d type %d class %d: reference %ld / %ld library %d / %d\n", a, c, t, k, (long)m_statData[a][c][t].count[k], (long)m_statData[a][c][t].diff[k], s[0], s[1]);
      }
      const SAOOffset& o = dst[a][c]; const hop_sao_param& q = coded[(size_t)a * 3 + c];
      bool same = o.modeIdc == q.mode && (o.modeIdc == SAO_MODE_OFF || (o.typeIdc == q.type && (o.modeIdc == SAO_MODE_MERGE || o.typeAuxInfo == q.aux)));
      if (same && o.modeIdc == SAO_MODE_NEW) for (int k = 0; k < 32; k++) same = same && o.offset[k] == q.offset[k];
      if (!same && bad++ < 20) fprintf(stderr, "hop pic check: SAO parameters CTU %d comp %d: reference mode %d type %d aux %d, library mode %d type %d aux %d\n", a, c, o.modeIdc, o.typeIdc, o.typeAuxInfo, q.mode, q.type, q.aux);
    }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) if (r->getLumaAddr()[(size_t)y * r->getStride() + x] != g_b.rec[0][(size_t)y * w + x] && bad++ < 30) fprintf(stderr, "hop pic check: SAO luma (%d, %d) reference %d library %d\n", x, y, r->getLumaAddr()[(size_t)y * r->getStride() + x], g_b.rec[0][(size_t)y * w + x]);
    for (int y = 0; y < h / 2; y++) for (int x = 0; x < w / 2; x++) {
      if (r->getCbAddr()[(size_t)y * r->getCStride() + x] != g_b.rec[1][(size_t)y * (w / 2) + x] && bad++ < 30) fprintf(stderr, "hop pic check: SAO Cb (%d, %d) differs\n", x, y);
      if (r->getCrAddr()[(size_t)y * r->getCStride() + x] != g_b.rec[2][(size_t)y * (w / 2) + x] && bad++ < 30) fprintf(stderr, "hop pic check: SAO Cr (%d, %d) differs\n", x, y);
    }
    fprintf(stderr, "hop pic check: SAO: %ld differences\n", bad);
    if (getenv("HOP_PIC_SAO_FUZZ")) {
      long modes[3] = { 0, 0, 0 }, types[5] = { 0, 0, 0, 0, 0 };
      for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) { modes[dst[a][c].modeIdc]++; if (dst[a][c].modeIdc == SAO_MODE_NEW) types[dst[a][c].typeIdc]++; }
      fprintf(stderr, "hop pic check: fuzz: off %ld new %ld merge %ld; edge 0/90/135/45 %ld %ld %ld %ld band %ld\n", modes[0], modes[1], modes[2], types[0], types[1], types[2], types[3], types[4]);
      if (const char* dp = getenv("HOP_PIC_SAO_DUMP")) if (FILE* fo = fopen(dp, "ab")) {
        const int32_t hd[7] = { w, h, n, p.slice_type, p.qp, (int32_t)p.rd_fraction, g_bitDepthY };
        fwrite(hd, 4, 7, fo); fwrite(p.lambda, 8, 3, fo);
        TComPicYuv* po = pPic->getPicYuvOrg();
        for (int y = 0; y < h; y++) fwrite(po->getLumaAddr() + (size_t)y * po->getStride(), 2, w, fo);
        for (int y = 0; y < h / 2; y++) fwrite(po->getCbAddr() + (size_t)y * po->getCStride(), 2, w / 2, fo);
        for (int y = 0; y < h / 2; y++) fwrite(po->getCrAddr() + (size_t)y * po->getCStride(), 2, w / 2, fo);
        for (int k = 0; k < 3; k++) fwrite(&before[k][0], 2, before[k].size(), fo);
        for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) for (int t = 0; t < 5; t++) for (int k = 0; k < 32; k++) { const int32_t s2[2] = { (int32_t)m_statData[a][c][t].count[k], (int32_t)m_statData[a][c][t].diff[k] }; fwrite(s2, 4, 2, fo); }
        for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) { hop_sao_param q; memset(&q, 0, sizeof(q)); q.mode = (int8_t)dst[a][c].modeIdc; q.type = (int8_t)dst[a][c].typeIdc; q.aux = (int8_t)dst[a][c].typeAuxInfo; for (int k = 0; k < 32; k++) q.offset[k] = (int8_t)dst[a][c].offset[k]; fwrite(&q, sizeof(q), 1, fo); }
        for (int y = 0; y < h; y++) fwrite(r->getLumaAddr() + (size_t)y * r->getStride(), 2, w, fo);
        for (int y = 0; y < h / 2; y++) fwrite(r->getCbAddr() + (size_t)y * r->getCStride(), 2, w / 2, fo);
        for (int y = 0; y < h / 2; y++) fwrite(r->getCrAddr() + (size_t)y * r->getCStride(), 2, w / 2, fo);
        fclose(fo);
      }
      fflush(stderr); _exit(bad ? 1 : 0);
    }
  }
  for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) {
    SAOOffset& o = dst[a][c]; const hop_sao_param& q = coded[(size_t)a * 3 + c];
    o.modeIdc = q.mode; o.typeIdc = q.type; o.typeAuxInfo = q.aux; for (int k = 0; k < MAX_NUM_SAO_CLASSES; k++) o.offset[k] = q.offset[k];
  }
  for (int y = 0; y < h; y++) memcpy(r->getLumaAddr() + (size_t)y * r->getStride(), &g_b.rec[0][(size_t)y * w], w * sizeof(Pel));
  for (int y = 0; y < h / 2; y++) { memcpy(r->getCbAddr() + (size_t)y * r->getCStride(), &g_b.rec[1][(size_t)y * (w / 2)], (w / 2) * sizeof(Pel));
                                    memcpy(r->getCrAddr() + (size_t)y * r->getCStride(), &g_b.rec[2][(size_t)y * (w / 2)], (w / 2) * sizeof(Pel)); }
  { double ps[3]; if (hop_psnr(g_b.ctx, NULL, ps) != HOP_OK) g_b.fail("hop_psnr"); fprintf(stderr, "hop pic psnr: %.4f %.4f %.4f\n", ps[0], ps[1], ps[2]); }   // TEncGOP::xCalculateAddPSNR prints the same three numbers
  g_b.sao++;
}
