// hop_ctx.hip -- context, picture residency, host-side logic and the C ABI glue of libhophip.so.
// gfx950 only.  No CPU compute path exists in this library: every hot-path entry point launches a
// HIP kernel and fails with HOP_ERR_DEVICE when there is no device.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <utility>
#include <algorithm>
#include "hop_dev.h"

static char g_create_err[512] = "";

int hop_set_err(hop_ctx* c, int code, const char* fmt, ...) {
  char* dst = c ? c->err : g_create_err;
  va_list ap; va_start(ap, fmt); vsnprintf(dst, 512, fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return hop_set_err((c), HOP_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

int hop_scratch(hop_ctx* c, size_t bytes, void** out) {
  if (bytes > c->scratch_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->scratch) HIPCHK(c, hipFree(c->scratch));
    c->scratch = nullptr; c->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(c, hipSetDevice(c->device));                                  // (the calling thread's current device may be another context's)
    HIPCHK(c, hipMalloc(&c->scratch, want));
    c->scratch_bytes = want;
  }
  *out = c->scratch;
  return HOP_OK;
}
static int hop_stage(hop_ctx* c, size_t bytes, void** out) {
  if (bytes > c->stage_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stage) HIPCHK(c, hipFree(c->stage));
    c->stage = nullptr; c->stage_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->stage, want));
    c->stage_bytes = want;
  }
  *out = c->stage;
  return HOP_OK;
}

extern "C" {

const char* hop_version(void) { return "hophip 0.1 gfx950"; }
const char* hop_last_error(const hop_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int hop_ctx_create(hop_ctx** out, int pic_w, int pic_h, int bit_depth_y, int bit_depth_c, int device) {
  if (!out) return hop_set_err(nullptr, HOP_ERR_ARG, "out is NULL");
  *out = nullptr;
  // picture sizes: the reference codes multiples of the minimum CU (8) only (TAppEncCfg padding)
  if (pic_w <= 0 || pic_h <= 0 || (pic_w & 7) || (pic_h & 7)) return hop_set_err(nullptr, HOP_ERR_ARG, "picture %dx%d must be a positive multiple of 8", pic_w, pic_h);
  if (bit_depth_y < 8 || bit_depth_y > 12 || bit_depth_c < 8 || bit_depth_c > 12) return hop_set_err(nullptr, HOP_ERR_ARG, "bit depth out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return hop_set_err(nullptr, HOP_ERR_DEVICE, "no HIP device: libhophip has no CPU path");
  if (device < 0 || device >= ndev) return hop_set_err(nullptr, HOP_ERR_ARG, "device %d of %d", device, ndev);
  hop_ctx* c = (hop_ctx*)calloc(1, sizeof(hop_ctx));
  c->pic_w = pic_w; c->pic_h = pic_h; c->bd_y = bit_depth_y; c->bd_c = bit_depth_c; c->device = device;
  c->stride_y = pic_w + 2 * HOP_MARGIN_Y; c->stride_c = (pic_w >> 1) + 2 * HOP_MARGIN_C;
  { const char* f = getenv("HOP_FUSED_LEAF"); c->fused_leaf_max = f ? atoi(f) : 8192; }
  { const char* f = getenv("HOP_WALK"); c->walk_max = f ? atoi(f) : 4096; }               // developer switch: the results do not depend on it
  { const char* f = getenv("HOP_SS_FAMILIES"); c->ss_families = !(f && f[0] == '0'); }   // developer switch: the results do not depend on it
  if (hipSetDevice(device) != hipSuccess) { free(c); return hop_set_err(nullptr, HOP_ERR_DEVICE, "hipSetDevice(%d) failed", device); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    hop_set_err(nullptr, HOP_ERR_DEVICE, "device %d is '%s', libhophip is built for gfx950 only", device, prop.gcnArchName);
    free(c); return HOP_ERR_DEVICE;
  }
  size_t ny = (size_t)pic_w * pic_h, nc = ny >> 2;
  size_t sy = (size_t)c->stride_y * (pic_h + 2 * HOP_MARGIN_Y), sc = (size_t)c->stride_c * ((pic_h >> 1) + 2 * HOP_MARGIN_C);
  // The context's own stream -- the spine's short requests: predictions, searches, block copies -- runs at the highest priority, the views' streams (the candidate
  // evaluations, hundreds of long workgroups in flight) at the lowest: without it a 10 us prediction kernel waits for a free compute unit behind them (measured on the
  // 7728-wide picture: 0.15 ms -> 0.03 ms per prediction request, 27 -> 35 CTU/s).  HOP_STREAM_PRIO=0 turns it off.
  int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const bool prio = !(getenv("HOP_STREAM_PRIO") && getenv("HOP_STREAM_PRIO")[0] == '0');
  hipError_t e = prio ? hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_hi) : hipStreamCreate(&c->stream);
  for (int k = 0; k < HOP_MAX_LANES - 1 && e == hipSuccess; k++) { e = hipStreamCreate(&c->xstream[k]); if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming); }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  { const char* f = getenv("HOP_LANES"); c->lanes = f ? atoi(f) : 2; if (c->lanes < 1) c->lanes = 1; if (c->lanes > HOP_MAX_LANES) c->lanes = HOP_MAX_LANES; }
  if (e == hipSuccess) e = hipMalloc((void**)&c->org_y, ny * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->org_cb, nc * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->org_cr, nc * 2);
  const size_t gy = (size_t)HOP_GUARD_ROWS * c->stride_y, gc = (size_t)HOP_GUARD_ROWS * c->stride_c;
  if (e == hipSuccess) e = hipMalloc((void**)&c->ss_alloc[0], (sy + 2 * gy) * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->ss_alloc[1], (sc + 2 * gc) * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->ss_alloc[2], (sc + 2 * gc) * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->pred[0], ny * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->pred[1], nc * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->pred[2], nc * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->rec[0], ny * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->rec[1], nc * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->rec[2], nc * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&c->entropy_bits, 128 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(c->entropy_bits, hop_entropy_bits_host(), 128 * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&c->rdoq_scans, HOP_RDOQ_SCAN_ENTRIES * sizeof(uint16_t));
  if (e == hipSuccess) {
    std::vector<uint16_t> tabs(HOP_RDOQ_SCAN_ENTRIES);
    hop_rdoq_build_scans(tabs.data());
    e = hipMemcpy(c->rdoq_scans, tabs.data(), tabs.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) { hop_set_err(nullptr, HOP_ERR_DEVICE, "allocation failed: %s", hipGetErrorString(e)); hop_ctx_destroy(c); return HOP_ERR_DEVICE; }
  c->ss_buf[0] = c->ss_alloc[0] + gy; c->ss_buf[1] = c->ss_alloc[1] + gc; c->ss_buf[2] = c->ss_alloc[2] + gc;
  c->ss00[0] = c->ss_buf[0] + (size_t)HOP_MARGIN_Y * c->stride_y + HOP_MARGIN_Y;
  c->ss00[1] = c->ss_buf[1] + (size_t)HOP_MARGIN_C * c->stride_c + HOP_MARGIN_C;
  c->ss00[2] = c->ss_buf[2] + (size_t)HOP_MARGIN_C * c->stride_c + HOP_MARGIN_C;
  *out = c;
  int r = hop_ssref_reset(c);
  if (r == HOP_OK) r = hop_sync(c);
  if (r != HOP_OK) { strncpy(g_create_err, c->err, 511); hop_ctx_destroy(c); *out = nullptr; }
  return r;
}

// A second handle on the same pictures with its own stream and work areas: requests issued through different views run concurrently on the device.  The caller keeps
// their rectangles apart (the RD spine's CTU rows are a wavefront lag apart) and destroys the views before the parent.
int hop_ctx_create_view(hop_ctx* parent, hop_ctx** out) {
  if (!parent || !out) return hop_set_err(parent, HOP_ERR_ARG, "hop_ctx_create_view: bad argument");          // (a view of a view is one more view of the same pictures)
  *out = nullptr;
  hop_ctx* c = (hop_ctx*)calloc(1, sizeof(hop_ctx));
  c->pic_w = parent->pic_w; c->pic_h = parent->pic_h; c->bd_y = parent->bd_y; c->bd_c = parent->bd_c; c->device = parent->device;
  c->stride_y = parent->stride_y; c->stride_c = parent->stride_c; c->sub_h = parent->sub_h; c->sub_pitch = parent->sub_pitch; c->slots = parent->slots; c->fused_leaf_max = parent->fused_leaf_max; c->walk_max = parent->walk_max; c->ss_families = parent->ss_families; c->lanes = 1; c->is_view = true;
  c->org_y = parent->org_y; c->org_cb = parent->org_cb; c->org_cr = parent->org_cr;
  for (int k = 0; k < 3; k++) { c->ss_alloc[k] = parent->ss_alloc[k]; c->ss_buf[k] = parent->ss_buf[k]; c->ss00[k] = parent->ss00[k]; c->pred[k] = parent->pred[k]; c->rec[k] = parent->rec[k]; }
  c->entropy_bits = parent->entropy_bits; c->rdoq_scans = parent->rdoq_scans; c->have_orig = parent->have_orig; c->stash = parent->stash; c->stash_slots = parent->stash_slots; c->coefpic = parent->coefpic; c->coef_stash = parent->coef_stash;
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) {
    int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const bool prio = !(getenv("HOP_STREAM_PRIO") && getenv("HOP_STREAM_PRIO")[0] == '0');
    e = prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  }
  for (int k = 0; k < HOP_MAX_LANES - 1 && e == hipSuccess; k++) { e = hipStreamCreateWithFlags(&c->xstream[k], hipStreamNonBlocking); if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming); }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e != hipSuccess) { hop_set_err(parent, HOP_ERR_DEVICE, "hop_ctx_create_view: %s", hipGetErrorString(e)); hop_ctx_destroy(c); return HOP_ERR_DEVICE; }
  *out = c;
  return HOP_OK;
}

// Several independent pictures of equal size in one context, so that one batch of requests serves CTUs of all of them (hop_encode_frame codes them side by side): the
// context's picture is their stack, picture k at rows k * sub_pitch .. k * sub_pitch + sub_h - 1, the rows between them unused.  The gap keeps every sample a picture's
// searches, margins and prefetches can touch (its margin + guard rows above and below) away from its neighbours', so each picture is coded exactly as in a context of its own.
// Candidate slots: the original, prediction and reconstruction pictures are laid out slots + 1 times one below the other (copy k at rows k * pic_h; the original is the
// same in all of them).  A request whose y coordinate is y + k * pic_h then works on copy k: candidates of one CU evaluated side by side predict, transform and
// reconstruct in copies of their own while searching the one SS reference (the predictor kernel takes the copy as hop_pred_job.dst_row_off).  Call before hop_upload_orig.
int hop_ctx_set_slots(hop_ctx* c, int slots) {
  if (!c || c->is_view || slots < 0 || slots > 128) return hop_set_err(c, HOP_ERR_ARG, "hop_ctx_set_slots: bad argument");
  if (slots == c->slots) return HOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t ny = (size_t)c->pic_w * c->pic_h * (slots + 1), nc = ny >> 2;
  int16_t** planes[9] = { &c->org_y, &c->org_cb, &c->org_cr, &c->pred[0], &c->pred[1], &c->pred[2], &c->rec[0], &c->rec[1], &c->rec[2] };
  for (int k = 0; k < 9; k++) {
    if (*planes[k]) (void)hipFree(*planes[k]);
    *planes[k] = nullptr;
    hipError_t e = hipMalloc((void**)planes[k], ((k % 3) ? nc : ny) * 2);
    if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_ctx_set_slots: %s", hipGetErrorString(e));
  }
  c->slots = slots; c->have_orig = false;
  return HOP_OK;
}

int hop_set_fused_leaf(hop_ctx* c, int max_tus) { if (!c || max_tus < 0) return hop_set_err(c, HOP_ERR_ARG, "hop_set_fused_leaf: bad argument"); c->fused_leaf_max = max_tus; return HOP_OK; }

int hop_stack_pitch(int sub_h) { return ((sub_h + 63) / 64) * 64 + 64 * ((2 * (HOP_MARGIN_Y + HOP_GUARD_ROWS) + 63) / 64); }
int hop_ctx_set_stack(hop_ctx* c, int sub_h, int sub_pitch) {
  if (!c || c->is_view) return hop_set_err(c, HOP_ERR_ARG, "hop_ctx_set_stack: bad argument");
  if (sub_h == 0 && sub_pitch == 0) { c->sub_h = c->sub_pitch = 0; return HOP_OK; }
  if (sub_h <= 0 || (sub_h & 7) || (sub_pitch & 63) || sub_pitch < hop_stack_pitch(sub_h) || c->pic_h < sub_h || (c->pic_h - sub_h) % sub_pitch)
    return hop_set_err(c, HOP_ERR_ARG, "hop_ctx_set_stack: a %d-row context is not a stack of %d-row pictures %d rows apart (pitch: a multiple of 64, at least hop_stack_pitch)", c->pic_h, sub_h, sub_pitch);
  c->sub_h = sub_h; c->sub_pitch = sub_pitch;
  return HOP_OK;
}

void hop_ctx_destroy(hop_ctx* c) {
  if (!c) return;
  if (c->stream) { (void)hipStreamSynchronize(c->stream); }
  if (c->is_view) {                                                     // pictures, tables and stash are the parent's
    c->org_y = c->org_cb = c->org_cr = nullptr; c->entropy_bits = nullptr; c->rdoq_scans = nullptr; c->stash = nullptr; c->coefpic = nullptr; c->coef_stash = nullptr;
    for (int k = 0; k < 3; k++) { c->ss_alloc[k] = nullptr; c->pred[k] = nullptr; c->rec[k] = nullptr; }
  }
  for (int i = 0; i < c->prof_cap; i++) { if (c->prof_recs[i].a) (void)hipEventDestroy(c->prof_recs[i].a); if (c->prof_recs[i].b) (void)hipEventDestroy(c->prof_recs[i].b); }
  free(c->prof_recs); free(c->rd_fraction);
  void* ptrs[] = { c->org_y, c->org_cb, c->org_cr, c->ss_alloc[0], c->ss_alloc[1], c->ss_alloc[2], c->pred[0], c->pred[1], c->pred[2],
                   c->rec[0], c->rec[1], c->rec[2], c->scratch, c->stage, c->rqt_buf, c->walk_buf, c->rdoq_scans, c->entropy_bits, c->stash, c->coefpic, c->coef_stash };
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (int k = 0; k < HOP_MAX_LANES - 1; k++) {
    if (c->xstream[k]) (void)hipStreamDestroy(c->xstream[k]);
    if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
    if (c->xscratch[k]) (void)hipFree(c->xscratch[k]);
    if (c->xrqt_buf[k]) (void)hipFree(c->xrqt_buf[k]);
    if (c->xwalk_buf[k]) (void)hipFree(c->xwalk_buf[k]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  free(c);
}

int hop_sync(hop_ctx* c) { if (!c) return HOP_ERR_ARG; HIPCHK(c, hipStreamSynchronize(c->stream)); return HOP_OK; }
void* hop_stream(hop_ctx* c) { return c ? (void*)c->stream : nullptr; }

int hop_upload_orig(hop_ctx* c, const int16_t* y, int stride_y, const int16_t* cb, const int16_t* cr, int stride_c) {
  if (!c || !y || !cb || !cr || stride_y < c->pic_w || stride_c < (c->pic_w >> 1)) return hop_set_err(c, HOP_ERR_ARG, "hop_upload_orig: bad argument");
  HIPCHK(c, hipMemcpy2DAsync(c->org_y, (size_t)c->pic_w * 2, y, (size_t)stride_y * 2, (size_t)c->pic_w * 2, c->pic_h, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->org_cb, (size_t)c->pic_w, cb, (size_t)stride_c * 2, (size_t)c->pic_w, c->pic_h >> 1, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->org_cr, (size_t)c->pic_w, cr, (size_t)stride_c * 2, (size_t)c->pic_w, c->pic_h >> 1, hipMemcpyHostToDevice, c->stream));
  for (int k = 1; k <= c->slots; k++) {                                  // the copies of the candidate slots
    const size_t ny = (size_t)c->pic_w * c->pic_h, nc = ny >> 2;
    HIPCHK(c, hipMemcpyAsync(c->org_y + k * ny, c->org_y, ny * 2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->org_cb + k * nc, c->org_cb, nc * 2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->org_cr + k * nc, c->org_cr, nc * 2, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_orig = true;
  return HOP_OK;
}

int hop_ssref_reset(hop_ctx* c) { if (!c) return HOP_ERR_ARG; return hop_launch_ssref_reset(c); }

static size_t plane_elems(const hop_ctx* c, int comp) {
  return comp == 0 ? (size_t)c->stride_y * (c->pic_h + 2 * HOP_MARGIN_Y) : (size_t)c->stride_c * ((c->pic_h >> 1) + 2 * HOP_MARGIN_C);
}
int hop_ssref_download(hop_ctx* c, int comp, int16_t* dst) {
  if (!c || !dst || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_download: bad argument");
  HIPCHK(c, hipMemcpyAsync(dst, c->ss_buf[comp], plane_elems(c, comp) * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_ssref_upload(hop_ctx* c, int comp, const int16_t* src) {
  if (!c || !src || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_upload: bad argument");
  HIPCHK(c, hipMemcpyAsync(c->ss_buf[comp], src, plane_elems(c, comp) * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_pred_download(hop_ctx* c, int comp, int16_t* dst) {
  if (!c || !dst || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_pred_download: bad argument");
  size_t n = comp == 0 ? (size_t)c->pic_w * c->pic_h : ((size_t)c->pic_w * c->pic_h) >> 2;
  HIPCHK(c, hipMemcpyAsync(dst, c->pred[comp], n * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

static int check_rects(hop_ctx* c, int n, const int32_t* r) {
  for (int i = 0; i < n; i++) {
    int x = r[4 * i], y = r[4 * i + 1], s = r[4 * i + 2];
    if (!(s == 8 || s == 16 || s == 32 || s == 64) || x < 0 || y < 0 || (x % s) || (y % s) || x + s > c->pic_w || y + s > c->pic_h)
      return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_commit_cus: CU %d (%d,%d,%d) is not a legal CU of a %dx%d picture", i, x, y, s, c->pic_w, c->pic_h);
  }
  return HOP_OK;
}

int hop_ssref_commit_cus(hop_ctx* c, int n, const int32_t* rect4, const int16_t* rec_y, const int16_t* rec_cb, const int16_t* rec_cr) {
  if (!c || n < 0 || (n && (!rect4 || !rec_y || !rec_cb || !rec_cr))) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_commit_cus: bad argument");
  if (n == 0) return HOP_OK;
  int r = check_rects(c, n, rect4); if (r) return r;
  // packed layout: CU i's luma block starts at sum_{k<i} size_k^2; rect4[4*i+3] receives that offset
  std::vector<int32_t> rr(rect4, rect4 + 4 * (size_t)n);
  size_t tot = 0;
  for (int i = 0; i < n; i++) { rr[4 * i + 3] = (int32_t)tot; tot += (size_t)rr[4 * i + 2] * rr[4 * i + 2]; }
  size_t bytes_r = (size_t)n * 16, bytes_y = tot * 2, bytes_c = (tot >> 2) * 2;
  size_t o_y = (bytes_r + 255) & ~(size_t)255, o_cb = (o_y + bytes_y + 255) & ~(size_t)255, o_cr = (o_cb + bytes_c + 255) & ~(size_t)255;
  void* st; r = hop_stage(c, o_cr + bytes_c, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, rr.data(), bytes_r, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_y, rec_y, bytes_y, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_cb, rec_cb, bytes_c, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_cr, rec_cr, bytes_c, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_ssref_commit(c, n, (const int32_t*)b, (const int16_t*)(b + o_y), (const int16_t*)(b + o_cb), (const int16_t*)(b + o_cr), 1);
  if (r) return r;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_ssref_commit_cus_device(hop_ctx* c, int n, const int32_t* d_rect4, const int16_t* d_rec_y, const int16_t* d_rec_cb, const int16_t* d_rec_cr) {
  if (!c || n < 0 || (n && (!d_rect4 || !d_rec_y || !d_rec_cb || !d_rec_cr))) return hop_set_err(c, HOP_ERR_ARG, "hop_ssref_commit_cus_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_ssref_commit(c, n, d_rect4, d_rec_y, d_rec_cb, d_rec_cr, 0);
}

// host logic (hop_set_search_range, hop_component_bits, hop_bits_gt, hop_me_finish, the CABAC bin helpers): host/hop_hostlogic.cpp

// ---------------------------------------------------------------------------------------------
// hot path entry points
// ---------------------------------------------------------------------------------------------
static const int kLegalDim[] = { 4, 8, 12, 16, 24, 32, 48, 64 };
static bool legal_dim(int v) { for (int d : kLegalDim) if (v == d) return true; return false; }

}  // extern "C"
int hop_check_pu_jobs(hop_ctx* c, int n, const hop_pu_job* jobs) {
  for (int i = 0; i < n; i++) {
    const hop_pu_job& j = jobs[i];
    if (!legal_dim(j.w) || !legal_dim(j.h) || j.pu_x < 0 || j.pu_y < 0 || (j.pu_x & 3) || (j.pu_y & 3) || j.pu_x + j.w > c->pic_w || j.pu_y + j.h > c->pic_h)
      return hop_set_err(c, HOP_ERR_ARG, "PU job %d: rectangle (%d,%d,%dx%d) is not a legal PU of a %dx%d picture", i, j.pu_x, j.pu_y, j.w, j.h, c->pic_w, c->pic_h);
    if (j.n_amvp < 0 || j.n_amvp > 2) return hop_set_err(c, HOP_ERR_ARG, "PU job %d: n_amvp %d", i, j.n_amvp);
    // Reach: the reference addresses the padded plane linearly, so a column overshoot wraps into the neighbouring
    // row exactly as it does in the reference; what must hold is that every access stays inside the allocation:
    // rows (window, block, +4 probes, GT patch H/2 + 8-tap reach) within the 80-row margin, columns within one pitch.
    if (j.rng_right >= j.rng_left && j.rng_bottom >= j.rng_top) {
      if (j.rng_right - j.rng_left > 256 || j.rng_bottom - j.rng_top > 256)
        return hop_set_err(c, HOP_ERR_ARG, "PU job %d: search window larger than 257x257 (SearchRange > 128)", i);
      const int lim = HOP_MARGIN_Y + HOP_GUARD_ROWS - 4;
      bool bad = j.pu_y + j.rng_top - j.h / 2 - 4 < -lim || j.pu_y + j.rng_bottom + j.h + j.h / 2 + 8 > c->pic_h + lim ||
                 j.pu_x + j.rng_left - j.w / 2 - 4 < -(c->stride_y - 8) || j.pu_x + j.rng_right + 2 * j.w + 8 > c->stride_y + c->pic_w - 8;
      for (int k = 0; k < j.n_amvp; k++) {   // AMVP start vectors of the GT search (TEncSearch.cpp:5144-5150)
        const int sx = (int)(int16_t)j.amvp[2 * k] >> 2, sy = (int)(int16_t)j.amvp[2 * k + 1] >> 2;
        bad = bad || j.pu_y + sy - j.h / 2 < -lim || j.pu_y + sy + j.h + j.h / 2 > c->pic_h + lim ||
              j.pu_x + sx - j.w / 2 - 4 < -(c->stride_y - 8) || j.pu_x + sx + 2 * j.w + 8 > c->stride_y + c->pic_w - 8;   // columns: within one pitch, as for the window (a neighbour's unclipped vector may point past the margin)
      }
      if (bad)
        return hop_set_err(c, HOP_ERR_ARG, "PU job %d: search range leaves the padded reference", i);
    }
  }
  return HOP_OK;
}
extern "C" {

static int me_pipeline(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_results, int stage) {
  int r = hop_launch_ss_search(c, n, d_jobs, d_results); if (r) return r;
  if (stage >= HOP_STAGE_FRAC) { r = hop_launch_frac(c, n, d_jobs, d_results); if (r) return r; }
  if (stage >= HOP_STAGE_GT) { r = hop_launch_gt(c, n, d_jobs, d_results); if (r) return r; }
  return HOP_OK;
}

int hop_set_lanes(hop_ctx* c, int lanes) {
  if (!c || lanes < 1 || lanes > HOP_MAX_LANES) return hop_set_err(c, HOP_ERR_ARG, "hop_set_lanes: 1..%d", HOP_MAX_LANES);
  c->lanes = lanes;
  return HOP_OK;
}

int hop_me_search_device(hop_ctx* c, int n, const hop_pu_job* d_jobs, hop_pu_result* d_results, int stage) {
  if (!c || n < 0 || stage < HOP_STAGE_INT || stage > HOP_STAGE_GT || (n && (!d_jobs || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_me_search_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_me_search: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  // Large batches are cut into parts (at multiples of 5 PUs, so that a list of whole CUs keeps its families) that run on
  // separate streams: PUs are independent, the results do not depend on the cut.
  const int parts = (n >= 16384) ? c->lanes : 1;
  const int per = ((n + parts - 1) / parts + 4) / 5 * 5;
  if (parts > 1) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int k = 0; k < parts - 1; k++) HIPCHK(c, hipStreamWaitEvent(c->xstream[k], c->ev_fork, 0));
  }
  int r = me_pipeline(c, std::min(per, n), d_jobs, d_results, stage); if (r) return r;
  for (int k = 0; k < parts - 1; k++) {
    const int o = (k + 1) * per, m = std::min(per, n - o);
    if (m <= 0) break;
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    r = me_pipeline(c, m, d_jobs + o, d_results + o, stage);
    hipError_t e = hipEventRecord(c->ev_join[k], c->stream);
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    if (r) return r;
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[k], 0);
    if (e != hipSuccess) return hop_set_err(c, HOP_ERR_DEVICE, "hop_me_search_device: stream join: %s", hipGetErrorString(e));
  }
  return HOP_OK;
}

int hop_me_search(hop_ctx* c, int n, const hop_pu_job* jobs, hop_pu_result* results, int stage) {
  if (!c || n < 0 || (n && (!jobs || !results))) return hop_set_err(c, HOP_ERR_ARG, "hop_me_search: bad argument");
  if (n == 0) return HOP_OK;
  int r = hop_check_pu_jobs(c, n, jobs); if (r) return r;
  size_t bj = (size_t)n * sizeof(hop_pu_job), o_r = (bj + 255) & ~(size_t)255, br = (size_t)n * sizeof(hop_pu_result);
  void* st; r = hop_stage(c, o_r + br, &st); if (r) return r;
  hop_pu_job* dj = (hop_pu_job*)st; hop_pu_result* dr = (hop_pu_result*)((char*)st + o_r);
  HIPCHK(c, hipMemcpyAsync(dj, jobs, bj, hipMemcpyHostToDevice, c->stream));
  r = hop_me_search_device(c, n, dj, dr, stage); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(results, dr, br, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_pred_inter_device(hop_ctx* c, int n, const hop_pred_job* d_jobs) {
  if (!c || n < 0 || (n && !d_jobs)) return hop_set_err(c, HOP_ERR_ARG, "hop_pred_inter_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_pred(c, n, d_jobs);
}

__global__ void k_gather_pred(const hop_pred_job* jobs, const int64_t* offs, const int16_t* py, const int16_t* pcb, const int16_t* pcr,
                              int pic_w, int16_t* oy, int16_t* ocb, int16_t* ocr) {
  hop_pred_job j = jobs[blockIdx.x];
  int64_t o = offs[blockIdx.x];
  for (int i = threadIdx.x; i < j.w * j.h; i += blockDim.x) {
    int r = i / j.w, cc = i % j.w;
    oy[o + i] = py[(size_t)(j.pu_y + r) * pic_w + j.pu_x + cc];
  }
  int cw = j.w >> 1, ch = j.h >> 1;
  for (int i = threadIdx.x; i < cw * ch; i += blockDim.x) {
    int r = i / cw, cc = i % cw;
    size_t s = (size_t)((j.pu_y >> 1) + r) * (pic_w >> 1) + (j.pu_x >> 1) + cc;
    ocb[(o >> 2) + i] = pcb[s]; ocr[(o >> 2) + i] = pcr[s];
  }
}

}  // extern "C"
// what every predictor job must satisfy before a kernel may run on it (shapes, candidate slot, the reach of the doubled patch + 8-tap filter inside the padded reference)
int hop_check_pred_jobs(hop_ctx* c, int n, const hop_pred_job* jobs) {
  for (int i = 0; i < n; i++) {
    const hop_pred_job& j = jobs[i];
    if (!legal_dim(j.w) || !legal_dim(j.h) || j.pu_x < 0 || j.pu_y < 0 || (j.pu_x & 3) || (j.pu_y & 3) || j.pu_x + j.w > c->pic_w || j.pu_y + j.h > c->pic_h)
      return hop_set_err(c, HOP_ERR_ARG, "pred job %d: illegal PU rectangle", i);
    if (j.dst_row_off < 0 || j.dst_row_off % c->pic_h || j.dst_row_off / c->pic_h > c->slots) return hop_set_err(c, HOP_ERR_ARG, "pred job %d: dst_row_off %d is not one of the context's %d candidate slots", i, j.dst_row_off, c->slots);
    // reach of the doubled patch + 8-tap filter must stay in the margin
    int ix = j.pu_x + (j.mv_x >> 2), iy = j.pu_y + (j.mv_y >> 2);
    const int lim = HOP_MARGIN_Y + HOP_GUARD_ROWS - 4;      // rows may use the guard band, columns wrap linearly like the reference
    if (iy - j.h / 2 - 4 < -lim || iy + j.h + j.h / 2 + 4 >= c->pic_h + lim || ix - j.w / 2 - 4 < -(c->stride_y - 8) || ix + 2 * j.w + 8 > c->stride_y + c->pic_w - 8)
      return hop_set_err(c, HOP_ERR_ARG, "pred job %d: motion vector leaves the padded reference", i);
  }
  return HOP_OK;
}
extern "C" {

int hop_pred_inter(hop_ctx* c, int n, const hop_pred_job* jobs, int16_t* out_y, int16_t* out_cb, int16_t* out_cr) {
  if (!c || n < 0 || (n && !jobs)) return hop_set_err(c, HOP_ERR_ARG, "hop_pred_inter: bad argument");
  if (n == 0) return HOP_OK;
  { int rc = hop_check_pred_jobs(c, n, jobs); if (rc) return rc; }
  std::vector<int64_t> offs(n);
  size_t tot = 0;
  for (int i = 0; i < n; i++) { offs[i] = (int64_t)tot; tot += (size_t)jobs[i].w * jobs[i].h; }
  size_t bj = (size_t)n * sizeof(hop_pred_job), o_o = (bj + 255) & ~(size_t)255, bo = (size_t)n * 8;
  size_t o_y = (o_o + bo + 255) & ~(size_t)255, o_cb = (o_y + tot * 2 + 255) & ~(size_t)255, o_cr = (o_cb + tot / 2 + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_cr + tot / 2 + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_pred(c, n, (const hop_pred_job*)b); if (r) return r;
  if (out_y && out_cb && out_cr) {
    HIPCHK(c, hipMemcpyAsync(b + o_o, offs.data(), bo, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_pred, dim3(n), dim3(256), 0, c->stream, (const hop_pred_job*)b, (const int64_t*)(b + o_o),
                       c->pred[0], c->pred[1], c->pred[2], c->pic_w, (int16_t*)(b + o_y), (int16_t*)(b + o_cb), (int16_t*)(b + o_cr));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out_y, b + o_y, tot * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_cb, b + o_cb, tot / 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_cr, b + o_cr, tot / 2, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_pred_upload(hop_ctx* c, int comp, const int16_t* src) {
  if (!c || !src || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_pred_upload: bad argument");
  size_t n = comp == 0 ? (size_t)c->pic_w * c->pic_h : ((size_t)c->pic_w * c->pic_h) >> 2;
  HIPCHK(c, hipMemcpyAsync(c->pred[comp], src, n * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_recon_upload(hop_ctx* c, int comp, const int16_t* src) {
  if (!c || !src || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_recon_upload: bad argument");
  size_t n = comp == 0 ? (size_t)c->pic_w * c->pic_h : ((size_t)c->pic_w * c->pic_h) >> 2;
  HIPCHK(c, hipMemcpyAsync(c->rec[comp], src, n * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_recon_download(hop_ctx* c, int comp, int16_t* dst) {
  if (!c || !dst || comp < 0 || comp > 2) return hop_set_err(c, HOP_ERR_ARG, "hop_recon_download: bad argument");
  size_t n = comp == 0 ? (size_t)c->pic_w * c->pic_h : ((size_t)c->pic_w * c->pic_h) >> 2;
  HIPCHK(c, hipMemcpyAsync(dst, c->rec[comp], n * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_tu_roundtrip(hop_ctx* c, int n, const hop_tu_job* jobs, hop_tu_result* results, int32_t* levels_out) {
  if (!c || n < 0 || (n && (!jobs || !results))) return hop_set_err(c, HOP_ERR_ARG, "hop_tu_roundtrip: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_tu_roundtrip: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<int64_t> offs(n);
  size_t tot = 0;
  for (int i = 0; i < n; i++) {
    const hop_tu_job& j = jobs[i];
    const int N = 1 << j.log2_size, sh = j.comp ? 1 : 0;
    if (j.comp < 0 || j.comp > 2 || j.log2_size < 2 || j.log2_size > 5 || j.x < 0 || j.y < 0 || ((j.x >> sh) & 3) || ((j.y >> sh) & 3) ||
        (j.x >> sh) + N > (c->pic_w >> sh) || (j.y >> sh) + N > (c->pic_h >> sh) || j.qp_scaled < 0 || j.qp_scaled > 87 || (j.use_dst && (j.log2_size != 2 || j.comp != 0)) || j.scan_idx < 0 || j.scan_idx > 2)
      return hop_set_err(c, HOP_ERR_ARG, "TU job %d: illegal transform unit", i);
    offs[i] = (int64_t)tot; tot += (size_t)N * N;
  }
  const size_t bj = (size_t)n * sizeof(hop_tu_job), o_r = (bj + 255) & ~(size_t)255, o_o = (o_r + (size_t)n * sizeof(hop_tu_result) + 255) & ~(size_t)255;
  const size_t o_l = (o_o + (size_t)n * 8 + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_l + (levels_out ? tot * 4 : 0) + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_o, offs.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_tu(c, n, (const hop_tu_job*)b, (hop_tu_result*)(b + o_r), levels_out ? (int32_t*)(b + o_l) : nullptr, (const int64_t*)(b + o_o));
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(results, b + o_r, (size_t)n * sizeof(hop_tu_result), hipMemcpyDeviceToHost, c->stream));
  if (levels_out) HIPCHK(c, hipMemcpyAsync(levels_out, b + o_l, tot * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_intra_rough(hop_ctx* c, int n, const hop_intra_job* jobs, uint32_t* satd_out) {
  if (!c || n < 0 || (n && (!jobs || !satd_out))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_rough: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_rough: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_intra_job& j = jobs[i];
    const int N = j.size;
    if (!(N == 4 || N == 8 || N == 16 || N == 32 || N == 64) || j.x < 0 || j.y < 0 || (j.x & 3) || (j.y & 3) || j.x + N > c->pic_w || j.y + N > c->pic_h)
      return hop_set_err(c, HOP_ERR_ARG, "intra job %d: illegal block", i);
    // an available unit must lie inside the picture (the reference derives availability from existing CUs)
    const int U = N / 4;
    for (int u = 0; u < 4 * U + 1; u++) if (j.flags[u]) {
      bool ok;
      if (u < 2 * U) ok = j.x > 0 && j.y + 4 * (2 * U - 1 - u) + 4 <= c->pic_h;
      else if (u == 2 * U) ok = j.x > 0 && j.y > 0;
      else ok = j.y > 0 && j.x + 4 * (u - 2 * U - 1) + 4 <= c->pic_w;
      if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra job %d: neighbour unit %d flagged available but outside the picture", i, u);
    }
  }
  const size_t bj = (size_t)n * sizeof(hop_intra_job), o_o = (bj + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_o + (size_t)n * 35 * 4, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_intra(c, n, (const hop_intra_job*)b, (uint32_t*)(b + o_o)); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(satd_out, b + o_o, (size_t)n * 35 * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_tu_roundtrip_device(hop_ctx* c, int n, const hop_tu_job* d_jobs, hop_tu_result* d_results, int32_t* d_levels, const int64_t* d_level_offsets) {
  if (!c || n < 0 || (n && (!d_jobs || !d_results)) || (d_levels && !d_level_offsets)) return hop_set_err(c, HOP_ERR_ARG, "hop_tu_roundtrip_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_tu_roundtrip: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  return hop_launch_tu(c, n, d_jobs, d_results, d_levels, d_level_offsets);
}

int hop_intra_rough_device(hop_ctx* c, int n, const hop_intra_job* d_jobs, uint32_t* d_satd) {
  if (!c || n < 0 || (n && (!d_jobs || !d_satd))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_rough_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_rough: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  return hop_launch_intra(c, n, d_jobs, d_satd);
}

int hop_rdoq_device(hop_ctx* c, int n, const hop_rdoq_job* d_jobs, const hop_estbits* d_tables, const int32_t* d_src, int32_t* d_dst, uint32_t* d_abs_sum) {
  if (!c || n < 0 || (n && (!d_jobs || !d_tables || !d_src || !d_dst || !d_abs_sum))) return hop_set_err(c, HOP_ERR_ARG, "hop_rdoq_device: bad argument");
  if (n == 0) return HOP_OK;
  void* work; int r = hop_scratch(c, hop_rdoq_work_bytes(n), &work); if (r) return r;
  return hop_launch_rdoq(c, n, d_jobs, d_tables, d_src, d_dst, d_abs_sum, work);
}

int hop_rdoq(hop_ctx* c, int n, const hop_rdoq_job* jobs, int n_tables, const hop_estbits* tables, size_t n_coeff,
             const int32_t* src, int32_t* dst, uint32_t* abs_sum) {
  if (!c || n < 0 || (n && (!jobs || !tables || !src || !dst || !abs_sum || n_tables <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_rdoq: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_rdoq_job& j = jobs[i];
    const size_t n2 = (size_t)1 << (2 * j.log2_size);
    if (j.log2_size < 2 || j.log2_size > 5 || j.comp < 0 || j.comp > 2 || (j.comp && j.log2_size == 5) || j.scan_idx < 0 || j.scan_idx > 2 ||
        j.tr_depth < 0 || j.tr_depth > 3 || j.qp_scaled < 0 || j.qp_scaled > 87 || j.bit_depth < 8 || j.bit_depth > 12 || !(j.lambda > 0.0) ||
        j.coeff_offset < 0 || (size_t)j.coeff_offset + n2 > n_coeff || j.estbits_index < 0 || j.estbits_index >= n_tables)
      return hop_set_err(c, HOP_ERR_ARG, "RDOQ job %d: illegal transform unit / table / offset", i);
  }
  const size_t bj = (size_t)n * sizeof(hop_rdoq_job), o_t = (bj + 255) & ~(size_t)255, bt = (size_t)n_tables * sizeof(hop_estbits);
  const size_t o_s = (o_t + bt + 255) & ~(size_t)255, o_d = (o_s + n_coeff * 4 + 255) & ~(size_t)255, o_a = (o_d + n_coeff * 4 + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_a + (size_t)n * 4 + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_t, tables, bt, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_s, src, n_coeff * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(b + o_d, 0, n_coeff * 4, c->stream));
  void* work; r = hop_scratch(c, hop_rdoq_work_bytes(n), &work); if (r) return r;
  r = hop_launch_rdoq(c, n, (const hop_rdoq_job*)b, (const hop_estbits*)(b + o_t), (const int32_t*)(b + o_s), (int32_t*)(b + o_d), (uint32_t*)(b + o_a), work);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(dst, b + o_d, n_coeff * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(abs_sum, b + o_a, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_coeff_bits_device(hop_ctx* c, int n, const hop_coeff_bits_job* d_jobs, const hop_cabac_ctx* d_ctx_in, const int32_t* d_coef,
                          uint64_t* d_bits, hop_cabac_ctx* d_ctx_out) {
  if (!c || n < 0 || (n && (!d_jobs || !d_ctx_in || !d_coef || !d_bits))) return hop_set_err(c, HOP_ERR_ARG, "hop_coeff_bits_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_coeff_bits(c, n, d_jobs, d_ctx_in, d_coef, (unsigned long long*)d_bits, d_ctx_out);
}

int hop_coeff_bits(hop_ctx* c, int n, const hop_coeff_bits_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, size_t n_coeff,
                   const int32_t* coef, uint64_t* bits, hop_cabac_ctx* ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !ctx_in || !coef || !bits || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_coeff_bits: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_coeff_bits_job& j = jobs[i];
    if (j.log2_size < 2 || j.log2_size > 5 || j.comp < 0 || j.comp > 2 || (j.comp && j.log2_size == 5) || j.scan_idx < 0 || j.scan_idx > 2 ||
        j.ctx_index < 0 || j.ctx_index >= n_ctx || j.coeff_offset < 0 || (size_t)j.coeff_offset + ((size_t)1 << (2 * j.log2_size)) > n_coeff)
      return hop_set_err(c, HOP_ERR_ARG, "coeff-bits job %d: illegal transform unit / context index / offset", i);
  }
  for (int k = 0; k < n_ctx; k++) for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
  const size_t bj = (size_t)n * sizeof(hop_coeff_bits_job), o_c = (bj + 255) & ~(size_t)255, bc = (size_t)n_ctx * sizeof(hop_cabac_ctx);
  const size_t o_s = (o_c + bc + 255) & ~(size_t)255, o_b = (o_s + n_coeff * 4 + 255) & ~(size_t)255, o_o = (o_b + (size_t)n * 8 + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_o + (ctx_out ? (size_t)n * sizeof(hop_cabac_ctx) : 0) + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, bc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_s, coef, n_coeff * 4, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_coeff_bits(c, n, (const hop_coeff_bits_job*)b, (const hop_cabac_ctx*)(b + o_c), (const int32_t*)(b + o_s), (unsigned long long*)(b + o_b),
                            ctx_out ? (hop_cabac_ctx*)(b + o_o) : nullptr);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(bits, b + o_b, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  if (ctx_out) HIPCHK(c, hipMemcpyAsync(ctx_out, b + o_o, (size_t)n * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_tu_rd_device(hop_ctx* c, int n, const hop_tu_rd_job* d_jobs, const hop_cabac_ctx* d_ctx_in, const int64_t* d_coef_offsets, size_t n_coeff,
                     int32_t* d_levels, hop_tu_rd_result* d_results) {
  if (!c || n < 0 || (n && (!d_jobs || !d_ctx_in || !d_coef_offsets || !d_levels || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_tu_rd_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_tu_rd: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  return hop_launch_tu_rd(c, n, d_jobs, d_ctx_in, d_coef_offsets, n_coeff, d_levels, d_results, 0);
}

int hop_intra_modes_device(hop_ctx* c, int n, const hop_intra_modes_job* d_jobs, const uint32_t* d_satd, hop_intra_modes_result* d_results) {
  if (!c || n < 0 || (n && (!d_jobs || !d_satd || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_modes_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_intra_modes(c, n, d_jobs, d_satd, d_results);
}

int hop_intra_modes(hop_ctx* c, int n, const hop_intra_modes_job* jobs, const uint32_t* satd, hop_intra_modes_result* results) {
  if (!c || n < 0 || (n && (!jobs || !satd || !results))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_modes: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_intra_modes_job& j = jobs[i];
    bool ok = j.pred_num >= 0 && j.pred_num <= 3 && j.mpm_cand >= 0 && j.mpm_cand <= j.pred_num && j.num_full_rd >= 1 && j.num_full_rd <= 8 && j.ctx_state >= 0 && j.ctx_state <= 127 &&
              j.frac_left >= 0 && j.frac_left < 32768 && j.sqrt_lambda > 0.0;
    for (int k = 0; k < j.pred_num && ok; k++) ok = j.preds[k] >= 0 && j.preds[k] < 35;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra modes job %d: illegal predictors / list size / context", i);
  }
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_s = al((size_t)n * sizeof(hop_intra_modes_job)), o_r = al(o_s + (size_t)n * 35 * 4);
  void* st; int r = hop_stage(c, o_r + (size_t)n * sizeof(hop_intra_modes_result) + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, (size_t)n * sizeof(hop_intra_modes_job), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_s, satd, (size_t)n * 35 * 4, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_intra_modes(c, n, (const hop_intra_modes_job*)b, (const uint32_t*)(b + o_s), (hop_intra_modes_result*)(b + o_r)); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(results, b + o_r, (size_t)n * sizeof(hop_intra_modes_result), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_rqt_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_cabac_ctx* d_ctx_in, hop_rqt_result* d_results, int32_t* d_coef_out,
                   hop_cabac_ctx* d_ctx_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_ctx_in || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_rqt: hop_upload_orig has not been called");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_device: illegal CU class");
  if (n == 0) return HOP_OK;
  const size_t wb = hop_rqt_work_bytes(cls->log2_cu, n);
  if (wb > c->rqt_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  return hop_launch_rqt_class(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->inter_split_flag ? 1 : 0, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, n, d_jobs,
                              d_ctx_in, d_results, d_coef_out, d_ctx_out, c->rqt_buf, c->rqt_bytes);
}

int hop_rqt_device_classes(hop_ctx* c, int n_classes, const int* n, const hop_rqt_job* const* d_jobs, const hop_rqt_job* cls, const hop_cabac_ctx* d_ctx_in,
                           hop_rqt_result* const* d_results, int32_t* const* d_coef_out, hop_cabac_ctx* const* d_ctx_out) {
  if (!c || n_classes < 0 || (n_classes && (!n || !d_jobs || !cls || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_device_classes: bad argument");
  if (n_classes == 0) return HOP_OK;
  // classes are independent chains of small kernels: each runs on its own stream (with its own scratch and state buffers)
  const int n_fork = std::min(n_classes, HOP_MAX_LANES) - 1;            // only the lanes this call uses (see hop_inter_cu_device_classes)
  if (n_fork > 0) HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  for (int k = 0; k < n_fork; k++) HIPCHK(c, hipStreamWaitEvent(c->xstream[k], c->ev_fork, 0));
  bool used[HOP_MAX_LANES - 1] = { false, false, false };
  int rc = HOP_OK;
  for (int i = 0; i < n_classes && rc == HOP_OK; i++) {
    const int lane = i % HOP_MAX_LANES;
    if (lane == 0) { rc = hop_rqt_device(c, n[i], d_jobs[i], cls + i, d_ctx_in, d_results[i], d_coef_out ? d_coef_out[i] : nullptr, d_ctx_out ? d_ctx_out[i] : nullptr); continue; }
    const int k = lane - 1;
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
    rc = hop_rqt_device(c, n[i], d_jobs[i], cls + i, d_ctx_in, d_results[i], d_coef_out ? d_coef_out[i] : nullptr, d_ctx_out ? d_ctx_out[i] : nullptr);
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
    used[k] = true;
  }
  for (int k = 0; k < HOP_MAX_LANES - 1; k++) {
    if (!used[k]) continue;
    hipError_t e = hipEventRecord(c->ev_join[k], c->xstream[k]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[k], 0);
    if (e != hipSuccess && rc == HOP_OK) rc = hop_set_err(c, HOP_ERR_DEVICE, "hop_rqt_device_classes: stream join: %s", hipGetErrorString(e));
  }
  return rc;
}

int hop_rqt(hop_ctx* c, int n, const hop_rqt_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, hop_rqt_result* results, int32_t* coef_out, hop_cabac_ctx* ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !ctx_in || !results || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_rqt: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i];
    const int S = 1 << j.log2_cu;
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
              j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 && j.log2_min_tu_in_cu <= j.log2_max_tu &&
              j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0;
    for (int k = 0; k < 3 && ok; k++) ok = j.qp_scaled[k] >= 0 && j.qp_scaled[k] <= 87 && j.lambda_rdoq[k] > 0.0;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "RQT job %d: illegal CU / transform-tree limits / snapshot / parameters", i);
    coff[i + 1] = coff[i] + (size_t)S * S * 3 / 2;
  }
  for (int k = 0; k < n_ctx; k++) for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
  // one pass per class of CUs (same size and transform-tree limits): the walk over the tree is the same for all of them
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.inter_split_flag == !f.inter_split_flag &&
          !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts) { done[i] = 1; idx.push_back(i); cls.push_back(j); }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_c = al((size_t)m * sizeof(hop_rqt_job)), o_r = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_x = al(o_o + (size_t)m * cu3 * 4), o_w = al(o_x + (size_t)m * sizeof(hop_cabac_ctx));
    void* st; int r = hop_stage(c, o_w + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    r = hop_rqt_device(c, m, (const hop_rqt_job*)(b + o_j), &f, (const hop_cabac_ctx*)(b + o_c), (hop_rqt_result*)(b + o_r), (int32_t*)(b + o_o), (hop_cabac_ctx*)(b + o_x));
    if (r) return r;
    std::vector<hop_rqt_result> rr(m); std::vector<int32_t> co((size_t)m * cu3); std::vector<hop_cabac_ctx> cx(m);
    HIPCHK(c, hipMemcpyAsync(rr.data(), b + o_r, (size_t)m * sizeof(hop_rqt_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(co.data(), b + o_o, (size_t)m * cu3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cx.data(), b + o_x, (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) {
      results[idx[t]] = rr[t];
      if (coef_out) memcpy(coef_out + coff[idx[t]], co.data() + (size_t)t * cu3, cu3 * 4);
      if (ctx_out) ctx_out[idx[t]] = cx[t];
    }
  }
  return HOP_OK;
}

int hop_rqt_finish_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, hop_rqt_result* d_results, int32_t* d_coef, const hop_cabac_ctx* d_ctx_after,
                          hop_cu_final* d_finals) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_results || !d_coef || !d_ctx_after || !d_finals))) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_finish_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_rqt_finish: hop_upload_orig has not been called");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_finish_device: illegal CU class");
  if (n == 0) return HOP_OK;
  const size_t wb = hop_rqt_finish_work_bytes(cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, n);
  if (wb > c->rqt_bytes) {                                         // the quadtree's state buffer is free again once its kernels are queued (same stream)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  return hop_launch_rqt_finish(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->use_ts ? 1 : 0, n, d_jobs, d_results, d_coef, d_ctx_after, d_finals, c->rqt_buf);
}

int hop_rqt_finish(hop_ctx* c, int n, const hop_rqt_job* jobs, hop_rqt_result* results, int32_t* coef, const hop_cabac_ctx* ctx_after, hop_cu_final* finals) {
  if (!c || n < 0 || (n && (!jobs || !results || !coef || !ctx_after || !finals))) return hop_set_err(c, HOP_ERR_ARG, "hop_rqt_finish: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_rqt_finish: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i];
    const int S = 1 << j.log2_cu;
    const bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
                    j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 && j.log2_min_tu_in_cu <= j.log2_max_tu && j.log2_cu - j.log2_min_tu_in_cu <= 3 &&
                    j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "RQT finish job %d: illegal CU / transform-tree limits / parameters", i);
    coff[i + 1] = coff[i] + (size_t)S * S * 3 / 2;
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_rqt_result> rr; std::vector<hop_cabac_ctx> cx;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.use_ts == !f.use_ts) {
        done[i] = 1; idx.push_back(i); cls.push_back(j); rr.push_back(results[i]); cx.push_back(ctx_after[i]);
      }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    std::vector<int32_t> co((size_t)m * cu3);
    for (int t = 0; t < m; t++) memcpy(co.data() + (size_t)t * cu3, coef + coff[idx[t]], cu3 * 4);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_r = al((size_t)m * sizeof(hop_rqt_job)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result)), o_x = al(o_o + (size_t)m * cu3 * 4);
    const size_t o_f = al(o_x + (size_t)m * sizeof(hop_cabac_ctx)), o_w = al(o_f + (size_t)m * sizeof(hop_cu_final));
    void* st; int r = hop_stage(c, o_w + hop_rqt_finish_work_bytes(f.log2_cu, f.log2_max_tu, f.log2_min_tu_in_cu, m) + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_r, rr.data(), (size_t)m * sizeof(hop_rqt_result), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_o, co.data(), (size_t)m * cu3 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_x, cx.data(), (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    r = hop_launch_rqt_finish(c, f.log2_cu, f.log2_max_tu, f.log2_min_tu_in_cu, f.use_ts ? 1 : 0, m, (const hop_rqt_job*)(b + o_j), (hop_rqt_result*)(b + o_r), (int32_t*)(b + o_o),
                              (const hop_cabac_ctx*)(b + o_x), (hop_cu_final*)(b + o_f), b + o_w);
    if (r) return r;
    std::vector<hop_cu_final> ff(m);
    HIPCHK(c, hipMemcpyAsync(rr.data(), b + o_r, (size_t)m * sizeof(hop_rqt_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(co.data(), b + o_o, (size_t)m * cu3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ff.data(), b + o_f, (size_t)m * sizeof(hop_cu_final), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) { results[idx[t]] = rr[t]; finals[idx[t]] = ff[t]; memcpy(coef + coff[idx[t]], co.data() + (size_t)t * cu3, cu3 * 4); }
  }
  return HOP_OK;
}

int hop_intra_cu_bits(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, int n_ctx,
                      const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !results || !coef || !ctx_in || !cu_ctx_in || !bits || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_bits: bad argument");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_intra_cu_syntax& y = syntax[i];
    const int parts = 1 << (2 * (j.log2_cu - 2));
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 &&
              j.log2_min_tu_in_cu <= j.log2_max_tu && j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && y.skip_ctx >= 0 && y.skip_ctx <= 2 &&
              y.tr_depth >= 0 && y.tr_depth <= 3 && y.part >= 0 && y.part < parts && (y.part % (parts >> (2 * y.tr_depth))) == 0 && (y.b_luma || y.b_chroma);
    for (int p = 0; p < (y.part_nxn ? 4 : 1) && ok; p++) ok = y.luma_dir[p] >= 0 && y.luma_dir[p] < 35 && y.pred_num[p] >= 0 && y.pred_num[p] <= 3;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra CU bits job %d: illegal CU class / node / syntax elements / snapshot", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1);
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_intra_cu_syntax> sy; std::vector<hop_rqt_result> rr;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts) {
        done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); rr.push_back(results[i]);
      }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    std::vector<int32_t> co((size_t)m * cu3);
    for (int t = 0; t < m; t++) memcpy(co.data() + (size_t)t * cu3, coef + coff[idx[t]], cu3 * 4);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_r = al(o_y + (size_t)m * sizeof(hop_intra_cu_syntax)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_c = al(o_o + (size_t)m * cu3 * 4), o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_b = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx));
    const size_t o_x = al(o_b + (size_t)m * 4), o_v = al(o_x + (size_t)m * sizeof(hop_cabac_ctx)), o_e = al(o_v + (size_t)m * sizeof(hop_cabac_cu_ctx));
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_intra_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_r, rr.data(), (size_t)m * sizeof(hop_rqt_result), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_o, co.data(), (size_t)m * cu3 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    r = hop_launch_intra_cu_bits(c, f.log2_cu, f.log2_max_tu, f.log2_min_tu_in_cu, f.sign_hide ? 1 : 0, f.use_ts ? 1 : 0, m, (const hop_rqt_job*)(b + o_j), (const hop_intra_cu_syntax*)(b + o_y),
                                 (const hop_rqt_result*)(b + o_r), (const int32_t*)(b + o_o), (const hop_cabac_ctx*)(b + o_c), (const hop_cabac_cu_ctx*)(b + o_u), (uint32_t*)(b + o_b),
                                 (hop_cabac_ctx*)(b + o_x), (hop_cabac_cu_ctx*)(b + o_v));
    if (r) return r;
    std::vector<uint32_t> bb(m); std::vector<hop_cabac_ctx> cx(m); std::vector<hop_cabac_cu_ctx> cv(m);
    HIPCHK(c, hipMemcpyAsync(bb.data(), b + o_b, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cx.data(), b + o_x, (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cv.data(), b + o_v, (size_t)m * sizeof(hop_cabac_cu_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) { bits[idx[t]] = bb[t]; if (ctx_out) ctx_out[idx[t]] = cx[t]; if (cu_ctx_out) cu_ctx_out[idx[t]] = cv[t]; }
  }
  return HOP_OK;
}

int hop_intra_rqt_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, int tr_depth, int check_first, const hop_intra_cu_syntax* d_syntax,
                         const hop_intra_rqt_opt* d_opts, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, hop_rqt_result* d_results, int32_t* d_coef_out,
                         hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_syntax || !d_opts || !d_ctx_in || !d_cu_ctx_in || !d_results))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_rqt_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_rqt: hop_upload_orig has not been called");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1 || tr_depth < 0 || tr_depth > 1 || cls->log2_cu - tr_depth < cls->log2_min_tu_in_cu)
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_rqt_device: illegal PU class");
  if (n == 0) return HOP_OK;
  const size_t wb = hop_intra_rqt_work_bytes(cls->log2_cu, n);
  if (wb > c->rqt_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  return hop_launch_intra_rqt(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, tr_depth, check_first ? 1 : 0, n, d_jobs, d_syntax,
                              d_opts, d_ctx_in, d_cu_ctx_in, d_results, d_coef_out, d_ctx_out, d_cu_ctx_out, c->rqt_buf, c->rqt_bytes, nullptr);
}

int hop_intra_rqt(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, int n_ctx, const hop_cabac_ctx* ctx_in,
                  const hop_cabac_cu_ctx* cu_ctx_in, hop_rqt_result* results, int32_t* coef_out, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !opts || !ctx_in || !cu_ctx_in || !results || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_rqt: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_rqt: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_intra_cu_syntax& y = syntax[i];
    const int S = 1 << j.log2_cu, parts = 1 << (2 * (j.log2_cu - 2));
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
              j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 && j.log2_min_tu_in_cu <= j.log2_max_tu &&
              j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0 && j.qp_scaled[0] >= 0 && j.qp_scaled[0] <= 87 && j.lambda_rdoq[0] > 0.0 &&
              y.skip_ctx >= 0 && y.skip_ctx <= 2 && (y.tr_depth == 0 || y.tr_depth == 1) && !y.part_nxn == !y.tr_depth && y.part >= 0 && y.part < parts &&
              (y.part % (parts >> (2 * y.tr_depth))) == 0 && j.log2_cu - y.tr_depth >= j.log2_min_tu_in_cu;
    for (int p = 0; p < (y.part_nxn ? 4 : 1) && ok; p++) ok = y.luma_dir[p] >= 0 && y.luma_dir[p] < 35 && y.pred_num[p] >= 0 && y.pred_num[p] <= 3;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra RQT job %d: illegal CU / PU node / transform-tree limits / syntax elements / snapshot / parameters", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1);
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first]; const int d0 = syntax[first].tr_depth, cf = opts[first].check_first ? 1 : 0;
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_intra_cu_syntax> sy; std::vector<hop_intra_rqt_opt> op;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts &&
          syntax[i].tr_depth == d0 && (opts[i].check_first ? 1 : 0) == cf) { done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); op.push_back(opts[i]); }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_p = al(o_y + (size_t)m * sizeof(hop_intra_cu_syntax)), o_c = al(o_p + (size_t)m * sizeof(hop_intra_rqt_opt));
    const size_t o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_r = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_x = al(o_o + (size_t)m * cu3 * 4), o_v = al(o_x + (size_t)m * sizeof(hop_cabac_ctx)), o_e = al(o_v + (size_t)m * sizeof(hop_cabac_cu_ctx));
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_intra_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_p, op.data(), (size_t)m * sizeof(hop_intra_rqt_opt), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_o, 0, (size_t)m * cu3 * 4, c->stream));
    r = hop_intra_rqt_device(c, m, (const hop_rqt_job*)(b + o_j), &f, d0, cf, (const hop_intra_cu_syntax*)(b + o_y), (const hop_intra_rqt_opt*)(b + o_p), (const hop_cabac_ctx*)(b + o_c),
                             (const hop_cabac_cu_ctx*)(b + o_u), (hop_rqt_result*)(b + o_r), (int32_t*)(b + o_o), (hop_cabac_ctx*)(b + o_x), (hop_cabac_cu_ctx*)(b + o_v));
    if (r) return r;
    std::vector<hop_rqt_result> rr(m); std::vector<int32_t> co((size_t)m * cu3); std::vector<hop_cabac_ctx> cx(m); std::vector<hop_cabac_cu_ctx> cv(m);
    HIPCHK(c, hipMemcpyAsync(rr.data(), b + o_r, (size_t)m * sizeof(hop_rqt_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(co.data(), b + o_o, (size_t)m * cu3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cx.data(), b + o_x, (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cv.data(), b + o_v, (size_t)m * sizeof(hop_cabac_cu_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) {
      results[idx[t]] = rr[t];
      if (coef_out) memcpy(coef_out + coff[idx[t]], co.data() + (size_t)t * cu3, cu3 * 4);
      if (ctx_out) ctx_out[idx[t]] = cx[t];
      if (cu_ctx_out) cu_ctx_out[idx[t]] = cv[t];
    }
  }
  return HOP_OK;
}

int hop_intra_luma_search_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, int part_nxn, int num_full_rd, const hop_intra_cu_syntax* d_syntax,
                                 const hop_intra_rqt_opt* d_opts, const hop_intra_search_job* d_sjobs, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in,
                                 hop_intra_search_result* d_sresults, hop_rqt_result* d_results, int32_t* d_coef_out, int16_t* d_reco_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_syntax || !d_opts || !d_sjobs || !d_ctx_in || !d_cu_ctx_in || !d_sresults || !d_results || !d_coef_out || !d_reco_out)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_luma_search_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_luma_search: hop_upload_orig has not been called");
  const int nxn = part_nxn ? 1 : 0;
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1 || cls->log2_cu - nxn < cls->log2_min_tu_in_cu || num_full_rd < 1 || num_full_rd > 8)
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_luma_search_device: illegal CU class");
  if (n == 0) return HOP_OK;
  const size_t wb = hop_intra_search_work_bytes(cls->log2_cu, n);
  if (wb > c->rqt_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  return hop_launch_intra_search(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, nxn, num_full_rd, n, d_jobs, d_syntax, d_opts,
                                 d_sjobs, d_ctx_in, d_cu_ctx_in, d_sresults, d_results, d_coef_out, d_reco_out, c->rqt_buf, c->rqt_bytes, nullptr);
}

int hop_intra_luma_search(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, const hop_intra_search_job* sjobs,
                          int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, hop_intra_search_result* sresults, hop_rqt_result* results,
                          int32_t* coef_out, int16_t* reco_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !opts || !sjobs || !ctx_in || !cu_ctx_in || !sresults || !results || !coef_out || !reco_out || n_ctx <= 0)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_luma_search: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_luma_search: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0), roff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_intra_cu_syntax& y = syntax[i]; const hop_intra_search_job& s = sjobs[i];
    const int S = 1 << j.log2_cu, nxn = y.part_nxn ? 1 : 0;
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
              j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 && j.log2_min_tu_in_cu <= j.log2_max_tu &&
              j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0 && j.qp_scaled[0] >= 0 && j.qp_scaled[0] <= 87 && j.lambda_rdoq[0] > 0.0 &&
              y.skip_ctx >= 0 && y.skip_ctx <= 2 && j.log2_cu - nxn >= j.log2_min_tu_in_cu && s.num_full_rd >= 1 && s.num_full_rd <= 8 && s.sqrt_lambda > 0.0;
    for (int p = 0; p < 4 && ok; p++) ok = s.left_dir[p] >= 0 && s.left_dir[p] < 35 && s.above_dir[p] >= 0 && s.above_dir[p] < 35;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra search job %d: illegal CU / transform-tree limits / neighbour directions / snapshot / parameters", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1); roff[i + 1] = roff[i] + ((size_t)1 << (2 * j.log2_cu));
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first]; const int nxn = syntax[first].part_nxn ? 1 : 0, nf = sjobs[first].num_full_rd;
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_intra_cu_syntax> sy; std::vector<hop_intra_rqt_opt> op; std::vector<hop_intra_search_job> sj;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts &&
          (syntax[i].part_nxn ? 1 : 0) == nxn && sjobs[i].num_full_rd == nf) { done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); op.push_back(opts[i]); sj.push_back(sjobs[i]); }
    }
    const int m = (int)idx.size();
    const size_t cu2 = (size_t)1 << (2 * f.log2_cu), cu3 = cu2 + (cu2 >> 1);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_p = al(o_y + (size_t)m * sizeof(hop_intra_cu_syntax)), o_s = al(o_p + (size_t)m * sizeof(hop_intra_rqt_opt));
    const size_t o_c = al(o_s + (size_t)m * sizeof(hop_intra_search_job)), o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_q = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx));
    const size_t o_r = al(o_q + (size_t)m * sizeof(hop_intra_search_result)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result)), o_k = al(o_o + (size_t)m * cu3 * 4);
    const size_t o_e = al(o_k + (size_t)m * cu2 * 2);
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_intra_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_p, op.data(), (size_t)m * sizeof(hop_intra_rqt_opt), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_s, sj.data(), (size_t)m * sizeof(hop_intra_search_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_q, 0, o_e - o_q, c->stream));
    r = hop_intra_luma_search_device(c, m, (const hop_rqt_job*)(b + o_j), &f, nxn, nf, (const hop_intra_cu_syntax*)(b + o_y), (const hop_intra_rqt_opt*)(b + o_p),
                                     (const hop_intra_search_job*)(b + o_s), (const hop_cabac_ctx*)(b + o_c), (const hop_cabac_cu_ctx*)(b + o_u), (hop_intra_search_result*)(b + o_q),
                                     (hop_rqt_result*)(b + o_r), (int32_t*)(b + o_o), (int16_t*)(b + o_k));
    if (r) return r;
    std::vector<hop_intra_search_result> sr(m); std::vector<hop_rqt_result> rr(m); std::vector<int32_t> co((size_t)m * cu3); std::vector<int16_t> rk((size_t)m * cu2);
    HIPCHK(c, hipMemcpyAsync(sr.data(), b + o_q, (size_t)m * sizeof(hop_intra_search_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rr.data(), b + o_r, (size_t)m * sizeof(hop_rqt_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(co.data(), b + o_o, (size_t)m * cu3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rk.data(), b + o_k, (size_t)m * cu2 * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) {
      sresults[idx[t]] = sr[t]; results[idx[t]] = rr[t];
      memcpy(coef_out + coff[idx[t]], co.data() + (size_t)t * cu3, cu3 * 4);
      memcpy(reco_out + roff[idx[t]], rk.data() + (size_t)t * cu2, cu2 * 2);
    }
  }
  return HOP_OK;
}

int hop_intra_chroma_search_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_intra_cu_syntax* d_syntax, const hop_intra_rqt_opt* d_opts,
                                   const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, hop_rqt_result* d_results, hop_intra_chroma_result* d_cresults,
                                   int32_t* d_coef_out, int16_t* d_reco_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_syntax || !d_opts || !d_ctx_in || !d_cu_ctx_in || !d_results || !d_cresults || !d_coef_out || !d_reco_out)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_chroma_search_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_chroma_search: hop_upload_orig has not been called");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_chroma_search_device: illegal CU class");
  if (n == 0) return HOP_OK;
  const size_t wb = hop_intra_chroma_work_bytes(cls->log2_cu, n);
  if (wb > c->rqt_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  return hop_launch_intra_chroma_search(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, n, d_jobs, d_syntax, d_opts, d_ctx_in,
                                        d_cu_ctx_in, d_results, d_cresults, d_coef_out, d_reco_out, c->rqt_buf, c->rqt_bytes, nullptr);
}

int hop_intra_chroma_search(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_intra_rqt_opt* opts, int n_ctx, const hop_cabac_ctx* ctx_in,
                            const hop_cabac_cu_ctx* cu_ctx_in, hop_rqt_result* results, hop_intra_chroma_result* cresults, int32_t* coef_out, int16_t* reco_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !opts || !ctx_in || !cu_ctx_in || !results || !cresults || !coef_out || !reco_out || n_ctx <= 0)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_chroma_search: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_chroma_search: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0), roff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_intra_cu_syntax& y = syntax[i];
    const int S = 1 << j.log2_cu, parts = 1 << (2 * (j.log2_cu - 2));
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
              j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 && j.log2_min_tu_in_cu <= j.log2_max_tu &&
              j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0 && y.luma_dir[0] >= 0 && y.luma_dir[0] < 35;
    for (int k = 1; k < 3 && ok; k++) ok = j.qp_scaled[k] >= 0 && j.qp_scaled[k] <= 87 && j.lambda_rdoq[k] > 0.0;
    // the luma tree must be a legal tree of the class: every depth within the limits, quadrants consistent
    for (int p = 0; p < parts && ok; p++) {
      const int d = results[i].tr_idx[p], lg = j.log2_cu - d;
      ok = d >= 0 && d <= 3 && lg >= j.log2_min_tu_in_cu && lg <= j.log2_max_tu && results[i].tr_idx[p - p % (parts >> (2 * d))] == d && results[i].tskip[0][p] <= 1;
    }
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra chroma search job %d: illegal CU / transform-tree limits / luma tree / snapshot / parameters", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1); roff[i + 1] = roff[i] + ((size_t)1 << (2 * j.log2_cu - 1));
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_intra_cu_syntax> sy; std::vector<hop_intra_rqt_opt> op; std::vector<hop_rqt_result> rr;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts) {
        done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); op.push_back(opts[i]); rr.push_back(results[i]);
      }
    }
    const int m = (int)idx.size();
    const size_t cu2 = (size_t)1 << (2 * f.log2_cu), cu3 = cu2 + (cu2 >> 1), h2x2 = cu2 >> 1;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_p = al(o_y + (size_t)m * sizeof(hop_intra_cu_syntax)), o_c = al(o_p + (size_t)m * sizeof(hop_intra_rqt_opt));
    const size_t o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_r = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx)), o_q = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_o = al(o_q + (size_t)m * sizeof(hop_intra_chroma_result)), o_k = al(o_o + (size_t)m * cu3 * 4), o_e = al(o_k + (size_t)m * h2x2 * 2);
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_intra_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_p, op.data(), (size_t)m * sizeof(hop_intra_rqt_opt), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_r, rr.data(), (size_t)m * sizeof(hop_rqt_result), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_q, 0, o_e - o_q, c->stream));
    r = hop_intra_chroma_search_device(c, m, (const hop_rqt_job*)(b + o_j), &f, (const hop_intra_cu_syntax*)(b + o_y), (const hop_intra_rqt_opt*)(b + o_p), (const hop_cabac_ctx*)(b + o_c),
                                       (const hop_cabac_cu_ctx*)(b + o_u), (hop_rqt_result*)(b + o_r), (hop_intra_chroma_result*)(b + o_q), (int32_t*)(b + o_o), (int16_t*)(b + o_k));
    if (r) return r;
    std::vector<hop_intra_chroma_result> cr(m); std::vector<int32_t> co((size_t)m * cu3); std::vector<int16_t> rk((size_t)m * h2x2);
    HIPCHK(c, hipMemcpyAsync(rr.data(), b + o_r, (size_t)m * sizeof(hop_rqt_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cr.data(), b + o_q, (size_t)m * sizeof(hop_intra_chroma_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(co.data(), b + o_o, (size_t)m * cu3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rk.data(), b + o_k, (size_t)m * h2x2 * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) {
      results[idx[t]] = rr[t]; cresults[idx[t]] = cr[t];
      memcpy(coef_out + coff[idx[t]] + cu2, co.data() + (size_t)t * cu3 + cu2, (cu2 >> 1) * 4);
      memcpy(reco_out + roff[idx[t]], rk.data() + (size_t)t * h2x2, h2x2 * 2);
    }
  }
  return HOP_OK;
}

int hop_intra_cu_total_bits_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_intra_cu_syntax* d_syntax, const hop_rqt_result* d_results,
                                   const int32_t* d_coef, const uint32_t* d_dist, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, uint32_t* d_bits, double* d_cost,
                                   hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_syntax || !d_results || !d_coef || !d_ctx_in || !d_cu_ctx_in || !d_bits))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_total_bits_device: bad argument");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_total_bits_device: illegal CU class");
  if (n == 0) return HOP_OK;
  return hop_launch_intra_cu_total(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, n, d_jobs, d_syntax, d_results, d_coef, d_ctx_in,
                                   d_cu_ctx_in, d_dist, d_bits, d_cost, d_ctx_out, d_cu_ctx_out);
}

int hop_intra_cu_total_bits(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_intra_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, const uint32_t* dist,
                            int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, double* cost, hop_cabac_ctx* ctx_out,
                            hop_cabac_cu_ctx* cu_ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !results || !coef || !dist || !ctx_in || !cu_ctx_in || !bits || !cost || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_total_bits: bad argument");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_intra_cu_syntax& y = syntax[i];
    const int parts = 1 << (2 * (j.log2_cu - 2));
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 &&
              j.log2_min_tu_in_cu <= j.log2_max_tu && j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && j.lambda_rd > 0.0 && y.skip_ctx >= 0 && y.skip_ctx <= 2;
    for (int p = 0; p < (y.part_nxn ? 4 : 1) && ok; p++) ok = y.luma_dir[p] >= 0 && y.luma_dir[p] < 35 && y.pred_num[p] >= 0 && y.pred_num[p] <= 3;
    for (int p = 0; p < parts && ok; p++) {
      const int d = results[i].tr_idx[p], lg = j.log2_cu - d;
      ok = d >= (y.part_nxn ? 1 : 0) && d <= 3 && lg >= j.log2_min_tu_in_cu && lg <= j.log2_max_tu && results[i].tr_idx[p - p % (parts >> (2 * d))] == d;
    }
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra CU bits job %d: illegal CU class / syntax elements / transform tree / snapshot", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1);
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_intra_cu_syntax> sy; std::vector<hop_rqt_result> rr; std::vector<uint32_t> dd;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts) {
        done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); rr.push_back(results[i]); dd.push_back(dist[i]);
      }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    std::vector<int32_t> co((size_t)m * cu3);
    for (int t = 0; t < m; t++) memcpy(co.data() + (size_t)t * cu3, coef + coff[idx[t]], cu3 * 4);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_r = al(o_y + (size_t)m * sizeof(hop_intra_cu_syntax)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_d = al(o_o + (size_t)m * cu3 * 4), o_c = al(o_d + (size_t)m * 4), o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_b = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx));
    const size_t o_k = al(o_b + (size_t)m * 4), o_x = al(o_k + (size_t)m * 8), o_v = al(o_x + (size_t)m * sizeof(hop_cabac_ctx)), o_e = al(o_v + (size_t)m * sizeof(hop_cabac_cu_ctx));
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_intra_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_r, rr.data(), (size_t)m * sizeof(hop_rqt_result), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_o, co.data(), (size_t)m * cu3 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_d, dd.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    r = hop_intra_cu_total_bits_device(c, m, (const hop_rqt_job*)(b + o_j), &f, (const hop_intra_cu_syntax*)(b + o_y), (const hop_rqt_result*)(b + o_r), (const int32_t*)(b + o_o),
                                       (const uint32_t*)(b + o_d), (const hop_cabac_ctx*)(b + o_c), (const hop_cabac_cu_ctx*)(b + o_u), (uint32_t*)(b + o_b), (double*)(b + o_k),
                                       (hop_cabac_ctx*)(b + o_x), (hop_cabac_cu_ctx*)(b + o_v));
    if (r) return r;
    std::vector<uint32_t> bb(m); std::vector<double> kk(m); std::vector<hop_cabac_ctx> cx(m); std::vector<hop_cabac_cu_ctx> cv(m);
    HIPCHK(c, hipMemcpyAsync(bb.data(), b + o_b, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(kk.data(), b + o_k, (size_t)m * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cx.data(), b + o_x, (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cv.data(), b + o_v, (size_t)m * sizeof(hop_cabac_cu_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) { bits[idx[t]] = bb[t]; cost[idx[t]] = kk[t]; if (ctx_out) ctx_out[idx[t]] = cx[t]; if (cu_ctx_out) cu_ctx_out[idx[t]] = cv[t]; }
  }
  return HOP_OK;
}

int hop_inter_cu_skip_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_cu_syntax* d_syntax, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in,
                             hop_cu_final* d_finals, uint32_t* d_bits, double* d_cost, hop_cabac_ctx* d_ctx_out, hop_cabac_cu_ctx* d_cu_ctx_out) {
  if (!c || n < 0 || (n && (!d_jobs || !d_syntax || !d_ctx_in || !d_cu_ctx_in || !d_finals || !d_bits || !d_cost))) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_skip_device: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_inter_cu_skip: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  return hop_launch_cu_skip(c, n, d_jobs, d_syntax, d_ctx_in, d_cu_ctx_in, d_finals, d_bits, d_cost, d_ctx_out, d_cu_ctx_out);
}

int hop_inter_cu_skip(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_cu_syntax* syntax, int n_ctx, const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in,
                      hop_cu_final* finals, uint32_t* bits, double* cost, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !ctx_in || !cu_ctx_in || !finals || !bits || !cost || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_skip: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_inter_cu_skip: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_cu_syntax& y = syntax[i];
    const int S = 1 << j.log2_cu;
    const bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.x >= 0 && j.y >= 0 && (j.x & (S - 1)) == 0 && (j.y & (S - 1)) == 0 && j.x + S <= c->pic_w && j.y + S <= c->pic_h &&
                    j.ctx_index >= 0 && j.ctx_index < n_ctx && j.lambda_rd > 0.0 && y.skip_ctx >= 0 && y.skip_ctx <= 2 && y.max_merge_cand >= 1 && y.max_merge_cand <= 5 &&
                    y.pu[0].merge_idx >= 0 && y.pu[0].merge_idx < y.max_merge_cand;
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "skip candidate %d: illegal CU / snapshot / merge index / parameters", i);
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_j = 0, o_y = al((size_t)n * sizeof(hop_rqt_job)), o_c = al(o_y + (size_t)n * sizeof(hop_cu_syntax)), o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx));
  const size_t o_f = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx)), o_b = al(o_f + (size_t)n * sizeof(hop_cu_final)), o_k = al(o_b + (size_t)n * 4), o_x = al(o_k + (size_t)n * 8);
  const size_t o_v = al(o_x + (size_t)n * sizeof(hop_cabac_ctx)), o_e = al(o_v + (size_t)n * sizeof(hop_cabac_cu_ctx));
  void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b + o_j, jobs, (size_t)n * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_y, syntax, (size_t)n * sizeof(hop_cu_syntax), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
  r = hop_inter_cu_skip_device(c, n, (const hop_rqt_job*)(b + o_j), (const hop_cu_syntax*)(b + o_y), (const hop_cabac_ctx*)(b + o_c), (const hop_cabac_cu_ctx*)(b + o_u),
                               (hop_cu_final*)(b + o_f), (uint32_t*)(b + o_b), (double*)(b + o_k), (hop_cabac_ctx*)(b + o_x), (hop_cabac_cu_ctx*)(b + o_v));
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(finals, b + o_f, (size_t)n * sizeof(hop_cu_final), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(bits, b + o_b, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cost, b + o_k, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  if (ctx_out) HIPCHK(c, hipMemcpyAsync(ctx_out, b + o_x, (size_t)n * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
  if (cu_ctx_out) HIPCHK(c, hipMemcpyAsync(cu_ctx_out, b + o_v, (size_t)n * sizeof(hop_cabac_cu_ctx), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// grows c->walk_buf (synchronises the stream first when it has to reallocate)
static int hop_walk_reserve(hop_ctx* c, size_t bytes) {
  if (bytes <= c->walk_bytes) return HOP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->walk_buf) HIPCHK(c, hipFree(c->walk_buf));
  c->walk_buf = nullptr; c->walk_bytes = 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(&c->walk_buf, bytes + bytes / 4));
  c->walk_bytes = bytes + bytes / 4;
  return HOP_OK;
}

// one class of intra candidates, device-resident from the rough search to the cost: luma search -> chroma search -> bits and cost, no host step in between
static int intra_candidate_chain(hop_ctx* c, const hop_intra_class& k, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in) {
  const hop_rqt_job* cls = &k.cls;
  const int nxn = k.part_nxn ? 1 : 0;
  if (k.n < 0 || (k.n && (!k.d_jobs || !k.d_syntax || !k.d_opts || !k.d_sjobs || !k.d_sresults || !k.d_results || !k.d_cresults || !k.d_coef || !k.d_reco_y || !k.d_reco_c ||
                          !k.d_syntax_out || !k.d_dist || !k.d_bits || !k.d_cost)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_device_classes: bad class descriptor");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1 || cls->log2_cu - nxn < cls->log2_min_tu_in_cu || k.num_full_rd < 1 || k.num_full_rd > 8)
    return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_device_classes: illegal CU class");
  if (k.n == 0) return HOP_OK;
  if (k.n <= c->walk_max && !getenv("HOP_WALK_NO_INTRA")) {              // the RD spine's batches: one kernel per candidate (k_walk.inl)
    int r = hop_walk_reserve(c, hop_intra_walk_bytes(c, cls->log2_cu, k.n, k.num_full_rd)); if (r) return r;
    return hop_launch_intra_walk(c, k, d_ctx_in, d_cu_ctx_in, c->walk_buf, c->walk_bytes);
  }
  size_t wb = hop_intra_search_work_bytes(cls->log2_cu, k.n); const size_t wc = hop_intra_chroma_work_bytes(cls->log2_cu, k.n);
  if (wc > wb) wb = wc;
  if (wb > c->rqt_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->rqt_buf) HIPCHK(c, hipFree(c->rqt_buf));
    c->rqt_buf = nullptr; c->rqt_bytes = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(&c->rqt_buf, wb + wb / 8));
    c->rqt_bytes = wb + wb / 8;
  }
  const int sh = cls->sign_hide ? 1 : 0, ts = cls->use_ts ? 1 : 0;
  int r = hop_launch_intra_search(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, sh, ts, nxn, k.num_full_rd, k.n, k.d_jobs, k.d_syntax, k.d_opts, k.d_sjobs, d_ctx_in, d_cu_ctx_in,
                                  k.d_sresults, k.d_results, k.d_coef, k.d_reco_y, c->rqt_buf, c->rqt_bytes, k.d_syntax_out);
  if (r) return r;
  r = hop_launch_intra_chroma_search(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, sh, ts, k.n, k.d_jobs, k.d_syntax_out, k.d_opts, d_ctx_in, d_cu_ctx_in, k.d_results,
                                     k.d_cresults, k.d_coef, k.d_reco_c, c->rqt_buf, c->rqt_bytes, k.d_syntax_out);
  if (r) return r;
  r = hop_launch_intra_dist_sum(c, k.n, k.d_sresults, k.d_cresults, k.d_dist);
  if (r) return r;
  // the chroma direction of the winner goes into the syntax elements before the bits are counted
  return hop_launch_intra_cu_total(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, sh, ts, k.n, k.d_jobs, k.d_syntax_out, k.d_results, k.d_coef, d_ctx_in, d_cu_ctx_in, k.d_dist,
                                   k.d_bits, k.d_cost, k.d_ctx_out, k.d_cu_ctx_out);
}

static int intra_classes_issue(hop_ctx* c, int n_classes, const hop_intra_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in);

int hop_intra_cu_device_classes(hop_ctx* c, int n_classes, const hop_intra_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in) {
  if (!c || n_classes < 0 || (n_classes && (!classes || !d_ctx_in || !d_cu_ctx_in))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_cu_device_classes: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_intra_cu_device_classes: hop_upload_orig has not been called");
  if (n_classes == 0) return HOP_OK;
  return intra_classes_issue(c, n_classes, classes, d_ctx_in, d_cu_ctx_in);
}

static int inter_candidate_chain(hop_ctx* c, const hop_inter_class& k, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in) {
  if (k.n < 0 || (k.n && (!k.d_jobs || !k.d_syntax || !k.d_results || !k.d_coef || !k.d_ctx_after || !k.d_finals || !k.d_bits || !k.d_skipped || !k.d_cost)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_device_classes: bad class descriptor");
  if (k.n == 0) return HOP_OK;
  if (k.n <= c->walk_max && !getenv("HOP_WALK_NO_INTER")) {              // the RD spine's batches: one kernel per candidate (k_walk.inl)
    const hop_rqt_job* cls = &k.cls;
    if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
        cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_device_classes: illegal CU class");
    if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_inter_cu_device_classes: hop_upload_orig has not been called");
    int r = hop_walk_reserve(c, hop_inter_walk_bytes(cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, k.n)); if (r) return r;
    return hop_launch_inter_walk(c, cls, k.n, k.d_jobs, k.d_syntax, d_ctx_in, d_cu_ctx_in, k.d_results, k.d_coef, k.d_ctx_after, k.d_finals, k.d_bits, k.d_skipped, k.d_cost, k.d_ctx_out,
                                 k.d_cu_ctx_out, c->walk_buf, c->walk_bytes);
  }
  int r = hop_rqt_device(c, k.n, k.d_jobs, &k.cls, d_ctx_in, k.d_results, k.d_coef, k.d_ctx_after); if (r) return r;
  r = hop_rqt_finish_device(c, k.n, k.d_jobs, &k.cls, k.d_results, k.d_coef, k.d_ctx_after, k.d_finals); if (r) return r;
  r = hop_inter_cu_bits_device(c, k.n, k.d_jobs, &k.cls, k.d_syntax, k.d_results, k.d_coef, d_ctx_in, d_cu_ctx_in, k.d_bits, k.d_skipped, k.d_ctx_out, k.d_cu_ctx_out); if (r) return r;
  return hop_launch_inter_cost(c, k.n, k.d_jobs, k.d_finals, k.d_bits, k.d_cost);
}

int hop_inter_cu_device_classes(hop_ctx* c, int n_classes, const hop_inter_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in) {
  if (!c || n_classes < 0 || (n_classes && (!classes || !d_ctx_in || !d_cu_ctx_in))) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_device_classes: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_inter_cu_device_classes: hop_upload_orig has not been called");
  if (n_classes == 0) return HOP_OK;
  auto issue = [&]() -> int {
    // only the lanes this call uses take part
    const int n_fork = std::min(n_classes, HOP_MAX_LANES) - 1;
    if (n_fork > 0) HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int k = 0; k < n_fork; k++) HIPCHK(c, hipStreamWaitEvent(c->xstream[k], c->ev_fork, 0));
    bool used[HOP_MAX_LANES - 1] = { false, false, false };
    int rc = HOP_OK;
    for (int i = 0; i < n_classes && rc == HOP_OK; i++) {
      const int lane = i % HOP_MAX_LANES;
      if (lane == 0) { rc = inter_candidate_chain(c, classes[i], d_ctx_in, d_cu_ctx_in); continue; }
      const int k = lane - 1;
      std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
      std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
      rc = inter_candidate_chain(c, classes[i], d_ctx_in, d_cu_ctx_in);
      std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
      std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
      used[k] = true;
    }
    for (int k = 0; k < HOP_MAX_LANES - 1; k++) {
      if (!used[k]) continue;
      hipError_t e = hipEventRecord(c->ev_join[k], c->xstream[k]);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[k], 0);
      if (e != hipSuccess && rc == HOP_OK) rc = hop_set_err(c, HOP_ERR_DEVICE, "hop_inter_cu_device_classes: stream join: %s", hipGetErrorString(e));
    }
    return rc;
  };
  return issue();
}

static int intra_classes_issue(hop_ctx* c, int n_classes, const hop_intra_class* classes, const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in) {
  const int n_fork = std::min(n_classes, HOP_MAX_LANES) - 1;            // only the lanes this call uses (see hop_inter_cu_device_classes)
  if (n_fork > 0) HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  for (int k = 0; k < n_fork; k++) HIPCHK(c, hipStreamWaitEvent(c->xstream[k], c->ev_fork, 0));
  bool used[HOP_MAX_LANES - 1] = { false, false, false };
  int rc = HOP_OK;
  for (int i = 0; i < n_classes && rc == HOP_OK; i++) {
    const int lane = i % HOP_MAX_LANES;
    if (lane == 0) { rc = intra_candidate_chain(c, classes[i], d_ctx_in, d_cu_ctx_in); continue; }
    const int k = lane - 1;
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
    rc = intra_candidate_chain(c, classes[i], d_ctx_in, d_cu_ctx_in);
    std::swap(c->stream, c->xstream[k]); std::swap(c->scratch, c->xscratch[k]); std::swap(c->scratch_bytes, c->xscratch_bytes[k]);
    std::swap(c->rqt_buf, c->xrqt_buf[k]); std::swap(c->rqt_bytes, c->xrqt_bytes[k]); std::swap(c->walk_buf, c->xwalk_buf[k]); std::swap(c->walk_bytes, c->xwalk_bytes[k]);
    used[k] = true;
  }
  for (int k = 0; k < HOP_MAX_LANES - 1; k++) {
    if (!used[k]) continue;
    hipError_t e = hipEventRecord(c->ev_join[k], c->xstream[k]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[k], 0);
    if (e != hipSuccess && rc == HOP_OK) rc = hop_set_err(c, HOP_ERR_DEVICE, "hop_intra_cu_device_classes: stream join: %s", hipGetErrorString(e));
  }
  return rc;
}

int hop_inter_cu_bits_device(hop_ctx* c, int n, const hop_rqt_job* d_jobs, const hop_rqt_job* cls, const hop_cu_syntax* d_syntax, const hop_rqt_result* d_results, const int32_t* d_coef,
                             const hop_cabac_ctx* d_ctx_in, const hop_cabac_cu_ctx* d_cu_ctx_in, uint32_t* d_bits, uint32_t* d_skipped, hop_cabac_ctx* d_ctx_out,
                             hop_cabac_cu_ctx* d_cu_ctx_out) {
  if (!c || n < 0 || !cls || (n && (!d_jobs || !d_syntax || !d_results || !d_coef || !d_ctx_in || !d_cu_ctx_in || !d_bits || !d_skipped)))
    return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_bits_device: bad argument");
  if (cls->log2_cu < 3 || cls->log2_cu > 6 || cls->log2_max_tu < 2 || cls->log2_max_tu > 5 || cls->log2_min_tu_in_cu < 2 || cls->log2_min_tu_in_cu > cls->log2_max_tu ||
      cls->log2_cu - cls->log2_min_tu_in_cu > 3 || cls->log2_cu - cls->log2_max_tu > 1) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_bits_device: illegal CU class");
  if (n == 0) return HOP_OK;
  return hop_launch_cu_bits(c, cls->log2_cu, cls->log2_max_tu, cls->log2_min_tu_in_cu, cls->inter_split_flag ? 1 : 0, cls->sign_hide ? 1 : 0, cls->use_ts ? 1 : 0, n, d_jobs, d_syntax,
                            d_results, d_coef, d_ctx_in, d_cu_ctx_in, d_bits, d_skipped, d_ctx_out, d_cu_ctx_out);
}

int hop_inter_cu_bits(hop_ctx* c, int n, const hop_rqt_job* jobs, const hop_cu_syntax* syntax, const hop_rqt_result* results, const int32_t* coef, int n_ctx,
                      const hop_cabac_ctx* ctx_in, const hop_cabac_cu_ctx* cu_ctx_in, uint32_t* bits, uint32_t* skipped, hop_cabac_ctx* ctx_out, hop_cabac_cu_ctx* cu_ctx_out) {
  if (!c || n < 0 || (n && (!jobs || !syntax || !results || !coef || !ctx_in || !cu_ctx_in || !bits || !skipped || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_inter_cu_bits: bad argument");
  if (n == 0) return HOP_OK;
  std::vector<size_t> coff(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const hop_rqt_job& j = jobs[i]; const hop_cu_syntax& y = syntax[i];
    bool ok = j.log2_cu >= 3 && j.log2_cu <= 6 && j.ctx_index >= 0 && j.ctx_index < n_ctx && j.log2_max_tu >= 2 && j.log2_max_tu <= 5 && j.log2_min_tu_in_cu >= 2 &&
              j.log2_min_tu_in_cu <= j.log2_max_tu && j.log2_cu - j.log2_min_tu_in_cu <= 3 && j.log2_cu - j.log2_max_tu <= 1 && y.part_size >= 0 && y.part_size <= 7 &&
              y.n_pu == (y.part_size == 0 ? 1 : y.part_size == 3 ? 4 : 2) && y.skip_ctx >= 0 && y.skip_ctx <= 2 && y.max_merge_cand >= 1 && y.max_merge_cand <= 5;
    for (int p = 0; p < y.n_pu && ok; p++) ok = y.pu[p].merge_flag ? (y.pu[p].merge_idx >= 0 && y.pu[p].merge_idx < 5) : (y.pu[p].mvp_idx >= 0 && y.pu[p].mvp_idx <= 1);
    if (!ok) return hop_set_err(c, HOP_ERR_ARG, "CU bits job %d: illegal CU class / syntax elements / snapshot", i);
    coff[i + 1] = coff[i] + (((size_t)3 << (2 * j.log2_cu)) >> 1);
  }
  for (int k = 0; k < n_ctx; k++) {
    for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
    for (int i = 0; i < 19; i++) if (cu_ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "CU context snapshot %d: state %d out of range", k, i);
  }
  std::vector<char> done(n, 0);
  for (int first = 0; first < n; first++) {
    if (done[first]) continue;
    const hop_rqt_job& f = jobs[first];
    std::vector<int> idx; std::vector<hop_rqt_job> cls; std::vector<hop_cu_syntax> sy; std::vector<hop_rqt_result> rr;
    for (int i = first; i < n; i++) {
      const hop_rqt_job& j = jobs[i];
      if (!done[i] && j.log2_cu == f.log2_cu && j.log2_max_tu == f.log2_max_tu && j.log2_min_tu_in_cu == f.log2_min_tu_in_cu && !j.inter_split_flag == !f.inter_split_flag &&
          !j.sign_hide == !f.sign_hide && !j.use_ts == !f.use_ts) { done[i] = 1; idx.push_back(i); cls.push_back(j); sy.push_back(syntax[i]); rr.push_back(results[i]); }
    }
    const int m = (int)idx.size();
    const size_t cu3 = ((size_t)3 << (2 * f.log2_cu)) / 2;
    std::vector<int32_t> co((size_t)m * cu3);
    for (int t = 0; t < m; t++) memcpy(co.data() + (size_t)t * cu3, coef + coff[idx[t]], cu3 * 4);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_j = 0, o_y = al((size_t)m * sizeof(hop_rqt_job)), o_r = al(o_y + (size_t)m * sizeof(hop_cu_syntax)), o_o = al(o_r + (size_t)m * sizeof(hop_rqt_result));
    const size_t o_c = al(o_o + (size_t)m * cu3 * 4), o_u = al(o_c + (size_t)n_ctx * sizeof(hop_cabac_ctx)), o_b = al(o_u + (size_t)n_ctx * sizeof(hop_cabac_cu_ctx));
    const size_t o_s = al(o_b + (size_t)m * 4), o_x = al(o_s + (size_t)m * 4), o_v = al(o_x + (size_t)m * sizeof(hop_cabac_ctx)), o_e = al(o_v + (size_t)m * sizeof(hop_cabac_cu_ctx));
    void* st; int r = hop_stage(c, o_e + 256, &st); if (r) return r;
    char* b = (char*)st;
    HIPCHK(c, hipMemcpyAsync(b + o_j, cls.data(), (size_t)m * sizeof(hop_rqt_job), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_y, sy.data(), (size_t)m * sizeof(hop_cu_syntax), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_r, rr.data(), (size_t)m * sizeof(hop_rqt_result), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_o, co.data(), (size_t)m * cu3 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, (size_t)n_ctx * sizeof(hop_cabac_ctx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_u, cu_ctx_in, (size_t)n_ctx * sizeof(hop_cabac_cu_ctx), hipMemcpyHostToDevice, c->stream));
    r = hop_launch_cu_bits(c, f.log2_cu, f.log2_max_tu, f.log2_min_tu_in_cu, f.inter_split_flag ? 1 : 0, f.sign_hide ? 1 : 0, f.use_ts ? 1 : 0, m, (const hop_rqt_job*)(b + o_j),
                           (const hop_cu_syntax*)(b + o_y), (const hop_rqt_result*)(b + o_r), (const int32_t*)(b + o_o), (const hop_cabac_ctx*)(b + o_c),
                           (const hop_cabac_cu_ctx*)(b + o_u), (uint32_t*)(b + o_b), (uint32_t*)(b + o_s), (hop_cabac_ctx*)(b + o_x), (hop_cabac_cu_ctx*)(b + o_v));
    if (r) return r;
    std::vector<uint32_t> bb(m), ss(m); std::vector<hop_cabac_ctx> cx(m); std::vector<hop_cabac_cu_ctx> cv(m);
    HIPCHK(c, hipMemcpyAsync(bb.data(), b + o_b, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ss.data(), b + o_s, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cx.data(), b + o_x, (size_t)m * sizeof(hop_cabac_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cv.data(), b + o_v, (size_t)m * sizeof(hop_cabac_cu_ctx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < m; t++) { bits[idx[t]] = bb[t]; skipped[idx[t]] = ss[t]; if (ctx_out) ctx_out[idx[t]] = cx[t]; if (cu_ctx_out) cu_ctx_out[idx[t]] = cv[t]; }
  }
  return HOP_OK;
}

int hop_tu_rd(hop_ctx* c, int n, const hop_tu_rd_job* jobs, int n_ctx, const hop_cabac_ctx* ctx_in, hop_tu_rd_result* results, int32_t* levels_out) {
  if (!c || n < 0 || (n && (!jobs || !ctx_in || !results || !levels_out || n_ctx <= 0))) return hop_set_err(c, HOP_ERR_ARG, "hop_tu_rd: bad argument");
  if (!c->have_orig) return hop_set_err(c, HOP_ERR_STATE, "hop_tu_rd: hop_upload_orig has not been called");
  if (n == 0) return HOP_OK;
  std::vector<int64_t> offs(n);
  size_t tot = 0;
  for (int i = 0; i < n; i++) {
    const hop_tu_rd_job& j = jobs[i];
    const int N = 1 << j.log2_size, sh = j.comp ? 1 : 0;
    if (j.comp < 0 || j.comp > 2 || j.log2_size < 2 || j.log2_size > 5 || (j.comp && j.log2_size == 5) || j.x < 0 || j.y < 0 || ((j.x >> sh) & 3) || ((j.y >> sh) & 3) ||
        (j.x >> sh) + N > (c->pic_w >> sh) || (j.y >> sh) + N > (c->pic_h >> sh) || j.qp_scaled < 0 || j.qp_scaled > 87 || j.tr_depth < 0 || j.tr_depth > 3 ||
        j.ctx_index < 0 || j.ctx_index >= n_ctx || j.bit_depth != (j.comp ? c->bd_c : c->bd_y) || !(j.lambda_rdoq > 0.0) || !(j.lambda_rd > 0.0) ||
        j.scan_idx < 0 || j.scan_idx > 2 || (!j.is_intra && (j.scan_idx || j.use_dst)) || (j.flags & ~3) ||
        ((j.flags & HOP_TU_RD_TS) && (j.log2_size != 2 || !j.use_ts)))
      return hop_set_err(c, HOP_ERR_ARG, "TU RD job %d: illegal transform unit / snapshot / parameters", i);
    offs[i] = (int64_t)tot; tot += (size_t)N * N;
  }
  for (int k = 0; k < n_ctx; k++) for (int i = 0; i < 150; i++) if (ctx_in[k].state[i] > 127) return hop_set_err(c, HOP_ERR_ARG, "context snapshot %d: state %d out of range", k, i);
  const size_t bj = (size_t)n * sizeof(hop_tu_rd_job), o_c = (bj + 255) & ~(size_t)255, bc = (size_t)n_ctx * sizeof(hop_cabac_ctx);
  const size_t o_o = (o_c + bc + 255) & ~(size_t)255, o_l = (o_o + (size_t)n * 8 + 255) & ~(size_t)255, o_r = (o_l + tot * 4 + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_r + (size_t)n * sizeof(hop_tu_rd_result) + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_c, ctx_in, bc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_o, offs.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_tu_rd(c, n, (const hop_tu_rd_job*)b, (const hop_cabac_ctx*)(b + o_c), (const int64_t*)(b + o_o), tot, (int32_t*)(b + o_l), (hop_tu_rd_result*)(b + o_r), 0);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(results, b + o_r, (size_t)n * sizeof(hop_tu_rd_result), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(levels_out, b + o_l, tot * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

int hop_intra_pred(hop_ctx* c, int n, const hop_intra_job* jobs, const int32_t* modes) {
  if (!c || n < 0 || (n && (!jobs || !modes))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_pred: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_intra_job& j = jobs[i];
    const int N = j.size;
    if (!(N == 4 || N == 8 || N == 16 || N == 32 || N == 64) || j.x < 0 || j.y < 0 || (j.x & 3) || (j.y & 3) || j.x + N > c->pic_w || j.y + N > c->pic_h || modes[i] < 0 || modes[i] > 34)
      return hop_set_err(c, HOP_ERR_ARG, "intra pred job %d: illegal block or mode", i);
    const int U = N / 4;
    for (int u = 0; u < 4 * U + 1; u++) if (j.flags[u]) {
      bool ok;
      if (u < 2 * U) ok = j.x > 0 && j.y + 4 * (2 * U - 1 - u) + 4 <= c->pic_h;
      else if (u == 2 * U) ok = j.x > 0 && j.y > 0;
      else ok = j.y > 0 && j.x + 4 * (u - 2 * U - 1) + 4 <= c->pic_w;
      if (!ok) return hop_set_err(c, HOP_ERR_ARG, "intra pred job %d: neighbour unit %d flagged available but outside the picture", i, u);
    }
  }
  const size_t bj = (size_t)n * sizeof(hop_intra_job), o_m = (bj + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_m + (size_t)n * 4 + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_m, modes, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_intra_pred(c, n, (const hop_intra_job*)b, (const int32_t*)(b + o_m)); if (r) return r;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_intra_pred_chroma(hop_ctx* c, int n, const hop_intra_job* jobs, const int32_t* modes) {
  if (!c || n < 0 || (n && (!jobs || !modes))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_pred_chroma: bad argument");
  if (n == 0) return HOP_OK;
  const int cw = c->pic_w >> 1, ch = c->pic_h >> 1;
  for (int i = 0; i < n; i++) {
    const hop_intra_job& j = jobs[i];
    const int N = j.size, x = j.x >> 1, y = j.y >> 1;
    if (!(N == 4 || N == 8 || N == 16 || N == 32) || j.x < 0 || j.y < 0 || (j.x & 7) || (j.y & 7) || x + N > cw || y + N > ch || modes[i] < 0 || modes[i] > 34)
      return hop_set_err(c, HOP_ERR_ARG, "chroma intra pred job %d: illegal block or mode", i);
    const int U = N / 2;
    for (int u = 0; u < 4 * U + 1; u++) if (j.flags[u]) {
      bool ok;
      if (u < 2 * U) ok = x > 0 && y + 2 * (2 * U - 1 - u) + 2 <= ch;
      else if (u == 2 * U) ok = x > 0 && y > 0;
      else ok = y > 0 && x + 2 * (u - 2 * U - 1) + 2 <= cw;
      if (!ok) return hop_set_err(c, HOP_ERR_ARG, "chroma intra pred job %d: neighbour unit %d flagged available but outside the picture", i, u);
    }
  }
  const size_t bj = (size_t)n * sizeof(hop_intra_job), o_m = (bj + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_m + (size_t)n * 4 + 256, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_m, modes, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_intra_pred_chroma(c, n, (const hop_intra_job*)b, (const int32_t*)(b + o_m)); if (r) return r;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}
int hop_intra_pred_chroma_device(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  if (!c || n < 0 || (n && (!d_jobs || !d_modes))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_pred_chroma_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_intra_pred_chroma(c, n, d_jobs, d_modes);
}
int hop_intra_pred_device(hop_ctx* c, int n, const hop_intra_job* d_jobs, const int32_t* d_modes) {
  if (!c || n < 0 || (n && (!d_jobs || !d_modes))) return hop_set_err(c, HOP_ERR_ARG, "hop_intra_pred_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_intra_pred(c, n, d_jobs, d_modes);
}

int hop_distortion_device(hop_ctx* c, int n, const hop_dist_job* d_jobs, uint32_t* d_out) {
  if (!c || n < 0 || (n && (!d_jobs || !d_out))) return hop_set_err(c, HOP_ERR_ARG, "hop_distortion_device: bad argument");
  if (n == 0) return HOP_OK;
  return hop_launch_dist(c, n, d_jobs, d_out);
}

int hop_distortion(hop_ctx* c, int n, const hop_dist_job* jobs, uint32_t* out) {
  if (!c || n < 0 || (n && (!jobs || !out))) return hop_set_err(c, HOP_ERR_ARG, "hop_distortion: bad argument");
  if (n == 0) return HOP_OK;
  for (int i = 0; i < n; i++) {
    const hop_dist_job& j = jobs[i];
    if (j.comp < 0 || j.comp > 2 || j.kind < 0 || j.kind > 2 || j.x < 0 || j.y < 0 || j.w <= 0 || j.h <= 0 || (j.w & 3) || (j.h & 3) || j.x + j.w > c->pic_w || j.y + j.h > c->pic_h)
      return hop_set_err(c, HOP_ERR_ARG, "dist job %d: bad rectangle/kind", i);
  }
  size_t bj = (size_t)n * sizeof(hop_dist_job), o_o = (bj + 255) & ~(size_t)255;
  void* st; int r = hop_stage(c, o_o + (size_t)n * 4, &st); if (r) return r;
  char* b = (char*)st;
  HIPCHK(c, hipMemcpyAsync(b, jobs, bj, hipMemcpyHostToDevice, c->stream));
  r = hop_launch_dist(c, n, (const hop_dist_job*)b, (uint32_t*)(b + o_o)); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(out, b + o_o, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

} // extern "C"
