// tools/spine_prof.cpp -- DEVELOPER TOOL: the host side of the RD spine alone.  Replays a request log (HOP_SPINE_LOG of a raster-order run of the same picture, e.g. from
// tests: HOP_SPINE_LOG=/tmp/l.bin with hop_spine_cpu_encode) through hopspine::Encoder with a backend that answers from the log, so that the time is the spine's own:
// candidate lists, CU bookkeeping, request marshalling.  Build with -pg for gprof:
//   g++ -std=c++11 -O2 -pg -I include -I hevc-hop_amd/host tools/spine_prof.cpp hevc-hop_amd/host/hop_spine.cpp hevc-hop_amd/host/hop_hostlogic.cpp -lpthread -o /tmp/spine_prof
//   /tmp/spine_prof W H log.bin [slots] [repeat]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "hop_spine.h"
using namespace hopspine;

class Replay : public BatchInner {
 public:
  explicit Replay(const char* path) : at_(0) { FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(1); } fseek(f, 0, SEEK_END); buf_.resize(ftell(f)); fseek(f, 0, SEEK_SET); if (fread(&buf_[0], 1, buf_.size(), f) != buf_.size()) exit(1); fclose(f); }
  void rewind() { at_ = 0; }
  void begin_frame() {}
  void me_search(int, int n, const hop_pu_job*, hop_pu_result* res) { next(0, res, n * sizeof(hop_pu_result)); }
  void pred_inter(int, int, const hop_pred_job*) { next(1, NULL, 0); }
  void distortion(int, int n, const hop_dist_job*, uint32_t* out) { next(2, out, n * 4); }
  void valid_pattern(int, int n, const int32_t*, uint8_t* out) { next(3, out, n); }
  void pred_cost(int, int n, const hop_pred_job*, int, uint32_t* out) { next(9, out, n * 4); }
  void inter_cu(int, const InterEval&, const Coder&, EvalResult& out) { next(4, &out, sizeof(out)); }
  void intra_cu(int, const IntraEval&, const Coder&, EvalResult& out) { next(5, &out, sizeof(out)); }
  void recon_save(int, int, int, int, int) {}
  void recon_restore(int, int, int, int, int) {}
  void commit(int, int, int, int) {}
  size_t records = 0;
 private:
  void next(int kind, void* out, size_t nb) {
    if (at_ + 16 > buf_.size()) { fprintf(stderr, "log exhausted\n"); exit(2); }
    int32_t h[2]; uint32_t z[2]; memcpy(h, &buf_[at_], 8); memcpy(z, &buf_[at_ + 8], 8);
    if (h[0] != kind || z[1] != nb) { fprintf(stderr, "record %zu: kind %d / %u bytes in the log, %d / %zu asked\n", records, h[0], z[1], kind, nb); exit(2); }
    if (nb) memcpy(out, &buf_[at_ + 16 + z[0]], nb);
    at_ += 16 + z[0] + z[1]; records++;
  }
  std::vector<char> buf_; size_t at_;
};

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: spine_prof W H log.bin [slots] [repeat]\n"); return 1; }
  const int W = atoi(argv[1]), H = atoi(argv[2]), slots = argc > 4 ? atoi(argv[4]) : 0, rep = argc > 5 ? atoi(argv[5]) : 1;
  Replay be(argv[3]);
  EncConfig cfg; default_hop_config(cfg, W, H, 32, 16); cfg.spec_slots = slots; cfg.slot_pitch = H; cfg.wpp = 1;   // (the log comes from hop_spine_cpu_encode_wpp with lag 0)
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  int n = 0;
  for (int r = 0; r < rep; r++) { be.rewind(); Encoder enc(cfg, &be); enc.encode_frame(0); n = enc.n_ctu(); }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%d CTUs x %d: %.3f s = %.2f ms of host spine per CTU, %zu requests per pass\n", n, rep, s, 1e3 * s / (n * rep), be.records / rep);
  return 0;
}
