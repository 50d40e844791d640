#!/usr/bin/env python3
"""tools/spine_logdiff.py A.log B.log -- first request two backends answered differently (HOP_SPINE_LOG of hevc-hop_amd/host/hop_spine.h:LogBackend)."""
import struct, sys
import numpy as np
def records(path):
    b = open(path, "rb").read(); o = 0
    while o + 16 <= len(b):
        kind, n, na, nb = struct.unpack_from("<iiII", b, o); o += 16
        yield kind, n, b[o:o + na], b[o + na:o + na + nb]; o += na + nb
NAMES = {0: "me_search", 1: "pred_inter", 2: "distortion", 3: "valid_pattern", 4: "inter_cu", 5: "intra_cu", 9: "pred_cost"}
def main():
    for i, (a, b) in enumerate(zip(records(sys.argv[1]), records(sys.argv[2]))):
        if a[0] != b[0] or a[2] != b[2]:
            print("request %d: the REQUESTS differ (%s vs %s): an earlier answer differed in a field the log's comparison below does not cover" % (i, NAMES.get(a[0]), NAMES.get(b[0]))); return 1
        if a[0] == 0:                                                   # ME results: a PU without a valid candidate carries nothing else
            ra, rb = np.frombuffer(a[3], '<i4').reshape(-1, 25), np.frombuffer(b[3], '<i4').reshape(-1, 25)
            if all((x[3] and y[3]) or np.array_equal(x, y) for x, y in zip(ra, rb)): continue
        if a[3] != b[3]:
            print("request %d (%s, n %d): answers differ" % (i, NAMES.get(a[0]), a[1]))
            x, y = np.frombuffer(a[3], np.uint8), np.frombuffer(b[3], np.uint8)
            d = np.nonzero(x != y)[0]
            print("  %d of %d answer bytes differ, first offsets %s" % (len(d), len(x), d[:24].tolist()))
            open("/tmp/spine_req.bin", "wb").write(a[2]); open("/tmp/spine_ans_a.bin", "wb").write(a[3]); open("/tmp/spine_ans_b.bin", "wb").write(b[3])
            if a[0] in (4, 5):
                hdr = np.frombuffer(a[2][:12], "<i4"); print("  job x, y, log2_cu:", hdr.tolist())
                for nm, z in (("A", a[3]), ("B", b[3])): print("  ", nm, "cost/bits/dist/skipped/root:", struct.unpack_from("<dIIii", z, 0))
            return 1
    print("logs equal over", i + 1, "requests"); return 0
if __name__ == "__main__":
    sys.exit(main())
