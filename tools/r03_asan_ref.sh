#!/bin/bash
# CPU only, build container: the UNMODIFIED reference encoder under AddressSanitizer (oracle/Makefile.ref target `asan`) on the 64x64 and 128x128 lenslet fixtures with the HOP
# configuration; the first report (ASan stops at it) goes to profiles/r03_asan_ref.txt.  It documents the out-of-bounds read of the reference's GT search that the tests'
# repeat-on-SIGSEGV loops refer to.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/oracle" -f Makefile.ref -j3 asan > /dev/null || exit 1
TD=$(mktemp -d)
python3 - "$TD" <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT/tests")
from hoputil import lenslet
Y, Cb, Cr = lenslet(128, 128, 16, 1234)
open(sys.argv[1] + "/in.yuv", "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
PY
cd "$TD"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:symbolize=1 "$ROOT/oracle/_ref/TAppEncoderRefAsan" -c /root/reference/cfg/3DHencoder_intra_main.cfg -i in.yuv -wdt 128 -hgt 128 -fr 30 -f 1 -q 32 --MIsize=16 -b s.bin -o rec.yuv > out.txt 2> asan.txt
echo "exit code $?" >> asan.txt
{ echo "# oracle/_ref/TAppEncoderRefAsan (the unmodified reference encoder, g++ -O1 -g -fsanitize=address, oracle/Makefile.ref target asan) on the 128x128 lenslet fixture"
  echo "# (hoputil.lenslet(128, 128, 16, 1234)), cfg/3DHencoder_intra_main.cfg --MIsize=16, QP 32.  tools/r03_asan_ref.sh.  First report (halt_on_error=1):"
  head -n 60 asan.txt; } > "$ROOT/profiles/r03_asan_ref.txt"
tail -n 3 asan.txt
