#!/usr/bin/env python3
"""oracle/make_golden24.py -- TEST INFRASTRUCTURE.  Goldens of the configuration bench.py measures (pitch-15 lenslets, --MIsize=15), made by the reference
encoder in the build container.  With MIsize 15 the micro-image candidates (TLibCommon/TComDataCU.cpp:2620-2748) yield vectors of -15 / -30 / -45 / -75 samples, which feed
AMVP, merge and the SS start; every earlier golden uses MIsize 16.

  make_golden24.py small          tests/golden/encoder_spine_mi15.npz: the shim encoder with its observers on (candidate trace of xCheckBestMode, per-CTU cost / bits /
                                  distortion, finished per-partition data -- the layout of encoder_spine.npz, oracle/make_golden19.py) for 200x136 in raster order (partial
                                  CTUs right and below), 192x128 and 448x192 with WaveFrontSynchro; + tests/golden/encoder_hop_pic_mi15.json: bitstream / reconstruction md5 of
                                  the UNMODIFIED encoder for the same pictures (for the picture-level binding, tests/test_encoder_pic.py / test_gpu_encoder_pic.py)
  make_golden24.py band ROWS      tests/golden/encoder_frame_mi15_rows<ROWS>.npz: cost.csv of the UNMODIFIED reference encoder (oracle/_ref/TAppEncoderRef) for the top ROWS
                                  CTU rows of bench.py's frame -- hoputil.lenslet(7728, 5368, 15, seed 2) -- coded as ONE picture of 7728 x 64 ROWS with WaveFrontSynchro, one
                                  substream per CTU row; ROWS = 84 is the whole frame (hours of one core).  bench.py compares the RD costs of the CTUs it
                                  retires with this file.  Kept: the costs (float64 per CTU, coding order), md5 of the input planes, of the bitstream and of the reconstruction.
Needs /root/reference (build container)."""
import hashlib, json, os, subprocess, sys, tempfile, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hoputil import lenslet, hop_encoder_args  # noqa: E402

SMALL = [(200, 136, 5, False), (192, 128, 7, True), (448, 192, 3, True)]      # W, H, seed, WaveFrontSynchro
FRAME_W, FRAME_H, PITCH, SEED = 7728, 5368, 15, 2                            # bench.py's frame


def md5(b):
    return hashlib.md5(b).hexdigest()


def run_ref(args, td, tries=6):
    """the unmodified encoder; its GT search reads past its reference picture buffer (profiles/r03_asan_ref.txt), which now and then kills the process: such a run is repeated"""
    for _ in range(tries):
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")] + args, cwd=td, capture_output=True, text=True)
        if r.returncode != -11: break
        for f in ("cost.csv", "qpValues.csv", "psnr.txt"):                       # side files are appended to
            if os.path.exists(os.path.join(td, f)): os.remove(os.path.join(td, f))
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return r


def small():
    from spine_check import run_reference, read_ctu_trace
    out, pics = {}, {}
    for W, H, seed, wpp in SMALL:
        key = "%dx%d_seed%d_mi15%s" % (W, H, seed, "_wpp" if wpp else "")
        with tempfile.TemporaryDirectory() as td:
            Y, Cb, Cr, _ = run_reference(W, H, seed, False, td, mi=15, wpp=wpp, pitch=15)
            text = open(os.path.join(td, "best.txt"), "rb").read()
            ctu = read_ctu_trace(os.path.join(td, "ctu.bin"))
        with tempfile.TemporaryDirectory() as td:                               # the unmodified encoder on the same input: cost.csv, bitstream, reconstruction
            raw = Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes()
            open(os.path.join(td, "in.yuv"), "wb").write(raw)
            over = {"WaveFrontSynchro": 1, "WaveFrontSubstreams": (H + 63) // 64} if wpp else {}
            run_ref(hop_encoder_args(W, H, mi=15, **over), td)
            ref_costs = [float(v) for v in open(os.path.join(td, "cost.csv")).read().strip().split(";")[1:]]
            b, r = open(os.path.join(td, "s.bin"), "rb").read(), open(os.path.join(td, "rec.yuv"), "rb").read()
            pics[key] = {"W": W, "H": H, "seed": seed, "pitch": 15, "mi": 15, "wpp": int(wpp), "input_md5": md5(raw), "bin_md5": md5(b), "rec_md5": md5(r), "bin_bytes": len(b)}
        assert ref_costs == [float(c) for c in ctu["cost"]], (key, "cost.csv of the unmodified encoder differs from the shim's")
        out[key + "/cost"] = ctu["cost"].astype(np.float64); out[key + "/bits"] = ctu["bits"].astype(np.uint32); out[key + "/dist"] = ctu["dist"].astype(np.uint32)
        out[key + "/parts"] = ctu["p"].astype(np.int16)
        out[key + "/trace"] = np.frombuffer(zlib.compress(text, 9), np.uint8)
        print(key, len(ctu), "CTUs", text.count(b"\n"), "candidates; cost.csv of the unmodified encoder matched", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_spine_mi15.npz"), **out)
    json.dump(pics, open(os.path.join(ROOT, "tests", "golden", "encoder_hop_pic_mi15.json"), "w"), indent=1, sort_keys=True)


def band(rows):
    Y, Cb, Cr = lenslet(FRAME_W, FRAME_H, PITCH, SEED)
    H = min(rows * 64, FRAME_H)
    Y, Cb, Cr = Y[:H], Cb[:H // 2], Cr[:H // 2]
    td = os.environ.get("GOLDEN24_DIR") or tempfile.mkdtemp(prefix="g24_")
    os.makedirs(td, exist_ok=True)
    raw = Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes()
    open(os.path.join(td, "in.yuv"), "wb").write(raw)
    t0 = time.time()
    r = run_ref(hop_encoder_args(FRAME_W, H, mi=15, WaveFrontSynchro=1, WaveFrontSubstreams=(H + 63) // 64), td)
    dt = time.time() - t0
    costs = np.array([float(v) for v in open(os.path.join(td, "cost.csv")).read().strip().split(";")[1:]], np.float64)
    n = ((FRAME_W + 63) // 64) * ((H + 63) // 64)
    assert len(costs) == n, (len(costs), n)
    tot = [ln for ln in r.stdout.split("\n") if "Total Time" in ln]
    meta = {"W": FRAME_W, "H": H, "pitch": PITCH, "seed": SEED, "mi": 15, "qp": 32, "rows": rows, "y_md5": md5(Y.tobytes()), "cb_md5": md5(Cb.tobytes()), "cr_md5": md5(Cr.tobytes()),
            "bin_md5": md5(open(os.path.join(td, "s.bin"), "rb").read()), "rec_md5": md5(open(os.path.join(td, "rec.yuv"), "rb").read()), "wall_s": dt,
            "total_time_line": tot[0].strip() if tot else "", "ctu_per_s_one_core": n / dt, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_frame_mi15_rows%d.npz" % rows), cost=costs, meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    print(json.dumps(meta), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "small": small()
    else: band(int(sys.argv[2]))
