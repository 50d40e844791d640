#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE's own code.

TEST INFRASTRUCTURE.  Needs oracle/_ref/libref_harness.so and the reference applications, which
`make -C oracle ref` compiles from the sources under /root/reference (they never travel with the
repository).  What is written is data only: seeded synthetic inputs and the outputs the reference's
functions produced for them.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz and encoder_*.json
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hoputil import (ROOT, Planes, I16P, lambda_for_qp, lenslet, p16, ref, ref_load_planes)  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SHAPES = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16), (8, 4), (4, 8),
          (64, 16), (64, 48), (16, 64), (48, 64), (32, 8), (32, 24), (8, 32), (24, 32), (16, 4), (16, 12), (4, 16), (12, 16)]


def cu_size_for(w, h):
    m = max(w, h)
    return 64 if m > 32 else 32 if m > 16 else 16 if m > 8 else 8


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def coded_planes(Y, Cb, Cr, W, H, ctu_r, ctu_c, partial):
    """SS reference as the encoder would hold it when CTU (ctu_r, ctu_c) is being coded: all CTU rows
    above complete, same row up to ctu_c, plus `partial` already-finalised CUs (x,y,size) of the
    current CTU; everything else the -1 sentinel; borders extended as TComPicYuv does."""
    pl = Planes(W, H)
    R = ref()
    R.ref_pic_create(W, H)
    R.ref_pic_reset()

    def put(x, y, sx, sy):
        ry = np.ascontiguousarray(Y[y:y + sy, x:x + sx])
        pl.y00()[y:y + sy, x:x + sx] = ry
        pl.bufCb[40 + y // 2:40 + (y + sy) // 2, 40 + x // 2:40 + (x + sx) // 2] = Cb[y // 2:(y + sy) // 2, x // 2:(x + sx) // 2]
        pl.bufCr[40 + y // 2:40 + (y + sy) // 2, 40 + x // 2:40 + (x + sx) // 2] = Cr[y // 2:(y + sy) // 2, x // 2:(x + sx) // 2]
    if ctu_r > 0:
        put(0, 0, W, ctu_r * 64)
    if ctu_c > 0:
        put(0, ctu_r * 64, ctu_c * 64, min(64, H - ctu_r * 64))
    for (x, y, s) in partial:
        put(x, y, s, s)
    ref_load_planes(R, pl)
    R.ref_pic_extend_border()
    st = ctypes.c_int()
    for comp, (buf, m) in enumerate(((pl.bufY, 80), (pl.bufCb, 40), (pl.bufCr, 40))):
        p = R.ref_pic_plane(comp, ctypes.byref(st))
        base = ctypes.addressof(p.contents) - (m * st.value + m) * 2
        buf[...] = np.ctypeslib.as_array(ctypes.cast(base, I16P), shape=buf.shape)
    return pl


def gen_me_chain():
    R = ref()
    qp = 32
    lam, lc = lambda_for_qp(qp)
    R.ref_set_lambda(lam)
    assert R.ref_lambda_motion_sad() == lc
    W, H = 320, 256
    Y, Cb, Cr = lenslet(W, H, 15, 2)
    rng = np.random.default_rng(11)
    rec = np.clip(Y + rng.integers(-3, 4, Y.shape), 0, 255).astype(np.int16)
    wInCtu = 5
    scen = []   # (ctu_r, ctu_c, partial CUs)
    scen.append((2, 3, []))
    scen.append((2, 3, [(192, 128, 32), (224, 128, 32)]))
    scen.append((1, 0, []))
    scen.append((0, 2, []))
    scen.append((0, 0, [(0, 0, 32)]))
    scen.append((3, 4, [(256, 192, 32), (288, 192, 16), (304, 192, 16)]))
    jobs, outs, planes_crc = [], [], []
    for si, (cr_, cc_, partial) in enumerate(scen):
        pl = coded_planes(rec, Cb, Cr, W, H, cr_, cc_, partial)
        planes_crc.append([crc(pl.bufY), crc(pl.bufCb), crc(pl.bufCr)])
        for k in range(14):
            w, h = SHAPES[(si * 14 + k) % len(SHAPES)]
            cuS = cu_size_for(w, h)
            cuX = cc_ * 64 + int(rng.integers(0, 64 // cuS)) * cuS
            cuY = cr_ * 64 + int(rng.integers(0, 64 // cuS)) * cuS
            second = int(rng.integers(0, 2))
            puX, puY = (cuX + (cuS - w), cuY + (cuS - h)) if second else (cuX, cuY)
            offx, offy = puX - cuX, puY - cuY
            if 4 * w == cuS and h == cuS and second == 0:
                offy = cuS      # SIZE_nLx2N quirk: riOffsetY = getHeight(0), TComDataCU.cpp:2283-2286
            firstRow = int(cuY == 0)
            firstCol = int(cuX == 0)
            pred = (int(rng.integers(-60, 60)), int(rng.integers(-260, -40))) if k % 3 else (0, 0)
            o6 = (ctypes.c_int * 6)()
            R.ref_set_search_range(W, H, cuX, cuY, cuS, cr_ * wInCtu + cc_, wInCtu, pred[0], pred[1], 128, offx, offy, firstRow, firstCol, o6)
            l, r, t, b, ox, oy = list(o6)
            nAmvp = 2
            amvp = [pred[0], pred[1], int(rng.integers(-80, 80)), int(rng.integers(-300, -60))]
            if k % 5 == 0:
                amvp[2:] = [0, 0]
            org = np.ascontiguousarray(Y[puY:puY + h, puX:puX + w])
            out = (ctypes.c_int64 * 32)()
            R.ref_me_pu(p16(org), w, puX, puY, w, h, l, r, t, b, ox, oy, pred[0], pred[1], nAmvp,
                        (ctypes.c_int * 4)(*amvp), 3, out)
            jobs.append([si, puX, puY, w, h, cuX, cuY, cuS, offx, offy, firstRow, firstCol, pred[0], pred[1],
                         l, r, t, b, ox, oy, nAmvp] + amvp)
            outs.append(list(out)[:27])
    np.savez_compressed(os.path.join(GOLD, "me_chain.npz"),
                        Y=Y.astype(np.uint8), Cb=Cb.astype(np.uint8), Cr=Cr.astype(np.uint8), rec=rec.astype(np.uint8),
                        scen=np.array([[a, b, len(c)] for a, b, c in scen], np.int32),
                        partial=np.array([p for _, _, c in scen for p in c], np.int32).reshape(-1, 3),
                        planes_crc=np.array(planes_crc, np.uint32),
                        jobs=np.array(jobs, np.int32), outs=np.array(outs, np.int64),
                        qp=np.int32(qp), lambda_cost=np.uint32(lc), W=np.int32(W), H=np.int32(H))
    print("me_chain:", len(jobs), "jobs; valid:", sum(1 for o in outs if not o[3]), "gt:", sum(1 for o in outs if o[9]))


def gen_pred_inter():
    R = ref()
    W, H = 320, 256
    Y, Cb, Cr = lenslet(W, H, 15, 2)
    pl = coded_planes(Y, Cb, Cr, W, H, 3, 2, [(128, 192, 32)])
    rng = np.random.default_rng(3)
    jobs, outs = [], []
    for trial in range(96):
        w, h = SHAPES[trial % len(SHAPES)]
        puX = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        puY = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        puX = (puX // 64) * 64 + min(puX % 64, 64 - w)
        puY = (puY // 64) * 64 + min(puY % 64, 64 - h)
        mvx, mvy = int(rng.integers(-60 * 4, 60 * 4)), int(rng.integers(-60 * 4, 60 * 4))
        mode = trial % 4
        if mode == 0:
            mvx &= ~3
            mvy &= ~3
        useGT = 1 if mode < 3 else 0
        g = rng.integers(-6, 7, 8)
        if trial % 5 != 1:      # affine (what the search/decoder produce); every 5th stays general
            g[6], g[7] = g[0] - g[2] + g[4], g[1] - g[3] + g[5]
        if trial % 7 == 0:
            g[:] = 0
        gt = (ctypes.c_int * 8)(*[int(v) for v in g])
        a = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        R.ref_pred_inter(puX, puY, w, h, mvx, mvy, useGT, gt, p16(a[0]), p16(a[1]), p16(a[2]))
        jobs.append([puX, puY, w, h, mvx, mvy, useGT] + [int(v) for v in g])
        outs.append(np.concatenate([x.ravel() for x in a]))
    flat = np.concatenate(outs)
    np.savez_compressed(os.path.join(GOLD, "pred_inter.npz"), Y=Y.astype(np.uint8), Cb=Cb.astype(np.uint8), Cr=Cr.astype(np.uint8),
                        bufY=pl.bufY, bufCb=pl.bufCb, bufCr=pl.bufCr, jobs=np.array(jobs, np.int32),
                        out_flat=flat.astype(np.int16), out_len=np.array([len(o) for o in outs], np.int32))
    print("pred_inter:", len(jobs), "jobs")


def gen_distortion():
    R = ref()
    rng = np.random.default_rng(21)
    rows = []
    A = rng.integers(-1, 256, (80, 96)).astype(np.int16)
    B = rng.integers(0, 256, (80, 96)).astype(np.int16)
    A10 = (A * 4 + rng.integers(0, 4, A.shape)).clip(-1, 1023).astype(np.int16)
    B10 = (B * 4 + rng.integers(0, 4, B.shape)).astype(np.int16)
    for i in range(len(SHAPES) * 2):
        w, h = SHAPES[i % len(SHAPES)]
        ax, ay, bx, by = [int(v) for v in rng.integers(0, 16, 4)]
        for bd, (a, b) in ((8, (A, B)), (10, (A10, B10))):
            pa = ctypes.cast(a.ctypes.data + (ay * 96 + ax) * 2, I16P)
            pb = ctypes.cast(b.ctypes.data + (by * 96 + bx) * 2, I16P)
            sad0 = R.ref_sad(pb, 96, pa, 96, w, h, bd, 0)
            sad1 = R.ref_sad(pb, 96, pa, 96, w, h, bd, 1) if h > 8 else 0
            hads = R.ref_hads(pb, 96, pa, 96, w, h, bd)
            sse = R.ref_sse(pb, 96, pa, 96, w, h, bd)
            chad = R.ref_calc_had(pb, 96, pa, 96, w, h, bd) if (w == h) else 0
            rows.append([w, h, ax, ay, bx, by, bd, sad0, sad1, hads, sse, chad])
    bits = [R.ref_component_bits(v) for v in range(-600, 601)]
    # homographies + warps
    hom = []
    for i in range(64):
        w, h = SHAPES[i % len(SHAPES)]
        d = rng.integers(-8, 9, 8)
        if i % 3:
            d[6], d[7] = d[0] - d[2] + d[4], d[1] - d[3] + d[5]
        x = [int(d[0]), int(d[2]) + 2 * w - 1, int(d[4]) + 2 * w - 1, int(d[6])]
        y = [int(d[1]), int(d[3]), int(d[5]) + 2 * h - 1, int(d[7]) + 2 * h - 1]
        hh = (ctypes.c_double * 9)()
        R.ref_calc_param_projective((ctypes.c_int * 4)(*x), (ctypes.c_int * 4)(*y), hh, 2 * w, 2 * h)
        patch = rng.integers(0, 256, (2 * h, 2 * w)).astype(np.int16)
        aux = np.zeros((h, w), np.int16)
        centre = ctypes.cast(patch.ctypes.data + ((h // 2) * 2 * w + w // 2) * 2, I16P)
        R.ref_projective_transform(centre, p16(aux), hh, 2 * w, 2 * h, 2 * w, (min(w, h) >> 1) * 2)
        hom.append([w, h] + x + y + [crc(patch), crc(aux)] + [int(np.frombuffer(np.float64(v).tobytes(), np.int64)[0]) for v in hh])
    np.savez_compressed(os.path.join(GOLD, "distortion.npz"), A=A, B=B, A10=A10, B10=B10,
                        rows=np.array(rows, np.int64), bits=np.array(bits, np.int32), hom=np.array(hom, np.int64),
                        hom_seed=np.int32(21))
    print("distortion:", len(rows), "rows;", len(hom), "homographies")


def gen_ssref_commit():
    R = ref()
    W, H = 200, 136      # not CTU-aligned, like the right/bottom edge of the 7728x5368 frame (multiples of 8)
    Y, Cb, Cr = lenslet(W, H, 15, 9)
    R.ref_pic_create(W, H)
    R.ref_pic_reset()
    order = [(0, 0, 64), (64, 0, 64), (128, 0, 64), (192, 0, 8), (192, 8, 8), (0, 64, 32), (32, 64, 32), (0, 96, 32),
             (32, 96, 16), (48, 96, 16), (32, 112, 8), (0, 128, 8), (192, 128, 8), (64, 64, 64), (128, 64, 64), (192, 64, 8)]
    st = ctypes.c_int()
    crcs = []
    for (x, y, s) in order:
        ry = np.ascontiguousarray(Y[y:y + s, x:x + s])
        rb = np.ascontiguousarray(Cb[y // 2:(y + s) // 2, x // 2:(x + s) // 2])
        rr = np.ascontiguousarray(Cr[y // 2:(y + s) // 2, x // 2:(x + s) // 2])
        R.ref_pic_commit_cu(x, y, s, p16(ry), p16(rb), p16(rr))
        row = []
        for comp, (shape, m) in enumerate((((H + 160, W + 160), 80), ((H // 2 + 80, W // 2 + 80), 40), ((H // 2 + 80, W // 2 + 80), 40))):
            p = R.ref_pic_plane(comp, ctypes.byref(st))
            base = ctypes.addressof(p.contents) - (m * st.value + m) * 2
            row.append(crc(np.ctypeslib.as_array(ctypes.cast(base, I16P), shape=shape)))
        crcs.append(row)
    np.savez_compressed(os.path.join(GOLD, "ssref_commit.npz"), Y=Y.astype(np.uint8), Cb=Cb.astype(np.uint8), Cr=Cr.astype(np.uint8),
                        order=np.array(order, np.int32), crcs=np.array(crcs, np.uint32), W=np.int32(W), H=np.int32(H))
    print("ssref_commit:", len(order), "commits")


def gen_encoder():
    """Whole-encoder goldens from the unmodified reference application (pins a0 for later rounds):
    bitstream/recon md5, per-CTU RD cost, PSNR, for small synthetic lenslets with the HOP cfg."""
    enc = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")
    dec = os.path.join(ROOT, "oracle", "_ref", "TAppDecoderRef")
    cfg = "/root/reference/cfg/3DHencoder_intra_main.cfg"
    res = {}
    for (W, H, seed) in ((64, 64, 1234), (128, 128, 1234), (192, 128, 7)):
        Y, Cb, Cr = lenslet(W, H, 16, seed)
        with tempfile.TemporaryDirectory() as td:
            yuv = os.path.join(td, "in.yuv")
            with open(yuv, "wb") as f:
                f.write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
            r = subprocess.run([enc, "-c", cfg, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", "32",
                                "--MIsize=16", "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"],
                               cwd=td, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            d = subprocess.run([dec, "-b", "s.bin", "-o", "dec.yuv"], cwd=td, capture_output=True, text=True)
            rec = open(os.path.join(td, "rec.yuv"), "rb").read()
            decd = open(os.path.join(td, "dec.yuv"), "rb").read()
            assert rec == decd, "encoder recon != decoder recon"
            tot = [ln for ln in r.stdout.splitlines() if "Total Time" in ln]
            poc = [ln for ln in r.stdout.splitlines() if ln.startswith("POC")]
            res["%dx%d_seed%d" % (W, H, seed)] = {
                "input_md5": hashlib.md5(open(yuv, "rb").read()).hexdigest(),
                "bin_md5": hashlib.md5(open(os.path.join(td, "s.bin"), "rb").read()).hexdigest(),
                "bin_bytes": os.path.getsize(os.path.join(td, "s.bin")),
                "rec_md5": hashlib.md5(rec).hexdigest(),
                "cost_csv": open(os.path.join(td, "cost.csv")).read() if os.path.exists(os.path.join(td, "cost.csv")) else None,
                "psnr_txt": open(os.path.join(td, "psnr.txt")).read() if os.path.exists(os.path.join(td, "psnr.txt")) else None,
                "poc_line": poc[0] if poc else None,
                "total_time_line_container": tot[0] if tot else None,
                "decoder_md5_ok": "ERROR" not in d.stdout,
            }
            print("encoder", W, H, res["%dx%d_seed%d" % (W, H, seed)]["poc_line"], tot)
    json.dump(res, open(os.path.join(GOLD, "encoder_hop_qp32.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["dist", "commit", "pred", "me", "enc"]
    if "dist" in which:
        gen_distortion()
    if "commit" in which:
        gen_ssref_commit()
    if "pred" in which:
        gen_pred_inter()
    if "me" in which:
        gen_me_chain()
    if "enc" in which:
        gen_encoder()
