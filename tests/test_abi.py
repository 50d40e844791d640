"""CPU: the C ABI as a binding sees it.

1. Layout: a C probe compiled by gcc against include/hophip.h prints sizeof and every field offset of the structs; they must equal hop_sizeof() of the built library and
   the mirrors hophip.py keeps (a silently re-ordered or re-typed field would otherwise only show up as wrong numbers on a GPU).
2. The reference-side binding (oracle/enc_shim_abi.cpp, what INTEGRATION.md describes, compiled against the reference's own headers): oracle/_ref/TAppEncoderAbi links the
   reference's objects with libhophip.so -- every hop_* symbol the binding calls is one the library exports -- and oracle/_ref/TAppEncoderAbiDry (the same binding over
   recording stand-ins) runs one picture here: every struct it marshals was filled with 0xA5 first, and no field may still hold that.  Needs /root/reference (skipped on the
   GPU box)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from hoputil import ROOT, lenslet

sys.path.insert(0, os.path.join(ROOT, "hevc-hop_amd"))
import hophip  # noqa: E402

REF = "/root/reference"
HDR = os.path.join(ROOT, "include", "hophip.h")


def _fields(m):
    if isinstance(m, np.dtype):
        return [(n, m.fields[n][1]) for n in m.names]
    return [(n, getattr(m, n).offset) for n, *_ in m._fields_]


def test_struct_layout_matches_the_header():
    L = hophip.load()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hophip.h"', 'int main(void) {']
    for name, m in hophip.MIRRORS.items():
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in _fields(m):
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "probe.c"), "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.dirname(HDR), "-o", os.path.join(td, "probe"), os.path.join(td, "probe.c")])
        out = subprocess.check_output([os.path.join(td, "probe")], text=True)
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for name, m in hophip.MIRRORS.items():
        assert got[(name, "sizeof")] == hophip.mirror_size(m) == L.hop_sizeof(name.encode()), name
        for f, off in _fields(m):
            assert got[(name, f)] == off, (name, f)
    # every struct the header declares is known to hop_sizeof, and an unknown name is refused
    declared = [n for n in re.findall(r"\}\s*(hop_\w+);", open(HDR).read()) if n != "hop_status"]
    assert len(declared) >= 29
    for n in declared:
        assert L.hop_sizeof(n.encode()) > 0, n
    assert L.hop_sizeof(b"hop_no_such_struct") == -1


def test_library_loads_and_exports_every_function_the_header_declares():
    """every hop_* function declared in include/hophip.h (comments stripped) is a dynamic symbol of the built libhophip.so, the library loads without a GPU (no compute
    call is made), and it exports nothing undeclared except the two internal entries named here"""
    L = hophip.load()
    text = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    declared = set(re.findall(r"\b(hop_\w+)\s*\(", text))
    assert len(declared) >= 96
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", hophip.LIB_PATH], text=True).splitlines() if " T hop_" in ln}
    assert declared <= exported, sorted(declared - exported)
    assert exported - declared <= {"hop_coef_put_device", "hop_fiber_switch"}, sorted(exported - declared)      # spine-internal helpers (k_spine.hip, the fibers' register switch)
    for n in declared:
        assert hasattr(L, n), n
    assert b"gfx950" in L.hop_version()


def _ref_build(*targets):
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4"] + ["_ref/" + t for t in targets], stdout=subprocess.DEVNULL)
    return [os.path.join(ROOT, "oracle", "_ref", t) for t in targets]


def test_reference_binding_links_against_the_library():
    (exe,) = _ref_build("TAppEncoderAbi")
    undefined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True).splitlines() if " hop_" in ln}
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", hophip.LIB_PATH], text=True).splitlines() if " T hop_" in ln}
    assert {"hop_ctx_create", "hop_upload_orig", "hop_ssref_reset", "hop_ssref_commit_cus", "hop_set_search_range", "hop_me_search", "hop_me_finish", "hop_pred_inter"} <= undefined
    assert undefined <= exported, undefined - exported
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "libhophip.so" in needed
    # the members the binding replaces are the binding's, not the reference's, in the linked program
    syms = subprocess.check_output(["nm", "-C", exe], text=True)
    for member in ("TEncSearch::xMotionEstimation", "TComPrediction::xPredInterLumaBlk", "TComPrediction::xPredInterChromaBlk", "TEncCu::xCopyYuv2SSRef"):
        assert re.search(r" T " + re.escape(member) + r"\(", syms), member


def test_picture_level_and_decoder_bindings_link_against_the_library():
    """oracle/_ref/TAppEncoderPic (compressCU, loopFilterPic, SAOProcess over the library) and oracle/_ref/TAppDecoderAbi (the decoder's predictor and SS-reference upkeep):
    every hop_* symbol they call is exported, and the replaced members in the linked programs are the bindings'"""
    enc, dec = _ref_build("TAppEncoderPic", "TAppDecoderAbi")
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", hophip.LIB_PATH], text=True).splitlines() if " T hop_" in ln}
    want = {enc: {"hop_encode_frame", "hop_levels_download", "hop_rd_fraction_download", "hop_recon_download", "hop_deblock_frame", "hop_sao_frame", "hop_sao_stats", "hop_psnr", "hop_upload_orig"},
            dec: {"hop_pred_inter", "hop_ssref_commit_cus", "hop_ssref_reset", "hop_ctx_create"}}
    members = {enc: ("TEncCu::compressCU", "TComLoopFilter::loopFilterPic", "TEncSampleAdaptiveOffset::SAOProcess"), dec: ("TComPrediction::xPredInterLumaBlk", "TDecCu::xFindSSRef2Copy")}
    for exe in (enc, dec):
        undefined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True).splitlines() if " hop_" in ln}
        assert want[exe] <= undefined <= exported, (exe, want[exe] - undefined, undefined - exported)
        syms = subprocess.check_output(["nm", "-C", exe], text=True)
        for m in members[exe]:
            assert re.search(r" T " + re.escape(m) + r"\(", syms), (exe, m)




def test_reference_binding_marshals_every_field():
    (exe,) = _ref_build("TAppEncoderAbiDry")
    W = H = 128
    Y, Cb, Cr = lenslet(W, H, 16, 1234)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        r = subprocess.run([exe, "-c", os.path.join(REF, "cfg", "3DHencoder_intra_main.cfg"), "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_ABI_DUMP=os.path.join(td, "dump.bin")))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        raw = open(os.path.join(td, "dump.bin"), "rb").read()
    recs = {1: [], 2: [], 3: [], 4: [], 5: []}
    o = 0
    while o < len(raw):
        kind, n = np.frombuffer(raw, "<i4", 2, o)
        recs[int(kind)].append(raw[o + 8:o + 8 + n]); o += 8 + int(n)
    assert len(recs[1]) == 1 and np.frombuffer(recs[1][0], "<i4").tolist() == [W, H, 8, 8, 0]          # hop_ctx_create once, with the picture's geometry
    assert len(recs[2]) == 1 and np.frombuffer(recs[2][0], "<i4")[2] == int(Y[0, 0])                     # hop_upload_orig once per picture, pointing at the original
    jobs = np.frombuffer(b"".join(recs[4]), hophip.PU_JOB_DTYPE)
    preds = np.frombuffer(b"".join(recs[5]), np.dtype([("pu_x", "<i4"), ("pu_y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4"), ("use_gt", "<i4"), ("gt", "<i4", 8), ("dst_row_off", "<i4")]))
    rects = np.frombuffer(b"".join(recs[3]), "<i4").reshape(-1, 4)
    assert len(jobs) > 300 and len(rects) >= 4
    poison = np.uint32(0xA5A5A5A5).astype(np.uint32)
    for arr in (jobs, preds):
        for f in arr.dtype.names:
            assert not np.any(arr[f].astype(np.int64) & 0xFFFFFFFF == int(poison)), f
    assert not np.any(rects.astype(np.int64) & 0xFFFFFFFF == int(poison))
    # and the values are the ones the interface documents: PUs inside the picture, shapes of the partition modes, the search window of a 2Nx2N PU at the origin, ...
    assert jobs["pu_x"].min() >= 0 and (jobs["pu_x"] + jobs["w"]).max() <= W and jobs["pu_y"].min() >= 0 and (jobs["pu_y"] + jobs["h"]).max() <= H
    assert set(jobs["w"].tolist()) <= {4, 8, 12, 16, 24, 32, 48, 64} and set(jobs["flags"].tolist()) == {hophip.HOP_FLAG_FEN | hophip.HOP_FLAG_HADME}
    assert np.all((jobs["n_amvp"] >= 0) & (jobs["n_amvp"] <= 2)) and np.all(jobs["lambda_cost"] > 0)
    assert np.all(jobs["rng_left"] <= jobs["rng_right"] + 1) and np.all(jobs["rng_left"] >= -128 - 64) and np.all(jobs["rng_right"] <= 128 + 64)
    assert set(map(tuple, rects[:, 2:].tolist())) <= {(8, 0), (16, 0), (32, 0), (64, 0)}
    covered = np.zeros((H, W), bool)
    for x, y, s, _ in rects.tolist():
        covered[y:y + s, x:x + s] = True
    assert covered.all()                                                                                 # every coded CU was committed to the resident SS reference
