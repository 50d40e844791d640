"""Shared helpers for the test-suite, the golden-vector generator and bench.py's cpu_baseline leg.

Everything here is TEST INFRASTRUCTURE: synthetic lenslet generator, ctypes bindings of the CPU
oracle (oracle/libhop_oracle.so) and -- only where /root/reference was available at build time --
of the reference-derived harness (oracle/_ref/libref_harness.so).
"""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libhop_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_harness.so")

MARGIN_Y = 80   # TLibCommon/TComPicYuv.cpp:82-85
MARGIN_C = 40

I16P = ctypes.POINTER(ctypes.c_int16)
I32P = ctypes.POINTER(ctypes.c_int32)
I64P = ctypes.POINTER(ctypes.c_int64)
F64P = ctypes.POINTER(ctypes.c_double)


def p16(a):
    assert a.dtype == np.int16
    return a.ctypes.data_as(I16P)


def lenslet(W, H, pitch=15, seed=2, bitdepth=8):
    """Synthetic lenslet frame: square-packed micro-images of `pitch` px,
    each a disparity-shifted view of one scene texture, radial vignetting, additive noise.
    Returns (Y, Cb, Cr) int16 arrays, 4:2:0."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    mx, my = np.floor(xx / pitch), np.floor(yy / pitch)          # micro-image index
    ux, uy = xx - mx * pitch - pitch / 2.0, yy - my * pitch - pitch / 2.0
    disp = 0.35
    sx, sy = mx * pitch * 0.12 + ux * disp * 3.0, my * pitch * 0.12 + uy * disp * 3.0   # scene coords

    def tex(ax, ay, r):
        t = np.zeros_like(ax)
        for _ in range(4):
            f = r.uniform(0.02, 0.35, 2)
            ph = r.uniform(0, 2 * np.pi, 2)
            t += r.uniform(0.4, 1.0) * np.sin(ax * f[0] + ph[0]) * np.cos(ay * f[1] + ph[1])
        return t / 4.0
    maxv = (1 << bitdepth) - 1
    vign = np.exp(-(ux ** 2 + uy ** 2) / (2 * (0.55 * pitch) ** 2))
    y = (0.5 + 0.45 * tex(sx, sy, rng)) * vign * maxv + rng.normal(0, 2.0 * (1 << (bitdepth - 8)), (H, W))
    Y = np.clip(np.rint(y), 0, maxv).astype(np.int16)
    cxx, cyy = sx[::2, ::2], sy[::2, ::2]
    mid = 1 << (bitdepth - 1)
    Cb = np.clip(np.rint(mid + 0.25 * maxv * tex(cxx, cyy, rng) * vign[::2, ::2]), 0, maxv).astype(np.int16)
    Cr = np.clip(np.rint(mid + 0.25 * maxv * tex(cxx, cyy, rng) * vign[::2, ::2]), 0, maxv).astype(np.int16)
    return Y, Cb, Cr


def sharp_frame(W, H, seed):
    """lenslet with isolated black / white samples and one-sample lines on top: content on which the 4x4 transform-skip variant wins
    (the plain lenslets never choose it)"""
    Y, Cb, Cr = lenslet(W, H, 16, seed)
    rng = np.random.default_rng(seed + 1000)
    Y, Cb, Cr = Y.copy(), Cb.copy(), Cr.copy()
    m = rng.random((H, W)) < 0.08
    Y[m] = rng.choice([0, 255], size=int(m.sum()))
    Y[10:H - 10:7, :] = 255 - Y[10:H - 10:7, :]
    mc = rng.random((H // 2, W // 2)) < 0.08
    Cb[mc] = rng.choice([16, 240], size=int(mc.sum()))
    Cr[mc] = rng.choice([16, 240], size=int(mc.sum()))
    return Y, Cb, Cr


class Planes:
    """SS-reference planes in the reference's own layout: margins 80/40, stride = W + 160 / W/2 + 80."""

    def __init__(self, W, H):
        self.W, self.H = W, H
        self.sy, self.sc = W + 2 * MARGIN_Y, (W >> 1) + 2 * MARGIN_C
        # the planes are views into larger arrays with GUARD sentinel rows above and below, as the library keeps them (HOP_GUARD_ROWS): the GT search of a PU at the
        # picture's top or bottom edge reads a row beyond the reference's own allocation (the reference reads whatever the heap holds there; its results do not depend on it)
        G = 64
        self._backY = np.full((H + 2 * MARGIN_Y + 2 * G, self.sy), -1, np.int16); self.bufY = self._backY[G:-G]
        self._backCb = np.full(((H >> 1) + 2 * MARGIN_C + 2 * G, self.sc), -1, np.int16); self.bufCb = self._backCb[G:-G]
        self._backCr = np.full(((H >> 1) + 2 * MARGIN_C + 2 * G, self.sc), -1, np.int16); self.bufCr = self._backCr[G:-G]

    def y00(self):
        return self.bufY[MARGIN_Y:, MARGIN_Y:]

    def ptr00(self, comp):
        b, m = ((self.bufY, MARGIN_Y), (self.bufCb, MARGIN_C), (self.bufCr, MARGIN_C))[comp]
        st = b.shape[1]
        return ctypes.cast(b.ctypes.data + (m * st + m) * 2, I16P)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "hop_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libhop_oracle.so"], stdout=subprocess.DEVNULL)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        for n in ("hop_o_sad", "hop_o_sse", "hop_o_hads", "hop_o_calc_had", "hop_o_component_bits", "hop_o_bits_gt", "hop_o_frac_search"):
            getattr(L, n).restype = ctypes.c_uint32
        _oracle = L
    return _oracle


_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref(bitdepth=8, use_had=1, fen=1, search_range=128):
    """The reference-derived harness (oracle/_ref); one configuration per process."""
    global _ref
    if _ref is None:
        L = ctypes.CDLL(REF_SO)
        L.ref_init(bitdepth, bitdepth, use_had, fen, search_range)
        L.ref_set_lambda.argtypes = [ctypes.c_double]
        L.ref_lambda_motion_sad.restype = ctypes.c_uint
        L.ref_pic_plane.restype = I16P
        for n in ("ref_sad", "ref_hads", "ref_sse", "ref_calc_had", "ref_component_bits", "ref_bits_gt"):
            getattr(L, n).restype = ctypes.c_uint
        _ref = L
    return _ref


def lambda_for_qp(qp):
    """TLibEncoder/TEncSlice.cpp:381-395 for an all-intra (ISS) slice at depth 0, no QP offsets:
    lambda = 0.57 * 2^((qp-12)/3); returns (lambda, m_uiLambdaMotionSAD) (TComRdCost.cpp:167-173)."""
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    return lam, int(np.floor(65536.0 * np.sqrt(lam)))


def ref_load_planes(L, pl):
    """Copy a Planes object into the harness picture (same layout, whole padded buffers)."""
    st = ctypes.c_int()
    stride = L.ref_pic_create(pl.W, pl.H)
    assert stride == pl.sy
    for comp, (buf, m) in enumerate(((pl.bufY, MARGIN_Y), (pl.bufCb, MARGIN_C), (pl.bufCr, MARGIN_C))):
        p = L.ref_pic_plane(comp, ctypes.byref(st))
        assert st.value == buf.shape[1]
        base = ctypes.addressof(p.contents) - (m * st.value + m) * 2
        ctypes.memmove(base, buf.ctypes.data, buf.nbytes)


# The HOP all-intra configuration (the settings cfg/3DHencoder_intra_main.cfg selects) as command-line options of the encoder application, so that a program built from
# the reference can be run where the reference tree itself is absent (the GPU box).  tests/test_encoder_pic.py checks here that they give the configuration file's bitstream.
HOP_ENCODER_OPTIONS = {
    "MaxCUWidth": 64, "MaxCUHeight": 64, "MaxPartitionDepth": 4, "QuadtreeTULog2MaxSize": 5, "QuadtreeTULog2MinSize": 2, "QuadtreeTUMaxDepthInter": 3, "QuadtreeTUMaxDepthIntra": 3,
    "IntraPeriod": 1, "HoloscopicIntra": 1, "MIMergeCand": 1, "DecodingRefreshType": 0, "GOPSize": 1,
    "FastSearch": 0, "SearchRange": 128, "HadamardME": 1, "FEN": 1, "FDM": 1,
    "MaxDeltaQP": 0, "MaxCuDQPDepth": 0, "DeltaQpRD": 0, "RDOQ": 1, "RDOQTS": 1,
    "DeblockingFilterControlPresent": 0, "LoopFilterOffsetInPPS": 0, "LoopFilterDisable": 0, "LoopFilterBetaOffset_div2": 0, "LoopFilterTcOffset_div2": 0,
    "InternalBitDepth": 8, "SAO": 1, "AMP": 1, "TransformSkip": 1, "TransformSkipFast": 1, "SAOLcuBoundary": 0,
    "SliceMode": 0, "SliceArgument": 1500, "LFCrossSliceBoundaryFlag": 1, "PCMEnabledFlag": 0,
    "UniformSpacingIdc": 0, "NumTileColumnsMinus1": 0, "NumTileRowsMinus1": 0, "LFCrossTileBoundaryFlag": 1, "WaveFrontSynchro": 0,
    "ScalingList": 0, "TransquantBypassEnableFlag": 0,
}


def hop_encoder_args(W, H, qp=32, mi=16, **over):
    o = dict(HOP_ENCODER_OPTIONS, **over)
    return ["--%s=%s" % kv for kv in o.items()] + ["-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", str(qp), "--MIsize=%d" % mi,
                                                   "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"]


def plain_encoder_args(W, H, qp, bit_depth, **over):
    """the plain HM intra configurations (cfg/encoder_intra_main.cfg, encoder_intra_main10.cfg: I slices, no SS / GT search) as options: what differs from the HOP list"""
    o = dict(HOP_ENCODER_OPTIONS, **over)
    for k in ("HoloscopicIntra", "MIMergeCand"): del o[k]
    o.update({"FastSearch": 1, "SearchRange": 64, "InternalBitDepth": bit_depth, "Profile": "main" if bit_depth == 8 else "main10"})
    return ["--%s=%s" % kv for kv in o.items()] + ["-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", str(qp), "--InputBitDepth=%d" % bit_depth,
                                                   "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"]


def pic_case_args(c):
    if "plain" in c:
        return plain_encoder_args(c["W"], c["H"], c["plain"][1], c["plain"][0], **c["over"]) + c["extra"]
    return hop_encoder_args(c["W"], c["H"], **c["over"]) + c["extra"]


# the pictures of the picture-level binding tests (golden: tests/golden/encoder_hop_pic.json, made by oracle/make_golden21.py with the unmodified reference encoder)
PIC_CASES = {
    "64x64_raster":    {"W": 64, "H": 64, "seed": 1234, "frames": 1, "over": {}, "extra": []},
    "192x128_raster":  {"W": 192, "H": 128, "seed": 7, "frames": 1, "over": {}, "extra": []},
    "192x128_wpp":     {"W": 192, "H": 128, "seed": 7, "frames": 1, "over": {"WaveFrontSynchro": 1, "WaveFrontSubstreams": 2}, "extra": []},
    "200x104_raster":  {"W": 200, "H": 104, "seed": 11, "frames": 1, "over": {}, "extra": []},                 # neither dimension a multiple of the CTU: partial CTUs right and below
    "448x192_wpp":     {"W": 448, "H": 192, "seed": 3, "frames": 1, "over": {"WaveFrontSynchro": 1, "WaveFrontSubstreams": 3}, "extra": []},
    "128x64_2frames":  {"W": 128, "H": 64, "seed": 5, "frames": 2, "over": {}, "extra": ["-f", "2"]},          # the context and the binding's buffers reused for a second picture
    # the plain HM intra configurations (BASELINE configs 1 and 4): I slices, 8 bit QP 32 and 10 bit QP 27, a picture with partial CTUs
    "136x72_plain8":   {"W": 136, "H": 72, "seed": 9, "frames": 1, "over": {}, "extra": [], "plain": (8, 32)},
    "136x72_plain10":  {"W": 136, "H": 72, "seed": 9, "frames": 1, "over": {}, "extra": [], "plain": (10, 27)},
    # the configuration bench.py measures: pitch-15 lenslets, --MIsize=15 (golden: tests/golden/encoder_hop_pic_mi15.json, oracle/make_golden24.py)
    "200x136_seed5_mi15":     {"W": 200, "H": 136, "seed": 5, "frames": 1, "pitch": 15, "over": {"mi": 15}, "extra": []},
    "192x128_seed7_mi15_wpp": {"W": 192, "H": 128, "seed": 7, "frames": 1, "pitch": 15, "over": {"mi": 15, "WaveFrontSynchro": 1, "WaveFrontSubstreams": 2}, "extra": []},
    "448x192_seed3_mi15_wpp": {"W": 448, "H": 192, "seed": 3, "frames": 1, "pitch": 15, "over": {"mi": 15, "WaveFrontSynchro": 1, "WaveFrontSubstreams": 3}, "extra": []},
}


def pic_case_input(c):
    raw = b""
    bd = c["plain"][0] if "plain" in c else 8
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    for f in range(c["frames"]):
        Y, Cb, Cr = lenslet(c["W"], c["H"], c.get("pitch", 16), c["seed"] + 100 * f, bitdepth=bd)
        raw += Y.astype(dt).tobytes() + Cb.astype(dt).tobytes() + Cr.astype(dt).tobytes()
    return raw


# ---- deblocking: fixtures from the reference's own loop filter (tests/golden/deblock_ref.npz, oracle/make_golden22.py) and random pictures ----
def deblock_cases():
    G = np.load(os.path.join(ROOT, "tests", "golden", "deblock_ref.npz"))
    keys = sorted(set(k.split("/")[0] for k in G.files))
    out = []
    for k in keys:
        W, H, qp, beta, tc, cbo, cro = [int(v) for v in G[k + "/geo"]]
        planes_in = [G[k + "/in_" + n].astype(np.int16).reshape((H, W) if n == "y" else (H // 2, W // 2)) for n in ("y", "cb", "cr")]
        planes_out = [G[k + "/out_" + n].astype(np.int16).reshape((H, W) if n == "y" else (H // 2, W // 2)) for n in ("y", "cb", "cr")]
        out.append((k, W, H, (qp, beta, tc, cbo, cro), np.ascontiguousarray(G[k + "/parts"]), planes_in, planes_out, int(G[k + "/bd"])))
    return out


def oracle_deblock(W, H, params, parts, planes, bit_depth=8, disable=0):
    """hop_o_deblock_frame on copies of the planes; parts: (n_ctu, 256, 44) uint8 or the structured array of hop_cu_part"""
    O = oracle()
    p = [np.ascontiguousarray(a, np.int16).copy() for a in planes]
    parts = np.ascontiguousarray(parts)
    O.hop_o_deblock_frame.argtypes = [ctypes.c_int] * 9 + [ctypes.c_void_p] * 4
    assert O.hop_o_deblock_frame(W, H, bit_depth, *[int(v) for v in params], disable, parts.ctypes.data, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data) == 0
    return p


def tile_deblock_case(case, nx, ny):
    """a picture of nx x ny copies of a CTU-aligned fixture picture (partition data and planes side by side): the seams are new edges between unrelated CUs"""
    _, W, H, params, parts, pin, _, _ = case
    assert W % 64 == 0 and H % 64 == 0
    wc, hc = W // 64, H // 64
    P = parts.reshape(hc, wc, 256, -1)
    big = np.ascontiguousarray(np.tile(P, (ny, nx, 1, 1))).reshape(ny * hc * nx * wc, 256, -1)
    return W * nx, H * ny, params, big, [np.ascontiguousarray(np.tile(a, (ny, nx))) for a in pin]


# ---- SAO: fixtures from the reference's own SAO encoder (tests/golden/sao_ref.npz, oracle/make_golden23.py) ----
SAO_PARAM_DTYPE = np.dtype([("mode", "i1"), ("type", "i1"), ("aux", "i1"), ("pad", "i1"), ("offset", "i1", 32)])          # hop_sao_param


class SaoParams(ctypes.Structure):                                                                                       # hop_sao_params
    _fields_ = [("lambda", ctypes.c_double * 3), ("enabled", ctypes.c_int32 * 3), ("slice_type", ctypes.c_int32), ("qp", ctypes.c_int32), ("rd_fraction", ctypes.c_uint32)]


def sao_cases():
    G = np.load(os.path.join(ROOT, "tests", "golden", "sao_ref.npz"))
    out = []
    for k in sorted(set(f.split("/")[0] for f in G.files)):
        W, H, n, st, qp, frac = [int(v) for v in G[k + "/geo"]]
        pl = lambda pre: [G[k + "/%s_%s" % (pre, c)].astype(np.int16).reshape((H, W) if c == "y" else (H // 2, W // 2)) for c in ("y", "cb", "cr")]
        out.append({"key": k, "W": W, "H": H, "n": n, "slice_type": st, "qp": qp, "rd_fraction": frac, "bd": int(G[k + "/bd"]), "lambda": [float(v) for v in G[k + "/lambda"]],
                    "org": pl("org"), "in": pl("in"), "out": pl("out"), "stats": np.ascontiguousarray(G[k + "/stats"]), "coded": np.ascontiguousarray(G[k + "/coded"]).view(SAO_PARAM_DTYPE).reshape(n, 3)})
    return out


def sao_params_of(case):
    return SaoParams((ctypes.c_double * 3)(*case["lambda"]), (ctypes.c_int32 * 3)(1, 1, 1), case["slice_type"], case["qp"], case["rd_fraction"])


def same_coded(a, b):
    """two arrays of hop_sao_param say the same thing to the entropy coder: mode; type for new and merge; band position and offsets for new"""
    for x, y in zip(a.reshape(-1), b.reshape(-1)):
        if x["mode"] != y["mode"]: return False
        if x["mode"] != 0 and x["type"] != y["type"]: return False
        if x["mode"] == 1 and (x["aux"] != y["aux"] or not np.array_equal(x["offset"], y["offset"])): return False
    return True
