"""CPU: the C-ABI shared library loads and exports every symbol include/hophip.h declares (no compute
calls here -- there is no GPU in the build container and the library has no CPU path)."""
import ctypes
import os
import re
import subprocess

from hoputil import ROOT


def _lib():
    so = os.path.join(ROOT, "hevc-hop_amd", "libhophip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hevc-hop_amd"), "-j8"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "hophip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(hop_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    L = _lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_host_logic_without_gpu():
    """host-side helpers of the ABI need no device: search-range derivation and bit costs against the oracle"""
    import numpy as np
    from hoputil import oracle
    L, O = _lib(), oracle()
    L.hop_component_bits.restype = ctypes.c_uint32
    for v in range(-2000, 2001):
        assert L.hop_component_bits(v) == O.hop_o_component_bits(v)
    rng = np.random.default_rng(0)
    for _ in range(2000):
        W, H = int(rng.integers(8, 1000)) * 8, int(rng.integers(8, 700)) * 8
        cuS = int(rng.choice([8, 16, 32, 64]))
        cuX, cuY = int(rng.integers(0, W // cuS)) * cuS, int(rng.integers(0, H // cuS)) * cuS
        wctu = (W + 63) // 64
        args = [W, H, cuX, cuY, cuS, (cuY // 64) * wctu + cuX // 64, wctu, int(rng.integers(-600, 600)), int(rng.integers(-600, 600)), 128,
                int(rng.integers(0, cuS // 4 + 1)) * 4 % cuS, int(rng.integers(0, cuS // 4 + 1)) * 4 % (cuS + 4), int(cuY == 0), int(cuX == 0)]
        a, b = (ctypes.c_int * 6)(), (ctypes.c_int * 6)()
        L.hop_set_search_range(*args, a)
        O.hop_o_set_search_range(*args, b)
        assert list(a) == list(b), args
    # without a device the context must refuse loudly (no CPU fallback)
    import torch
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        assert L.hop_ctx_create(ctypes.byref(h), 64, 64, 8, 8, 0) != 0
        L.hop_last_error.restype = ctypes.c_char_p
        assert b"no HIP device" in L.hop_last_error(None) or b"device" in L.hop_last_error(None)


def test_cabac_host_functions_vs_golden():
    """hop_cabac_init / hop_cabac_est_bits are host logic of the library (no device needed): against the vectors made by the
    reference's own TEncSbac (tests/golden/cabac.npz)"""
    import ctypes
    import numpy as np
    from goldutil import load
    L = _lib()
    L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.hop_cabac_est_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    g = load("cabac.npz")
    for st in range(5):
        for qp in range(52):
            b = np.full(152, 0xEE, np.uint8)
            assert L.hop_cabac_init(b.ctypes.data, st, qp) == 0
            assert np.array_equal(b[:150], g["init"][st, qp]) and b[150] == 0 and b[151] == 0
    assert L.hop_cabac_init(b.ctypes.data, 5, 30) != 0
    for st, (w, comp), want in zip(g["est_states"], g["est_par"], g["est_out"]):
        s152 = np.zeros(152, np.uint8); s152[:150] = st
        e = np.full(244, 0x5A5A, np.int32)
        assert L.hop_cabac_est_bits(s152.ctypes.data, int(w), int(comp), e.ctypes.data) == 0
        assert np.array_equal(e, want), (w, comp)
    assert L.hop_cabac_est_bits(s152.ctypes.data, 32, 1, e.ctypes.data) != 0     # no chroma 32x32


def test_cabac_cu_init_vs_golden():
    """hop_cabac_cu_init (host logic): the CU-level context sets of hop_cabac_cu_ctx against the states the reference's own ContextModel3DBuffer::initBuffer
    produced for every slice type and QP (tests/golden/cabac_cu.npz, made by oracle/make_golden4.py through ref_cabac_cu_init)"""
    import numpy as np
    from goldutil import load
    L = _lib()
    L.hop_cabac_cu_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    g = load("cabac_cu.npz")["init"]
    for st in range(5):
        for qp in range(52):
            b = np.full(20, 0xEE, np.uint8)
            assert L.hop_cabac_cu_init(b.ctypes.data, st, qp) == 0 and np.array_equal(b[:19], g[st, qp]) and b[19] == 0, (st, qp)
    assert L.hop_cabac_cu_init(b.ctypes.data, 7, 30) != 0
