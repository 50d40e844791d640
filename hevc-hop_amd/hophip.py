"""ctypes binding of libhophip.so (include/hophip.h) for tests/ and bench.py.

This is plumbing, not the product: the product is the shared library and the C++ host mirror in
hevc-hop_amd/host/.  There is no CPU fallback here or in the library -- loading fails loudly when the
library has not been built, and every hot-path call fails with HOP_ERR_DEVICE without a gfx950 GPU.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhophip.so")

HOP_STAGE_INT, HOP_STAGE_FRAC, HOP_STAGE_GT = 1, 2, 3
HOP_FLAG_FEN, HOP_FLAG_HADME = 1, 2
HOP_TU_RD_TS, HOP_TU_RD_KEEP = 1, 2
HOP_DIST_SAD, HOP_DIST_SSE, HOP_DIST_HADS = 0, 1, 2


class PuJob(ctypes.Structure):
    _fields_ = [("pu_x", ctypes.c_int32), ("pu_y", ctypes.c_int32), ("w", ctypes.c_int32), ("h", ctypes.c_int32),
                ("rng_left", ctypes.c_int32), ("rng_right", ctypes.c_int32), ("rng_top", ctypes.c_int32), ("rng_bottom", ctypes.c_int32),
                ("off_x", ctypes.c_int32), ("off_y", ctypes.c_int32), ("pred_x", ctypes.c_int32), ("pred_y", ctypes.c_int32),
                ("lambda_cost", ctypes.c_uint32), ("n_amvp", ctypes.c_int32), ("amvp", ctypes.c_int32 * 4), ("flags", ctypes.c_int32)]


class PuResult(ctypes.Structure):
    _fields_ = [("mv_int", ctypes.c_int32 * 2), ("sad", ctypes.c_uint32), ("not_valid", ctypes.c_int32),
                ("half", ctypes.c_int32 * 2), ("qter", ctypes.c_int32 * 2), ("frac_cost", ctypes.c_uint32),
                ("gt_flag", ctypes.c_int32), ("gt", ctypes.c_int32 * 8), ("cost", ctypes.c_uint32),
                ("mv_final", ctypes.c_int32 * 2), ("half_final", ctypes.c_int32 * 2), ("qter_final", ctypes.c_int32 * 2)]


class PredJob(ctypes.Structure):
    _fields_ = [("pu_x", ctypes.c_int32), ("pu_y", ctypes.c_int32), ("w", ctypes.c_int32), ("h", ctypes.c_int32),
                ("mv_x", ctypes.c_int32), ("mv_y", ctypes.c_int32), ("use_gt", ctypes.c_int32), ("gt", ctypes.c_int32 * 8), ("dst_row_off", ctypes.c_int32)]


class DistJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("w", ctypes.c_int32), ("h", ctypes.c_int32),
                ("comp", ctypes.c_int32), ("kind", ctypes.c_int32)]


class TuJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("comp", ctypes.c_int32), ("log2_size", ctypes.c_int32),
                ("use_dst", ctypes.c_int32), ("transform_skip", ctypes.c_int32), ("qp_scaled", ctypes.c_int32), ("is_i_slice", ctypes.c_int32),
                ("sign_hide", ctypes.c_int32), ("scan_idx", ctypes.c_int32)]


class TuResult(ctypes.Structure):
    _fields_ = [("abs_sum", ctypes.c_uint32), ("sse", ctypes.c_uint32)]


class IntraJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("size", ctypes.c_int32), ("strong", ctypes.c_int32), ("flags", ctypes.c_uint8 * 68)]


RDOQ_JOB_DTYPE = np.dtype([("log2_size", "<i4"), ("comp", "<i4"), ("is_intra", "<i4"), ("scan_idx", "<i4"), ("tr_depth", "<i4"), ("qp_scaled", "<i4"),
                           ("bit_depth", "<i4"), ("sign_hide", "<i4"), ("lambda", "<f8"), ("coeff_offset", "<i8"), ("estbits_index", "<i4"), ("reserved", "<i4")])
ESTBITS_INTS = 4 + 84 + 32 + 32 + 48 + 12 + 24 + 8       # hop_estbits as a flat int32 array
COEFF_BITS_JOB_DTYPE = np.dtype([("log2_size", "<i4"), ("comp", "<i4"), ("scan_idx", "<i4"), ("sign_hide", "<i4"), ("use_ts", "<i4"), ("ts_flag", "<i4"),
                                 ("ctx_index", "<i4"), ("cbf_ctx_plus1", "<i4"), ("coeff_offset", "<i8")])
CABAC_CTX_BYTES = 152
TU_RD_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("comp", "<i4"), ("log2_size", "<i4"), ("qp_scaled", "<i4"), ("tr_depth", "<i4"), ("ctx_index", "<i4"),
                            ("sign_hide", "<i4"), ("use_ts", "<i4"), ("bit_depth", "<i4"), ("is_intra", "<i4"), ("scan_idx", "<i4"), ("use_dst", "<i4"), ("flags", "<i4"),
                            ("lambda_rdoq", "<f8"), ("lambda_rd", "<f8"), ("dist_weight", "<f8")])
RQT_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("log2_cu", "<i4"), ("qp_scaled", "<i4", (3,)), ("ctx_index", "<i4"), ("sign_hide", "<i4"), ("use_ts", "<i4"),
                          ("log2_max_tu", "<i4"), ("log2_min_tu_in_cu", "<i4"), ("inter_split_flag", "<i4"), ("lambda_rd", "<f8"), ("lambda_rdoq", "<f8", (3,)),
                          ("dist_weight", "<f8", (2,))])
RQT_RESULT_DTYPE = np.dtype([("cost", "<f8"), ("bits", "<u4"), ("dist", "<u4"), ("zero_dist", "<u4"), ("pad", "<u4"), ("tr_idx", "u1", (256,)), ("cbf", "u1", (3, 256)),
                             ("tskip", "u1", (3, 256))])
CABAC_CU_CTX_BYTES = 20
CU_SYNTAX_DTYPE = np.dtype([("part_size", "<i4"), ("n_pu", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("amp_acc", "<i4"), ("is_min_cu", "<i4"), ("max_merge_cand", "<i4"),
                            ("pu", [("merge_flag", "<i4"), ("merge_idx", "<i4"), ("mvd", "<i4", (2,)), ("mvp_idx", "<i4"), ("gt_flag", "<i4"), ("gt", "<i4", (8,))], (4,))])
INTRA_MODES_JOB_DTYPE = np.dtype([("preds", "<i4", (3,)), ("pred_num", "<i4"), ("mpm_cand", "<i4"), ("num_full_rd", "<i4"), ("ctx_state", "<i4"), ("frac_left", "<i4"), ("sqrt_lambda", "<f8")])
INTRA_MODES_RESULT_DTYPE = np.dtype([("n", "<u4"), ("modes", "<u4", (11,)), ("costs", "<f8", (8,))])
INTRA_CU_SYNTAX_DTYPE = np.dtype([("part_nxn", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("is_min_cu", "<i4"), ("luma_dir", "<i4", (4,)), ("preds", "<i4", (4, 3)),
                                  ("pred_num", "<i4", (4,)), ("chroma_is_dm", "<i4"), ("chroma_dir", "<i4"), ("tr_depth", "<i4"), ("part", "<i4"), ("b_luma", "<i4"), ("b_chroma", "<i4")])
INTRA_RQT_OPT_DTYPE = np.dtype([("check_first", "<i4"), ("ts_fast", "<i4"), ("strong", "<i4"), ("pad", "<i4"), ("avail", "<u8", (341,))])   # hop_intra_rqt_opt
INTRA_SEARCH_JOB_DTYPE = np.dtype([("left_dir", "<i4", (4,)), ("above_dir", "<i4", (4,)), ("rough_flags", "u1", (4, 68)), ("sqrt_lambda", "<f8"), ("num_full_rd", "<i4"),
                                   ("pad", "<i4")])                                                   # hop_intra_search_job
INTRA_SEARCH_RESULT_DTYPE = np.dtype([("best_dir", "<i4", (4,)), ("n_cand", "<i4", (4,)), ("dist", "<u4"), ("pad", "<u4")])
TU_RD_RESULT_DTYPE = np.dtype([("abs_sum", "<u4"), ("cbf", "<u4"), ("dist", "<u4"), ("zero_dist", "<u4"), ("nonzero_dist", "<u4"), ("bits", "<u4"),
                               ("null_bits", "<u4"), ("pad", "<u4"), ("cost", "<f8")])
TU_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("comp", "<i4"), ("log2_size", "<i4"), ("use_dst", "<i4"), ("transform_skip", "<i4"),
                         ("qp_scaled", "<i4"), ("is_i_slice", "<i4"), ("sign_hide", "<i4"), ("scan_idx", "<i4")])
INTRA_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("size", "<i4"), ("strong", "<i4"), ("flags", "u1", (68,))])
assert TU_JOB_DTYPE.itemsize == ctypes.sizeof(TuJob) and INTRA_JOB_DTYPE.itemsize == ctypes.sizeof(IntraJob)
PU_JOB_DTYPE = np.dtype([("pu_x", "<i4"), ("pu_y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("rng_left", "<i4"), ("rng_right", "<i4"),
                         ("rng_top", "<i4"), ("rng_bottom", "<i4"), ("off_x", "<i4"), ("off_y", "<i4"), ("pred_x", "<i4"), ("pred_y", "<i4"),
                         ("lambda_cost", "<u4"), ("n_amvp", "<i4"), ("amvp", "<i4", (4,)), ("flags", "<i4")])
PU_RESULT_DTYPE = np.dtype([("mv_int", "<i4", (2,)), ("sad", "<u4"), ("not_valid", "<i4"), ("half", "<i4", (2,)), ("qter", "<i4", (2,)),
                            ("frac_cost", "<u4"), ("gt_flag", "<i4"), ("gt", "<i4", (8,)), ("cost", "<u4"), ("mv_final", "<i4", (2,)),
                            ("half_final", "<i4", (2,)), ("qter_final", "<i4", (2,))])
assert PU_JOB_DTYPE.itemsize == ctypes.sizeof(PuJob) and PU_RESULT_DTYPE.itemsize == ctypes.sizeof(PuResult)

CU_PART_DTYPE = np.dtype([("depth", "u1"), ("pred_mode", "u1"), ("part_size", "u1"), ("skip", "u1"), ("merge_flag", "u1"), ("merge_idx", "u1"), ("gt_flag", "u1"), ("inter_dir", "u1"),
                          ("ref_idx", "i1"), ("mvp_idx", "i1"), ("mvp_num", "i1"), ("luma_dir", "u1"), ("chroma_dir", "u1"), ("tr_idx", "u1"), ("cbf", "u1", 3), ("tskip", "u1", 3),
                          ("mv", "<i2", 2), ("mvd", "<i2", 2), ("gt", "<i2", 8)])                           # hop_cu_part


class EncParams(ctypes.Structure):       # hop_enc_params
    _fields_ = [("qp", ctypes.c_int32), ("mi_size", ctypes.c_int32), ("first_ctus", ctypes.c_int32), ("wpp", ctypes.c_int32), ("wavefront_lag", ctypes.c_int32),
                ("plain_intra", ctypes.c_int32), ("streams", ctypes.c_int32), ("trace_path", ctypes.c_char_p)]


SAO_PARAM_DTYPE = np.dtype([("mode", "i1"), ("type", "i1"), ("aux", "i1"), ("pad", "i1"), ("offset", "i1", 32)])          # hop_sao_param


class SaoParams(ctypes.Structure):       # hop_sao_params
    _fields_ = [("lambda", ctypes.c_double * 3), ("enabled", ctypes.c_int32 * 3), ("slice_type", ctypes.c_int32), ("qp", ctypes.c_int32), ("rd_fraction", ctypes.c_uint32)]


class DeblockParams(ctypes.Structure):   # hop_deblock_params
    _fields_ = [("qp", ctypes.c_int32), ("beta_offset_div2", ctypes.c_int32), ("tc_offset_div2", ctypes.c_int32), ("cb_qp_offset", ctypes.c_int32), ("cr_qp_offset", ctypes.c_int32),
                ("disable", ctypes.c_int32)]


_I16P = ctypes.POINTER(ctypes.c_int16)


class HopError(RuntimeError):
    pass


# every mirror of a hophip.h struct kept in this file, by the C name hop_sizeof() knows it under; load() refuses a library whose structs have another size
MIRRORS = {"hop_pu_job": PU_JOB_DTYPE, "hop_pu_result": PU_RESULT_DTYPE, "hop_pred_job": PredJob, "hop_dist_job": DistJob, "hop_tu_job": TU_JOB_DTYPE, "hop_tu_result": TuResult,
           "hop_intra_job": INTRA_JOB_DTYPE, "hop_rdoq_job": RDOQ_JOB_DTYPE, "hop_coeff_bits_job": COEFF_BITS_JOB_DTYPE, "hop_tu_rd_job": TU_RD_JOB_DTYPE,
           "hop_tu_rd_result": TU_RD_RESULT_DTYPE, "hop_intra_modes_job": INTRA_MODES_JOB_DTYPE, "hop_intra_modes_result": INTRA_MODES_RESULT_DTYPE, "hop_rqt_job": RQT_JOB_DTYPE,
           "hop_rqt_result": RQT_RESULT_DTYPE, "hop_cu_syntax": CU_SYNTAX_DTYPE, "hop_intra_cu_syntax": INTRA_CU_SYNTAX_DTYPE, "hop_intra_rqt_opt": INTRA_RQT_OPT_DTYPE,
           "hop_intra_search_job": INTRA_SEARCH_JOB_DTYPE, "hop_intra_search_result": INTRA_SEARCH_RESULT_DTYPE, "hop_cu_part": CU_PART_DTYPE, "hop_enc_params": EncParams,
           "hop_deblock_params": DeblockParams, "hop_sao_param": SAO_PARAM_DTYPE, "hop_sao_params": SaoParams}


def mirror_size(m):
    return m.itemsize if isinstance(m, np.dtype) else ctypes.sizeof(m)


def load():
    global LIB_PATH
    if os.environ.get("HOP_LIB"): LIB_PATH = os.environ["HOP_LIB"]      # (development: a differently built library)
    if not os.path.exists(LIB_PATH):
        raise HopError("libhophip.so is not built (run `make -C hevc-hop_amd` or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    L.hop_sizeof.argtypes = [ctypes.c_char_p]
    for name, m in MIRRORS.items():
        if L.hop_sizeof(name.encode()) != mirror_size(m):
            raise HopError("%s: this binding lays the struct out in %d bytes, libhophip.so in %d" % (name, mirror_size(m), L.hop_sizeof(name.encode())))
    L.hop_last_error.restype = ctypes.c_char_p
    L.hop_last_error.argtypes = [ctypes.c_void_p]
    L.hop_version.restype = ctypes.c_char_p
    L.hop_stream.restype = ctypes.c_void_p
    L.hop_stream.argtypes = [ctypes.c_void_p]
    L.hop_component_bits.restype = ctypes.c_uint32
    L.hop_bits_gt.restype = ctypes.c_uint32
    for n in ("hop_ctx_destroy", "hop_sync", "hop_ssref_reset"):
        getattr(L, n).argtypes = [ctypes.c_void_p]
    L.hop_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int] * 5
    L.hop_upload_orig.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hop_ssref_commit_cus.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.hop_ssref_commit_cus_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.hop_ssref_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.hop_ssref_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.hop_pred_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.hop_me_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hop_me_search_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hop_pred_inter.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.hop_pred_inter_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.hop_distortion.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_pred_jobs_from_results_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.hop_tu_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_intra_rough.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_rdoq.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 3
    L.hop_rdoq_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
    L.hop_rdoq.restype = L.hop_rdoq_device.restype = ctypes.c_int
    L.hop_tu_rd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    L.hop_tu_rd_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 2
    L.hop_tu_rd.restype = L.hop_tu_rd_device.restype = ctypes.c_int
    L.hop_intra_pred.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_intra_pred_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_intra_pred.restype = L.hop_intra_pred_device.restype = ctypes.c_int
    L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.hop_cabac_est_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.hop_coeff_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 3
    L.hop_coeff_bits_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
    for n in ("hop_cabac_init", "hop_cabac_est_bits", "hop_coeff_bits", "hop_coeff_bits_device"):
        getattr(L, n).restype = ctypes.c_int
    L.hop_tu_roundtrip_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.hop_intra_rough_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_distortion_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for n in ("hop_tu_roundtrip_device", "hop_intra_rough_device", "hop_distortion_device"):
        getattr(L, n).restype = ctypes.c_int
    for n in ("hop_recon_upload", "hop_recon_download", "hop_pred_upload"):
        getattr(L, n).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        getattr(L, n).restype = ctypes.c_int
    L.hop_tu_roundtrip.restype = ctypes.c_int
    L.hop_intra_rough.restype = ctypes.c_int
    L.hop_enumerate_ctu_jobs.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hop_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hop_profile_reset.argtypes = [ctypes.c_void_p]
    L.hop_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for n in ("hop_ctx_create", "hop_sync", "hop_ssref_reset", "hop_upload_orig", "hop_ssref_commit_cus", "hop_ssref_commit_cus_device",
              "hop_ssref_download", "hop_ssref_upload", "hop_pred_download", "hop_me_search", "hop_me_search_device", "hop_pred_inter",
              "hop_pred_inter_device", "hop_distortion", "hop_pred_jobs_from_results_device", "hop_enumerate_ctu_jobs",
              "hop_profile_enable", "hop_profile_reset", "hop_profile_read"):
        getattr(L, n).restype = ctypes.c_int
    L.hop_set_search_range.argtypes = [ctypes.c_int] * 14 + [ctypes.POINTER(ctypes.c_int)]
    L.hop_me_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32,
                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    return L


class Context:
    """Thin RAII wrapper over hop_ctx_* for the tests and the bench."""

    def __init__(self, pic_w, pic_h, bit_depth=8, device=0, lib=None, pictures=1, slots=0):
        """pictures > 1: a stacked context (hop_ctx_set_stack) of that many independent pic_w x pic_h pictures; self.H is the height of the stack, self.sub_h / self.pitch
        the pictures' height and spacing, picture k at rows k * pitch."""
        self.L = lib or load()
        self.h = ctypes.c_void_p()
        self.pictures, self.sub_h, self.pitch = pictures, pic_h, 0
        if pictures > 1:
            self.pitch = self.L.hop_stack_pitch(pic_h)
            pic_h = (pictures - 1) * self.pitch + pic_h
        self.W, self.H = pic_w, pic_h
        r = self.L.hop_ctx_create(ctypes.byref(self.h), pic_w, pic_h, bit_depth, bit_depth, device)
        if r != 0:
            raise HopError("hop_ctx_create: %d %s" % (r, self.L.hop_last_error(None).decode()))
        if pictures > 1:
            self.L.hop_ctx_set_stack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            self._chk(self.L.hop_ctx_set_stack(self.h, self.sub_h, self.pitch), "hop_ctx_set_stack")
        self.slots = slots
        if slots:                     # candidate slots (hop_ctx_set_slots): hop_encode_frame evaluates the SS/GT candidates of a CU side by side
            self.L.hop_ctx_set_slots.argtypes = [ctypes.c_void_p, ctypes.c_int]
            self._chk(self.L.hop_ctx_set_slots(self.h, slots), "hop_ctx_set_slots")

    def stack(self, planes, chroma=False):
        """the pictures' planes (a list of arrays) laid out as the context's stack"""
        sh, pt = (self.sub_h >> 1, self.pitch >> 1) if chroma else (self.sub_h, self.pitch)
        out = np.zeros(((self.pictures - 1) * pt + sh, planes[0].shape[1]), np.int16)
        for k, a in enumerate(planes):
            out[k * pt:k * pt + sh] = a
        return out

    def unstack(self, plane, chroma=False):
        sh, pt = (self.sub_h >> 1, self.pitch >> 1) if chroma else (self.sub_h, self.pitch)
        return [plane[k * pt:k * pt + sh] for k in range(self.pictures)]

    def close(self):
        if self.h:
            self.L.hop_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r, what):
        if r != 0:
            raise HopError("%s: %d %s" % (what, r, self.L.hop_last_error(self.h).decode()))

    def upload_orig(self, Y, Cb, Cr):
        Y, Cb, Cr = (np.ascontiguousarray(a, np.int16) for a in (Y, Cb, Cr))
        self._chk(self.L.hop_upload_orig(self.h, Y.ctypes.data, Y.shape[1], Cb.ctypes.data, Cr.ctypes.data, Cb.shape[1]), "hop_upload_orig")

    def ssref_reset(self):
        self._chk(self.L.hop_ssref_reset(self.h), "hop_ssref_reset")
        self._chk(self.L.hop_sync(self.h), "hop_sync")

    def ssref_commit(self, rects, recY, recCb, recCr):
        r4 = np.zeros((len(rects), 4), np.int32)
        r4[:, :3] = np.asarray(rects, np.int32).reshape(-1, 3)
        recY, recCb, recCr = (np.ascontiguousarray(a, np.int16) for a in (recY, recCb, recCr))
        self._chk(self.L.hop_ssref_commit_cus(self.h, len(rects), r4.ctypes.data, recY.ctypes.data, recCb.ctypes.data, recCr.ctypes.data), "hop_ssref_commit_cus")

    def ssref_download(self, comp):
        shape = (self.H + 160, self.W + 160) if comp == 0 else (self.H // 2 + 80, self.W // 2 + 80)
        a = np.empty(shape, np.int16)
        self._chk(self.L.hop_ssref_download(self.h, comp, a.ctypes.data), "hop_ssref_download")
        return a

    def ssref_upload(self, comp, buf):
        buf = np.ascontiguousarray(buf, np.int16)
        self._chk(self.L.hop_ssref_upload(self.h, comp, buf.ctypes.data), "hop_ssref_upload")

    def me_search(self, jobs, stage=HOP_STAGE_GT):
        jobs = np.ascontiguousarray(jobs, PU_JOB_DTYPE)
        res = np.zeros(len(jobs), PU_RESULT_DTYPE)
        self._chk(self.L.hop_me_search(self.h, len(jobs), jobs.ctypes.data, res.ctypes.data, stage), "hop_me_search")
        return res

    def pred_inter(self, jobs):
        n = len(jobs)
        arr = (PredJob * n)(*jobs)
        tot = sum(j.w * j.h for j in jobs)
        oy, ocb, ocr = np.empty(tot, np.int16), np.empty(tot // 4, np.int16), np.empty(tot // 4, np.int16)
        self._chk(self.L.hop_pred_inter(self.h, n, ctypes.addressof(arr), oy.ctypes.data, ocb.ctypes.data, ocr.ctypes.data), "hop_pred_inter")
        return oy, ocb, ocr

    def pred_download(self, comp):
        shape = (self.H, self.W) if comp == 0 else (self.H // 2, self.W // 2)
        a = np.empty(shape, np.int16)
        self._chk(self.L.hop_pred_download(self.h, comp, a.ctypes.data), "hop_pred_download")
        return a

    def distortion(self, jobs):
        n = len(jobs)
        arr = (DistJob * n)(*jobs)
        out = np.zeros(n, np.uint32)
        self._chk(self.L.hop_distortion(self.h, n, ctypes.addressof(arr), out.ctypes.data), "hop_distortion")
        return out

    def plane_upload(self, what, comp, a):
        a = np.ascontiguousarray(a, np.int16)
        self._chk(getattr(self.L, "hop_%s_upload" % what)(self.h, comp, a.ctypes.data), "hop_%s_upload" % what)

    def recon_download(self, comp):
        shape = (self.H, self.W) if comp == 0 else (self.H // 2, self.W // 2)
        a = np.empty(shape, np.int16)
        self._chk(self.L.hop_recon_download(self.h, comp, a.ctypes.data), "hop_recon_download")
        return a

    def encode_frame(self, qp=32, mi_size=16, first_ctus=0, trace_path=None, wpp=0, wavefront_lag=0, streams=0, plain_intra=0):
        """hop_encode_frame: the RD spine over the kernels for the resident original.  Returns per-CTU (cost, bits, dist), parts (n_ctu, 256) CU_PART_DTYPE, candidates."""
        L = self.L
        L.hop_encode_frame.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_void_p]
        assert L.hop_sizeof_cu_part() == CU_PART_DTYPE.itemsize
        n = ((self.W + 63) // 64) * ((self.sub_h + 63) // 64) * self.pictures      # a stacked context: picture k's CTUs at [k * n / pictures, ...)
        cost = np.zeros(n, np.float64); bits = np.zeros(n, np.uint32); dist = np.zeros(n, np.uint32); parts = np.zeros((n, 256), CU_PART_DTYPE)
        nc = ctypes.c_uint64(0)
        p = EncParams(qp, mi_size, first_ctus, wpp, wavefront_lag, plain_intra, streams, trace_path.encode() if trace_path else None)
        self._chk(L.hop_encode_frame(self.h, ctypes.byref(p), cost.ctypes.data, bits.ctypes.data, dist.ctypes.data, parts.ctypes.data, ctypes.byref(nc)), "hop_encode_frame")
        return cost, bits, dist, parts, int(nc.value)

    def encode_progress(self):
        """hop_encode_progress: CTUs the running (or last) hop_encode_frame has retired; callable from another thread"""
        v = ctypes.c_int64(0)
        self.L.hop_encode_progress.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_encode_progress(self.h, ctypes.byref(v)), "hop_encode_progress")
        return int(v.value)

    def encode_cancel(self):
        """hop_encode_cancel: the running hop_encode_frame (wavefront mode) starts no further CTU and returns what it has"""
        self.L.hop_encode_cancel.argtypes = [ctypes.c_void_p]
        self._chk(self.L.hop_encode_cancel(self.h), "hop_encode_cancel")

    def set_shard(self, rank, world, allgather=None):
        """hop_encode_set_shard: the next encode_frame (wavefront mode) codes the CTU rows r % world == rank of ONE picture and exchanges every step's finished CTUs through
        `allgather` (an object with a ctypes callback .fn of type shard.ALLGATHER_FN, e.g. shard.TorchAllgather); every rank ends with the whole picture"""
        self.L.hop_encode_set_shard.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._shard = allgather                      # (keeps the callback alive)
        fn = ctypes.cast(allgather.fn, ctypes.c_void_p) if allgather is not None else None
        self._chk(self.L.hop_encode_set_shard(self.h, rank, world, fn, None), "hop_encode_set_shard")

    def levels_download(self):
        """hop_levels_download: (n_ctu, 6144) int32 -- per CTU 4096 luma + 1024 Cb + 1024 Cr levels in the reference's TComDataCU layout"""
        n = ((self.W + 63) // 64) * ((self.sub_h + 63) // 64) * self.pictures
        a = np.zeros((n, 6144), np.int32)
        self.L.hop_levels_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_levels_download(self.h, a.ctypes.data), "hop_levels_download")
        return a

    def deblock_frame(self, parts, qp, beta_offset_div2=0, tc_offset_div2=0, cb_qp_offset=0, cr_qp_offset=0, disable=0):
        """hop_deblock_frame: the deblocking filter over the resident reconstruction picture(s), in place; parts as encode_frame returned them"""
        parts = np.ascontiguousarray(parts)
        n = ((self.W + 63) // 64) * ((self.sub_h + 63) // 64) * self.pictures
        assert parts.nbytes == n * 256 * CU_PART_DTYPE.itemsize, (parts.shape, parts.dtype, n)
        p = DeblockParams(qp, beta_offset_div2, tc_offset_div2, cb_qp_offset, cr_qp_offset, disable)
        self.L.hop_deblock_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_deblock_frame(self.h, ctypes.byref(p), parts.ctypes.data), "hop_deblock_frame")

    def _n_ctu(self):
        return ((self.W + 63) // 64) * ((self.sub_h + 63) // 64) * self.pictures

    def sao_stats(self):
        """hop_sao_stats: (n_ctu, 3, 5, 32, 2) int32 -- per CTU, component, type, class: count and sum of (original - reconstruction)"""
        a = np.zeros((self._n_ctu(), 3, 5, 32, 2), np.int32)
        self.L.hop_sao_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_sao_stats(self.h, a.ctypes.data), "hop_sao_stats")
        return a

    def sao_frame(self, lambdas, slice_type, qp, rd_fraction, enabled=(1, 1, 1)):
        """hop_sao_frame: statistics, decision, offsets for the resident picture(s); returns the coded parameters (n_ctu, 3) SAO_PARAM_DTYPE"""
        p = SaoParams((ctypes.c_double * 3)(*lambdas), (ctypes.c_int32 * 3)(*enabled), slice_type, qp, rd_fraction)
        coded = np.zeros((self._n_ctu(), 3), SAO_PARAM_DTYPE)
        self.L.hop_sao_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_sao_frame(self.h, ctypes.byref(p), coded.ctypes.data), "hop_sao_frame")
        return coded

    def sao_apply(self, recon):
        recon = np.ascontiguousarray(recon)
        assert recon.nbytes == self._n_ctu() * 3 * SAO_PARAM_DTYPE.itemsize
        self.L.hop_sao_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_sao_apply(self.h, recon.ctypes.data), "hop_sao_apply")

    def psnr(self):
        """hop_psnr: (ssd (pictures, 3) uint64, psnr (pictures, 3) float64) between the resident original and the reconstruction"""
        ssd = np.zeros((self.pictures, 3), np.uint64); ps = np.zeros((self.pictures, 3), np.float64)
        self.L.hop_psnr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_psnr(self.h, ssd.ctypes.data, ps.ctypes.data), "hop_psnr")
        return ssd, ps

    def rd_fraction_download(self):
        """hop_rd_fraction_download: per CTU the fraction of a bit the RD search's counting coder carries when the CTU is done"""
        n = ((self.W + 63) // 64) * ((self.sub_h + 63) // 64) * self.pictures
        a = np.zeros(n, np.uint16)
        self.L.hop_rd_fraction_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_rd_fraction_download(self.h, a.ctypes.data), "hop_rd_fraction_download")
        return a

    def encode_stats(self):
        ms, calls = (ctypes.c_double * 16)(), (ctypes.c_double * 16)()
        self.L.hop_encode_stats(ms, calls)
        names = ("me_search", "pred_inter", "distortion", "valid_pattern", "inter_cu", "inter_cu_skip", "intra_cu", "recon_stash", "commit", "evaluation_wait")
        d = {k: {"ms": ms[i], "calls": int(calls[i])} for i, k in enumerate(names)}
        d["rendezvous"] = {"rounds": int(calls[14]), "requests": int(calls[15]), "serve_ms": float(ms[14]), "run_ms": float(ms[13]),
                           "searches_reaching_below": int(calls[10]), "first_ctu_reaching_below": int(ms[10]), "searches_reaching_above": int(calls[11]), "first_ctu_reaching_above": int(ms[11])}
        return d

    def tu_roundtrip(self, jobs, want_levels=True):
        n = len(jobs)
        arr = (TuJob * n)(*jobs)
        res = (TuResult * n)()
        tot = sum((1 << j.log2_size) ** 2 for j in jobs)
        lv = np.zeros(tot, np.int32) if want_levels else None
        self._chk(self.L.hop_tu_roundtrip(self.h, n, ctypes.addressof(arr), ctypes.addressof(res), lv.ctypes.data if want_levels else None), "hop_tu_roundtrip")
        return [(r.abs_sum, r.sse) for r in res], lv

    def intra_rough(self, jobs):
        n = len(jobs)
        arr = (IntraJob * n)(*jobs)
        out = np.zeros((n, 35), np.uint32)
        self._chk(self.L.hop_intra_rough(self.h, n, ctypes.addressof(arr), out.ctypes.data), "hop_intra_rough")
        return out

    def rdoq(self, jobs, tables, src):
        """jobs: RDOQ_JOB_DTYPE array; tables: (n_tables, ESTBITS_INTS) int32; src: all coefficients (int32) -> levels, abs_sum"""
        jobs = np.ascontiguousarray(jobs, RDOQ_JOB_DTYPE); tables = np.ascontiguousarray(tables, np.int32); src = np.ascontiguousarray(src, np.int32)
        assert tables.ndim == 2 and tables.shape[1] == ESTBITS_INTS
        dst = np.zeros(len(src), np.int32); asum = np.zeros(len(jobs), np.uint32)
        self._chk(self.L.hop_rdoq(self.h, len(jobs), jobs.ctypes.data, len(tables), tables.ctypes.data, len(src), src.ctypes.data, dst.ctypes.data,
                                  asum.ctypes.data), "hop_rdoq")
        return dst, asum

    def coeff_bits(self, jobs, ctx_in, coef, want_ctx=True):
        """jobs: COEFF_BITS_JOB_DTYPE; ctx_in: (n_ctx, 152) uint8 snapshots; coef: all levels (int32) -> bits (uint64, Q15), ctx after each TU"""
        jobs = np.ascontiguousarray(jobs, COEFF_BITS_JOB_DTYPE); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); coef = np.ascontiguousarray(coef, np.int32)
        assert ctx_in.ndim == 2 and ctx_in.shape[1] == CABAC_CTX_BYTES
        bits = np.zeros(len(jobs), np.uint64); out = np.zeros((len(jobs), CABAC_CTX_BYTES), np.uint8) if want_ctx else None
        self._chk(self.L.hop_coeff_bits(self.h, len(jobs), jobs.ctypes.data, len(ctx_in), ctx_in.ctypes.data, len(coef), coef.ctypes.data, bits.ctypes.data,
                                        out.ctypes.data if want_ctx else None), "hop_coeff_bits")
        return bits, out

    def tu_rd(self, jobs, ctx_in):
        """jobs: TU_RD_JOB_DTYPE; ctx_in: (n_ctx, 152) uint8 -> results (TU_RD_RESULT_DTYPE), levels (all TUs, job order)"""
        jobs = np.ascontiguousarray(jobs, TU_RD_JOB_DTYPE); ctx_in = np.ascontiguousarray(ctx_in, np.uint8)
        res = np.zeros(len(jobs), TU_RD_RESULT_DTYPE); lv = np.zeros(int(np.sum(1 << (2 * jobs["log2_size"].astype(np.int64)))), np.int32)
        self._chk(self.L.hop_tu_rd(self.h, len(jobs), jobs.ctypes.data, len(ctx_in), ctx_in.ctypes.data, res.ctypes.data, lv.ctypes.data), "hop_tu_rd")
        return res, lv

    def rqt(self, jobs, ctx_in):
        """jobs: RQT_JOB_DTYPE; ctx_in: (n_ctx, 152) uint8 -> results (RQT_RESULT_DTYPE), chosen levels (1.5 * size^2 per CU, job order), coder states out"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); ctx_in = np.ascontiguousarray(ctx_in, np.uint8)
        res = np.zeros(len(jobs), RQT_RESULT_DTYPE); co = np.zeros(int(np.sum(3 << (2 * jobs["log2_cu"].astype(np.int64) - 1))), np.int32)
        cx = np.zeros((len(jobs), CABAC_CTX_BYTES), np.uint8)
        self.L.hop_rqt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        self._chk(self.L.hop_rqt(self.h, len(jobs), jobs.ctypes.data, len(ctx_in), ctx_in.ctypes.data, res.ctypes.data, co.ctypes.data, cx.ctypes.data), "hop_rqt")
        return res, co, cx

    def rqt_finish(self, jobs, res, coef, ctx_after):
        """the tail of encodeResAndCalcRdInterCU on the outputs of rqt(): returns (results, levels) updated in place where the zero residual
        wins, and per CU (root_cbf, dist Y, Cb, Cr); the reconstruction is in the context's reconstruction picture"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); res = np.ascontiguousarray(res, RQT_RESULT_DTYPE).copy(); coef = np.ascontiguousarray(coef, np.int32).copy()
        cx = np.ascontiguousarray(ctx_after, np.uint8); fin = np.zeros((len(jobs), 4), np.uint32)
        self.L.hop_rqt_finish.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
        self._chk(self.L.hop_rqt_finish(self.h, len(jobs), jobs.ctypes.data, res.ctypes.data, coef.ctypes.data, cx.ctypes.data, fin.ctypes.data), "hop_rqt_finish")
        return res, coef, fin

    def inter_cu_bits(self, jobs, syntax, res, coef, ctx_in, cu_ctx_in):
        """CU-level syntax bits (xAddSymbolBitsInter): returns bits, skipped, coder states (n, 152) and CU-level states (n, 20) afterwards"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, CU_SYNTAX_DTYPE); res = np.ascontiguousarray(res, RQT_RESULT_DTYPE)
        coef = np.ascontiguousarray(coef, np.int32); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); bits = np.zeros(n, np.uint32); sk = np.zeros(n, np.uint32); cx = np.zeros((n, CABAC_CTX_BYTES), np.uint8); cu = np.zeros((n, CABAC_CU_CTX_BYTES), np.uint8)
        self.L.hop_inter_cu_bits.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 6
        self._chk(self.L.hop_inter_cu_bits(self.h, n, jobs.ctypes.data, syntax.ctypes.data, res.ctypes.data, coef.ctypes.data, len(ctx_in), ctx_in.ctypes.data, cu_ctx_in.ctypes.data,
                                           bits.ctypes.data, sk.ctypes.data, cx.ctypes.data, cu.ctypes.data), "hop_inter_cu_bits")
        return bits, sk, cx, cu

    def intra_modes(self, jobs, satd):
        jobs = np.ascontiguousarray(jobs, INTRA_MODES_JOB_DTYPE); satd = np.ascontiguousarray(satd, np.uint32); res = np.zeros(len(jobs), INTRA_MODES_RESULT_DTYPE)
        self.L.hop_intra_modes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_intra_modes(self.h, len(jobs), jobs.ctypes.data, satd.ctypes.data, res.ctypes.data), "hop_intra_modes")
        return res

    def intra_cu_bits(self, jobs, syntax, res, coef, ctx_in, cu_ctx_in):
        """xGetIntraBitsQT: returns bits, coder states (n, 152) and CU-level states (n, 20) afterwards"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, INTRA_CU_SYNTAX_DTYPE); res = np.ascontiguousarray(res, RQT_RESULT_DTYPE)
        coef = np.ascontiguousarray(coef, np.int32); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); bits = np.zeros(n, np.uint32); cx = np.zeros((n, CABAC_CTX_BYTES), np.uint8); cu = np.zeros((n, CABAC_CU_CTX_BYTES), np.uint8)
        self.L.hop_intra_cu_bits.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 5
        self._chk(self.L.hop_intra_cu_bits(self.h, n, jobs.ctypes.data, syntax.ctypes.data, res.ctypes.data, coef.ctypes.data, len(ctx_in), ctx_in.ctypes.data, cu_ctx_in.ctypes.data,
                                           bits.ctypes.data, cx.ctypes.data, cu.ctypes.data), "hop_intra_cu_bits")
        return bits, cx, cu

    def intra_rqt(self, jobs, syntax, opts, ctx_in, cu_ctx_in):
        """xRecurIntraCodingQT (luma): returns results (RQT_RESULT_DTYPE), chosen levels (1.5 size^2 per job, the PU's luma partitions filled), coder states (n, 152)
        and CU-level states (n, 20) afterwards; the context's reconstruction picture holds the chosen trees"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, INTRA_CU_SYNTAX_DTYPE); opts = np.ascontiguousarray(opts, INTRA_RQT_OPT_DTYPE)
        ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); res = np.zeros(n, RQT_RESULT_DTYPE); cx = np.zeros((n, CABAC_CTX_BYTES), np.uint8); cu = np.zeros((n, CABAC_CU_CTX_BYTES), np.uint8)
        coef = np.zeros(int(sum((3 << (2 * int(j["log2_cu"]))) // 2 for j in jobs)), np.int32)
        self.L.hop_intra_rqt.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 6
        self._chk(self.L.hop_intra_rqt(self.h, n, jobs.ctypes.data, syntax.ctypes.data, opts.ctypes.data, len(ctx_in), ctx_in.ctypes.data, cu_ctx_in.ctypes.data, res.ctypes.data,
                                       coef.ctypes.data, cx.ctypes.data, cu.ctypes.data), "hop_intra_rqt")
        return res, coef, cx, cu

    def intra_luma_search(self, jobs, syntax, opts, sjobs, ctx_in, cu_ctx_in):
        """estIntraPredQT (luma): returns search results (INTRA_SEARCH_RESULT_DTYPE), arrays (RQT_RESULT_DTYPE), levels (1.5 size^2 per job, luma filled) and the CUs'
        luma reconstruction planes (size^2 per job); the context's reconstruction picture is updated as the reference updates it"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, INTRA_CU_SYNTAX_DTYPE); opts = np.ascontiguousarray(opts, INTRA_RQT_OPT_DTYPE)
        sjobs = np.ascontiguousarray(sjobs, INTRA_SEARCH_JOB_DTYPE); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); sres = np.zeros(n, INTRA_SEARCH_RESULT_DTYPE); res = np.zeros(n, RQT_RESULT_DTYPE)
        coef = np.zeros(int(sum((3 << (2 * int(j["log2_cu"]))) // 2 for j in jobs)), np.int32); reco = np.zeros(int(sum(1 << (2 * int(j["log2_cu"])) for j in jobs)), np.int16)
        self.L.hop_intra_luma_search.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 6
        self._chk(self.L.hop_intra_luma_search(self.h, n, jobs.ctypes.data, syntax.ctypes.data, opts.ctypes.data, sjobs.ctypes.data, len(ctx_in), ctx_in.ctypes.data,
                                               cu_ctx_in.ctypes.data, sres.ctypes.data, res.ctypes.data, coef.ctypes.data, reco.ctypes.data), "hop_intra_luma_search")
        return sres, res, coef, reco

    def intra_chroma_search(self, jobs, syntax, opts, res, ctx_in, cu_ctx_in):
        """estIntraPredChromaQT: res holds tr_idx / tskip[0] of the luma search; returns (mode, dist) pairs, the arrays incl. cbf[1..2] / tskip[1..2], levels (1.5 size^2 per
        job, the chroma parts filled) and the chroma reconstruction planes (size^2 / 2 per job: Cb, Cr)"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, INTRA_CU_SYNTAX_DTYPE); opts = np.ascontiguousarray(opts, INTRA_RQT_OPT_DTYPE)
        res = np.ascontiguousarray(res, RQT_RESULT_DTYPE).copy(); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); cres = np.zeros(n, np.dtype([("best_mode", "<i4"), ("dist", "<u4")]))
        coef = np.zeros(int(sum((3 << (2 * int(j["log2_cu"]))) // 2 for j in jobs)), np.int32); reco = np.zeros(int(sum((1 << (2 * int(j["log2_cu"]))) // 2 for j in jobs)), np.int16)
        self.L.hop_intra_chroma_search.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 6
        self._chk(self.L.hop_intra_chroma_search(self.h, n, jobs.ctypes.data, syntax.ctypes.data, opts.ctypes.data, len(ctx_in), ctx_in.ctypes.data, cu_ctx_in.ctypes.data,
                                                 res.ctypes.data, cres.ctypes.data, coef.ctypes.data, reco.ctypes.data), "hop_intra_chroma_search")
        return cres, res, coef, reco

    def intra_cu_total_bits(self, jobs, syntax, res, coef, dist, ctx_in, cu_ctx_in):
        """the counting part of xCheckRDCostIntra: returns bits, costs, coder states (n, 152) and CU-level states (n, 20) afterwards"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, INTRA_CU_SYNTAX_DTYPE); res = np.ascontiguousarray(res, RQT_RESULT_DTYPE)
        coef = np.ascontiguousarray(coef, np.int32); dist = np.ascontiguousarray(dist, np.uint32); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); bits = np.zeros(n, np.uint32); cost = np.zeros(n, np.float64); cx = np.zeros((n, CABAC_CTX_BYTES), np.uint8); cu = np.zeros((n, CABAC_CU_CTX_BYTES), np.uint8)
        self.L.hop_intra_cu_total_bits.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 6
        self._chk(self.L.hop_intra_cu_total_bits(self.h, n, jobs.ctypes.data, syntax.ctypes.data, res.ctypes.data, coef.ctypes.data, dist.ctypes.data, len(ctx_in), ctx_in.ctypes.data,
                                                 cu_ctx_in.ctypes.data, bits.ctypes.data, cost.ctypes.data, cx.ctypes.data, cu.ctypes.data), "hop_intra_cu_total_bits")
        return bits, cost, cx, cu

    def inter_cu_skip(self, jobs, syntax, ctx_in, cu_ctx_in):
        """encodeResAndCalcRdInterCU without residual: returns (n, 4) root_cbf | dist Y, Cb, Cr, bits, costs, coder states (n, 152), CU-level states (n, 20)"""
        jobs = np.ascontiguousarray(jobs, RQT_JOB_DTYPE); syntax = np.ascontiguousarray(syntax, CU_SYNTAX_DTYPE); ctx_in = np.ascontiguousarray(ctx_in, np.uint8); cu_ctx_in = np.ascontiguousarray(cu_ctx_in, np.uint8)
        n = len(jobs); fin = np.zeros((n, 4), np.uint32); bits = np.zeros(n, np.uint32); cost = np.zeros(n, np.float64)
        cx = np.zeros((n, CABAC_CTX_BYTES), np.uint8); cu = np.zeros((n, CABAC_CU_CTX_BYTES), np.uint8)
        self.L.hop_inter_cu_skip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7
        self._chk(self.L.hop_inter_cu_skip(self.h, n, jobs.ctypes.data, syntax.ctypes.data, len(ctx_in), ctx_in.ctypes.data, cu_ctx_in.ctypes.data, fin.ctypes.data, bits.ctypes.data,
                                           cost.ctypes.data, cx.ctypes.data, cu.ctypes.data), "hop_inter_cu_skip")
        return fin, bits, cost, cx, cu

    def intra_pred(self, jobs, modes):
        n = len(jobs)
        arr = (IntraJob * n)(*jobs); m = np.ascontiguousarray(modes, np.int32)
        self._chk(self.L.hop_intra_pred(self.h, n, ctypes.addressof(arr), m.ctypes.data), "hop_intra_pred")

    def intra_pred_chroma(self, jobs, modes):
        n = len(jobs)
        arr = (IntraJob * n)(*jobs); m = np.ascontiguousarray(modes, np.int32)
        self.L.hop_intra_pred_chroma.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.hop_intra_pred_chroma(self.h, n, ctypes.addressof(arr), m.ctypes.data), "hop_intra_pred_chroma")

    def sync(self):
        self._chk(self.L.hop_sync(self.h), "hop_sync")
