"""ctypes binding of the ORACLE (oracle/libhmo.so).  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NPART = 256
TRAINING, VERIFYING, TESTING = 0, 1, 2      # fork states (getCurrentState, tools_YS.cpp:1237)


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("qp", C.c_int), ("slice_ctus", C.c_int),
                ("transform_skip", C.c_int), ("transform_skip_fast", C.c_int), ("sign_hiding", C.c_int),
                ("strong_smoothing", C.c_int), ("lambda_", C.c_double), ("sqrt_lambda", C.c_double),
                ("chroma_weight", C.c_double), ("rdoq_lambda", C.c_double * 3), ("qp_c", C.c_int),
                ("slice_type", C.c_int), ("search_range", C.c_int), ("fast_search", C.c_int), ("fast_enc", C.c_int),
                ("had_me", C.c_int), ("fdm", C.c_int), ("max_merge_cand", C.c_int), ("amp", C.c_int), ("tmvp", C.c_int), ("rdoq", C.c_int), ("rdoq_ts", C.c_int), ("lambda_override", C.c_double),
                ("lambda_motion_sad", C.c_uint), ("lambda_motion_sse", C.c_uint), ("cabac_b_table", C.c_int), ("search_state_per_slice", C.c_int)]


class Ctu(C.Structure):
    _fields_ = [("depth", C.c_uint8 * NPART), ("width", C.c_uint8 * NPART), ("height", C.c_uint8 * NPART),
                ("skip", C.c_uint8 * NPART), ("part_size", C.c_int8 * NPART), ("pred_mode", C.c_int8 * NPART),
                ("tq_bypass", C.c_uint8 * NPART), ("qp", C.c_int8 * NPART), ("chroma_qp_adj", C.c_uint8 * NPART),
                ("tr_idx", C.c_uint8 * NPART), ("tskip", (C.c_uint8 * NPART) * 3), ("cbf", (C.c_uint8 * NPART) * 3),
                ("intra_dir", (C.c_uint8 * NPART) * 2), ("ipcm", C.c_uint8 * NPART),
                ("merge_flag", C.c_uint8 * NPART), ("merge_idx", C.c_uint8 * NPART), ("inter_dir", C.c_uint8 * NPART),
                ("mvp_idx", C.c_int8 * NPART), ("ref_idx", C.c_int8 * NPART),
                ("mv", (C.c_int16 * 2) * NPART), ("mvd", (C.c_int16 * 2) * NPART),
                ("coeff_y", C.c_int32 * 4096), ("coeff_cb", C.c_int32 * 1024), ("coeff_cr", C.c_int32 * 1024),
                ("total_cost", C.c_double), ("total_dist", C.c_uint32), ("total_bits", C.c_uint32),
                ("total_bins", C.c_uint32)]


SLICE_I, SLICE_P = 0, 1
NCTX = 176                                   # HMO_NCTX: 160 intra contexts + the inter syntax (hmo.h)


class Cabac(C.Structure):
    _fields_ = [("ctx", C.c_uint8 * NCTX), ("frac", C.c_uint64)]


class CU(C.Structure):
    """HmoCU (hmo_int.h): a per-depth working CU, arrays relative to the CU's first partition."""
    _fields_ = [("depth_cu", C.c_int), ("x", C.c_int), ("y", C.c_int), ("zidx", C.c_int), ("nparts", C.c_int),
                ("cost", C.c_double), ("dist", C.c_uint32), ("bits", C.c_uint32), ("bins", C.c_uint32),
                ("depth", C.c_uint8 * NPART), ("part_size", C.c_int8 * NPART), ("pred_mode", C.c_int8 * NPART),
                ("tr_idx", C.c_uint8 * NPART), ("tskip", (C.c_uint8 * NPART) * 3), ("cbf", (C.c_uint8 * NPART) * 3),
                ("intra_dir", (C.c_uint8 * NPART) * 2),
                ("skip", C.c_uint8 * NPART), ("merge_flag", C.c_uint8 * NPART), ("merge_idx", C.c_uint8 * NPART),
                ("inter_dir", C.c_uint8 * NPART), ("mvp_idx", C.c_int8 * NPART), ("ref_idx", C.c_int8 * NPART),
                ("mv", (C.c_int16 * 2) * NPART), ("mvd", (C.c_int16 * 2) * NPART),
                ("coef", (C.c_int32 * 4096) * 3)]


class Yuv(C.Structure):
    _fields_ = [("y", C.c_uint8 * 4096), ("u", C.c_uint8 * 1024), ("v", C.c_uint8 * 1024)]


SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN = 0, 1, 2, 3
SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N = 4, 5, 6, 7
CI_CURR_BEST, CI_NEXT_BEST, CI_TEMP_BEST, CI_CHROMA_INTRA, CI_QT_TRAFO_TEST, CI_QT_TRAFO_ROOT = range(6)
EV_INTRA_BEGIN, EV_INTRA_END, EV_INTER_BEGIN, EV_INTER_END, EV_MERGE_BEGIN, EV_MERGE_END, EV_CU_BEGIN, EV_CU_DONE = range(8)
TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


def load():
    lib = C.CDLL(os.path.join(_HERE, "libhmo.so"))
    lib.hmo_create.restype = C.c_void_p
    lib.hmo_create.argtypes = [C.POINTER(Params)]
    lib.hmo_destroy.argtypes = [C.c_void_p]
    lib.hmo_params_default.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
    lib.hmo_params_finish.argtypes = [C.POINTER(Params)]
    lib.hmo_set_planes.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.hmo_set_ref_planes.argtypes = [C.c_void_p] + [C.c_void_p] * 3
    lib.hmo_test_n_sad.restype = C.c_uint64
    lib.hmo_test_n_sad.argtypes = [C.c_void_p]
    lib.hmo_compress_ctu.argtypes = [C.c_void_p, C.c_int]
    lib.hmo_compress_frame.argtypes = [C.c_void_p]
    lib.hmo_get_ctu.restype = C.POINTER(Ctu)
    lib.hmo_get_ctu.argtypes = [C.c_void_p, C.c_int]
    lib.hmo_get_cabac.restype = C.POINTER(Cabac)
    lib.hmo_get_cabac.argtypes = [C.c_void_p]
    lib.hmo_num_ctus.argtypes = [C.c_void_p]
    lib.hmo_ctu_replay_bits.restype = C.c_uint32
    lib.hmo_obf_prepass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.hmo_tcm_threshold.restype = C.c_double
    lib.hmo_tcm_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.hmo_ctu_replay_bits.argtypes = [C.c_void_p, C.c_int]
    lib.hmo_deblock.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.hmo_deblock_pic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.hmo_sao_picture.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7
    lib.hmo_sao_stats.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hmo_set_pu_trace.argtypes = [C.c_void_p, C.c_void_p]
    lib.hmo_set_decision.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.hmo_get_verify.argtypes = [C.c_void_p, C.c_void_p]
    lib.hmo_decision_switch.argtypes = [C.c_void_p] * 5
    lib.hmo_set_trace.argtypes = [C.c_void_p, TRACE_FN, C.c_void_p]
    lib.hmo_test_cu.restype = C.POINTER(CU)
    lib.hmo_test_cu.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.hmo_test_reco.restype = C.POINTER(Yuv)
    lib.hmo_test_reco.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.hmo_test_slot.restype = C.POINTER(Cabac)
    lib.hmo_test_slot.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.hmo_test_cur_ctu.argtypes = [C.c_void_p]
    lib.hmo_test_last_luma_dist.argtypes = [C.c_void_p]
    lib.hmo_test_last_luma_dist.restype = C.c_uint32
    return lib


PU_TRACE_DTYPE = np.dtype([("valid", np.uint8), ("best_mode", np.uint8), ("n_rmd", np.uint8), ("n_rd", np.uint8), ("rd_mode", np.uint8, (12,)),
                           ("best_dist", np.uint32), ("pad", np.uint32), ("best_cost", np.float64), ("rmd_cost", np.float64, (8,))])   # = fcu_pu_trace
PUS_PER_CTU = 341


class Encoder:
    """One chain: a frame (or its slices) decided CTU by CTU in raster order."""

    def __init__(self, Y, U, V, qp, slice_ctus=0, ref=None, col=None, refs=None, ref_pocs=None, poc=None, col_ref_pocs=None, **flags):
        """ref = (Y, U, V) planes of the reference picture makes this a P picture (slice_type P, list 0, index 0);
        col = the reference picture's decided CTUs (bytes of its Ctu array: all_ctus_bytes()) switches TMVP on;
        refs = [(Y, U, V), ...] (up to 4) with ref_pocs = their POCs and poc = this picture's: several reference pictures
        (RefPicList0 in that order; refs[0] is also the collocated picture, col_ref_pocs = the POCs ITS list 0 named);
        flags: any Params field (search_range, fast_enc, lambda_override, ...)."""
        if refs is not None:
            assert ref is None and 1 <= len(refs) <= 4 and len(ref_pocs) == len(refs) and poc is not None
            ref = refs[0]
        self.lib = load()
        h, w = Y.shape
        self.p = Params()
        self.lib.hmo_params_default(C.byref(self.p), w, h, qp)
        self.p.slice_ctus = slice_ctus
        if ref is not None:
            self.p.slice_type = SLICE_P
        if col is not None:
            self.p.tmvp = 1
        known = {n for n, _ in Params._fields_}
        for k, v in flags.items():
            if k not in known:
                raise TypeError(f"unknown oracle parameter {k!r}")
            setattr(self.p, k, v)
        self.lib.hmo_params_finish(C.byref(self.p))
        self.org = [np.ascontiguousarray(a, dtype=np.uint8) for a in (Y, U, V)]
        self.rec = [np.zeros_like(a) for a in self.org]
        self.h = self.lib.hmo_create(C.byref(self.p))
        self.lib.hmo_set_planes(self.h, *[a.ctypes.data for a in self.org], *[a.ctypes.data for a in self.rec])
        self.n_ctu = self.lib.hmo_num_ctus(self.h)
        self.ref = None
        if ref is not None:
            self.ref = [np.ascontiguousarray(a, dtype=np.uint8) for a in ref]
            self.lib.hmo_set_ref_planes(self.h, *[a.ctypes.data for a in self.ref])
        self.refs = None
        if refs is not None:
            self.refs = [[np.ascontiguousarray(a, dtype=np.uint8) for a in r] for r in refs]
            self.lib.hmo_set_ref_picture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            for i, r in enumerate(self.refs):
                self.lib.hmo_set_ref_picture(self.h, i, *[a.ctypes.data for a in r], int(ref_pocs[i]))
            self.lib.hmo_set_poc.argtypes = [C.c_void_p, C.c_int, C.c_int]
            self.lib.hmo_set_poc(self.h, int(poc), len(refs))
            crp = np.ascontiguousarray(col_ref_pocs if col_ref_pocs is not None else [ref_pocs[0] - 1], np.int32)
            self.lib.hmo_set_col_pocs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            self.lib.hmo_set_col_pocs(self.h, int(ref_pocs[0]), crp.ctypes.data, len(crp))
        self.col = None
        if col is not None:
            assert len(col) == C.sizeof(Ctu) * self.n_ctu
            self.col = np.frombuffer(bytes(col), dtype=np.uint8).copy()
            self.lib.hmo_set_col.argtypes = [C.c_void_p, C.c_void_p]
            self.lib.hmo_set_col(self.h, self.col.ctypes.data)

    def all_ctus_bytes(self):
        """the decided picture's Ctu array as bytes (same layout as the engine's fcu_ctu_out array): the motion field a later
        picture's TMVP reads"""
        return b"".join(C.string_at(C.addressof(self.ctu(a)), C.sizeof(Ctu)) for a in range(self.n_ctu))

    def compress_ctu(self, a):
        self.lib.hmo_compress_ctu(self.h, a)

    def compress_frame(self):
        self.lib.hmo_compress_frame(self.h)

    def ctu(self, a):
        return self.lib.hmo_get_ctu(self.h, a).contents

    def ctu_arrays(self, a):
        c = self.ctu(a)
        out = {}
        for name, _ in Ctu._fields_:
            v = getattr(c, name)
            out[name] = np.ctypeslib.as_array(v).copy() if hasattr(v, "_length_") else v
        return out

    def cabac(self, full=False):
        """context states of [0][CI_CURR_BEST] after the last CTU: the 160 an I slice uses (all NCTX with full=True) + Q15 counter"""
        c = self.lib.hmo_get_cabac(self.h).contents
        a = np.ctypeslib.as_array(c.ctx).copy()
        return (a if full else a[:160]), int(c.frac)

    # -- test hooks (hmo_int.h: trace)
    def set_trace(self, fn, cu_events=False):
        """fn(event, depth, arg) around every CU candidate (EV_* above); cu_events: also EV_CU_BEGIN / EV_CU_DONE around
        the candidates of every CU (arg = eParentPartSize)"""
        self._trace = TRACE_FN(lambda user, ev, d, a: fn(ev, d, a) if (cu_events or ev < EV_CU_BEGIN) else None)
        self.lib.hmo_set_trace(self.h, self._trace, None)

    def test_cu(self, depth, best=False):
        return self.lib.hmo_test_cu(self.h, depth, int(best)).contents

    def test_reco(self, depth, best=False):
        return self.lib.hmo_test_reco(self.h, depth, int(best)).contents

    def test_slot(self, depth, ci):
        c = self.lib.hmo_test_slot(self.h, depth, ci).contents
        return np.ctypeslib.as_array(c.ctx).copy(), int(c.frac)

    def cur_ctu(self):
        return self.lib.hmo_test_cur_ctu(self.h)

    def last_luma_dist(self):
        """luma distortion of the last intra candidate (what estIntraPredLumaQT left in the CU before chroma)"""
        return self.lib.hmo_test_last_luma_dist(self.h)

    def set_decision(self, state, obf=None, sw_skip=(0, 0, 0, 0), sw_term=(0, 0, 0, 0), depth_exception=0):
        """Fork state of the frame + Naive decision switches per depth + the frame's OBF map (int16 [h/4, w/4])."""
        self._obf = None if obf is None else np.ascontiguousarray(obf, dtype=np.int16)
        assert state == TRAINING or self._obf is not None
        sk, te = np.ascontiguousarray(sw_skip, np.uint8), np.ascontiguousarray(sw_term, np.uint8)
        self.lib.hmo_set_decision(self.h, state, sk.ctypes.data, te.ctypes.data, depth_exception,
                                  None if self._obf is None else self._obf.ctypes.data)

    def verify_counts(self):
        """g_iVerResult[depth][TP, FP, TN, FN, FPLoss, FNLoss] since set_decision."""
        v = np.zeros((4, 6), np.float64)
        self.lib.hmo_get_verify(self.h, v.ctypes.data)
        return v

    def set_int_mv(self, xy):
        """m_integerMv2Nx2N[list 0][r] as the previous picture (or whoever ran the search last) left it: [(x, y)] per reference index"""
        a = (C.c_int * 8)(*[int(v) for p in (list(xy) + [(0, 0)] * 4)[:4] for v in p])
        self.lib.hmo_set_int_mv.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hmo_set_int_mv(self.h, a)

    def test_int_mv(self):
        xy = (C.c_int * 8)()
        self.lib.hmo_test_int_mv.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hmo_test_int_mv(self.h, xy)
        return [(int(xy[2 * r]), int(xy[2 * r + 1])) for r in range(4)]       # per reference index

    def enable_pu_trace(self):
        """per-PU record of the luma search (BASELINE configs[1]): structured array [n_ctu, 341], filled as CTUs are decided"""
        self.pu_trace = np.zeros((self.n_ctu, PUS_PER_CTU), PU_TRACE_DTYPE)
        self.lib.hmo_set_pu_trace(self.h, self.pu_trace.ctypes.data)
        return self.pu_trace

    def deblock(self, beta_offset_div2=0, tc_offset_div2=0):
        """In-loop deblocking of the decided picture, in place on self.rec (TComLoopFilter::loopFilterPic)."""
        self.lib.hmo_deblock(self.h, beta_offset_div2, tc_offset_div2)

    def replay_bits(self, a):
        return self.lib.hmo_ctu_replay_bits(self.h, a)

    def close(self):
        if self.h:
            self.lib.hmo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ldp_slice(poc, base_qp, gop=((3, 0.4624), (2, 0.4624), (3, 0.4624), (1, 0.578)), had_me=True):
    """Slice QP and lambda of picture `poc` under HM's encoder_lowdelay_P_main GOP table (QP offset, QP factor of Frame1..4)
    as TEncSlice::initEncSlice derives them (TEncSlice.cpp:560-740): returns (slice_type, qp, lambda)."""
    n = len(gop)
    if poc == 0:
        scale = 1.0 - min(0.5, max(0.0, 0.05 * (n - 1)))
        return SLICE_I, base_qp, 0.57 * scale * 2.0 ** ((base_qp - 12) / 3.0)
    off, factor = gop[(poc - 1) % n]
    qp = base_qp + off
    r = poc % n
    depth = 0
    if r:
        step, i = n, n >> 1
        while i >= 1:
            j, hit = i, False
            while j < n:
                if j == r:
                    hit = True
                    break
                j += step
            step >>= 1
            depth += 1
            if hit:
                break
            i >>= 1
    lam = factor * 2.0 ** ((qp - 12) / 3.0)
    if depth > 0:
        lam *= min(4.0, max(2.0, (qp - 12) / 6.0))
    if not had_me:
        lam *= 0.95
    return SLICE_P, qp, lam


def ldp_layer(poc, n=4):
    """pic->getSlice(0)->getDepth(): temporal layer of picture `poc` inside its GOP (TEncSlice.cpp:236-262)"""
    r = poc % n
    if r == 0:
        return 0
    depth, step, i = 0, n, n >> 1
    while i >= 1:
        j = i
        while j < n:
            if j == r:
                return depth + 1
            j += step
        step >>= 1
        depth += 1
        i >>= 1
    return depth


def slice_lambdas(qp, lam):
    """TComSlice::getLambdas(): [lambda, lambda / w, lambda / w], w = 2^((QP - QPc) / 3) (TEncSlice::setUpLambda, TEncSlice.cpp:163-186)"""
    p = Params()
    load().hmo_params_default(C.byref(p), 64, 64, qp)
    w = 2.0 ** ((qp - p.qp_c) / 3.0)
    return [lam, lam / w, lam / w]


SAO_DISABLE_RATE = (0.75, 0.5, 0.5)                         # SAO_ENCODING_RATE / _CHROMA, TypeDef.h:201-204


class SaoState:
    """m_saoDisabledRate of TEncSampleAdaptiveOffset across the pictures of a sequence (decidePicParams :363-395, update :895-917)"""

    def __init__(self):
        self.rate = np.zeros((3, 8), np.float64)

    def enabled(self, layer):
        return [0 if (layer > 0 and self.rate[c][layer - 1] > SAO_DISABLE_RATE[c]) else 1 for c in range(3)]

    def update(self, layer, off_count, n_ctu):
        for c in range(3):
            self.rate[c][layer] = float(off_count[c]) / float(n_ctu)


def sao_picture(org, rec, qp, slice_type, lam, enabled=(1, 1, 1), slice_ctus=0, want_stats=False):
    """SAO of one deblocked picture, in place on `rec` (three uint8 planes).  Returns (params int32 [n_ctu, 3, 35] =
    mode, type, aux, offset[32] as signalled; off_count[3]; stats int64 [n_ctu, 3, 2, 5, 32] (diff, count) or None)."""
    lib = load()
    h, w = org[0].shape
    n = ((w + 63) // 64) * ((h + 63) // 64)
    for a in list(org) + list(rec):
        assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    po = (C.c_void_p * 3)(*[a.ctypes.data for a in org])
    pr = (C.c_void_p * 3)(*[a.ctypes.data for a in rec])
    lams = (C.c_double * 3)(*slice_lambdas(qp, lam))
    en = (C.c_int * 3)(*[int(v) for v in enabled])
    params = np.zeros((n, 3, 35), np.int32)
    stats = np.zeros((n, 3, 2, 5, 32), np.int64) if want_stats else None
    off = (C.c_int * 3)()
    lib.hmo_sao_picture(w, h, slice_ctus, qp, slice_type, C.cast(lams, C.c_void_p), C.cast(en, C.c_void_p), C.cast(po, C.c_void_p), C.cast(pr, C.c_void_p),
                        params.ctypes.data, stats.ctypes.data if want_stats else None, C.cast(off, C.c_void_p))
    return params, list(off), stats


def obf_prepass(Y):
    """Outlier-block-flag map of a luma plane (fork pre-pass): (obf int16 [h/4, w/4], yc float64[16])."""
    lib = load()
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    h, w = Y.shape
    obf = np.zeros((h // 4, w // 4), np.int16)
    yc = np.zeros(16, np.float64)
    err = lib.hmo_obf_prepass(Y.ctypes.data, w, h, w, obf.ctypes.data, yc.ctypes.data)
    assert err == 0, "amplitude beyond the reference's bucket array"
    return obf, yc


def decision_switch(ver, th_skip=(0, 0, 0, 0), th_term=(0, 0, 0, 0)):
    """SetDecisionSwitch: (sw_skip[4], sw_term[4]) from the Verifying frame's counters."""
    lib = load()
    v = np.ascontiguousarray(ver, np.float64)
    a, b = np.ascontiguousarray(th_skip, np.float64), np.ascontiguousarray(th_term, np.float64)
    sk, te = np.zeros(4, np.uint8), np.zeros(4, np.uint8)
    lib.hmo_decision_switch(v.ctypes.data, a.ctypes.data, b.ctypes.data, sk.ctypes.data, te.ctypes.data)
    return sk, te


def deblock_pic(ctus, width, height, rec, beta_offset_div2=0, tc_offset_div2=0):
    """Deblocks (Y, U, V) uint8 planes in place given the picture's per-CTU decisions: `ctus` is a ctypes array of
    Ctu (or anything with the same memory layout, e.g. the engine's fcu_ctu_out array as bytes)."""
    lib = load()
    buf = np.frombuffer(ctus, dtype=np.uint8) if not isinstance(ctus, np.ndarray) else ctus
    assert buf.nbytes == C.sizeof(Ctu) * ((width + 63) // 64) * ((height + 63) // 64)
    buf = np.ascontiguousarray(buf)
    for a in rec:
        assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    lib.hmo_deblock_pic(buf.ctypes.data, width, height, rec[0].ctypes.data, rec[1].ctypes.data, rec[2].ctypes.data,
                        beta_offset_div2, tc_offset_div2)
