"""ctypes binding of tests/emu/libfcu_emu.so (TEST-ONLY wave emulator of the engine source)."""
import ctypes as C
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "oracle"))
from hmo_py import Ctu  # same TComDataCU layout as include/fcu.h:fcu_ctu_out


def load():
    lib = C.CDLL(os.path.join(_HERE, "emu", "libfcu_emu.so"))
    lib.fcu_emu_create.restype = C.c_void_p
    lib.fcu_emu_create.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7
    lib.fcu_emu_destroy.argtypes = [C.c_void_p]
    lib.fcu_emu_compress_ctu.argtypes = [C.c_void_p, C.c_int]
    lib.fcu_emu_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcu_emu_set_decision.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.fcu_emu_get_verify.argtypes = [C.c_void_p, C.c_void_p]
    lib.fcu_emu_tu_trials.restype = C.c_ulonglong
    lib.fcu_emu_tu_trials.argtypes = [C.c_void_p]
    lib.fcu_emu_set_p.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcu_emu_set_lambda.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.fcu_emu_get_state_full.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcu_emu_set_rdoq.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fcu_emu_set_amp.argtypes = [C.c_void_p, C.c_int]
    lib.fcu_emu_set_cabac_b.argtypes = [C.c_void_p, C.c_int]
    lib.fcu_emu_set_col.argtypes = [C.c_void_p, C.c_void_p]
    return lib


REF_MARGIN = 80


def pad_planes(planes):
    """numpy twin of fcu_pad_reference (TComPicYuv::extendPicBorder): border replicated 80 / 40 samples"""
    return [np.ascontiguousarray(np.pad(p, REF_MARGIN >> (1 if k else 0), mode="edge")) for k, p in enumerate(planes)]


class EmuEncoder:
    def __init__(self, Y, U, V, qp, slice_ctus=0, tools=-1, ref=None, lam=None, search_range=64, fast_search=0, rdoq=1, rdoq_ts=1, col=None, amp=0, cabac_b_table=0,
                 refs=None, ref_pocs=None, poc=None, col_ref_pocs=None):
        """ref = (Y, U, V) of the reference picture makes this a P picture (lam = its slice lambda); refs / ref_pocs / poc /
        col_ref_pocs: several reference pictures, as hmo_py.Encoder takes them"""
        if refs is not None:
            assert ref is None and len(refs) == len(ref_pocs) and poc is not None
            ref = refs[0]
        self.lib = load()
        h, w = Y.shape
        self.org = [np.ascontiguousarray(a, dtype=np.uint8) for a in (Y, U, V)]
        self.rec = [np.zeros_like(a) for a in self.org]
        self.n_ctu = ((w + 63) // 64) * ((h + 63) // 64)
        self.out = (Ctu * self.n_ctu)()
        self.h = self.lib.fcu_emu_create(w, h, qp, slice_ctus, tools, *[a.ctypes.data for a in self.org],
                                         *[a.ctypes.data for a in self.rec], C.addressof(self.out))
        self._rdoq = (rdoq, rdoq_ts)
        self._amp = amp
        self._cabac_b = cabac_b_table
        self._col = None if col is None else np.frombuffer(bytes(col), dtype=np.uint8).copy()      # TMVP: the reference picture's Ctu array
        if ref is not None:
            self.pad = pad_planes([np.ascontiguousarray(a, dtype=np.uint8) for a in ref])
            self.lib.fcu_emu_set_p(self.h, qp, float(lam), search_range, fast_search, *[a.ctypes.data for a in self.pad])
            if refs is not None:
                self.pads = [pad_planes([np.ascontiguousarray(a, dtype=np.uint8) for a in r]) for r in refs]
                ptrs = (C.c_void_p * (3 * len(refs)))(*[a.ctypes.data for r in self.pads for a in r])
                pocs = np.ascontiguousarray(ref_pocs, np.int32)
                crp = np.ascontiguousarray(col_ref_pocs if col_ref_pocs is not None else [ref_pocs[0] - 1], np.int32)
                self.lib.fcu_emu_set_refs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
                self.lib.fcu_emu_set_refs(self.h, len(refs), ptrs, pocs.ctypes.data, int(poc), crp.ctypes.data, len(crp))
        elif lam is not None:
            self.lib.fcu_emu_set_lambda(self.h, qp, float(lam))

    def enable_pu_trace(self):
        import hmo_py
        self.pu_trace = np.zeros((self.n_ctu, hmo_py.PUS_PER_CTU), hmo_py.PU_TRACE_DTYPE)
        self.lib.fcu_emu_set_pu_trace.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.fcu_emu_set_pu_trace(self.h, self.pu_trace.ctypes.data)
        return self.pu_trace

    def compress_ctu(self, a):
        self.lib.fcu_emu_set_rdoq(self.h, *self._rdoq)           # (set_p / set_lambda rebuild the parameter block)
        self.lib.fcu_emu_set_amp(self.h, self._amp)
        self.lib.fcu_emu_set_cabac_b(self.h, self._cabac_b)
        if self._col is not None:
            self.lib.fcu_emu_set_col(self.h, self._col.ctypes.data)
        self.lib.fcu_emu_compress_ctu(self.h, a)

    def set_decision(self, state, obf=None, sw_skip=(0, 0, 0, 0), sw_term=(0, 0, 0, 0), depth_exception=0):
        self._obf = None if obf is None else np.ascontiguousarray(obf, dtype=np.int16)
        sk, te = np.ascontiguousarray(sw_skip, np.uint8), np.ascontiguousarray(sw_term, np.uint8)
        self.lib.fcu_emu_set_decision(self.h, state, sk.ctypes.data, te.ctypes.data, depth_exception,
                                      None if self._obf is None else self._obf.ctypes.data)

    def verify_counts(self):
        v = np.zeros((4, 6), np.float64)
        self.lib.fcu_emu_get_verify(self.h, v.ctypes.data)
        return v

    def compress_frame(self):
        for a in range(self.n_ctu):
            self.compress_ctu(a)

    def ctu_arrays(self, a):
        c = self.out[a]
        res = {}
        for name, _ in Ctu._fields_:
            v = getattr(c, name)
            res[name] = np.ctypeslib.as_array(v).copy() if hasattr(v, "_length_") else v
        return res

    def cabac(self, full=False):
        ctx = np.zeros(176 if full else 160, np.uint8)
        frac = C.c_uint64(0)
        (self.lib.fcu_emu_get_state_full if full else self.lib.fcu_emu_get_state)(self.h, ctx.ctypes.data, C.byref(frac))
        return ctx, int(frac.value)
