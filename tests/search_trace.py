"""Shared by oracle/ref/make_golden_search.py (generator, build container) and tests/test_golden_search.py (checker):
the per-candidate record layout, the oracle -> record reduction, and the binding that shows an oracle state to the
REFERENCE's search code (oracle/_ref/libhmleaf.so; only used where /root/reference was present at build time)."""
import ctypes as C
import os
import zlib

import numpy as np

import hmo_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# HM context order (TEncSbac.cpp:56-96) <- oracle order (oracle/hmo.h); (oracle index, HM index)
O2HM = ([(i, i) for i in range(3)] + [(3, 8), (4, 13), (5, 14)] + [(6 + i, 28 + i) for i in range(10)] +
        [(16 + i, 38 + i) for i in range(3)] + [(19 + i, 42 + i) for i in range(4)] + [(23 + i, 46 + i) for i in range(44)] +
        [(67 + i, 90 + i) for i in range(30)] + [(97 + i, 120 + i) for i in range(30)] + [(127 + i, 150 + i) for i in range(24)] +
        [(151 + i, 174 + i) for i in range(6)] + [(157, 183), (158, 184)] +
        # inter syntax: skip 3..5, merge flag 6, merge idx 7, pred mode 12, part size 9..11, mvd 26..27, ref 24..25, mvp idx 180, root cbf 41
        [(160 + i, 3 + i) for i in range(3)] + [(163, 6), (164, 7), (165, 12)] + [(166 + i, 9 + i) for i in range(3)] +
        [(169, 26), (170, 27), (171, 24), (172, 25), (173, 180), (174, 41)])
O_IDX = np.array([a for a, _ in O2HM])
HM_IDX = np.array([b for _, b in O2HM])

FIELDS = ["ctu", "zidx", "depth", "part_size", "dist_luma", "dist", "bits", "bins", "cost_lo", "cost_hi",
          "dir0", "dir1", "dir2", "dir3", "chroma_dir", "crc_tree", "crc_coef", "crc_reco", "crc_coder"]


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def _cost_words(cost):
    u = np.array([cost], np.float64).view(np.uint32)
    return int(u[0]), int(u[1])


def record_from_oracle(enc, depth, part_size, dist_luma=None):
    """The candidate the oracle has just searched (temp CU of `depth`, reco_temp, slot [depth][CI_TEMP_BEST])."""
    cu = enc.test_cu(depth, best=False)
    n, s = cu.nparts, 64 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    q = n // 4 if part_size == hmo_py.SIZE_NxN else 0
    dirs = [int(cu.intra_dir[0][k * q]) for k in range(4)]
    tree = crc(A(cu.tr_idx)[:n], A(cu.cbf)[:, :n], A(cu.tskip)[:, :n], A(cu.intra_dir)[:, :n])
    coef = crc(A(cu.coef)[0, :s * s], A(cu.coef)[1, :s * s // 4], A(cu.coef)[2, :s * s // 4])
    r = enc.test_reco(depth, best=False)
    h = s // 2
    reco = crc(A(r.y).reshape(64, 64)[:s, :s], A(r.u).reshape(32, 32)[:h, :h], A(r.v).reshape(32, 32)[:h, :h])
    ctx, frac = enc.test_slot(depth, hmo_py.CI_TEMP_BEST)
    coder = crc(ctx[:159], np.array([frac], np.uint64))
    lo, hi = _cost_words(cu.cost)
    dl = enc.last_luma_dist() if dist_luma is None else dist_luma
    return np.array([enc.cur_ctu(), cu.zidx, depth, part_size, dl, cu.dist, cu.bits, cu.bins, lo, hi] + dirs +
                    [int(cu.intra_dir[1][0]), tree, coef, reco, coder], np.uint32)


def fmt(rec):
    return " ".join(f"{k}={int(v)}" for k, v in zip(FIELDS, rec))


# ------------------------------------------------------------------------------------------------------------------
class RefCuOut(C.Structure):
    """RefCuOut of oracle/ref/ref_driver.cpp."""
    _fields_ = [("dist", C.c_uint), ("bits", C.c_uint), ("bins", C.c_uint), ("dist_luma", C.c_uint), ("cost", C.c_double),
                ("luma_dir", C.c_uint8 * 256), ("chroma_dir", C.c_uint8 * 256), ("tr_idx", C.c_uint8 * 256),
                ("cbf", (C.c_uint8 * 256) * 3), ("tskip", (C.c_uint8 * 256) * 3),
                ("skip", C.c_uint8 * 256), ("merge_flag", C.c_uint8 * 256), ("merge_idx", C.c_uint8 * 256),
                ("inter_dir", C.c_uint8 * 256), ("part_size", C.c_uint8 * 256), ("pred_mode", C.c_uint8 * 256),
                ("mvp_idx", C.c_int8 * 256), ("ref_idx", C.c_int8 * 256),
                ("mv", (C.c_int16 * 2) * 256), ("mvd", (C.c_int16 * 2) * 256),
                ("coef", (C.c_int32 * 4096) * 3), ("reco", (C.c_uint8 * 4096) * 3), ("pred", (C.c_uint8 * 4096) * 3)]


class RefSearch:
    """The reference's TEncSearch behind ref_driver.cpp, fed with oracle states."""

    def __init__(self, w, h, qp, org):
        self.L = L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhmleaf.so"))
        L.ref_coder_get.restype = C.c_ulonglong
        L.ref_coder_set.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ulonglong]
        L.ref_coder_get.argtypes = [C.c_int, C.c_int, C.c_void_p]
        self.w, self.h, self.qp = w, h, qp
        self.n_ctu = L.ref_setup(w, h, qp)
        assert L.ref_search_setup(64, 0, 1, 1, 1) == 0
        for c in range(3):
            L.ref_set_org(c, np.ascontiguousarray(org[c]).ctypes.data_as(C.c_void_p))
        # HM-order states of a freshly reset coder: the contexts the oracle does not model keep these
        L.ref_cabac_reset()
        st = np.zeros(512, np.uint8)
        self.n_hm = L.ref_cabac_states(st.ctypes.data_as(C.c_void_p))
        self.base = st[:self.n_hm].copy()
        self.w_ctu = (w + 63) // 64

    def to_hm(self, ctx):
        st = self.base.copy()
        st[HM_IDX] = ctx[O_IDX]
        return st

    def from_hm(self, st):
        ctx = np.zeros(hmo_py.NCTX, np.uint8)
        ctx[O_IDX] = st[HM_IDX]
        return ctx

    def load_state(self, enc, depth, n_back=None):
        """Reconstruction planes, decided CTU arrays (current CTU and its left / above row neighbours) and the coder slot
        [depth][CI_CURR_BEST] of the oracle -> the reference's picture / RD coder."""
        L, vp = self.L, lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        for c in range(3):
            L.ref_set_rec(c, vp(enc.rec[c]))
        cur = enc.cur_ctu()
        for a in range(max(0, cur - self.w_ctu - 1), cur + 1):
            c = enc.ctu_arrays(a)
            for fid, name in ((0, "depth"), (1, "part_size"), (2, "pred_mode"), (5, "tr_idx")):
                L.ref_set_ctu_field(a, fid, vp(np.ascontiguousarray(c[name]).view(np.uint8)))
            for k in range(2):
                L.ref_set_ctu_field(a, 3 + k, vp(c["intra_dir"][k]))
            for k in range(3):
                L.ref_set_ctu_field(a, 6 + k, vp(c["tskip"][k]))
                L.ref_set_ctu_field(a, 9 + k, vp(c["cbf"][k]))
        ctx, frac = enc.test_slot(depth, hmo_py.CI_CURR_BEST)
        st = self.to_hm(ctx)
        L.ref_coder_set(depth, hmo_py.CI_CURR_BEST, st.ctypes.data_as(C.c_void_p), frac)

    def intra_cu(self, ctu, zidx, depth, part_size, stage=2):
        out = RefCuOut()
        assert self.L.ref_intra_cu(ctu, zidx, depth, part_size, stage, C.byref(out)) == 0
        return out

    def coder(self, depth, ci):
        st = np.zeros(512, np.uint8)
        frac = self.L.ref_coder_get(depth, ci, st.ctypes.data_as(C.c_void_p))
        return self.from_hm(st[:self.n_hm]), int(frac)


def record_from_ref(r, ref, depth, ctu, zidx, part_size):
    n, s = 256 >> (2 * depth), 64 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    q = n // 4 if part_size == hmo_py.SIZE_NxN else 0
    dirs = [int(r.luma_dir[k * q]) for k in range(4)]
    intra_dir = np.stack([A(r.luma_dir)[:n], A(r.chroma_dir)[:n]])
    tree = crc(A(r.tr_idx)[:n], A(r.cbf)[:, :n], A(r.tskip)[:, :n], intra_dir)
    coef = crc(A(r.coef)[0, :s * s], A(r.coef)[1, :s * s // 4], A(r.coef)[2, :s * s // 4])
    h = s // 2
    reco = crc(A(r.reco)[0, :s * s].reshape(s, s), A(r.reco)[1, :h * h].reshape(h, h), A(r.reco)[2, :h * h].reshape(h, h))
    ctx, frac = ref.coder(depth, hmo_py.CI_TEMP_BEST)
    coder = crc(ctx[:159], np.array([frac], np.uint64))
    lo, hi = _cost_words(r.cost)
    return np.array([ctu, zidx, depth, part_size, r.dist_luma, r.dist, r.bits, r.bins, lo, hi] + dirs +
                    [int(r.chroma_dir[0]), tree, coef, reco, coder], np.uint32)
