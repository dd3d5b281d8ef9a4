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
O_SORTED = np.sort(O_IDX)                       # every modelled context (the pads 159, 175 are not)

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


IFIELDS = ["ctu", "zidx", "depth", "kind", "arg", "dist", "bits", "cost_lo", "cost_hi", "skip", "merge0", "midx0", "mvx0", "mvy0",
           "merge1", "midx1", "mvx1", "mvy1", "crc_motion", "crc_tree", "crc_coef", "crc_reco", "crc_coder"]


def _inter_record(ctu, zidx, depth, kind, arg, n, s, dist, bits, cost, skip, merge_flag, merge_idx, mvp_idx, ref_idx, mv, mvd, inter_dir,
                  part_size, pred_mode, tr_idx, cbf, tskip, coef, reco, coder):
    """kind 0 = xCheckRDCostInter (arg = part size), 1 = merge candidate (arg = 2 * cand + noResidual)"""
    last = n - 1
    motion = crc(skip[:n], merge_flag[:n], merge_idx[:n], mvp_idx[:n], ref_idx[:n], mv[:n], mvd[:n], inter_dir[:n], part_size[:n], pred_mode[:n])
    tree = crc(tr_idx[:n], cbf[:, :n], tskip[:, :n])
    lo, hi = _cost_words(cost)
    u = lambda v: int(v) & 0xffffffff
    return np.array([ctu, zidx, depth, kind, arg, dist, bits, lo, hi, int(skip[0]), int(merge_flag[0]), int(merge_idx[0]), u(mv[0][0]), u(mv[0][1]),
                     int(merge_flag[last]), int(merge_idx[last]), u(mv[last][0]), u(mv[last][1]), motion, tree, coef, reco, coder], np.uint32)


def inter_record_from_oracle(enc, depth, kind, arg):
    cu = enc.test_cu(depth, best=False)
    n, s, h = cu.nparts, 64 >> depth, 32 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    coef = crc(A(cu.coef)[0, :s * s], A(cu.coef)[1, :s * s // 4], A(cu.coef)[2, :s * s // 4])
    r = enc.test_reco(depth, best=False)
    reco = crc(A(r.y).reshape(64, 64)[:s, :s], A(r.u).reshape(32, 32)[:h, :h], A(r.v).reshape(32, 32)[:h, :h])
    ctx, frac = enc.test_slot(depth, hmo_py.CI_TEMP_BEST)
    coder = crc(ctx[O_SORTED], np.array([frac], np.uint64))
    return _inter_record(enc.cur_ctu(), cu.zidx, depth, kind, arg, n, s, cu.dist, cu.bits, cu.cost, A(cu.skip), A(cu.merge_flag), A(cu.merge_idx),
                         A(cu.mvp_idx), A(cu.ref_idx), A(cu.mv), A(cu.mvd), A(cu.inter_dir), A(cu.part_size).view(np.uint8), A(cu.pred_mode).view(np.uint8),
                         A(cu.tr_idx), A(cu.cbf), A(cu.tskip), coef, reco, coder)


def inter_record_from_ref(r, ref, depth, ctu, zidx, kind, arg):
    n, s, h = 256 >> (2 * depth), 64 >> depth, 32 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    coef = crc(A(r.coef)[0, :s * s], A(r.coef)[1, :s * s // 4], A(r.coef)[2, :s * s // 4])
    reco = crc(A(r.reco)[0, :s * s].reshape(s, s), A(r.reco)[1, :h * h].reshape(h, h), A(r.reco)[2, :h * h].reshape(h, h))
    ctx, frac = ref.coder(depth, hmo_py.CI_TEMP_BEST)
    coder = crc(ctx[O_SORTED], np.array([frac], np.uint64))
    return _inter_record(ctu, zidx, depth, kind, arg, n, s, r.dist, r.bits, r.cost, A(r.skip), A(r.merge_flag), A(r.merge_idx), A(r.mvp_idx), A(r.ref_idx),
                         A(r.mv), A(r.mvd), A(r.inter_dir), A(r.part_size), A(r.pred_mode), A(r.tr_idx), A(r.cbf), A(r.tskip), coef, reco, coder)


def ifmt(rec):
    return " ".join(f"{k}={int(np.int32(v)) if k.startswith('mv') else int(v)}" for k, v in zip(IFIELDS, rec))


def moving_frame(synth, gen, w, h, seed, i, shift=(3, 1)):
    """picture i of the synthetic lowdelay_P clip: the generator's picture shifted by `shift`
    samples per picture, plus fresh +-2 noise on luma.  A generator name "shear_<gen>" moves bands of the picture with a
    second vector (rows 48..63 of every 64 in the left half, columns 0..15 of every 64 in the right half): motion
    boundaries at quarter positions of the CUs, the case asymmetric partitions exist for."""
    if gen.startswith("shear_"):
        a = moving_frame(synth, gen[6:], w, h, seed, i, shift)
        b = moving_frame(synth, gen[6:], w, h, seed + 1, i, (-shift[1] - 1, shift[0]))
        yy, xx = np.mgrid[0:h, 0:w]
        m = np.where(xx < w // 2, (yy % 64) >= 48, (xx % 64) < 16)
        mc = m[::2, ::2]
        return (np.where(m, b[0], a[0]), np.where(mc, b[1], a[1]), np.where(mc, b[2], a[2]))
    Y, U, V = getattr(synth, gen)(w + 32, h + 32, seed=seed)
    dx, dy = (shift[0] * i) % 32, (shift[1] * i) % 32
    dx = dx & ~1                                                     # horizontal shift even (chroma uses the halved shift); vertical free
    rng = np.random.default_rng(1000 + 17 * seed + i)
    Yc = Y[dy:dy + h, dx:dx + w].astype(np.int16) + rng.integers(-2, 3, (h, w))
    cx, cy = dx // 2, dy // 2
    return (np.clip(Yc, 0, 255).astype(np.uint8), np.ascontiguousarray(U[cy:cy + h // 2, cx:cx + w // 2]),
            np.ascontiguousarray(V[cy:cy + h // 2, cx:cx + w // 2]))


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

    def __init__(self, w, h, qp, org, search_range=64, fast_search=0, amp=0):
        self.L = L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhmleaf.so"))
        L.ref_coder_get.restype = C.c_ulonglong
        L.ref_coder_set.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ulonglong]
        L.ref_coder_get.argtypes = [C.c_int, C.c_int, C.c_void_p]
        self.w, self.h, self.qp = w, h, qp
        self.n_ctu = L.ref_setup(w, h, qp)
        assert L.ref_search_setup(search_range, fast_search, 1, 1, 1) == 0
        self.fast_search = fast_search
        L.ref_set_amp(1 if amp else 0)
        self.is_p = False
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

    # ---- P pictures
    def setup_p(self, ref_planes, lam):
        """slice type P with `ref_planes` as the one reference picture and the slice lambda (TEncSlice::setUpLambda)"""
        vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        self.L.ref_setup_p.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        assert self.L.ref_setup_p(vp(ref_planes[0]), vp(ref_planes[1]), vp(ref_planes[2]), float(lam)) == 0
        self.L.ref_cabac_reset()                                # P-slice init states for the contexts the oracle does not model
        st = np.zeros(512, np.uint8)
        self.L.ref_cabac_states(st.ctypes.data_as(C.c_void_p))
        self.base = st[:self.n_hm].copy()
        self.is_p = True

    def setup_p_multi(self, refs, ref_pocs, poc, lam):
        """slice type P with several reference pictures: RefPicList0[k] = refs[k] at POC ref_pocs[k], this picture at `poc`"""
        vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        self._keep = [np.ascontiguousarray(a) for r in refs for a in r]
        arr = (C.c_void_p * len(self._keep))(*[a.ctypes.data for a in self._keep])
        pocs = np.ascontiguousarray(ref_pocs, np.int32)
        self.L.ref_setup_p_multi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        assert self.L.ref_setup_p_multi(len(refs), arr, pocs.ctypes.data, int(poc), float(lam)) == 0
        self.L.ref_cabac_reset()
        st = np.zeros(512, np.uint8)
        self.L.ref_cabac_states(st.ctypes.data_as(C.c_void_p))
        self.base = st[:self.n_hm].copy()
        self.is_p = True

    def setup_col_multi(self, ctus_bytes, poc, col_poc, col_ref_pocs):
        """TMVP with several references: the collocated picture = refs[0], decided with the reference POCs `col_ref_pocs`"""
        L, vp = self.L, lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        n = C.sizeof(hmo_py.Ctu)
        for a in range(self.n_ctu):
            c = hmo_py.Ctu.from_buffer_copy(ctus_bytes[a * n:(a + 1) * n])
            L.ref_set_col_ctu(a, vp(np.ctypeslib.as_array(c.pred_mode).copy()), vp(np.ctypeslib.as_array(c.mv).copy()), vp(np.ctypeslib.as_array(c.ref_idx).copy()))
        crp = np.ascontiguousarray(col_ref_pocs, np.int32)
        L.ref_col_finish_multi(int(poc), int(col_poc), crp.ctypes.data_as(C.c_void_p), len(crp))

    def setup_col(self, ctus_bytes, poc):
        """TMVP: the reference picture's decided CTUs (Encoder.all_ctus_bytes()) become the collocated picture's motion field"""
        L, vp = self.L, lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        n = C.sizeof(hmo_py.Ctu)
        for a in range(self.n_ctu):
            c = hmo_py.Ctu.from_buffer_copy(ctus_bytes[a * n:(a + 1) * n])
            L.ref_set_col_ctu(a, vp(np.ctypeslib.as_array(c.pred_mode).copy()), vp(np.ctypeslib.as_array(c.mv).copy()), vp(np.ctypeslib.as_array(c.ref_idx).copy()))
        L.ref_col_finish(poc)

    def load_inter_state(self, enc):
        """inter fields of the decided CTUs (current CTU and its left / above-row neighbours)"""
        L, vp = self.L, lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        cur = enc.cur_ctu()
        for a in range(max(0, cur - self.w_ctu - 1), cur + 1):
            c = enc.ctu_arrays(a)
            L.ref_set_ctu_inter(a, vp(c["skip"]), vp(c["inter_dir"]), vp(c["merge_flag"]), vp(c["mv"]), vp(c["ref_idx"]))
        if self.fast_search:                                    # TZ search state carried from the oracle's own search
            for r, (x, y) in enumerate(enc.test_int_mv()):
                L.ref_set_int_mv(r, x, y)

    def deblock(self, enc, beta=0, tc=0):
        """the reference's loopFilterPic on the oracle's decided picture (its arrays + un-filtered reconstruction)"""
        L, vp = self.L, lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        for a in range(enc.n_ctu):
            c = enc.ctu_arrays(a)
            for fid, name in ((0, "depth"), (1, "part_size"), (2, "pred_mode"), (5, "tr_idx")):
                L.ref_set_ctu_field(a, fid, vp(np.ascontiguousarray(c[name]).view(np.uint8)))
            for k in range(3):
                L.ref_set_ctu_field(a, 9 + k, vp(c["cbf"][k]))
            L.ref_set_ctu_qp(a, vp(np.ascontiguousarray(c["qp"]).astype(np.int8)))
            L.ref_set_ctu_inter(a, vp(c["skip"]), vp(c["inter_dir"]), vp(c["merge_flag"]), vp(c["mv"]), vp(c["ref_idx"]))
        for k in range(3):
            L.ref_set_rec(k, vp(enc.rec[k]))
        L.ref_deblock(beta, tc)
        out = [np.zeros_like(r) for r in enc.rec]
        for k in range(3):
            L.ref_get_rec(k, out[k].ctypes.data_as(C.c_void_p))
        return out

    def inter_cu(self, ctu, zidx, depth, part_size):
        out = RefCuOut()
        assert self.L.ref_inter_cu(ctu, zidx, depth, part_size, C.byref(out)) == 0
        return out

    def merge_cu(self, ctu, zidx, depth, cand, no_res):
        out = RefCuOut()
        cands = np.zeros(15, np.int32)
        n = self.L.ref_merge_cu(ctu, zidx, depth, cand, no_res, C.byref(out), cands.ctypes.data_as(C.c_void_p))
        assert n == 5, n
        return out, cands.reshape(5, 3)

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
