"""The HM adapter executed end to end on the GPU: `TEncCu::compressCtu(TComDataCU *)` of adapter/TEncCuFcu.cpp -- the
method TEncSlice::compressSlice calls (TEncSlice.cpp:1468) -- driven on the REFERENCE's own objects (TComPic, TComSlice,
TComDataCU, TEncCfg, TComRdCost, TComTrQuant of oracle/_ref/libhmleaf.so, which is the reference compiled in this repository's
container plus the adapter; oracle/ref/ref_driver.cpp: ref_adapter_compress_ctu).  The adapter uploads the source and reference
planes it finds in the reference's picture buffers, reads the slice parameters from the reference's objects, calls libfcu.so
(fcu_chain_begin / fcu_chain_set_references / fcu_compress_ctu), and marshals the decided CTU and its reconstruction back.
Checked after every CTU: the adapter's encodeCtu walk over the TComDataCU it filled, on the reference's entropy coder, must
arrive at the oracle's Q15 counter and context states (every coded field of the CTU); after every picture PicYuvRec must equal
the oracle's reconstruction.  Pictures: intra (partial CTUs on both borders) and a lowdelay_P clip with two reference
pictures, TZ search, AMP and TMVP (the collocated motion field is the adapter's own previous picture in HBM)."""
import ctypes as C
import os

import numpy as np
import pytest

import hmo_py
import search_trace as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF = os.path.join(ROOT, "oracle", "_ref", "libhmleaf.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(LEAF), reason="oracle/_ref/libhmleaf.so (built where /root/reference exists) did not travel")]


@pytest.fixture(autouse=True, scope="module")
def _torch_runtime_first():
    """libhmleaf.so pulls in the system's libamdhip64 through libfcu.so; torch ships its own copy.  Whichever is loaded first serves
    the whole process, and a process that initialised the system runtime first leaves torch without a device.  Every other GPU
    test goes through torch, so it is made the first here as well."""
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")


def _ctu_through_the_adapter(R, o, a, tag):
    L = R.L
    assert L.ref_adapter_compress_ctu(a) == 0
    o.compress_ctu(a)
    raw = C.create_string_buffer(C.sizeof(hmo_py.Ctu))
    ahead = L.fcu_adapter_last_record(raw)
    assert ahead >= 0
    got = hmo_py.Ctu.from_buffer_copy(raw.raw)
    for name, _ in hmo_py.Ctu._fields_:                         # the record itself, field by field, before it goes through the reference's coder
        x, y = getattr(got, name), getattr(o.ctu(a), name)
        same = (bytes(x) == bytes(y)) if hasattr(x, "_length_") else (x == y)
        assert same, (tag, a, "fcu_ctu_out field", name, "picture-ahead" if ahead else "per CTU")
    L.ref_cabac_reset_bits()
    L.ref_adapter_encode_ctu(a)
    ctx, frac = o.cabac(full=True)
    stt = np.zeros(512, np.uint8)
    n = L.ref_cabac_states(stt.ctypes.data_as(C.c_void_p))
    assert L.ref_cabac_frac() == frac, (tag, a, "Q15 bit counter of the reference coder after the adapter's encodeCtu")
    assert np.array_equal(R.from_hm(stt[:n])[st.O_SORTED], ctx[st.O_SORTED]), (tag, a, "context states")


def _same_reconstruction(R, o, tag):
    for c in range(3):
        got = np.zeros_like(o.rec[c])
        R.L.ref_get_rec(c, got.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, o.rec[c]), (tag, "PicYuvRec", c)


@pytest.mark.parametrize("gen,w,h,qp", [("mixed", 136, 72, 32), ("textured", 200, 136, 27)])
def test_intra_picture_through_compressCtu(pkg, gen, w, h, qp):
    f = getattr(pkg.synth, gen)(w, h, seed=3)
    o = hmo_py.Encoder(*f, qp)
    R = st.RefSearch(w, h, qp, f)
    R.L.ref_cabac_frac.restype = C.c_ulonglong
    R.L.ref_adapter_release()
    R.L.ref_cabac_reset()
    for a in range(o.n_ctu):
        _ctu_through_the_adapter(R, o, a, gen)
    _same_reconstruction(R, o, gen)
    R.L.ref_adapter_release()


@pytest.mark.parametrize("gen,w,h,qp,sl", [("textured", 200, 136, 27, 5), ("mixed", 264, 136, 32, 2)])
def test_slices_of_a_picture_decided_ahead(pkg, gen, w, h, qp, sl):
    """SliceMode 1: with the first CTU of the first slice the adapter binds every slice of the picture as a chain of its own,
    decides them in one launch and then serves compressCtu from the results.  The reference's slice object is moved from
    slice to slice as TEncGOP's slice loop does; every CTU and the picture must equal the oracle run with the same slices."""
    f = getattr(pkg.synth, gen)(w, h, seed=6)
    o = hmo_py.Encoder(*f, qp, slice_ctus=sl)
    R = st.RefSearch(w, h, qp, f)
    L = R.L
    L.ref_cabac_frac.restype = C.c_ulonglong
    L.ref_adapter_release()
    L.ref_adapter_slices(sl)
    try:
        for first in range(0, o.n_ctu, sl):
            n = min(sl, o.n_ctu - first)
            L.ref_set_slice_range(first, n)
            L.ref_cabac_reset()                                  # TEncSlice::compressSlice: resetEntropy at the start of a slice
            for a in range(first, first + n):
                _ctu_through_the_adapter(R, o, a, f"{gen} slice at {first}")
        _same_reconstruction(R, o, gen)
    finally:
        L.ref_adapter_slices(0)
        L.ref_adapter_release()


@pytest.mark.parametrize("w,h,sl", [(136, 72, 0), (136, 72, 2), (192, 128, 3)])
def test_lowdelay_p_clip_through_compressCtu(pkg, w, h, sl):
    """sl = 0: one slice per picture.  136x72 with 2 CTUs per slice: slices begin on the 8-sample-wide CTUs of the last column, so
    the adapter runs them one after the other and carries the TZ search state (m_integerMv2Nx2N) from slice to slice and from
    picture to picture as HM's encoder object does.  192x128 with 3 CTUs per slice: every slice begins on a whole CTU and the
    picture's slices are decided ahead in one launch."""
    gen, base_qp, n_pic, sr, nref = "shear_mixed", 30, 4, 16, 2
    search_state = [(0, 0)] * 4
    dpb = []                                                    # (poc, deblocked planes, the POCs its list 0 named)
    prev_ctus = None
    n_inter = n_far = 0
    L = None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, 9, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        R = st.RefSearch(w, h, qp, f, search_range=sr, fast_search=1, amp=1)
        if L is None:
            R.L.ref_adapter_release()
        L = R.L
        L.ref_cabac_frac.restype = C.c_ulonglong
        L.ref_adapter_slices(sl)
        if poc == 0:
            o = hmo_py.Encoder(*f, qp, slice_ctus=sl, lambda_override=lam)
            L.ref_set_poc(0)
            L.ref_set_lambda.argtypes = [C.c_double]
            L.ref_set_lambda(float(lam))
            pocs = []
        else:
            rl = dpb[-nref:][::-1]
            pocs = [r[0] for r in rl]
            crp = rl[0][2] or [rl[0][0] - 1]
            o = hmo_py.Encoder(*f, qp, slice_ctus=sl, refs=[r[1] for r in rl], ref_pocs=pocs, poc=poc, col=prev_ctus, col_ref_pocs=crp,
                               lambda_override=lam, search_range=sr, fast_search=1, amp=1)
            o.set_int_mv(search_state)                          # the encoder's search state crosses pictures (TEncSearch.h:123)
            R.setup_p_multi([r[1] for r in rl], pocs, poc, lam)
            R.setup_col_multi(prev_ctus, poc, pocs[0], crp)
        for first in range(0, o.n_ctu, sl or o.n_ctu):
            n = min(sl or o.n_ctu, o.n_ctu - first)
            if sl:
                L.ref_set_slice_range(first, n)
            L.ref_cabac_reset()
            for a in range(first, first + n):
                _ctu_through_the_adapter(R, o, a, f"poc{poc}")
                A = o.ctu_arrays(a)
                n_inter += int((A["pred_mode"] == 0).sum())
                n_far += int(((A["pred_mode"] == 0) & (A["ref_idx"] > 0)).sum())
        _same_reconstruction(R, o, f"poc{poc}")
        prev_ctus = o.all_ctus_bytes()
        search_state = o.test_int_mv()
        o.deblock()
        dpb.append((poc, [p.copy() for p in o.rec], pocs))
    assert n_inter > 0 and n_far > 0
    L.ref_adapter_slices(0)
    L.ref_adapter_release()
