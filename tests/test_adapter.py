"""The HM adapter (adapter/TEncCuFcu.cpp + adapter/fcu_marshal.h: TEncCu's public methods over libfcu.so) against the
REFERENCE's classes.  build_ref.sh compiles the adapter against the reference's own headers and links it into
oracle/_ref/libhmleaf.so next to the reference's TComDataCU / TEncEntropy / TEncSbac; here fcu_ctu_out records (same bytes
the engine publishes; the oracle's HmoCtu has the same layout) go through the adapter's marshalling into a real TComDataCU
and through the adapter's TEncCu::encodeCtu into the reference's entropy coder:
  * intra pictures: after every CTU the Q15 counter and all context states must equal tests/golden/syntax_*.npz (produced
    by the reference's coder, driven node by node by make_golden_syntax.py -- an independent walk);
  * P pictures: the reference coder, fed by the adapter, must arrive at the state the oracle's own CTU replay reaches
    (skip / merge / AMVP syntax, inter transform trees) -- which pins that replay through the reference's TEncSbac.
No GPU here: compressCtu itself (the device calls) runs in tests/test_gpu_adapter.py, on the same reference objects."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import hmo_py
import search_trace as st
from test_golden_leaf import HM2O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF = os.path.join(ROOT, "oracle", "_ref", "libhmleaf.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LEAF), reason="oracle/_ref/libhmleaf.so is built where /root/reference exists")
GOLD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "syntax_*.npz")))


def _raw(enc, a):
    return C.string_at(C.addressof(enc.ctu(a)), C.sizeof(hmo_py.Ctu))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_intra_ctus_through_the_adapter_into_the_reference_coder(built, pkg, path):
    g = np.load(path)
    w, h, qp = int(g["width"]), int(g["height"]), int(g["qp"])
    f = getattr(pkg.synth, str(g["generator"]))(w, h, seed=int(g["seed"]))
    assert C.sizeof(hmo_py.Ctu) == pkg.engine.CTU_OUT_BYTES
    o = hmo_py.Encoder(*f, qp)
    R = st.RefSearch(w, h, qp, f)
    L = R.L
    L.ref_cabac_frac.restype = C.c_ulonglong
    L.ref_cabac_reset()
    for a in range(o.n_ctu):
        o.compress_ctu(a)
        L.ref_adapter_marshal(a, _raw(o, a))
        L.ref_cabac_reset_bits()
        L.ref_adapter_encode_ctu(a)
        assert L.ref_cabac_frac() == int(g["frac"][a]), (a, "Q15 bit counter")
        assert L.ref_cabac_bits() == int(g["bits"][a]) == o.replay_bits(a)
        stt = np.zeros(512, np.uint8)
        n = L.ref_cabac_states(stt.ctypes.data_as(C.c_void_p))
        assert np.array_equal(stt[:n], g["states"][a]), (a, "context states")


@pytest.mark.parametrize("gen,w,h,base_qp,n_pic,sr,amp", [("mixed", 136, 72, 27, 3, 8, 0), ("textured", 128, 64, 35, 3, 16, 0),
                                                           ("shear_textured", 192, 128, 27, 3, 16, 1)])      # amp: asymmetric partitions in the coded CTUs
def test_p_ctus_through_the_adapter_into_the_reference_coder(built, pkg, gen, w, h, base_qp, n_pic, sr, amp):
    prev = None
    n_inter = n_amp = 0
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, 5, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        o = hmo_py.Encoder(*f, qp, lambda_override=lam) if poc == 0 else hmo_py.Encoder(*f, qp, ref=prev, lambda_override=lam, search_range=sr, amp=amp)
        if poc:
            R = st.RefSearch(w, h, qp, f, search_range=sr, amp=amp)
            R.setup_p(prev, lam)
            L = R.L
            L.ref_cabac_frac.restype = C.c_ulonglong
            L.ref_cabac_reset()
        for a in range(o.n_ctu):
            o.compress_ctu(a)
            if not poc:
                continue
            L.ref_adapter_marshal(a, _raw(o, a))
            L.ref_cabac_reset_bits()
            L.ref_adapter_encode_ctu(a)
            ctx, frac = o.cabac(full=True)
            stt = np.zeros(512, np.uint8)
            n = L.ref_cabac_states(stt.ctypes.data_as(C.c_void_p))
            got = R.from_hm(stt[:n])
            assert L.ref_cabac_frac() == frac, (poc, a, "Q15 bit counter")
            assert np.array_equal(got[st.O_SORTED], ctx[st.O_SORTED]), (poc, a, "context states")
            A = o.ctu_arrays(a)
            n_inter += int((A["pred_mode"] == 0).sum())
            n_amp += int(((A["pred_mode"] == 0) & (A["part_size"] >= 4) & (A["part_size"] <= 7)).sum())
        o.deblock()
        prev = [p.copy() for p in o.rec]
    assert n_inter > 0 and (n_amp > 0) == bool(amp)


def test_plane_converters_and_islice_cost(built, pkg):
    w, h = 200, 136                                            # partial CTUs on both borders
    f = pkg.synth.mixed(w, h, 4)
    R = st.RefSearch(w, h, 30, f)
    L = R.L
    for c in range(3):                                         # 8-bit plane -> PicYuvRec (CTU blocks) -> 8-bit plane
        out = np.zeros_like(f[c])
        L.ref_adapter_planes_roundtrip(c, f[c].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, f[c])
    # updateCtuDataISlice: 8x8 Hadamard amplitudes without DC over the whole 8x8 blocks of the CTU's original luma
    from scipy.linalg import hadamard
    H = hadamard(8).astype(np.int64)
    wc = (w + 63) // 64
    for a in range(R.n_ctu):
        x0, y0 = (a % wc) * 64, (a // wc) * 64
        bw, bh = min(64, w - x0), min(64, h - y0)
        want = 0
        for by in range(0, bh - 7, 8):
            for bx in range(0, bw - 7, 8):
                t = H @ f[0][y0 + by:y0 + by + 8, x0 + bx:x0 + bx + 8].astype(np.int64) @ H.T
                want += (int(np.abs(t).sum() - abs(t[0, 0])) + 2) >> 2
        assert L.ref_adapter_isl_cost(a, bw, bh) == want, a
