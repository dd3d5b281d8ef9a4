"""P pictures through the ENGINE SOURCE on the CPU wave emulator (tests/emu, test-only build of csrc/fcu_engine.h +
fcu_inter.h) and the deblocking kernel source compiled for the CPU (tests/emu/dbk_emu.cpp): every fcu_ctu_out field,
the reconstruction and the CABAC state against the oracle CTU by CTU, and every deblocked P picture against the CRC the
REFERENCE's own TComLoopFilter produced for the same clip (tests/golden/inter_*.npz)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import emu_py
import hmo_py
import search_trace as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.CASES


def _tmvp(case):
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.TMVP.get(case, 0)


def _mref(case):
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.MREF.get(case, 1)


def _fast(case):
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.FAST_SEARCH.get(case, 0)


def _amp(case):
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.AMP.get(case, 0)


def dbk_emu(out_arr, rec, w, h, beta=0, tc=0):
    lib = C.CDLL(os.path.join(ROOT, "tests", "emu", "libdbk_emu.so"))
    lib.dbk_emu.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
    lib.dbk_emu(C.addressof(out_arr), rec[0].ctypes.data, rec[1].ctypes.data, rec[2].ctypes.data, w, h, beta, tc)


@pytest.mark.parametrize("case", ["mixed_qp27", "textured_qp37", "tz_mixed_qp27", "tz_textured_qp32", "tmvp_mixed_qp30", "tmvp_textured_qp35",
                                  "amp_textured_qp27", "amp_mixed_qp30", "amp_shear_qp27",
                                  "mr2_mixed_qp30", "mr4_textured_qp32", "mr3_amp_shear_qp27"])     # mr<n>: n reference pictures in list 0
def test_emulated_engine_p_pictures(case, built, pkg):
    gen, w, h, base_qp, seed, n_pic, sr = _cases()[case]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"inter_{case}.npz"))
    prev = None
    prev_ctus = None
    n_inter = n_skip = n_amp = n_far = 0
    nref = _mref(case)
    dpb = []                                                     # (poc, deblocked planes, the POCs its list 0 named)
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, seed, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        if poc == 0:
            o, e = hmo_py.Encoder(*f, qp, lambda_override=lam), emu_py.EmuEncoder(*f, qp, lam=lam)
        else:
            col = prev_ctus if _tmvp(case) else None
            kw = dict(ref=prev)
            if nref > 1:
                rl = dpb[-nref:][::-1]                           # RefPicList0: most recent first
                kw = dict(refs=[r[1] for r in rl], ref_pocs=[r[0] for r in rl], poc=poc, col_ref_pocs=rl[0][2] or [rl[0][0] - 1])
            o = hmo_py.Encoder(*f, qp, col=col, lambda_override=lam, search_range=sr, fast_search=_fast(case), amp=_amp(case), **kw)
            e = emu_py.EmuEncoder(*f, qp, lam=lam, search_range=sr, fast_search=_fast(case), col=col, amp=_amp(case), **kw)
        for a in range(o.n_ctu):
            o.compress_ctu(a)
            e.compress_ctu(a)
            A, B = o.ctu_arrays(a), e.ctu_arrays(a)
            for k, v in A.items():
                assert (np.array_equal(v, B[k]) if isinstance(v, np.ndarray) else v == B[k]), (poc, a, k)
            (ca, fa), (cb, fb) = o.cabac(full=True), e.cabac(full=True)
            assert fa == fb and np.array_equal(ca[st.O_SORTED], cb[st.O_SORTED]), (poc, a, "CABAC state")
            n_inter += int((A["pred_mode"] == 0).sum())
            n_skip += int(A["skip"].sum())
            n_far += int(((A["ref_idx"] > 0) & (A["pred_mode"] == 0)).sum())
            n_amp += int(((A["part_size"] >= 4) & (A["part_size"] <= 7) & (A["pred_mode"] == 0)).sum()) if _amp(case) else 0
        for p, q in zip(o.rec, e.rec):
            assert np.array_equal(p, q), (poc, "reconstruction")
        prev_ctus = bytes(e.out)                                 # the engine's own array is the next picture's motion field (TMVP)
        assert prev_ctus == o.all_ctus_bytes()
        dbk_emu(e.out, e.rec, w, h)                              # kernel source on the CPU, in place
        if poc:
            assert [st.crc(p) for p in e.rec] == [int(v) for v in g[f"deblock_{poc}"][:3]], (poc, "deblocked picture vs the reference's loop filter")
        o.deblock()
        for p, q in zip(o.rec, e.rec):
            assert np.array_equal(p, q), (poc, "deblocked picture vs oracle")
        prev = [a.copy() for a in e.rec]
        dpb.append((poc, prev, kw.get("ref_pocs", [poc - 1]) if poc else []))
    assert n_inter > 0 and n_skip > 0
    if nref > 1:
        print("partitions predicted from a picture other than the nearest:", n_far)
        assert n_far > 0
    if case == "amp_shear_qp27":
        print("partitions of asymmetric CUs:", n_amp)
        assert n_amp > 0                                         # asymmetric partitions survive into the decided pictures


def test_p_slice_started_from_the_b_tables(built, pkg):
    """cabac_init_flag: a P picture whose contexts start from the B-slice tables (the encoder's choice after the previous
    slice, TEncSbac.cpp:111-115).  The initial states of both tables are pinned by the reference's own resetEntropy at every
    QP (tests/golden/cabac_init.npz, oracle/ref/make_golden_cabac_init.py); the engine source on the wave emulator follows
    the oracle through a P picture started that way -- and decides differently from the same picture started from the P
    tables."""
    import ctypes as C
    g = np.load(os.path.join(ROOT, "tests", "golden", "cabac_init.npz"))["states"]
    lib = hmo_py.load()
    for qp in range(52):
        for k, (stype, b) in enumerate(((hmo_py.SLICE_I, 0), (hmo_py.SLICE_P, 0), (hmo_py.SLICE_P, 1))):
            c = hmo_py.Cabac()
            lib.hmo_cabac_init_tab(C.byref(c), qp, stype, b)
            idx = st.O_SORTED if k else st.O_SORTED[st.O_SORTED < 159]
            assert np.array_equal(np.ctypeslib.as_array(c.ctx)[idx], g[qp, k][idx]), (qp, k)
    assert (g[32, 1] != g[32, 2]).sum() > 30
    w, h, base_qp, sr = 128, 64, 30, 8
    f0, f1 = st.moving_frame(pkg.synth, "mixed", w, h, 5, 0), st.moving_frame(pkg.synth, "mixed", w, h, 5, 1)
    _, qp0, lam0 = hmo_py.ldp_slice(0, base_qp)
    o = hmo_py.Encoder(*f0, qp0, lambda_override=lam0)
    o.compress_frame(); o.deblock()
    prev = [a.copy() for a in o.rec]
    _, qp, lam = hmo_py.ldp_slice(1, base_qp)
    states = {}
    for b in (0, 1):
        o = hmo_py.Encoder(*f1, qp, ref=prev, lambda_override=lam, search_range=sr, fast_search=1, cabac_b_table=b)
        e = emu_py.EmuEncoder(*f1, qp, ref=prev, lam=lam, search_range=sr, fast_search=1, cabac_b_table=b)
        for a in range(o.n_ctu):
            o.compress_ctu(a); e.compress_ctu(a)
            A, B = o.ctu_arrays(a), e.ctu_arrays(a)
            for k, v in A.items():
                assert (np.array_equal(v, B[k]) if isinstance(v, np.ndarray) else v == B[k]), (b, a, k)
            (ca, fa), (cb, fb) = o.cabac(full=True), e.cabac(full=True)
            assert fa == fb and np.array_equal(ca[st.O_SORTED], cb[st.O_SORTED]), (b, a, "CABAC state")
        states[b] = (o.cabac(full=True), [o.ctu_arrays(a)["total_bits"] for a in range(o.n_ctu)])
    assert states[0][1] != states[1][1] or not np.array_equal(states[0][0][0], states[1][0][0])


def test_random_small_clips_on_the_emulator(built, pkg):
    """A CPU-side slice of tests/parity_sweep_ldp.py: random small lowdelay clips (global or sheared motion, QP, search range,
    TZ / full search, TMVP, AMP, slices) through the engine source on the wave emulator against the oracle, every field of
    every CTU.  Fixed seed: the cases are the same in every run."""
    rng = np.random.default_rng(8081)
    for case in range(16):
        w, h = int(rng.integers(8, 21)) * 8, int(rng.integers(8, 17)) * 8
        base_qp = int(rng.integers(18, 42))
        gen = ["smooth", "mixed", "textured"][int(rng.integers(0, 3))]
        if rng.random() < 0.5:
            gen = "shear_" + gen
        sr = int(rng.choice([4, 8, 16]))
        fast, tmvp, amp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        n_ctu = ((w + 63) // 64) * ((h + 63) // 64)
        sl = int(rng.choice([0, 1, n_ctu]))
        seed = int(rng.integers(0, 1000))
        prev = prev_ctus = None
        for poc in range(2):
            f = st.moving_frame(pkg.synth, gen, w, h, seed, poc, shift=(int(rng.integers(-4, 5)), int(rng.integers(-2, 3))) if poc else (0, 0))
            _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
            if not 0 <= qp <= 51:
                break
            if poc == 0:
                o, e = hmo_py.Encoder(*f, qp, slice_ctus=sl, lambda_override=lam), emu_py.EmuEncoder(*f, qp, slice_ctus=sl, lam=lam)
            else:
                col = prev_ctus if tmvp else None
                o = hmo_py.Encoder(*f, qp, slice_ctus=sl, ref=prev, col=col, lambda_override=lam, search_range=sr, fast_search=fast, amp=amp)
                e = emu_py.EmuEncoder(*f, qp, slice_ctus=sl, ref=prev, lam=lam, search_range=sr, fast_search=fast, col=col, amp=amp)
            for a in range(o.n_ctu):
                o.compress_ctu(a)
                e.compress_ctu(a)
                A, B = o.ctu_arrays(a), e.ctu_arrays(a)
                for k, v in A.items():
                    assert (np.array_equal(v, B[k]) if isinstance(v, np.ndarray) else v == B[k]), (case, gen, w, h, poc, a, k)
            for p, q in zip(o.rec, e.rec):
                assert np.array_equal(p, q), (case, poc, "reconstruction")
            prev_ctus = bytes(e.out)
            dbk_emu(e.out, e.rec, w, h)
            o.deblock()
            for p, q in zip(o.rec, e.rec):
                assert np.array_equal(p, q), (case, poc, "deblocked picture")
            prev = [a.copy() for a in e.rec]
