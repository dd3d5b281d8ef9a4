"""GPU parity of the P-slice path (BASELINE configs[4]): the HIP engine through the C ABI against the oracle (which is
pinned candidate by candidate by the reference's own inter search, tests/test_golden_inter.py) -- every fcu_ctu_out field
incl. motion, the reconstruction, the CABAC state, the deblocked pictures; and the deblocked pictures of the 416x240 clip
against the CRCs the reference's own loop filter produced (tests/golden/inter_smooth416_qp32.npz)."""
import os

import numpy as np
import pytest

import hmo_py
import search_trace as st

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same_ctu(got, want, tag):
    for k, v in want.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, got[k]), f"{tag}: field {k} differs at {np.argwhere(v != got[k])[:4].tolist()}"
        else:
            assert v == got[k], f"{tag}: {k}: engine {got[k]} oracle {v}"


@pytest.mark.parametrize("gen,w,h,base_qp,n_pic,sr,fast,tmvp,amp,btab", [c + (0,) for c in [("mixed", 136, 72, 27, 3, 8, 0, 0, 0), ("textured", 192, 128, 32, 3, 16, 0, 0, 0), ("smooth", 128, 64, 37, 5, 64, 0, 0, 0),
                                                                        ("mixed", 136, 72, 27, 3, 16, 1, 0, 0), ("textured", 192, 128, 32, 3, 64, 1, 0, 0),      # fast = 1: TZ search (FastSearch 1)
                                                                        ("mixed", 192, 128, 30, 4, 16, 1, 1, 0), ("textured", 136, 72, 35, 4, 32, 0, 1, 0),     # tmvp = 1: temporal candidates
                                                                        ("shear_textured", 192, 128, 27, 3, 16, 1, 0, 1), ("shear_mixed", 136, 72, 32, 4, 32, 1, 1, 1),   # amp = 1: asymmetric partitions
                                                                        ("mixed", 192, 128, 30, 3, 16, 0, 0, 1)]] +
                         [("mixed", 136, 72, 30, 3, 16, 1, 1, 1, 1)])     # btab = 1: P pictures started from the B-slice context tables (cabac_init_flag)
def test_p_pictures_ctu_by_ctu(pkg, gen, w, h, base_qp, n_pic, sr, fast, tmvp, amp, btab):
    """compressCtu-shaped calls: every CTU of every picture of a short lowdelay_P clip, CABAC state after every CTU."""
    eng = pkg.CuEngine(w, h, max_chains=1)
    prev, prev_pad, prev_out, prev_ctus = None, None, None, None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, 5, poc)
        fp = pkg.engine.ldp_slice(base_qp, poc)
        fp.search_range = sr
        fp.fast_search = fast
        fp.tmvp = 1 if (tmvp and poc) else 0
        fp.amp = amp
        fp.cabac_b_table = 1 if (btab and poc) else 0
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        assert fp.qp == qp and fp.lambda_ == lam and fp.slice_type == (0 if poc == 0 else 1)
        eng.init_chain(0, f, fp.qp, params=fp, ref=prev_pad, col=prev_out if (tmvp and poc) else None)
        ref = hmo_py.Encoder(*f, qp, lambda_override=lam) if poc == 0 else \
            hmo_py.Encoder(*f, qp, ref=prev, col=prev_ctus if tmvp else None, lambda_override=lam, search_range=sr, fast_search=fast, amp=amp, cabac_b_table=btab)
        for a in range(eng.n_ctu):
            got = eng.compress_ctu(0, a)
            ref.compress_ctu(a)
            _same_ctu(got, ref.ctu_arrays(a), f"{gen} poc{poc} ctu{a}")
            (ce, fe), (co, fo) = eng.ctx_state(0, full=True), ref.cabac(full=True)
            assert fe == fo and np.array_equal(ce[st.O_SORTED], co[st.O_SORTED]), f"CABAC state poc{poc} ctu{a}"
        for p, q in zip(eng.rec_planes(0), ref.rec):
            assert np.array_equal(p, q), f"reconstruction poc{poc}"
        prev_out, prev_ctus = eng._keep[0][2], ref.all_ctus_bytes()     # the picture's fcu_ctu_out array stays in HBM as the next picture's motion field
        assert bytes(prev_out.cpu().numpy()) == prev_ctus
        eng.deblock(0)
        eng.sync()
        ref.deblock()
        for p, q in zip(eng.rec_planes(0), ref.rec):
            assert np.array_equal(p, q), f"deblocked picture poc{poc}"
        prev = [a.copy() for a in ref.rec]
        prev_pad = eng.pad_reference(eng._keep[0][1])
        import emu_py
        for t, q in zip(prev_pad, emu_py.pad_planes(prev)):       # fcu_pad_reference == replicated border
            assert np.array_equal(t.cpu().numpy(), q.ravel())
    eng.destroy()


@pytest.mark.parametrize("case", ["mr2_mixed_qp30", "mr4_textured_qp32", "mr3_amp_shear_qp27"])
def test_several_reference_pictures(pkg, case):
    """RefPicList0 with 2..4 pictures (the lowdelay cfg lists four): reference-index loop, ref_idx syntax, POC-scaled spatial and
    collocated predictors.  The clips are those of tests/golden/inter_mr*.npz, where every candidate of the oracle is checked
    against the reference's own search; here the HIP engine against the oracle CTU by CTU, and its deblocked pictures against the
    CRCs of the reference's loop filter."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgi", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    gen, w, h, base_qp, seed, n_pic, sr = m.CASES[case]
    nref, tmvp, fast, amp = m.MREF[case], m.TMVP.get(case, 0), m.FAST_SEARCH.get(case, 0), m.AMP.get(case, 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"inter_{case}.npz"))
    eng = pkg.CuEngine(w, h, max_chains=1)
    dpb = []                                                    # (poc, deblocked planes, padded device planes, the POCs its list 0 named)
    prev_out, prev_ctus, n_far = None, None, 0
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, seed, poc)
        fp = pkg.engine.ldp_slice(base_qp, poc)
        fp.search_range, fp.fast_search, fp.amp, fp.tmvp = sr, fast, amp, 1 if (tmvp and poc) else 0
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        if poc == 0:
            eng.init_chain(0, f, fp.qp, params=fp)
            ref = hmo_py.Encoder(*f, qp, lambda_override=lam)
            pocs = []
        else:
            rl = dpb[-nref:][::-1]
            pocs = [r[0] for r in rl]
            crp = rl[0][3] or [rl[0][0] - 1]
            eng.init_chain(0, f, fp.qp, params=fp, refs=[r[2] for r in rl], ref_pocs=pocs, poc=poc, col_ref_pocs=crp, col=prev_out if tmvp else None)
            ref = hmo_py.Encoder(*f, qp, refs=[r[1] for r in rl], ref_pocs=pocs, poc=poc, col_ref_pocs=crp, col=prev_ctus if tmvp else None,
                                 lambda_override=lam, search_range=sr, fast_search=fast, amp=amp)
        for a in range(eng.n_ctu):
            got = eng.compress_ctu(0, a)
            ref.compress_ctu(a)
            want = ref.ctu_arrays(a)
            _same_ctu(got, want, f"{case} poc{poc} ctu{a}")
            (ce, fe), (co, fo) = eng.ctx_state(0, full=True), ref.cabac(full=True)
            assert fe == fo and np.array_equal(ce[st.O_SORTED], co[st.O_SORTED]), f"CABAC state poc{poc} ctu{a}"
            n_far += int(((want["ref_idx"] > 0) & (want["pred_mode"] == 0)).sum())
        prev_out, prev_ctus = eng._keep[0][2], ref.all_ctus_bytes()
        assert bytes(prev_out.cpu().numpy()) == prev_ctus
        eng.deblock(0)
        eng.sync()
        planes = [p.copy() for p in eng.rec_planes(0)]
        if poc:
            assert [st.crc(p) for p in planes] == [int(v) for v in g[f"deblock_{poc}"][:3]], "deblocked picture vs the reference's own loop filter"
        dpb.append((poc, planes, eng.pad_reference(eng._keep[0][1]), pocs))
    assert n_far > 0                                           # partitions predicted from a picture other than the nearest
    eng.destroy()


def test_lowdelay_driver_with_the_cfg_reference_picture_sets(pkg):
    """LowDelayPDecider with n_refs = 4 and the reference picture sets of HM's lowdelay_P cfg (previous picture + GOP-boundary
    pictures; pictures 6.. have three references), two slices per picture, TZ + AMP + TMVP: every picture against the oracle
    run with the same lists."""
    gen, w, h, base_qp, n_pic, sr, sl = "mixed", 192, 128, 30, 8, 16, 3
    dec = pkg.lowdelay.LowDelayPDecider(w, h, base_qp, n_clips=1, search_range=sr, slice_ctus=sl, fast_search=1, amp=True, tmvp=True, n_refs=4)
    dpb = {}
    prev_ctus = None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, 11, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        r = dec.decide_picture([f])[0]
        rl = pkg.lowdelay.ref_pocs(poc, 4)
        if poc == 0:
            ref = hmo_py.Encoder(*f, qp, slice_ctus=sl, lambda_override=lam)
        else:
            assert r.get("ref_pocs", rl) == rl
            ref = hmo_py.Encoder(*f, qp, slice_ctus=sl, refs=[dpb[q][0] for q in rl], ref_pocs=rl, poc=poc, col_ref_pocs=dpb[rl[0]][1] or [rl[0] - 1],
                                 col=prev_ctus, lambda_override=lam, search_range=sr, fast_search=1, amp=1, search_state_per_slice=1)
        ref.compress_frame()
        for a in range(ref.n_ctu):
            _same_ctu(dec.eng.ctu_out(0, a), ref.ctu_arrays(a), f"poc{poc} ctu{a}")
        prev_ctus = ref.all_ctus_bytes()
        ref.deblock()
        for p, q in zip(r["rec"], ref.rec):
            assert np.array_equal(p.cpu().numpy(), q), f"deblocked picture poc{poc}"
        dpb[poc] = ([a.copy() for a in ref.rec], rl)
    assert len(pkg.lowdelay.ref_pocs(7, 4)) == 3 and len(dec.dpb[0]) <= 4
    dec.close()


def test_slice_chains_that_begin_on_partial_ctus(pkg):
    """Slices decided side by side as independent chains start the TZ search's carried start point (m_integerMv2Nx2N) from zero;
    HM's encoder carries it from the slice before.  The two differ only where a slice begins with a CTU too small for a 64x64 CU:
    136x72 with two CTUs per slice puts slice starts on the 8-sample-wide last column.  The oracle follows the engine's convention
    with search_state_per_slice = 1 (with 0 -- HM's carry -- picture 3 of this clip differs in two CTUs)."""
    gen, w, h, base_qp, n_pic, sr, nref, sl = "shear_mixed", 136, 72, 30, 4, 16, 2, 2
    dec = pkg.lowdelay.LowDelayPDecider(w, h, base_qp, n_clips=1, search_range=sr, slice_ctus=sl, fast_search=1, amp=True, tmvp=True, n_refs=nref, rps="recent")
    dpb, prev_ctus = [], None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, 9, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        dec.decide_picture([f])
        if poc == 0:
            o, pocs = hmo_py.Encoder(*f, qp, slice_ctus=sl, lambda_override=lam), []
        else:
            rl = dpb[-nref:][::-1]
            pocs = [x[0] for x in rl]
            o = hmo_py.Encoder(*f, qp, slice_ctus=sl, refs=[x[1] for x in rl], ref_pocs=pocs, poc=poc, col=prev_ctus, col_ref_pocs=rl[0][2] or [rl[0][0] - 1],
                               lambda_override=lam, search_range=sr, fast_search=1, amp=1, search_state_per_slice=1)
        o.compress_frame()
        for a in range(o.n_ctu):
            _same_ctu(dec.eng.ctu_out(0, a), o.ctu_arrays(a), f"poc{poc} ctu{a}")
        prev_ctus = o.all_ctus_bytes()
        o.deblock()
        dpb.append((poc, [p.copy() for p in o.rec], pocs))
    dec.close()


def test_ldp_416x240_clip_matches_oracle_and_reference_loop_filter(pkg):
    """The >= 3-picture 416x240 lowdelay_P clip of the golden fixture through the batched driver (one launch per picture)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgi", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    gen, w, h, base_qp, seed, n_pic, sr = m.CASES["smooth416_qp32"]
    g = np.load(os.path.join(ROOT, "tests", "golden", "inter_smooth416_qp32.npz"))
    dec = pkg.lowdelay.LowDelayPDecider(w, h, base_qp, n_clips=1, search_range=sr, fast_search=0)
    prev = None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, gen, w, h, seed, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        r = dec.decide_picture([f])[0]
        ref = hmo_py.Encoder(*f, qp, lambda_override=lam) if poc == 0 else hmo_py.Encoder(*f, qp, ref=prev, lambda_override=lam, search_range=sr)
        ref.compress_frame()
        for a in range(ref.n_ctu):
            _same_ctu(dec.eng.ctu_out(0, a), ref.ctu_arrays(a), f"poc{poc} ctu{a}")
        for p, q in zip(r["rec_unfiltered"], ref.rec):
            assert np.array_equal(p.cpu().numpy(), q)
        ref.deblock()
        planes = [p.cpu().numpy() for p in r["rec"]]
        for p, q in zip(planes, ref.rec):
            assert np.array_equal(p, q)
        if poc:
            assert [st.crc(p) for p in planes] == [int(v) for v in g[f"deblock_{poc}"][:3]], "deblocked P picture vs the reference's own loop filter"
        prev = [a.copy() for a in ref.rec]
    dec.close()


@pytest.mark.parametrize("gen,fast,amp,rows,sr,tmvp", [("textured", 0, 0, (0, 17, 33), 16, 0), ("shear_textured", 1, 1, (0, 16, 33), 16, 0),
                                                       ("textured", 1, 0, (17,), 64, 0), ("shear_textured", 1, 1, (9, 33), 64, 1)])
def test_4k_pair_ctu_rows(pkg, gen, fast, amp, rows, sr, tmvp):
    """BASELINE configs[4] at full size: a 3840x2160 picture pair.  Picture 0 (intra) is decided, deblocked and padded on
    the GPU (that path has its own 4K parity test); picture 1 (P, one CTU row per slice) is compared with the oracle on
    whole CTU rows: top, interior and the partial bottom row at SearchRange 16 with the full search and with TZ search +
    asymmetric partitions (content whose motion boundaries sit on CU quarters); an interior row at the bench's and the
    reference cfg's SearchRange 64 with TZ search; an interior and the bottom row at SearchRange 64 with TZ + AMP + TMVP (the
    collocated picture = picture 0's fcu_ctu_out array in HBM)."""
    w, h, base_qp, sl = 3840, 2160, 32, 60
    dec = pkg.lowdelay.LowDelayPDecider(w, h, base_qp, n_clips=1, search_range=sr, slice_ctus=sl, fast_search=fast, amp=bool(amp), tmvp=bool(tmvp))
    f0 = st.moving_frame(pkg.synth, gen, w, h, 7, 0)
    r0 = dec.decide_picture([f0])[0]
    prev = [p.cpu().numpy() for p in r0["rec"]]
    f1 = st.moving_frame(pkg.synth, gen, w, h, 7, 1)
    r1 = dec.decide_picture([f1])[0]
    _, qp, lam = hmo_py.ldp_slice(1, base_qp)
    col = bytes(r0["out"].cpu().numpy()) if tmvp else None
    ref = hmo_py.Encoder(*f1, qp, slice_ctus=sl, ref=prev, col=col, lambda_override=lam, search_range=sr, fast_search=fast, amp=amp,
                         search_state_per_slice=1)           # the rows are slice chains decided side by side (the bottom row begins on 48-sample-high CTUs)
    n_inter = 0
    for row in rows:
        for a in range(row * 60, row * 60 + 60):
            ref.compress_ctu(a)
            want = ref.ctu_arrays(a)
            _same_ctu(dec.eng.ctu_out(0, a), want, f"4K P picture ctu{a}")
            n_inter += int((want["pred_mode"] == 0).sum())
        y0, y1 = row * 64, min(h, row * 64 + 64)
        for k, (p, q) in enumerate(zip(r1["rec_unfiltered"], ref.rec)):
            a0, a1 = (y0 >> (1 if k else 0)), (y1 >> (1 if k else 0))
            assert np.array_equal(p.cpu().numpy()[a0:a1], q[a0:a1]), f"reconstruction rows of CTU row {row}"
    assert n_inter > 0
    dec.close()
