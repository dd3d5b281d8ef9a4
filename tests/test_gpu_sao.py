"""GPU parity of sample adaptive offset through the C ABI (fcu_sao): against the fixture the REFERENCE's own
TEncSampleAdaptiveOffset produced (tests/golden/sao.npz), batched with mixed parameters against the oracle, the whole
intra pipeline (decide -> deblock -> SAO on the device) against the PSNR the reference encoder printed for BASELINE
config 0's frame, and a lowdelay_P clip whose references are the SAO-filtered pictures."""
import os

import numpy as np
import pytest
import torch

import hmo_py
import search_trace as st
import test_sao as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = lambda planes: [torch.as_tensor(np.ascontiguousarray(p)).cuda() for p in planes]


@pytest.mark.parametrize("name", list(T.M.CASES))
def test_fcu_sao_matches_the_reference(name, pkg):
    case = T.M.CASES[name]
    gen, w, h, qp, seed, slice_ctus, slice_type, layer, dis = case
    g = np.load(os.path.join(ROOT, "tests", "golden", "sao.npz"))
    f, rec, n = T.M.decided_picture(pkg, case)
    eng = pkg.CuEngine(w, h, max_chains=1)
    rate = pkg.engine.SaoRate()
    if layer > 0:
        rate.rate[:, layer - 1] = dis
    enabled = rate.enabled(layer)
    assert enabled == [int(v) for v in g[name + "/enabled"]], "fcu_sao_enabled"
    d_rec = dev(rec)
    coded, off, ms = eng.sao([{"org": dev(f), "rec": d_rec, "qp": qp, "lambda_": 0.57 * 2.0 ** ((qp - 12) / 3.0), "slice_type": slice_type,
                               "slice_ctus": slice_ctus, "enabled": enabled}], timed=True)
    assert np.array_equal(T.M.canon(pkg.engine.sao_coded_to_array(coded[0])), g[name + "/params"]), "signalled parameters"
    assert [st.crc(p.cpu().numpy()) for p in d_rec] == [int(v) for v in g[name + "/planes_crc"]], "filtered planes"
    rate.update(layer, off[0], n)
    assert np.array_equal(rate.rate[:, layer], g[name + "/rate"]), "fcu_sao_update_rate"
    assert len(ms) == 4 and all(m >= 0 for m in ms)
    eng.destroy()


def test_batched_pictures_with_their_own_parameters(pkg):
    """one fcu_sao call over pictures that differ in content, QP, slice type, slices and switches == the oracle one by one"""
    w, h = 192, 136
    eng = pkg.CuEngine(w, h, max_chains=1)
    specs = [("mixed", 27, 0, 0, (1, 1, 1)), ("textured", 32, 1, 2, (1, 1, 0)), ("smooth", 37, 0, 0, (0, 1, 1)), ("mixed", 22, 1, 3, (1, 0, 0))]
    pics, want = [], []
    for i, (gen, qp, stype, sl, en) in enumerate(specs):
        f = getattr(pkg.synth, gen)(w, h, 11 + i)
        enc = hmo_py.Encoder(*f, qp, slice_ctus=sl)
        enc.compress_frame()
        enc.deblock()
        lam = 0.6 * 2.0 ** ((qp - 12) / 3.0)
        rec = [p.copy() for p in enc.rec]
        params, off, _ = hmo_py.sao_picture(f, rec, qp, stype, lam, enabled=en, slice_ctus=sl)
        want.append((params, off, rec))
        pics.append({"org": dev(f), "rec": dev(enc.rec), "qp": qp, "lambda_": lam, "slice_type": stype, "slice_ctus": sl, "enabled": en})
    coded, off, _ = eng.sao(pics)
    for i, (params, o, rec) in enumerate(want):
        assert np.array_equal(T.M.canon(pkg.engine.sao_coded_to_array(coded[i])), T.M.canon(params)), i
        assert list(off[i]) == o, i
        for a, b in zip(pics[i]["rec"], rec):
            assert np.array_equal(a.cpu().numpy(), b), i
    eng.destroy()


def test_device_pipeline_reproduces_the_reference_runs_psnr(pkg):
    """decide -> deblock -> SAO of the survey's 416x240 frame, all on the device, prints the reference
    encoder's Y/U/V PSNR"""
    f = pkg.synth.survey_frame(416, 240, 1234)
    eng = pkg.CuEngine(416, 240, max_chains=1)
    rec, out = eng.init_chain(0, f, 32)
    eng.compress_chains(0, 1, eng.n_ctu)
    eng.deblock(0)
    eng.sao([{"org": eng._keep[0][0], "rec": rec, "qp": 32, "lambda_": 0.57 * 2.0 ** ((32 - 12) / 3.0)}])
    assert [f"{T.psnr(a, b.cpu().numpy()):.4f}" for a, b in zip(f, rec)] == ["32.3524", "41.0897", "41.1974"]
    eng.destroy()


def test_lowdelay_clip_with_sao_references(pkg):
    """lowdelay_P as close to the reference's cfg as the path goes: SAO 1 (every picture's reference is the deblocked AND
    SAO-filtered predecessor; the slice switches follow m_saoDisabledRate of the lower temporal layer), TZ search, TMVP (the
    previous picture's fcu_ctu_out array is the collocated motion field)"""
    w, h, base_qp, sr, n_pic = 192, 128, 30, 16, 4
    dec = pkg.lowdelay.LowDelayPDecider(w, h, base_qp, n_clips=1, search_range=sr, sao=True, tmvp=True)
    state, prev, prev_ctus = hmo_py.SaoState(), None, None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, "mixed", w, h, 9, poc)
        stype, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        r = dec.decide_picture([f])[0]
        ref = hmo_py.Encoder(*f, qp, lambda_override=lam) if poc == 0 else hmo_py.Encoder(*f, qp, ref=prev, col=prev_ctus, lambda_override=lam, search_range=sr, fast_search=1)
        ref.compress_frame()
        prev_ctus = ref.all_ctus_bytes()
        assert bytes(r["out"].cpu().numpy()) == prev_ctus, poc
        ref.deblock()
        layer = hmo_py.ldp_layer(poc)
        assert layer == pkg.engine.ldp_layer(poc)
        en = state.enabled(layer)
        assert en == r["sao_enabled"], poc
        rec = [p.copy() for p in ref.rec]
        params, off, _ = hmo_py.sao_picture(f, rec, qp, stype, lam, enabled=en)
        state.update(layer, off, ref.n_ctu)
        assert np.array_equal(T.M.canon(pkg.engine.sao_coded_to_array(r["sao"])), T.M.canon(params)), poc
        for a, b in zip(r["rec"], rec):
            assert np.array_equal(a.cpu().numpy(), b), poc
        prev = rec
    dec.close()
