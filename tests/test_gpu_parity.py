"""GPU parity: the HIP engine (through the C ABI of libfcu.so) against the oracle, bit-exact
on every TComDataCU array, the reconstruction and the CABAC state after each CTU."""
import numpy as np
import pytest

import hmo_py

pytestmark = pytest.mark.gpu

FIELDS = ["depth", "width", "height", "skip", "part_size", "pred_mode", "tq_bypass", "qp", "chroma_qp_adj", "tr_idx",
          "tskip", "cbf", "intra_dir", "ipcm", "coeff_y", "coeff_cb", "coeff_cr", "total_cost", "total_dist",
          "total_bits", "total_bins"]


def _compare_ctu(got, want, tag):
    for k in FIELDS:
        v = want[k]
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, got[k]), f"{tag}: field {k} differs at {np.argwhere(v != got[k])[:4].tolist()}"
        else:
            assert v == got[k], f"{tag}: {k}: engine {got[k]} oracle {v}"


CASES = [
    # (generator, w, h, qp, slice_ctus)
    ("mixed", 128, 64, 32, 0),
    ("smooth", 192, 128, 27, 0),
    ("textured", 128, 128, 22, 0),
    ("mixed", 136, 72, 37, 0),      # partial CTUs right and bottom (forced splits)
    ("mixed", 256, 64, 32, 2),      # SliceMode 1: two CTUs per slice
    ("textured", 128, 64, 5, 0),    # ends of the QP range
    ("mixed", 128, 64, 51, 0),
]


@pytest.mark.parametrize("gen,w,h,qp,sl", CASES)
def test_ctu_by_ctu_matches_oracle(pkg, gen, w, h, qp, sl):
    Y, U, V = getattr(pkg.synth, gen)(w, h, seed=11)
    eng = pkg.CuEngine(w, h, max_chains=1)
    eng.init_chain(0, (Y, U, V), qp=qp, slice_ctus=sl)
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    for a in range(eng.n_ctu):
        got = eng.compress_ctu(0, a)          # TEncCu::compressCtu-shaped call through the C ABI
        ref.compress_ctu(a)
        _compare_ctu(got, ref.ctu_arrays(a), f"{gen} {w}x{h} qp{qp} ctu{a}")
        ctx_e, frac_e = eng.ctx_state(0)
        ctx_o, frac_o = ref.cabac()
        assert np.array_equal(ctx_e, ctx_o) and frac_e == frac_o, f"CABAC state after CTU {a}"
    for p, q in zip(eng.rec_planes(0), ref.rec):
        assert np.array_equal(p, q)
    eng.destroy()


def test_batched_chains_equal_single(pkg):
    """Many chains per launch (different QPs / contents) give the same result as one by one."""
    w, h = 128, 64
    frames = [getattr(pkg.synth, g)(w, h, seed=s) for g, s in (("mixed", 1), ("smooth", 2), ("textured", 3), ("mixed", 4))]
    qps = [22, 27, 32, 37]
    eng = pkg.CuEngine(w, h, max_chains=4)
    for i, (f, qp) in enumerate(zip(frames, qps)):
        eng.init_chain(i, f, qp=qp)
    eng.compress_chains(0, 4, eng.n_ctu)
    eng.sync()
    for i, (f, qp) in enumerate(zip(frames, qps)):
        ref = hmo_py.Encoder(*f, qp)
        ref.compress_frame()
        for a in range(eng.n_ctu):
            _compare_ctu(eng.ctu_out(i, a), ref.ctu_arrays(a), f"chain{i} ctu{a}")
        for p, q in zip(eng.rec_planes(i), ref.rec):
            assert np.array_equal(p, q)
    ms, n = eng.kernel_ms()
    assert n == 1 and ms > 0
    eng.destroy()


def test_raster_order_is_enforced(pkg):
    Y, U, V = pkg.synth.smooth(128, 64, seed=5)
    eng = pkg.CuEngine(128, 64, max_chains=1)
    eng.init_chain(0, (Y, U, V), qp=32)
    with pytest.raises(pkg.FcuError):
        eng.compress_ctu(0, 1)                # CTU 0 has not been decided yet
    eng.destroy()


def test_full_4k_frame_as_slice_chains(pkg):
    """BASELINE configs[2] size: one 3840x2160 frame, SliceMode 1 with one CTU row (60 CTUs) per slice, decided as 34
    independent chains in one launch sequence -- every one of the 2040 CTUs (partial bottom row included) bit-exact
    against the oracle run on the same frame with the same slicing, plus the reconstruction planes."""
    import torch
    w, h, qp, sl = 3840, 2160, 32, 60
    Y, U, V = pkg.synth.textured(w, h, seed=7)
    eng = pkg.CuEngine(w, h, max_chains=34)
    n_sl, rec, out = eng.init_slice_chains(0, (Y, U, V), qp, sl)
    assert n_sl == 34 and eng.n_ctu == 2040
    eng.compress_chains(0, n_sl, sl)
    eng.sync()
    assert [eng.position(k) for k in range(n_sl)] == [min((k + 1) * sl, eng.n_ctu) for k in range(n_sl)]
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    ref.compress_frame()
    raw = out.cpu().numpy()
    nbytes = pkg.engine.CTU_OUT_BYTES
    for a in range(eng.n_ctu):
        got = pkg.engine.ctu_to_dict(pkg.engine.CtuOut.from_buffer_copy(raw[a * nbytes:(a + 1) * nbytes].tobytes()))
        _compare_ctu(got, ref.ctu_arrays(a), f"4K ctu{a}")
    for p, q in zip([t.cpu().numpy() for t in rec], ref.rec):
        assert np.array_equal(p, q)
    # the chain of the last slice ends in the same coder state as the oracle (which ran the slices in order)
    ctx_e, frac_e = eng.ctx_state(n_sl - 1)
    ctx_o, frac_o = ref.cabac()
    assert np.array_equal(ctx_e, ctx_o) and frac_e == frac_o
    eng.destroy()


@pytest.mark.parametrize("qp", [22, 27, 37])
def test_4k_slices_at_other_qps(pkg, qp):
    """The first six CTU rows of the 4K frame as six slice chains at the other ends of the QP range."""
    w, h, sl, rows = 3840, 2160, 60, 6
    Y, U, V = pkg.synth.textured(w, h, seed=8)
    eng = pkg.CuEngine(w, h, max_chains=rows)
    rec, out = eng.init_chain(0, (Y, U, V), qp, slice_ctus=sl)
    planes = eng._keep[0][0]
    eng.set_range(0, 0, sl)
    for k in range(1, rows):
        eng.init_chain(k, planes, qp, slice_ctus=sl, rec=rec, out=out)
        eng.set_range(k, k * sl, sl)
    eng.compress_chains(0, rows, sl)
    eng.sync()
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    raw = out.cpu().numpy()
    nbytes = pkg.engine.CTU_OUT_BYTES
    for a in range(rows * sl):
        ref.compress_ctu(a)
        got = pkg.engine.ctu_to_dict(pkg.engine.CtuOut.from_buffer_copy(raw[a * nbytes:(a + 1) * nbytes].tobytes()))
        _compare_ctu(got, ref.ctu_arrays(a), f"4K qp{qp} ctu{a}")
    for p, q in zip([t.cpu().numpy() for t in rec], ref.rec):
        assert np.array_equal(p[:rows * 64 >> (0 if p.shape[1] == w else 1)], q[:rows * 64 >> (0 if q.shape[1] == w else 1)])
    eng.destroy()


def test_baseline_config0_416x240(pkg):
    """BASELINE configs[0]: one 416x240 frame (the plumbing frame of SURVEY.md 8d), QP 32, one slice -- whole frame,
    partial CTUs on the right (416 = 6.5 CTUs) and at the bottom (240 = 3.75 CTUs)."""
    w, h, qp = 416, 240, 32
    Y, U, V = pkg.synth.smooth(w, h)
    eng = pkg.CuEngine(w, h, max_chains=1)
    eng.init_chain(0, (Y, U, V), qp=qp)
    eng.compress_chains(0, 1, eng.n_ctu)
    eng.sync()
    ref = hmo_py.Encoder(Y, U, V, qp)
    ref.compress_frame()
    assert eng.n_ctu == 28
    for a in range(eng.n_ctu):
        _compare_ctu(eng.ctu_out(0, a), ref.ctu_arrays(a), f"416x240 ctu{a}")
    for p, q in zip(eng.rec_planes(0), ref.rec):
        assert np.array_equal(p, q)
    ctx_e, frac_e = eng.ctx_state(0)
    ctx_o, frac_o = ref.cabac()
    assert np.array_equal(ctx_e, ctx_o) and frac_e == frac_o
    eng.destroy()


def test_baseline_config1_1080p(pkg):
    """BASELINE configs[1] size: one 1920x1080 frame, QP 32, one CTU row per slice = 17 chains (the last row is
    partial: 1080 = 16.875 CTUs), all 510 CTUs against the oracle."""
    w, h, qp, sl = 1920, 1080, 32, 30
    Y, U, V = pkg.synth.textured(w, h, seed=9)
    eng = pkg.CuEngine(w, h, max_chains=17)
    n_sl, rec, out = eng.init_slice_chains(0, (Y, U, V), qp, sl)
    assert n_sl == 17 and eng.n_ctu == 510
    eng.compress_chains(0, n_sl, sl)
    eng.sync()
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    ref.compress_frame()
    raw = out.cpu().numpy()
    nbytes = pkg.engine.CTU_OUT_BYTES
    for a in range(eng.n_ctu):
        got = pkg.engine.ctu_to_dict(pkg.engine.CtuOut.from_buffer_copy(raw[a * nbytes:(a + 1) * nbytes].tobytes()))
        _compare_ctu(got, ref.ctu_arrays(a), f"1080p ctu{a}")
    for p, q in zip([t.cpu().numpy() for t in rec], ref.rec):
        assert np.array_equal(p, q)
    eng.destroy()


def test_chain_range_must_follow_slices(pkg):
    w, h = 256, 128
    Y, U, V = pkg.synth.mixed(w, h, seed=3)
    eng = pkg.CuEngine(w, h, max_chains=1)
    eng.init_chain(0, (Y, U, V), qp=32, slice_ctus=4)
    with pytest.raises(pkg.FcuError):
        eng.set_range(0, 2, 4)          # does not start at a slice boundary
    with pytest.raises(pkg.FcuError):
        eng.set_range(0, 4, 3)          # does not end at one
    eng.set_range(0, 4, 4)
    assert eng.position(0) == 4
    eng.destroy()


@pytest.mark.parametrize("ts,tsf,sbh,strong", [(0, 0, 1, 1), (1, 0, 0, 0)])
def test_tool_flags_through_the_abi(pkg, ts, tsf, sbh, strong):
    """fcu_frame_params tool switches (TransformSkip / TransformSkipFast / SignHideFlag / StrongIntraSmoothing)."""
    w, h, qp = 128, 64, 27
    Y, U, V = pkg.synth.mixed(w, h, seed=9)
    eng = pkg.CuEngine(w, h, max_chains=1)
    eng.init_chain(0, (Y, U, V), qp=qp, transform_skip=ts, transform_skip_fast=tsf, sign_hiding=sbh, strong_intra_smoothing=strong)
    ref = hmo_py.Encoder(Y, U, V, qp, transform_skip=ts, transform_skip_fast=tsf, sign_hiding=sbh, strong_smoothing=strong)
    for a in range(eng.n_ctu):
        got = eng.compress_ctu(0, a)
        ref.compress_ctu(a)
        _compare_ctu(got, ref.ctu_arrays(a), f"tools {ts}{tsf}{sbh}{strong} ctu{a}")
    ctx_e, frac_e = eng.ctx_state(0)
    ctx_o, frac_o = ref.cabac()
    assert np.array_equal(ctx_e, ctx_o) and frac_e == frac_o
    eng.destroy()


# ---- the fork's fast CU-size decision (Verifying / Testing states, Naive model on the OBF map) ----------------------
def _decide_frame(pkg, eng, yuv, qp, state, obf_dev, sw=((0, 0, 0, 0), (0, 0, 0, 0)), dex=0):
    eng.init_chain(0, yuv, qp=qp)
    eng.set_decision(0, state, obf_dev, *sw, depth_exception=dex)
    eng.compress_chains(0, 1, eng.n_ctu)
    eng.sync()
    return [eng.ctu_out(0, a) for a in range(eng.n_ctu)]


@pytest.mark.parametrize("gen,w,h,qp", [("smooth", 192, 128, 32), ("mixed", 136, 72, 27)])
def test_decision_states_match_oracle(pkg, gen, w, h, qp):
    """OBF map from the device pre-pass -> Verifying frame (counters) -> switches -> Testing frames: every state of
    the fork's xCompressCU hooks bit-exact against the oracle, through the C ABI."""
    eng_mod = pkg.engine
    Y, U, V = getattr(pkg.synth, gen)(w, h, seed=5)
    obf_o, _ = hmo_py.obf_prepass(Y)
    eng = pkg.CuEngine(w, h, max_chains=1)
    obf_t, _, _ = eng.obf_prepass(Y)                      # the map the engine itself produces (also off the CTU grid)
    obf_dev = obf_t[0].contiguous()
    assert np.array_equal(obf_dev.cpu().numpy(), obf_o)

    def check(state, sw=((0, 0, 0, 0), (0, 0, 0, 0)), dex=0):
        got = _decide_frame(pkg, eng, (Y, U, V), qp, state, obf_dev, sw, dex)
        ref = hmo_py.Encoder(Y, U, V, qp)
        ref.set_decision(state, obf_o, *sw, depth_exception=dex)
        ref.compress_frame()
        for a in range(eng.n_ctu):
            _compare_ctu(got[a], ref.ctu_arrays(a), f"{gen} state{state} sw{sw} ctu{a}")
        for p, q in zip(eng.rec_planes(0), ref.rec):
            assert np.array_equal(p, q)
        ctx_e, frac_e = eng.ctx_state(0)
        ctx_o, frac_o = ref.cabac()
        assert np.array_equal(ctx_e, ctx_o) and frac_e == frac_o
        return ref

    ref = check(eng_mod.VERIFYING)
    ver = eng.verify_counts(0)
    assert np.array_equal(ver, ref.verify_counts())       # TP/FP/TN/FN and the f64 RD-loss sums
    sw = eng_mod.decision_switch(ver)
    check(eng_mod.TESTING, sw)
    check(eng_mod.TESTING, ((1, 1, 1, 1), (1, 1, 1, 1)))
    check(eng_mod.TESTING, ((0, 1, 0, 1), (1, 0, 1, 0)), dex=1)
    check(eng_mod.TRAINING)                               # back to exhaustive on the same chain
    eng.destroy()


def test_decision_needs_the_obf_map(pkg):
    Y, U, V = pkg.synth.smooth(128, 64, seed=5)
    eng = pkg.CuEngine(128, 64, max_chains=1)
    with pytest.raises(pkg.FcuError):
        eng.set_decision(0, pkg.engine.TESTING, None)     # chain not bound
    eng.init_chain(0, (Y, U, V), qp=32)
    with pytest.raises(pkg.FcuError):
        eng.set_decision(0, pkg.engine.TESTING, None)     # Testing without a map
    eng.destroy()


def test_4k_testing_state_obeys_the_rule(pkg):
    """Full-size property: six CTU rows of the 4K frame in the Testing state with every switch on, OBF map from the
    device pre-pass.  The published quadtrees must follow the Naive rule everywhere, the pruned search must be
    cheaper in TU trials than the exhaustive one, and all 360 CTUs are compared with the oracle in the same state."""
    from test_decision import check_rule
    w, h, sl, rows, qp = 3840, 2160, 60, 6, 32
    Y, U, V = pkg.synth.textured(w, h, seed=7)
    eng = pkg.CuEngine(w, h, max_chains=rows)
    obf_t, _, _ = eng.obf_prepass(Y)
    obf_dev = obf_t[0].contiguous()
    obf = obf_dev.cpu().numpy()
    sw = ((1, 1, 1, 1), (1, 1, 1, 1))
    trials = {}
    for state in (pkg.engine.TRAINING, pkg.engine.TESTING):
        rec, out = eng.init_chain(0, (Y, U, V), qp, slice_ctus=sl)
        planes = eng._keep[0][0]
        for k in range(rows):
            if k:
                eng.init_chain(k, planes, qp, slice_ctus=sl, rec=rec, out=out)
            eng.set_range(k, k * sl, sl)
            eng.set_decision(k, state, obf_dev, *sw)
        eng.compress_chains(0, rows, sl)
        eng.sync()
        trials[state] = sum(eng.debug_counters(k)[16] for k in range(rows))
    raw = out.cpu().numpy()
    nbytes = pkg.engine.CTU_OUT_BYTES
    got = [pkg.engine.ctu_to_dict(pkg.engine.CtuOut.from_buffer_copy(raw[a * nbytes:(a + 1) * nbytes].tobytes())) for a in range(rows * sl)]
    assert check_rule(got, obf, w, h, *sw) >= rows * sl
    assert trials[pkg.engine.TESTING] < trials[pkg.engine.TRAINING]
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    ref.set_decision(hmo_py.TESTING, obf, *sw)
    for a in range(rows * sl):                            # all six slices
        ref.compress_ctu(a)
        _compare_ctu(got[a], ref.ctu_arrays(a), f"4K testing ctu{a}")
    eng.destroy()


# ---- in-loop deblocking (TComLoopFilter::loopFilterPic) -------------------------------------------------------------
def test_deblock_matches_reference_golden(pkg):
    """fcu_deblock on the inputs of tests/golden/deblock_*.npz against the planes the reference's own TComLoopFilter
    produced for them (generator: oracle/ref/make_golden_deblock.py): pinned parity, through the C ABI."""
    import glob
    import os
    import torch
    from test_deblock import ctus_from_golden
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "deblock_*.npz")))
    assert len(paths) == 6
    for path in paths:
        g = np.load(path)
        w, h = int(g["width"]), int(g["height"])
        eng = pkg.CuEngine(w, h, max_chains=1)
        out = torch.frombuffer(bytearray(bytes(ctus_from_golden(g))), dtype=torch.uint8).cuda()
        rec = [torch.as_tensor(np.ascontiguousarray(g["rec_" + c])).cuda() for c in "yuv"]
        eng.deblock(out=out, rec=rec, beta_offset_div2=int(g["beta_offset_div2"]), tc_offset_div2=int(g["tc_offset_div2"]))
        eng.sync()
        for k, c in enumerate("yuv"):
            got, want = rec[k].cpu().numpy(), g["out_" + c]
            assert np.array_equal(got, want), f"{os.path.basename(path)} plane {c}: {np.argwhere(got != want)[:4].tolist()}"
        eng.destroy()


def test_decide_then_deblock_1080p(pkg):
    """The picture pipeline of TEncGOP::compressGOP for an intra picture: all slices decided (17 slice chains), then
    the loop filter over the whole picture -- planes against the oracle doing the same on the CPU."""
    w, h, qp, sl = 1920, 1080, 32, 30
    Y, U, V = pkg.synth.textured(w, h, seed=9)
    eng = pkg.CuEngine(w, h, max_chains=17)
    n_sl, rec, out = eng.init_slice_chains(0, (Y, U, V), qp, sl)
    eng.compress_chains(0, n_sl, sl)
    ms = eng.deblock(0, timed=True)                        # same stream order: after the decisions
    ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl)
    ref.compress_frame()
    before = ref.rec[0].copy()
    ref.deblock()
    assert not np.array_equal(before, ref.rec[0])
    for p, q in zip([t.cpu().numpy() for t in rec], ref.rec):
        assert np.array_equal(p, q)
    assert ms[0] > 0 and ms[1] > 0
    eng.destroy()
