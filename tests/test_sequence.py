"""Picture-level driver (fast-cu-decision-hevc_amd/sequence.py): the fork's per-picture schedule as host logic (CPU), the
.yuv reader / writer, and -- on the GPU -- a short sequence through SequenceDecider against the oracle driven by the same
schedule."""
import numpy as np
import pytest

import hmo_py


def test_schedule_follows_the_reference_cycle(built, pkg):
    seq = pkg.sequence
    s = seq.FastDecisionSchedule(period=6, n_training=2, n_verifying=2, decision_switch=hmo_py.decision_switch)
    good = np.zeros((4, 6)); good[:, 0] = 9; good[:, 1] = 1; good[:, 2] = 9; good[:, 3] = 1          # precision 0.9 both ways
    half = np.zeros((4, 6)); half[:, 0] = 1; half[:, 1] = 9; half[:, 2] = 1; half[:, 3] = 9          # 0.1
    states, switches = [], []
    for poc in range(13):
        st, sk, te = s.begin_picture(poc)
        states.append(st)
        switches.append((sk.tolist(), te.tolist()))
        if st == seq.VERIFYING:
            # first period: two good Verifying pictures; second period: a good and a bad one (sum: 10/20 = 0.5 < 0.8)
            s.end_picture(poc, good if (poc < 6 or poc % 6 == 2) else half)
        else:
            s.end_picture(poc)
    assert states == [0, 0, 1, 1, 2, 2, 0, 0, 1, 1, 2, 2, 0]
    on, off = ([1, 1, 1, 1], [1, 1, 1, 1]), ([0, 0, 0, 0], [0, 0, 0, 0])
    # switches appear only after the LAST Verifying picture of a period and vanish at the start of the next period
    assert switches[:4] == [off] * 4 and switches[4:6] == [on] * 2 and switches[6:10] == [off] * 4
    assert switches[10:12] == [off] * 2            # second period: summed precision 0.5 stays below the threshold
    with pytest.raises(ValueError):
        t = seq.FastDecisionSchedule(period=4, n_training=1, n_verifying=1, decision_switch=hmo_py.decision_switch)
        t.begin_picture(1)
        t.end_picture(1)                           # a Verifying picture without counters


def test_yuv_round_trip(pkg, tmp_path):
    seq = pkg.sequence
    frames = [pkg.synth.mixed(64, 32, seed=s) for s in (1, 2, 3)]
    path = tmp_path / "clip_64x32.yuv"
    with open(path, "wb") as f:
        for fr in frames:
            seq.write_yuv420(f, fr)
    assert path.stat().st_size == 3 * 64 * 32 * 3 // 2
    for i, fr in enumerate(frames):
        got = seq.read_yuv420(str(path), 64, 32, i)
        assert all(np.array_equal(a, b) for a, b in zip(got, fr))
    assert seq.read_yuv420(str(path), 64, 32, 3) is None


@pytest.mark.gpu
def test_sequence_decider_matches_oracle_over_a_period(pkg, tmp_path):
    """Seven pictures of a 192x128 clip read from a .yuv file, period 5 = 1 Training + 1 Verifying + 3 Testing, one CTU
    row per slice, deblocking on: states, switches, depth maps and deblocked planes against the oracle run picture by
    picture under the same schedule (its own counters)."""
    seq = pkg.sequence
    w, h, qp, n = 192, 128, 32, 7
    path = tmp_path / "clip.yuv"
    with open(path, "wb") as f:
        for i in range(n):
            seq.write_yuv420(f, pkg.synth.smooth(w, h, seed=40 + i // 3))      # content changes every third picture
    dec = seq.SequenceDecider(w, h, qp, slice_ctus=3, fast=True, schedule=seq.FastDecisionSchedule(period=5, n_training=1, n_verifying=1))
    assert dec.slice_mode.startswith("SliceMode 1") and dec.n_slices == 2
    ref_sched = seq.FastDecisionSchedule(period=5, n_training=1, n_verifying=1, decision_switch=hmo_py.decision_switch)
    states = []
    for i in range(n):
        yuv = seq.read_yuv420(str(path), w, h, i)
        got = dec.decide(yuv)
        st, sk, te = ref_sched.begin_picture(i)
        states.append(st)
        assert got["state"] == st and got["sw_skip"].tolist() == sk.tolist() and got["sw_term"].tolist() == te.tolist()
        ref = hmo_py.Encoder(*yuv, qp, slice_ctus=dec.slice_ctus)
        if st != seq.TRAINING:
            ref.set_decision(st, hmo_py.obf_prepass(yuv[0])[0], sk, te)
        ref.compress_frame()
        ver = ref.verify_counts() if st == seq.VERIFYING else None
        if ver is not None:
            # counts are exact; the f64 loss sums are added per slice chain and then across chains, the oracle adds
            # them CU by CU over the whole picture
            assert np.array_equal(got["verify"][:, :4], ver[:, :4])
            assert np.allclose(got["verify"][:, 4:], ver[:, 4:], rtol=1e-12, atol=0)
        ref_sched.end_picture(i, ver)
        for a in range(ref.n_ctu):
            assert np.array_equal(got["depth"][a], ref.ctu_arrays(a)["depth"]), (i, a)
        ref.deblock()
        for p, q in zip([t.cpu().numpy() for t in got["rec"]], ref.rec):
            assert np.array_equal(p, q), i
    assert states == [0, 1, 2, 2, 2, 0, 1]
    dec.close()


@pytest.mark.gpu
def test_pictures_side_by_side_equal_one_by_one(pkg):
    """decide_group: the pictures the schedule allows to run together (here 1 Training, 1 Verifying, then 4 Testing in one
    launch) give what picture-by-picture deciding gives."""
    seq = pkg.sequence
    w, h, qp = 192, 128, 32
    clips = [pkg.synth.smooth(w, h, seed=60 + i) for i in range(6)]
    mk = lambda: seq.FastDecisionSchedule(period=6, n_training=1, n_verifying=1)
    one = seq.SequenceDecider(w, h, qp, fast=True, schedule=mk())
    ref = [one.decide(c) for c in clips]
    ref = [{k: (v if k not in ("rec",) else [t.cpu().numpy() for t in v]) for k, v in r.items()} for r in ref]
    one.close()
    many = seq.SequenceDecider(w, h, qp, fast=True, schedule=mk(), in_flight=8)
    got, sizes, i = [], [], 0
    while i < len(clips):
        n = many.group_size()
        sizes.append(n)
        got += many.decide_group(clips[i:i + n])
        i += n
    assert sizes == [1, 1, 4]
    for a, b in zip(ref, got):
        assert a["state"] == b["state"] and a["sw_term"].tolist() == b["sw_term"].tolist()
        assert np.array_equal(a["depth"], b["depth"])
        for p, q in zip(a["rec"], [t.cpu().numpy() for t in b["rec"]]):
            assert np.array_equal(p, q)
    many.close()


@pytest.mark.gpu
def test_command_line_front_end(pkg, tmp_path):
    """tools/fcu_decide.py end to end on a three-picture 128x64 clip: depth maps and the deblocked .yuv it writes equal
    what SequenceDecider returns in-process."""
    import os
    import subprocess
    import sys
    seq = pkg.sequence
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w, h, qp = 128, 64, 30
    clips = [pkg.synth.mixed(w, h, seed=70 + i) for i in range(3)]
    src, rec, dep = tmp_path / "in.yuv", tmp_path / "rec.yuv", tmp_path / "depth.npy"
    with open(src, "wb") as f:
        for c in clips:
            seq.write_yuv420(f, c)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fcu_decide.py"), "-i", str(src), "-w", str(w), "-h", str(h),
                        "-q", str(qp), "--fast", "--period", "3", "--training", "1", "--verifying", "1",
                        "--rec", str(rec), "--depth", str(dep)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "3 pictures decided" in r.stdout
    dec = seq.SequenceDecider(w, h, qp, fast=True, schedule=seq.FastDecisionSchedule(3, 1, 1))
    depth = np.load(dep)
    for i, c in enumerate(clips):
        want = dec.decide(c)
        assert np.array_equal(depth[i], want["depth"])
        got = seq.read_yuv420(str(rec), w, h, i)
        for p, q in zip(got, [t.cpu().numpy() for t in want["rec"]]):
            assert np.array_equal(p, q)
    dec.close()
