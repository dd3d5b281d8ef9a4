#!/usr/bin/env python3
"""Generates tests/golden/cu_<case>.npz by RUNNING THE REFERENCE'S OWN CU-LEVEL FUNCTIONS -- TEncCu::xCheckRDCostMerge2Nx2N
(the whole function: candidate loop, FastDecisionForMerge early-outs), xCheckRDCostInter, xCheckRDCostIntra, xCheckBestMode
(best / temp swap, TEMP_BEST -> NEXT_BEST coder hand-over) and deriveTestModeAMP (TEncCu.cpp:381,1900,2025,2064,2213) --
compiled in place from the reference's TEncCu.cpp without the body of xCompressCU (oracle/ref/build_ref.sh) and driven by
oracle/ref/ref_driver.cpp:ref_cu_run.

A case is an I picture or a short lowdelay_P clip decided by the oracle.  For every CU inside the picture, at every depth,
the oracle's state before the CU's first candidate (picture reconstruction, decided neighbours, coder slot
[depth][CI_CURR_BEST], the TZ search's carried vector) is shown to the reference, which evaluates ALL candidates of that CU
with its own functions; the fixture stores the CU that survived in the reference (prediction mode, partition size, skip /
merge, distortion, bits, cost, CRC-32s of motion, modes, TU tree, coefficients, reconstruction and of the coder in
[depth][CI_NEXT_BEST]).  tests/test_golden_cu.py re-runs the oracle and compares its surviving CU (best[depth] when the
candidates are done, before the split flag is priced) record by record.

Run in the build container only:  python oracle/ref/make_golden_cu.py [case]
"""
import importlib.util
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {      # name: (generator, width, height, base QP, seed, pictures, search range, fast search, amp, tmvp)
    "intra_mixed_qp32": ("mixed", 136, 72, 32, 5, 1, 0, 0, 0, 0),             # I picture, partial CTUs
    "intra_textured_qp22": ("textured", 128, 64, 22, 7, 1, 0, 0, 0, 0),
    "ldp_mixed_qp27": ("mixed", 136, 72, 27, 31, 3, 8, 0, 0, 0),              # full search; intra CUs inside P pictures
    "ldp_tz_textured_qp32": ("textured", 128, 128, 32, 7, 3, 32, 1, 0, 0),    # TZ search
    "ldp_amp_shear_qp27": ("shear_textured", 128, 128, 27, 9, 3, 16, 1, 1, 1),  # AMP + TZ + TMVP on motion boundaries: asymmetric partitions win
    "ldp_amp_mixed_qp35": ("mixed", 192, 64, 35, 13, 4, 16, 0, 1, 0),         # AMP, full search, high QP: skip / root-cbf-zero CUs, FDM early-outs
}
FIELDS = ["ctu", "zidx", "depth", "parent_part_size", "pred_mode", "part_size", "skip", "merge0", "midx0", "dist", "bits", "cost_lo", "cost_hi",
          "crc_motion", "crc_tree", "crc_coef", "crc_reco", "crc_coder"]


def _record(st, ctu, zidx, depth, parent, n, s, dist, bits, cost, skip, merge_flag, merge_idx, mvp_idx, ref_idx, mv, mvd, inter_dir, part_size, pred_mode,
            tr_idx, cbf, tskip, intra_dir, coef, reco, coder):
    inter = int(pred_mode[0]) == 0
    motion = st.crc(skip[:n], merge_flag[:n], merge_idx[:n], mvp_idx[:n], ref_idx[:n], mv[:n], mvd[:n], inter_dir[:n]) if inter else 0
    tree = st.crc(part_size[:n], pred_mode[:n], tr_idx[:n], cbf[:, :n], tskip[:, :n], intra_dir[:, :n] if not inter else np.zeros(1, np.uint8))
    lo, hi = st._cost_words(cost)
    return np.array([ctu, zidx, depth, parent & 0xff, int(pred_mode[0]), int(part_size[0]), int(skip[0]), int(merge_flag[0]), int(merge_idx[0]), dist, bits, lo, hi,
                     motion, tree, coef, reco, coder], np.uint32)


def record_from_oracle(st, hmo_py, enc, depth, parent):
    cu = enc.test_cu(depth, best=True)
    n, s, h = cu.nparts, 64 >> depth, 32 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    coef = st.crc(A(cu.coef)[0, :s * s], A(cu.coef)[1, :s * s // 4], A(cu.coef)[2, :s * s // 4])
    r = enc.test_reco(depth, best=True)
    reco = st.crc(A(r.y).reshape(64, 64)[:s, :s], A(r.u).reshape(32, 32)[:h, :h], A(r.v).reshape(32, 32)[:h, :h])
    ctx, frac = enc.test_slot(depth, hmo_py.CI_NEXT_BEST)
    coder = st.crc(ctx[st.O_SORTED], np.array([frac], np.uint64))
    return _record(st, enc.cur_ctu(), cu.zidx, depth, parent, n, s, cu.dist, cu.bits, cu.cost, A(cu.skip), A(cu.merge_flag), A(cu.merge_idx), A(cu.mvp_idx), A(cu.ref_idx),
                   A(cu.mv), A(cu.mvd), A(cu.inter_dir), A(cu.part_size).view(np.uint8), A(cu.pred_mode).view(np.uint8), A(cu.tr_idx), A(cu.cbf), A(cu.tskip), A(cu.intra_dir), coef, reco, coder)


def record_from_ref(st, hmo_py, r, ref, depth, ctu, zidx, parent):
    n, s, h = 256 >> (2 * depth), 64 >> depth, 32 >> depth
    A = lambda f: np.ctypeslib.as_array(f)
    coef = st.crc(A(r.coef)[0, :s * s], A(r.coef)[1, :s * s // 4], A(r.coef)[2, :s * s // 4])
    reco = st.crc(A(r.reco)[0, :s * s].reshape(s, s), A(r.reco)[1, :h * h].reshape(h, h), A(r.reco)[2, :h * h].reshape(h, h))
    ctx, frac = ref.coder(depth, hmo_py.CI_NEXT_BEST)
    coder = st.crc(ctx[st.O_SORTED], np.array([frac], np.uint64))
    intra_dir = np.stack([A(r.luma_dir), A(r.chroma_dir)])
    return _record(st, ctu, zidx, depth, parent, n, s, r.dist, r.bits, r.cost, A(r.skip), A(r.merge_flag), A(r.merge_idx), A(r.mvp_idx), A(r.ref_idx), A(r.mv), A(r.mvd),
                   A(r.inter_dir), A(r.part_size), A(r.pred_mode), A(r.tr_idx), A(r.cbf), A(r.tskip), intra_dir, coef, reco, coder)


def run_case(case, with_ref):
    """Decides the clip with the oracle; returns (records per picture, mismatches against the reference when with_ref)."""
    import ctypes as C
    import hmo_py
    import search_trace as st
    spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "fast-cu-decision-hevc_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    gen, w, h, base_qp, seed, n_pic, sr, fast, amp, tmvp = CASES[case]
    prev = prev_ctus = None
    out, bad = [], [0]
    for poc in range(n_pic):
        f = st.moving_frame(synth, gen, w, h, seed, poc) if n_pic > 1 else getattr(synth, gen)(w, h, seed=seed)
        stype, qp, lam = hmo_py.ldp_slice(poc, base_qp) if n_pic > 1 else (hmo_py.SLICE_I, base_qp, None)
        is_p = poc > 0
        if not is_p:
            enc = hmo_py.Encoder(*f, qp, lambda_override=lam) if lam is not None else hmo_py.Encoder(*f, qp)
        else:
            enc = hmo_py.Encoder(*f, qp, ref=prev, col=prev_ctus if tmvp else None, lambda_override=lam, search_range=sr, fast_search=fast, amp=amp)
        ref = None
        traced = is_p or n_pic == 1                            # picture 0 of a clip (intra with the clip's lambda) only provides the reference picture
        if with_ref and traced:
            ref = st.RefSearch(w, h, qp, f, search_range=max(sr, 8), fast_search=fast, amp=amp)
            if is_p:
                ref.setup_p(prev, lam)
                if tmvp:
                    ref.setup_col(prev_ctus, poc)
            ref.L.ref_cu_run.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
        recs, pending = [], {}

        def on_event(ev, depth, arg, enc=enc, ref=ref, recs=recs, pending=pending, is_p=is_p):
            if ev == hmo_py.EV_CU_BEGIN and ref:
                ref.load_state(enc, depth)
                if is_p:
                    ref.load_inter_state(enc)
                cu = enc.test_cu(depth, best=True)
                o, log = st.RefCuOut(), np.zeros(16, np.int32)
                ref.L.ref_cu_run(enc.cur_ctu(), cu.zidx, depth, arg & 0xff if arg >= 0 else 8, (1 if is_p else 0) | (2 if amp else 0), C.byref(o), log.ctypes.data_as(C.c_void_p))
                pending[depth] = record_from_ref(st, hmo_py, o, ref, depth, enc.cur_ctu(), cu.zidx, arg)
            elif ev == hmo_py.EV_CU_DONE:
                mine = record_from_oracle(st, hmo_py, enc, depth, arg)
                rec = pending.pop(depth) if ref else mine
                recs.append(rec)
                if not np.array_equal(rec, mine):
                    bad[0] += 1
                    if bad[0] <= 5:
                        print("MISMATCH poc", poc, "\n  ref   ", dict(zip(FIELDS, rec.tolist())), "\n  oracle", dict(zip(FIELDS, mine.tolist())))

        if traced:
            enc.set_trace(on_event, cu_events=True)
        enc.compress_frame()
        out.append(np.stack(recs) if recs else np.zeros((0, len(FIELDS)), np.uint32))
        prev_ctus = enc.all_ctus_bytes()
        enc.deblock()
        prev = [a.copy() for a in enc.rec]
    return out, bad[0]


def one(case):
    recs, bad = run_case(case, True)
    G = {"fields": np.array(FIELDS), "case": np.array(list(map(str, CASES[case])))}
    for poc, r in enumerate(recs):
        G[f"cu_{poc}"] = r
    np.savez_compressed(os.path.join(OUT, f"cu_{case}.npz"), **G)
    print(case, "CUs per picture", [len(r) for r in recs], "inter / intra winners", [(int((r[:, 4] == 0).sum()), int((r[:, 4] == 1).sum())) for r in recs],
          "skip winners", [int(r[:, 6].sum()) for r in recs], "oracle mismatches", bad)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(1 if one(sys.argv[1]) else 0)
    for case in CASES:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), case])
