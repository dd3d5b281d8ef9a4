#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (a measurement script, not part of the test suite): picture sizes that are not
multiples of the CTU, the whole QP range, all content generators, slices, tool flags, the three fork states with
random switches, then deblocking -- every fcu_ctu_out field, the reconstruction, the coder state and the deblocked
planes must be identical.  Prints one line per case and a summary; exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=2026)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    import hmo_py
    pkg = g.load_package()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        w, h = int(rng.integers(8, 41)) * 8, int(rng.integers(8, 25)) * 8
        qp = int(rng.integers(0, 52))
        gen = ["smooth", "mixed", "textured"][int(rng.integers(0, 3))]
        w_ctu, n_ctu = (w + 63) // 64, ((w + 63) // 64) * ((h + 63) // 64)
        sl = int(rng.choice([0, 1, w_ctu, 2 * w_ctu]))
        flags = dict(transform_skip=int(rng.integers(0, 2)), transform_skip_fast=int(rng.integers(0, 2)),
                     sign_hiding=int(rng.integers(0, 2)), strong_intra_smoothing=int(rng.integers(0, 2)))
        state = int(rng.integers(0, 3))
        sk, te = rng.integers(0, 2, 4).astype(np.uint8), rng.integers(0, 2, 4).astype(np.uint8)
        dex = int(rng.integers(0, 2))
        boff, toff = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
        Y, U, V = getattr(pkg.synth, gen)(w, h, seed=int(rng.integers(0, 10000)))
        obf, _ = hmo_py.obf_prepass(Y)
        eng = pkg.CuEngine(w, h, max_chains=1)
        eng.init_chain(0, (Y, U, V), qp=qp, slice_ctus=sl, **flags)
        if state:
            eng.set_decision(0, state, torch.as_tensor(obf).cuda(), sk, te, depth_exception=dex)
        eng.compress_chains(0, 1, n_ctu)
        eng.sync()
        oflags = dict(flags)
        oflags["strong_smoothing"] = oflags.pop("strong_intra_smoothing")
        ref = hmo_py.Encoder(Y, U, V, qp, slice_ctus=sl, **oflags)
        if state:
            ref.set_decision(state, obf, sk, te, depth_exception=dex)
        ref.compress_frame()
        diffs = []
        for a in range(n_ctu):
            got, want = eng.ctu_out(0, a), ref.ctu_arrays(a)
            for k, v in want.items():
                same = np.array_equal(v, got[k]) if isinstance(v, np.ndarray) else v == got[k]
                if not same:
                    diffs.append(f"ctu{a}.{k}")
        if any(not np.array_equal(p, q) for p, q in zip(eng.rec_planes(0), ref.rec)):
            diffs.append("rec")
        ce, fe = eng.ctx_state(0)
        co, fo = ref.cabac()
        if not (np.array_equal(ce, co) and fe == fo):
            diffs.append("cabac")
        if state == 1 and not np.array_equal(eng.verify_counts(0), ref.verify_counts()):
            diffs.append("verify")
        eng.deblock(0, boff, toff)
        eng.sync()
        ref.deblock(boff, toff)
        if any(not np.array_equal(p, q) for p, q in zip(eng.rec_planes(0), ref.rec)):
            diffs.append("deblock")
        eng.destroy()
        bad += bool(diffs)
        print(f"case {case:3d} {gen:8s} {w}x{h} qp{qp:2d} slice_ctus {sl} flags {list(flags.values())} state {state} "
              f"sw {sk.tolist()}/{te.tolist()} dex {dex} dbk {boff}/{toff}: {'OK' if not diffs else 'MISMATCH ' + ','.join(diffs[:6])}", flush=True)
    print(f"{args.cases - bad} of {args.cases} cases identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
