#!/usr/bin/env python3
"""Throughput of the SAO kernels (fcu_sao) on 4K pictures -- a measurement script, not a test.
`--pics` pictures in one call: source = the textured generator, "reconstruction" = the source low-passed and re-quantised
(a stand-in for a decided, deblocked picture; SAO's work does not depend on where the distortion came from).
Prints one JSON line with the four kernel times (statistics, candidates, decision, application) and their HBM figures."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pics", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    w, h, qp = 3840, 2160, 32
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    dev = torch.device("cuda", 0)
    eng = pkg.CuEngine(w, h, max_chains=1)
    pics = []
    for i in range(args.pics):
        org = [torch.from_numpy(p).to(dev) for p in pkg.synth.textured(w, h, seed=7 + i)]
        rec = []
        for p in org:
            f = p.float()[None, None]
            f = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(f, (1, 1, 1, 1), mode="replicate"), 3, 1)[0, 0]
            rec.append(((f / 6).round() * 6).clamp(0, 255).to(torch.uint8).contiguous())
        pics.append({"org": org, "rec": rec, "qp": qp, "lambda_": lam})
    best = None
    for _ in range(args.reps):
        work = [dict(p, rec=[r.clone() for r in p["rec"]]) for p in pics]      # SAO is in place
        torch.cuda.synchronize()
        _, off, ms = eng.sao(work, timed=True)
        torch.cuda.synchronize()
        if best is None or sum(ms) < sum(best):
            best = ms
    n = args.pics
    plane = w * h * 3 // 2
    # algorithmic bytes: statistics read source + reconstruction; application reads and writes the reconstruction
    res = {"pictures": n, "width": w, "height": h, "kernel_ms": {"stats": best[0], "cands": best[1], "decide": best[2], "apply": best[3]},
           "ms_per_picture": sum(best) / n, "offset_ctus": off.tolist()[:2],
           "stats_GBps": 2 * plane * n / (best[0] * 1e-3) / 1e9, "apply_GBps": 2 * plane * n / (best[3] * 1e-3) / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
