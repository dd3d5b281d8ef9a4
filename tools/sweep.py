#!/usr/bin/env python3
"""Development sweep: intra CTUs/s against the number of chains in flight.
  python tools/sweep.py [--frames 16] [--counts 64,256,1024] [--ctus 4]
Same workload as bench.py's intra configuration (4K, QP 22..37, SliceMode-1 slices of 120 CTUs, lowest QP first)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--counts", default="64,256,1024")
    ap.add_argument("--ctus", type=int, default=4)
    ap.add_argument("--first", type=int, default=0, help="CTUs decided (untimed) before the timed launch")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    dev = torch.device("cuda", 0)
    W, H, sl = 3840, 2160, 120
    n_ctu = 60 * 34; n_sl = 17; qps = [22, 27, 32, 37]
    seeds = list(range(7, 7 + args.frames))
    chain_list = sorted(((seed, qp, k) for seed in seeds for qp in qps for k in range(n_sl)), key=lambda c: (c[1], c[0], c[2]))
    counts = [int(c) for c in args.counts.split(",")]
    n_chains = min(len(chain_list), max(counts))
    chain_list = chain_list[:n_chains]
    eng = pkg.CuEngine(W, H, max_chains=n_chains)
    nb = pkg.engine.CTU_OUT_BYTES
    frames = {s: bench.gen_textured_gpu(torch, dev, W, H, s) for s in set(c[0] for c in chain_list)}
    recs = {}; outs = {}
    for seed, qp, k in chain_list:
        if (seed, qp) not in recs:
            recs[(seed, qp)] = [torch.zeros_like(p) for p in frames[seed]]
            outs[(seed, qp)] = torch.zeros(nb * n_ctu, dtype=torch.uint8, device=dev)
    def bind(ci):
        seed, qp, k = chain_list[ci]
        eng.init_chain(ci, frames[seed], qp=qp, slice_ctus=sl, rec=recs[(seed, qp)], out=outs[(seed, qp)])
        eng.set_range(ci, k * sl, min(sl, n_ctu - k * sl))
    res = {}
    for nw in (1,):
        for n in counts:
            if n > n_chains:
                continue
            for ci in range(n):
                bind(ci)
            if args.first:
                eng.compress_chains(0, n, args.first)
            eng.sync()
            t = time.perf_counter()
            eng.compress_chains(0, n, args.ctus)
            eng.sync()
            dt = time.perf_counter() - t
            res[f"n{n}"] = round(n * args.ctus / dt, 1)
            print(f"chains {n:5d}: {n * args.ctus / dt:9.1f} CTU/s  ({dt * 1e3 / args.ctus:.1f} ms per CTU step)", flush=True)
    print(json.dumps(res))
    eng.destroy()

if __name__ == "__main__":
    main()
