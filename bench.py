#!/usr/bin/env python3
"""bench.py -- CTUs/sec of the CU depth/mode RDO decision (TEncCu::compressCtu equivalent).

Workload (BASELINE.json configs[2]): 3840x2160 all-intra, QP {22,27,32,37}, full depth-0..3
quadtree + chroma RDO, synthetic "textured" YUV, resident in HBM before the
timed region.  A chain = (frame, QP) = one I slice, the unit HM decides strictly serially;
`--frames` frames x 4 QPs chains are in flight per GPU, one wavefront each; by default that is twice the number of
wave slots of the device, ordered longest (lowest QP) first, so that the slots the short high-QP chains free early are
refilled by the hardware dispatcher.  A *step* advances every chain by `--ctus-per-step` CTUs (compressCtu + encodeCtu
replay per CTU) in one launch.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; every rank
   owns its own frames -- no collective on the data path; weak scaling)

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CTU = 55300          # src 12288 + neighbours 1024 + rec 12288 + coeff 24576 + meta 5120
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")


def measured_traffic(chains, ctus_per_step):
    """HBM-side bytes per launch of the engine kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_summary.json: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when the summary is for another workload shape."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        if d.get("chains_per_launch") != chains or d.get("ctus_per_chain_per_launch") != ctus_per_step:
            return None
        return float(d["traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def gen_textured_gpu(torch, dev, w, h, seed):
    """The SURVEY 8d 'textured' generator evaluated on the device (torch RNG)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    y = torch.arange(h, device=dev, dtype=torch.float32)[:, None]
    x = torch.arange(w, device=dev, dtype=torch.float32)[None, :]
    noise = torch.randn((h, w), generator=g, device=dev) * 18 * (1 + ((x // 64 + y // 64) % 3)) / 2
    Y = 128 + 50 * torch.sin(x / 9 + y / 31) + 35 * torch.cos(y / 7) * torch.sin(x / 53) + noise
    cy = torch.arange(h // 2, device=dev, dtype=torch.float32)[:, None]
    cx = torch.arange(w // 2, device=dev, dtype=torch.float32)[None, :]
    U = 128 + 25 * torch.sin(cx / 23) + torch.randn((h // 2, w // 2), generator=g, device=dev) * 5 + 0 * cy
    V = 128 + 25 * torch.cos(cy / 19) + torch.randn((h // 2, w // 2), generator=g, device=dev) * 5 + 0 * cx
    f = lambda a: a.round().clamp(0, 255).to(torch.uint8).contiguous()
    return f(Y), f(U), f(V)


def cpu_baseline(frame_np, qp, budget_s=15.0, gpu_ctus=()):
    """The oracle (plain-C restatement of the reference loop) on this box's host cores, 1 thread,
    on the first CTUs of the same workload.  A reported baseline, not the target.  As the checker it also compares the
    CU split decisions (depth per 4x4 partition, and the NxN flag of 8x8 CUs) the GPU published for the first CTUs of
    the same chain -- the second half of BASELINE's metric."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hmo_py
    enc = hmo_py.Encoder(*frame_np, qp)
    t0 = time.time()
    n = 0
    while n < enc.n_ctu and (time.time() - t0) < budget_s:
        enc.compress_ctu(n)
        n += 1
    dt = time.time() - t0
    same = total = 0
    for a, got in enumerate(gpu_ctus):
        if a >= n:
            break
        want = enc.ctu_arrays(a)
        same += int(((want["depth"] == got["depth"]) & (want["part_size"] == got["part_size"])).sum())
        total += want["depth"].size
    res = {"value": n / dt, "unit": "CTUs/sec", "cores": 1, "kind": "port",
           "sample": f"first {n} CTUs of frame 0 (3840x2160, QP{qp}), oracle/libhmo.so single thread, {dt:.1f}s"}
    match = {"ctus": min(n, len(gpu_ctus)), "partitions_equal": same, "partitions": total,
             "frac": (same / total) if total else None,
             "what": f"depth + part_size per 4x4 partition, GPU chain (frame 0, QP{qp}) vs the oracle on the same CTUs"}
    return res, match


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=2048, help="frames per GPU (x4 QPs = chains per GPU)")
    ap.add_argument("--ctus-per-step", type=int, default=1)
    ap.add_argument("--qps", default="22,27,32,37")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", choices=["frame", "qp"], default="qp",
                    help="chain order inside a launch: frame-major or (default) QP-major, lowest QP = longest chains first; "
                         "with more chains than wave slots the hardware dispatcher then backfills the slots freed by the "
                         "short high-QP chains")
    ap.add_argument("--state", choices=["training", "testing"], default="training",
                    help="fork state of the timed steps.  training (default, the headline metric): exhaustive RDO.  "
                         "testing: the fork's Naive pruning; the OBF maps come from the device pre-pass, the first warm-up "
                         "step runs in the Verifying state and its counters set the per-depth switches (SetDecisionSwitch)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # FCU_BENCH_REHEARSAL=1 (never set by the driver): rehearse the N>1 code path on a one-GPU box -- all ranks share
    # cuda:0 and rendezvous over gloo, since RCCL refuses two ranks on one device.  The reported number is then not a
    # scaling measurement and says so in `config`.
    rehearsal = os.environ.get("FCU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    qps = [int(q) for q in args.qps.split(",")]
    W, H = args.width, args.height
    n_chains = args.frames * len(qps)
    total_ctus_chain = (args.warmup + args.steps) * args.ctus_per_step
    eng = pkg.CuEngine(W, H, max_chains=n_chains, device=local)
    assert total_ctus_chain <= eng.n_ctu, "bench walks past the end of the frame"
    # inputs resident in HBM before timing; every chain gets its own reconstruction plane set,
    # the 4 QP chains of a frame share its source planes
    out_bytes = pkg.engine.CTU_OUT_BYTES * total_ctus_chain
    frames, cache = [], {}
    chain_list = pkg.sharding.chains_for_rank(args.frames, qps, rank)
    if args.order == "qp":
        chain_list = sorted(chain_list, key=lambda c: (c[1], c[0]))
    seed_index = {}
    for ci, (seed, qp) in enumerate(chain_list):
        if seed not in cache:
            cache[seed] = gen_textured_gpu(torch, dev, W, H, seed=seed)
            seed_index[seed] = len(frames)
            frames.append(cache[seed])
        fr = cache[seed]
        rec = [torch.zeros_like(p) for p in fr]
        out = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
        eng.init_chain(ci, fr, qp=qp, rec=rec, out=out)
    torch.cuda.synchronize()

    def step():
        eng.compress_chains(0, n_chains, args.ctus_per_step)

    switches = None
    if args.state == "testing":
        # untimed: OBF pre-pass of every frame, one Verifying step, switches from its counters, then Testing
        assert args.warmup >= 1, "--state testing uses the first warm-up step as the Verifying step"
        obf = [eng.obf_prepass(fr[0])[0][0].contiguous() for fr in frames]
        chain_obf = [obf[seed_index[seed]] for seed, _ in chain_list]
        for ci in range(n_chains):
            eng.set_decision(ci, pkg.engine.VERIFYING, chain_obf[ci])
        step()
        ver = eng.verify_counts(0, n_chains)
        switches = pkg.engine.decision_switch(ver)
        for ci in range(n_chains):
            eng.set_decision(ci, pkg.engine.TESTING, chain_obf[ci], *switches)
    for _ in range(args.warmup - (1 if args.state == "testing" else 0)):
        step()
    eng.sync()
    eng.kernel_ms()                       # drop warm-up launches from the event accumulator
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = pkg.sharding.reduce_step_time(dist, time.perf_counter() - t0, None if rehearsal else dev)
    kernel_ms, launches = eng.kernel_ms()

    ctus_per_step_gpu = n_chains * args.ctus_per_step
    total = ctus_per_step_gpu * args.steps * world
    value = total / dt
    if rank == 0:
        achieved = (ALGO_BYTES_PER_CTU * ctus_per_step_gpu) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        res = {
            "metric": "CTUs/sec (RDO decision only) at 4K all-intra", "value": value, "unit": "CTUs/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32+f64",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} all-intra QP{{{args.qps}}}, full depth-0..3 quadtree + chroma RDO "
                                   f"(BASELINE configs[2]); {args.frames} frames x {len(qps)} QPs = {n_chains} chains/GPU, "
                                   f"{args.ctus_per_step} CTU/chain/step, {args.order}-major order",
                       "chains_per_gpu": n_chains, "ctus_per_step": ctus_per_step_gpu * world,
                       **({"rehearsal": "all ranks on one GPU over gloo: not a scaling measurement"} if rehearsal else {}),
                       "state": args.state if switches is None else
                       f"testing (Naive switches skip2Nx2N={switches[0].tolist()} terminate={switches[1].tolist()} from a Verifying step)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic(n_chains, args.ctus_per_step) if switches is None else None,
                         "kernel": "fcu_ctu_engine", "kernel_ms": kernel_ms, "launches": launches},
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg runs at N=1 only
            fr0 = [p.cpu().numpy() for p in frames[0]]
            gpu_ctus = []
            seed0 = chain_list[0][0]                      # frames[0]: the lowest seed in either chain order
            if args.state == "training" and (seed0, 32) in chain_list:
                c32 = chain_list.index((seed0, 32))
                gpu_ctus = [eng.ctu_out(c32, a) for a in range(total_ctus_chain)]
            res["cpu_baseline"], res["split_flag_match"] = cpu_baseline(fr0, 32, gpu_ctus=gpu_ctus)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    eng.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
