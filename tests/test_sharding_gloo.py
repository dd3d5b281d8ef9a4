"""N>1 path on CPU: two gloo ranks each decide their own shard of chains (with the emulated engine) and
the union equals the single-process result; the step time is the MAX over ranks."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, zlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
import emu_py
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sums = []
for seed, qp in pkg.sharding.chains_for_rank(2, [27, 37], rank):
    Y, U, V = pkg.synth.mixed(64, 64, seed=seed)
    e = emu_py.EmuEncoder(Y, U, V, qp)
    e.compress_frame()
    a = e.ctu_arrays(0)
    sums.append(zlib.crc32(a["depth"].tobytes() + a["intra_dir"].tobytes() + a["coeff_y"].tobytes() + e.rec[0].tobytes()))
t = torch.tensor(sums, dtype=torch.int64)
out = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(out, t)
step = pkg.sharding.reduce_step_time(dist, 1.0 + rank)
if rank == 0:
    import json
    print("RESULT " + json.dumps([[o.tolist() for o in out], step]))
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(built, pkg, tmp_path):
    import zlib
    import emu_py
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    import json
    got = json.loads(line[len("RESULT "):])
    sums, step = got
    assert step == 2.0                                   # MAX over ranks of (1.0, 2.0)
    seen = set()
    for rank in range(2):
        want = []
        for seed, qp in pkg.sharding.chains_for_rank(2, [27, 37], rank):
            assert (seed, qp) not in seen                # shards are disjoint
            seen.add((seed, qp))
            Y, U, V = pkg.synth.mixed(64, 64, seed=seed)
            e = emu_py.EmuEncoder(Y, U, V, qp)
            e.compress_frame()
            a = e.ctu_arrays(0)
            want.append(zlib.crc32(a["depth"].tobytes() + a["intra_dir"].tobytes() + a["coeff_y"].tobytes() + e.rec[0].tobytes()))
        assert sums[rank] == want
