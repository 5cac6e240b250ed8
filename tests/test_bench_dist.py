"""bench.py's multi-GPU launch path on CPU (gloo, world size 2): `--gpus N` without a launcher starts N ranks as a child
torch.distributed.run before anything touches torch or HIP; the ranks shard the batch (weak: --batch per rank, strong:
--total-batch over all ranks), run the timed steps of bench.run_steps between barriers, take the MAX over ranks and gather
the result records.  The per-shard compute is the CPU ORACLE here (a stand-in for bench.GpuShard that only tests may
use -- the product's compute is the HIP library and needs a GPU): what is under test is the launch / shard / time /
gather skeleton that the driver's SCALE run depends on."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench
import osqp_recursive_ldl_amd as R
import oracle_bindings as ob

class OracleShard:                                    # CPU stand-in for bench.GpuShard (same interface)
    def __init__(self, wl, lo, count, kw):
        self.wl, self.lo, self.count, self.kw, self.calls = wl, lo, count, kw, 0
        self.res = None
    def step(self):
        self.calls += 1
        rs = []
        for b in range(self.lo, self.lo + self.count):
            P, q, A, l, u = self.wl.instance(b)
            rs.append(ob.OracleOSQP(P, q, A, l, u, **self.kw).solve())
        t = lambda a, dt=torch.float64: torch.tensor(np.array(a), dtype=dt)
        self.res = dict(x=t([r["x_iter"] for r in rs]), y=t([r["y_iter"] for r in rs]), obj=t([r["obj"] for r in rs]),
                        pri_res=t([r["pri_res"] for r in rs]), dua_res=t([r["dua_res"] for r in rs]),
                        iter=t([r["iter"] for r in rs], torch.int32), status=t([r["status"] for r in rs], torch.int32))
        return self.res
    def finish(self): pass
    def results(self): return self.res
    def device_sync(self): pass

args = bench.parse_args(sys.argv[1:])
world, rank, local_rank = bench.check_world(args)     # exits with 2 when --gpus disagrees with the launcher
dist.init_process_group("gloo")
n, m = 6, 9
total, sizes, scaling = bench.shard_plan(args, world)
wl = R.workloads.SharedPatternQPs(n=n, m=m, density=0.4, pattern_seed=4)
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=args.iters, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
shard = OracleShard(wl, sum(sizes[:rank]), sizes[rank], kw)
elapsed, per_rank, res, gather_ms = bench.run_steps(args, shard, n, m, sizes, scaling, world)
assert shard.calls == args.warmup + args.steps
assert res["x"].shape == (total, n) and len(per_rank) == world and elapsed == max(per_rank) and gather_ms is not None
ref = ob.OracleOSQP(*wl.instance(total - 1), **kw).solve()   # the last instance of the whole batch lives on the last rank
assert np.array_equal(res["x"][total - 1].numpy(), ref["x_iter"])
if rank == 0:
    print(json.dumps({"metric": "TEST-STUB (CPU oracle, not a measurement)", "n_gpus": world, "scaling": scaling, "batch_total": total,
                      "batch_per_gpu": sizes, "per_rank_seconds": per_rank, "steps": args.steps, "value": total * args.steps / elapsed}))
dist.destroy_process_group()
'''


def run_launcher(tmp_path, argv, env_extra=None):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT % {"root": ROOT})
    drv = tmp_path / "drv.py"
    drv.write_text("import sys\nsys.path.insert(0, %r)\nimport bench\nargs = bench.parse_args(sys.argv[1:])\n"
                   "sys.exit(bench.launch_ranks(args, sys.argv[1:], script=%r))\n" % (ROOT, str(script)))
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, str(drv)] + argv, capture_output=True, text=True, timeout=600, env=env)


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_gpus_2_launches_two_ranks_that_shard_time_and_gather(tmp_path, mode):
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--iters", "20"]
    argv += ["--batch", "3"] if mode == "weak" else ["--total-batch", "7"]
    out = run_launcher(tmp_path, argv)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["per_rank_seconds"]) == 2 and line["scaling"] == mode
    assert (line["batch_total"], line["batch_per_gpu"]) == ((6, [3, 3]) if mode == "weak" else (7, [4, 3]))


def test_gpus_disagreeing_with_the_launcher_is_refused(tmp_path):
    """A rank whose --gpus differs from WORLD_SIZE must not report (the round-1 bench printed n_gpus = WORLD_SIZE whatever --gpus said)."""
    sys.path.insert(0, ROOT)
    import bench
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; bench.check_world(bench.parse_args(['--gpus', '4']))" % ROOT],
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True)
    assert r.returncode == 2 and "refusing" in r.stderr
    assert bench.shard_plan(bench.parse_args(["--gpus", "8", "--total-batch", "65536"]), 8) == (65536, [8192] * 8, "strong")   # BASELINE config 4
    assert bench.shard_plan(bench.parse_args(["--gpus", "8"]), 8) == (8 * 4096, [4096] * 8, "weak")
