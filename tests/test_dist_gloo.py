"""Multi-process path on CPU (gloo, world size 2): batch sharding + the path's single collective (all-gather of
the packed result records).  The per-shard compute is the CPU ORACLE here (tests may use it; the product's
compute is the HIP library and needs a GPU) -- what is under test is osqp_recursive_ldl_amd.dist."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import osqp_recursive_ldl_amd as R
from osqp_recursive_ldl_amd import dist as rd
import oracle_bindings as ob
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, n, m = 7, 8, 10                                    # 7 instances over 2 ranks: ragged shards 4 + 3
wl = R.workloads.SharedPatternQPs(n=n, m=m, density=0.3, pattern_seed=3)
lo, hi = rd.shard_range(B, rank, world)
Px, Ax, q, l, u = rd.sharded_values(wl.values, B, rank, world)
assert Px.shape[0] == hi - lo
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=50, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
def solve(b):
    P, qq, A, ll, uu = wl.instance(b)
    r = ob.OracleOSQP(P, qq, A, ll, uu, **kw).solve()
    return r
rs = [solve(b) for b in range(lo, hi)]
t = lambda a, dt=torch.float64: torch.tensor(np.array(a), dtype=dt)
res = dict(x=t([r["x_iter"] for r in rs]), y=t([r["y_iter"] for r in rs]), obj=t([r["obj"] for r in rs]),
           pri_res=t([r["pri_res"] for r in rs]), dua_res=t([r["dua_res"] for r in rs]),
           iter=t([r["iter"] for r in rs], torch.int32), status=t([r["status"] for r in rs], torch.int32))
full = rd.gather_results(res, n, m, sizes=rd.shard_sizes(B, world))
assert full["x"].shape == (B, n) and full["y"].shape == (B, m)
for b in range(B):                                    # every rank holds the whole batch, in order
    r = solve(b)
    assert np.allclose(full["x"][b].numpy(), r["x_iter"], rtol=0, atol=0)
    assert int(full["iter"][b]) == r["iter"] and int(full["status"][b]) == r["status"]
# equal shards take the tensor path
res2 = {k: v[:3] for k, v in res.items()}
full2 = rd.gather_results(res2, n, m)
assert full2["x"].shape == (3 * world, n)
# the pipelined gather of the benchmark (dist.ResultGather: one pack call + an asynchronous all_gather_into_tensor, two buffer
# pairs): four steps with results that differ per step and rank; after finish() every rank holds the LAST step's records of all
# ranks in rank order, and the buffers of earlier steps were reused without mixing steps
step = [0]
def pack(rec):
    rec.copy_(rd.pack_results({k: (v[:3] + (100 * step[0] + rank) if v.dtype == torch.float64 else v[:3]) for k, v in res.items()}, n, m))
gat = rd.ResultGather(pack, 3, n, m, "cpu")
for s_ in range(4):
    step[0] = s_
    gat.gather()
last = gat.finish()
ref = [torch.empty((3, n + m + 5), dtype=torch.float64) for _ in range(world)]
mine = rd.pack_results({k: (v[:3] + (300 + rank) if v.dtype == torch.float64 else v[:3]) for k, v in res.items()}, n, m)
dist.all_gather(ref, mine)
assert torch.equal(last["x"], torch.cat(ref, 0)[:, :n]) and torch.equal(last["obj"], torch.cat(ref, 0)[:, n + m])
assert last["x"].shape == (3 * world, n)
dist.barrier()
if rank == 0: print("DIST_OK")
dist.destroy_process_group()
'''


def test_shard_ranges_cover_the_batch():
    sys.path.insert(0, ROOT)
    from osqp_recursive_ldl_amd import dist as rd
    for B in (1, 7, 64, 65536):
        for W in (1, 2, 3, 8):
            r = [rd.shard_range(B, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[k][1] == r[k + 1][0] for k in range(W - 1))
            sz = rd.shard_sizes(B, W)
            assert sum(sz) == B and max(sz) - min(sz) <= 1


def test_gloo_world_size_2_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_OK" in out.stdout
