"""The multi-GPU leg in plain C (csrc/rldl_dist.c): RCCL opened with dlopen, communicator from a 128-byte unique id, the path's one
collective = an all-gather of the packed result records.  One rank on the one GPU of the test box: the collective itself runs (RCCL
initialises, ncclAllGather executes on the device) and must hand back this rank's own records in order."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


def test_c_level_rccl_gather_with_one_rank():
    import osqp_recursive_ldl_amd as R
    from osqp_recursive_ldl_amd import _lib
    L = _lib.lib()
    wl = R.workloads.SharedPatternQPs()
    B = 96
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=2000, check_termination=25, adaptive_rho=1, eps_abs=1e-4, eps_rel=1e-4, warm_start=0, scaling=10)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    uid = C.create_string_buffer(128)
    rc = L.osqp_dist_unique_id(uid)
    if rc == 3:
        pytest.skip("librccl.so is not on this box")
    assert rc == 0
    d = C.c_void_p()
    assert L.osqp_dist_init(C.byref(d), uid, 0, 1, None) == 0
    reclen = int(L.osqp_dist_record_len(w.h))
    assert reclen == wl.n + wl.m + 5
    out = torch.full((B, reclen), float("nan"), dtype=torch.float64, device="cuda:0")
    for _ in range(2):                                            # (the second call reuses the packed-record buffer)
        assert L.osqp_dist_gather_results(d, w.h, C.c_void_p(out.data_ptr())) == 0
        torch.cuda.synchronize()
    assert torch.equal(out[:, :wl.n], r["x"]) and torch.equal(out[:, wl.n:wl.n + wl.m], r["y"])
    assert torch.equal(out[:, wl.n + wl.m], r["obj"]) and torch.equal(out[:, wl.n + wl.m + 1], r["pri_res"]) and torch.equal(out[:, wl.n + wl.m + 2], r["dua_res"])
    assert torch.equal(out[:, wl.n + wl.m + 3].to(torch.int32), r["iter"]) and torch.equal(out[:, wl.n + wl.m + 4].to(torch.int32), r["status"])
    # pack only (callers with their own collective, bench.py): the same records as the torch packing of dist.py
    from osqp_recursive_ldl_amd import dist as rdist
    rec = torch.full((B, reclen), float("nan"), dtype=torch.float64, device="cuda:0")
    w.pack_results(rec)
    torch.cuda.synchronize()
    assert torch.equal(rec, out) and torch.equal(rec, rdist.pack_results(r, wl.n, wl.m))
    # error behaviour: bad rank / size are refused before RCCL is touched
    bad = C.c_void_p()
    assert L.osqp_dist_init(C.byref(bad), uid, 2, 2, None) == 1 and not bad.value
    L.osqp_dist_free(d)
    w.cleanup()
