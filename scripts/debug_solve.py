"""Compare the device solve against the CPU plan emulator with phases skipped (RLDL_DBG mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
from osqp_recursive_ldl_amd.linsys import plan_export
import oracle_bindings as ob
dbg = int(os.environ.get("RLDL_DBG", "0"))
n, m = int(os.environ.get("DN", "20")), int(os.environ.get("DM", "35"))
wl = R.workloads.SharedPatternQPs(n=n, m=m, density=0.2, pattern_seed=5)
B = 2
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rho = np.full((B, m), 0.1)
ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), 1e-6, t(rho), polish=0)
sym = ls.export_symbolic(); pl = plan_export(wl.P_pattern, wl.A_pattern)
rhs = np.random.default_rng(1).standard_normal((B, n + m))
out = ls.solve(t(rhs)).cpu().numpy()
f = ls.export_factor(0)
S = np.zeros(pl["nS"]); S[pl["LtoS"]] = f["Lx"]
w = pl["blob"]; u16 = lambda off: w[off:].view(np.uint16)
fsig, bsig, fcol = u16(pl["po_fsig"]), u16(pl["po_bsig"]), u16(pl["po_fcol"])
xs = rhs[0][sym["perm"]].copy(); Dinv = f["Dinv"]
ng = pl["ngroups"]; gs = w[pl["po_gstart"]:pl["po_gstart"] + ng + 1]
for k in range(ng):
    g0, g = int(gs[k]), int(gs[k + 1] - gs[k])
    fs0, fs1 = int(w[pl["po_fsp"] + k]), int(w[pl["po_fsp"] + k + 1])
    if fs1 > fs0 and not dbg & 1:
        ga = np.array([xs[fsig[g0 + i]] for i in range(g)])
        for tt in range(fs0, fs1):
            base, cnt = int(w[pl["po_fsb"] + tt]), int(w[pl["po_fsc"] + tt])
            for i in range(cnt): ga[i] -= S[base + i] * xs[fcol[base + i]]
        for i in range(g): xs[fsig[g0 + i]] = ga[i]
    if w[pl["po_gflag"] + k] and not dbg & 2:
        Tb = int(w[pl["po_gToff"] + k]); acc = xs[g0:g0 + g].copy()
        for a in range(g - 1):
            for i in range(a + 1, g): acc[i] -= S[Tb + i * (i - 1) // 2 + a] * acc[a]
        xs[g0:g0 + g] = acc
for k in range(ng - 1, -1, -1):
    g0, g = int(gs[k]), int(gs[k + 1] - gs[k])
    bs0, bs1 = int(w[pl["po_bsp"] + k]), int(w[pl["po_bsp"] + k + 1])
    scaled = False
    if bs1 > bs0 and not dbg & 4:
        gb = np.array([xs[bsig[g0 + i]] * Dinv[bsig[g0 + i]] for i in range(g)])
        for tt in range(bs0, bs1):
            base, cnt = int(w[pl["po_bsb"] + tt]), int(w[pl["po_bsc"] + tt])
            for i in range(cnt):
                rs = int(w[pl["po_brs"] + base + i]) & 0xffffffff
                gb[i] -= S[rs >> 16] * xs[rs & 0xffff]
        for i in range(g): xs[bsig[g0 + i]] = gb[i]
        scaled = True
    acc = xs[g0:g0 + g].copy() if scaled else xs[g0:g0 + g] * Dinv[g0:g0 + g]
    if w[pl["po_gflag"] + k] and not dbg & 8:
        Tb = int(w[pl["po_gToff"] + k])
        for il in range(g - 1, 0, -1):
            for j in range(il): acc[j] -= S[Tb + il * (il - 1) // 2 + j] * acc[il]
    xs[g0:g0 + g] = acc
sol = np.zeros(n + m); sol[sym["perm"]] = xs
exp = np.concatenate([sol[:n], rhs[0][n:] + sol[n:] / rho[0]])
d = np.abs(out[0] - exp)
print("dbg", dbg, "groups", [(int(gs[k]), int(gs[k+1])) for k in range(ng)], "max err", d.max(), "bad idx (perm pos)", [int(np.where(sym["perm"] == i)[0][0]) for i in np.where(d > 1e-9)[0]][:20])
