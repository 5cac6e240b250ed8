#!/usr/bin/env python3
"""Register / occupancy report of the gfx950 kernels (compile-only, no GPU): hipcc -Rpass-analysis=kernel-resource-usage on
csrc/rldl_kernels.hip, one line per kernel whose demangled name contains any of the given substrings.
usage: scripts/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "osqp_recursive_ldl_amd", "csrc", "rldl_kernels.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
pats = sys.argv[1:] or [""]
for b in re.split(r"(?=remark: [^\n]*Function Name:)", r.stderr):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^\(anonymous namespace\)::|^void \(anonymous namespace\)::", "", name).split("(")[0]
    if not any(p in name for p in pats):
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print("%-44s VGPR %3s AGPR %3s SGPR %3s spill %3s scratch %4s B/lane, %s waves/SIMD, LDS %s B" % (
        name[:44], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
