#!/usr/bin/env python3
"""Instruction mix of the loops of one gfx950 kernel (compile-only, no GPU).
usage: scripts/isa_loops.py <mangled-name substring> [--dump] [--lds-seq]   (compiles csrc/rldl_kernels.hip to /tmp/rldl_kernels.s once per change)
--lds-seq: per loop, the order of its LDS operations as the hardware will see them (a wave's LDS operations execute in program order),
run-length encoded: R = ds_read, W = ds_write, A = ds_add / ds_max (atomic), | = s_waitcnt lgkmcnt.  Lanes hand values to other lanes
through W and A, so a phase's reads must not appear in front of the previous phase's W / A: the sequence is the evidence."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "osqp_recursive_ldl_amd", "csrc", "rldl_kernels.hip")
out = "/tmp/rldl_kernels.s"
if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                           src, "-o", out], stderr=subprocess.DEVNULL)
txt = open(out).read()
pat = sys.argv[1]
for m in re.finditer(r"^(_Z\S*):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", txt, re.S | re.M):
    if pat not in m.group(1):
        continue
    lines = m.group(2).split("\n")
    print(m.group(1), len(lines), "lines")
    labels = {re.match(r"^(\.LBB\S+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"^(\.LBB\S+):", l)}
    cnt = lambda seg, p: sum(1 for x in seg if re.search(p, x))
    def mix(seg):
        return ("scratch %d ds_read %d ds_write %d ds_add %d global %d s_load %d fma64 %d mul64 %d VALU %d SALU %d waitcnt %d" % (
            cnt(seg, r"scratch_"), cnt(seg, r"\bds_read"), cnt(seg, r"\bds_write"), cnt(seg, r"ds_add|ds_max"), cnt(seg, r"\bglobal_"),
            cnt(seg, r"\bs_load"), cnt(seg, "v_fma_f64"), cnt(seg, "v_mul_f64"), cnt(seg, r"^\s+v_"), cnt(seg, r"^\s+s_"), cnt(seg, "s_waitcnt")))
    print("whole:", mix(lines))
    for i, l in enumerate(lines):
        mm = re.search(r"s_c?branch\S*\s+(\.LBB\S+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a = labels[mm.group(1)]
            print("loop %s [%d, %d] %d lines:" % (mm.group(1), a, i, i - a), mix(lines[a:i + 1]))
            if "--dump" in sys.argv:
                print("\n".join(lines[a:i + 1]))
            if "--lds-seq" in sys.argv:
                seq = []
                for x in lines[a:i + 1]:
                    c = "R" if re.search(r"\bds_read", x) else "W" if re.search(r"\bds_write", x) else "A" if re.search(r"ds_add|ds_max", x) else \
                        "|" if re.search(r"s_waitcnt.*lgkmcnt", x) else None
                    if c is None:
                        continue
                    if seq and seq[-1][0] == c:
                        seq[-1][1] += 1
                    else:
                        seq.append([c, 1])
                print("  LDS order:", " ".join("%s%s" % (c, n if n > 1 else "") for c, n in seq if c != "|" or True).replace("| ", "|"))
