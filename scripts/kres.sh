#!/bin/bash
# compile the kernels for gfx950 and print register / spill usage of the kernels matching $1 (default: arrow admm)
cd /root/repo/osqp_recursive_ldl_amd/csrc || exit 1
pat=${1:-k_arrow_admmILi3ELi24}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only rldl_kernels.hip -o /tmp/k.s -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
grep -i "error" -A4 /tmp/res.txt | head -20
grep -A12 "$pat" /tmp/res.txt | grep -i "Function Name\|VGPRs:\|Spill\|Scratch\|Occupancy" | sed 's/.*remark: *//' | head -12
