cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/mp; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mstats -- python3 scripts/pmc_run_mpc.py > $O/mstats.log 2>&1 &&
cp $O/mstats/*/*kernel_stats.csv $O/mpc_kernel_stats.csv &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/m1 -- python3 scripts/pmc_run_mpc.py > $O/m1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/m2 -- python3 scripts/pmc_run_mpc.py > $O/m2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/m3 -- python3 scripts/pmc_run_mpc.py > $O/m3.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/m4 -- python3 scripts/pmc_run_mpc.py > $O/m4.log 2>&1 &&
python3 scripts/pmc_reduce.py mpc $O/pmc_mpc.json $O/m1 $O/m2 $O/m3 $O/m4 > /dev/null && head -12 $O/mpc_kernel_stats.csv
