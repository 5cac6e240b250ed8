#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): PMC passes (fused kernel: SQ counters; solve kernel: HBM
# traffic), kernel stats of the bench command, the bench line, secondary configs.  Outputs under gpurun_out/final; copy the
# summaries into profiles/ (the *.json files are what bench.py reads back).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/sq1 -- python3 scripts/pmc_run.py > $O/sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python3 scripts/pmc_run.py > $O/sq2.log 2>&1 &&
python3 scripts/pmc_reduce.py fused $O/pmc_fused.json $O/sq1 $O/sq2 > /dev/null &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 scripts/pmc_run_solve.py > $O/pf.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 scripts/pmc_run_solve.py > $O/pw.log 2>&1 &&
python3 scripts/pmc_reduce.py traffic $O/pmc_traffic_solve.json $O/pf $O/pw k_tile_solve3 $((20104 * 4096)) 4096 > /dev/null &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mstats -- python3 scripts/pmc_run_mpc.py > $O/mstats.log 2>&1 &&
cp $O/mstats/*/*kernel_stats.csv $O/mpc_kernel_stats.csv &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/m1 -- python3 scripts/pmc_run_mpc.py > $O/m1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/m2 -- python3 scripts/pmc_run_mpc.py > $O/m2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/m3 -- python3 scripts/pmc_run_mpc.py > $O/m3.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/m4 -- python3 scripts/pmc_run_mpc.py > $O/m4.log 2>&1 &&
python3 scripts/pmc_reduce.py mpc $O/pmc_mpc.json $O/m1 $O/m2 $O/m3 $O/m4 > /dev/null &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1 &&
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv &&
timeout -k 10 600 python3 bench.py > $O/bench.log 2>&1 && tail -1 $O/bench.log > $O/bench.json &&
timeout -k 10 600 python3 scripts/bench_configs.py all > $O/configs.log 2>&1 && grep '^{' $O/configs.log > $O/configs.jsonl &&
timeout -k 10 300 python3 scripts/solve_probe.py --batches 1024,4096,8192 --json $O/solve_timeline.json > $O/solve_probe.log 2>&1 &&
timeout -k 10 300 python3 scripts/factor_bench.py 64 1024 4096 > $O/factor_timeline.txt 2>&1
echo "rc=$?"; cut -c1-300 $O/bench.json; head -c 400 $O/pmc_traffic_solve.json
rm -rf $O/sq1 $O/sq2 $O/pf $O/pw $O/stats $O/mstats $O/m1 $O/m2 $O/m3 $O/m4
