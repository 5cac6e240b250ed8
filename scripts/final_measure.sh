#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): PMC traffic passes, kernel stats, bench line, secondary configs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 scripts/pmc_run.py > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 scripts/pmc_run.py > $O/pmc_write.log 2>&1 &&
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1 &&
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv &&
timeout -k 10 600 python3 bench.py > $O/bench.log 2>&1 && tail -1 $O/bench.log > $O/bench.json &&
timeout -k 10 600 python3 scripts/bench_configs.py all > $O/configs.log 2>&1 && grep '^{' $O/configs.log > $O/configs.jsonl
echo "rc=$?"; cut -c1-200 $O/bench.json; cat $O/pmc_traffic.json | head -c 600
rm -rf $O/pmc_fetch $O/pmc_write $O/stats
