#!/bin/bash
# GPU test-suite under each kernel-selection switch (fallback paths are product code); run on the GPU box from the repo root.
for sw in RLDL_NO_TILE RLDL_NO_MFMA RLDL_NO_STAGE_PROD RLDL_NO_ARROW RLDL_NO_ARROW_FACTOR RLDL_NO_STAGE_FACTOR RLDL_STAGE_LDS RLDL_NO_STAGE RLDL_NO_STAGE_SOLVE RLDL_HORIZON_FULL RLDL_SCALE_LOOPS RLDL_CHECK_STAGED RLDL_ITERS_PER_LAUNCH RLDL_HORIZON_MULTI RLDL_SOLVE_V2 RLDL_PROD_V1 RLDL_TILE_SCATTER RLDL_SPLIT_INVERT RLDL_SOLVE_BEGIN_LAUNCH RLDL_ASSEMBLE_SCATTER; do
  env $sw=1 timeout -k 10 300 python -m pytest tests -x -q -m gpu > gpurun_out/sw_$sw.log 2>&1
  echo "$sw: $(tail -1 gpurun_out/sw_$sw.log)"
done
