#!/bin/bash
# Round-6 evidence on ONE box: GPU suite + smoke + default bench + rocprofv3 kernel statistics (gpu_final.sh), the counter
# passes of C3 (HBM with the trace figure, MFMA, SQ) and of the configs[4] shard (SQ), then same-box A/B of the round-6
# switches: tile map of k_round (HMX_ROUND_GA), pre-split Z planes and fused table of the wide regime.
export TMPDIR=/tmp
bash scripts/gpu_final.sh
bash scripts/gpu_pmc.sh
bash scripts/gpu_pmc_mfma.sh
bash scripts/gpu_pmc_sq.sh
bash scripts/gpu_pmc_sq_c5.sh
bash scripts/gpu_r6_envmatrix.sh "default:;classic_tile_map:HMX_ROUND_GA=0" "c3:10 c2:40 c4:5 c4x1:2" 2
cp gpurun_out/envmatrix.txt gpurun_out/ab_tile_map_final.txt
bash scripts/gpu_r6_envmatrix.sh "default:;launch_per_block:HMX_WIDE_SWEEP=0;lists_beside_sweep:HMX_LISTS_BESIDE_RTZ=0;launch_per_block_lists_beside_blocks:HMX_WIDE_SWEEP=0 HMX_LISTS_BESIDE_RTZ=0;z_rows_split_per_pass:HMX_RTZW_ZF=0;table_launches:HMX_FUSE_TABLE=0;round5_wide_path:HMX_RTZW_ZF=0 HMX_FUSE_TABLE=0" "c5:3" 2
cp gpurun_out/envmatrix.txt gpurun_out/ab_wide_final.txt
