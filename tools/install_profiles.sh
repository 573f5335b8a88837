#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh <tag>`: copy the collected files into profiles/<tag>_* (tracked) and regenerate the
# generated number blocks of DESIGN.md / README.md from them:   bash tools/install_profiles.sh r03
set -e
TAG=$1
cd "$(dirname "$0")/.."
SRC=gpurun_out/profiles_$TAG
cp $SRC/bench.json profiles/${TAG}_bench.json
cp $SRC/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $SRC/pmc_summary.txt profiles/${TAG}_render_pass_pmc.txt
cp $SRC/cv_kernel_stats_128x256.csv profiles/${TAG}_costvolume_unet2_kernel_stats_128x256.csv
{ echo "# tools/time_cost_volume.py on MI355X ($TAG, tools/profile_round.sh)"; grep -E "hypotheses|cost volume|CostRegNet|UNet2" $SRC/cv_128x256.log | grep -v "^first-generation"; grep "^first-generation" $SRC/cv_128x256.log; cat $SRC/cv_64x128.log; } > profiles/${TAG}_costvolume_timings.txt
for f in bench_config3 bench_config3_costregnet bench_config4_fine bench_config5_1gpu bench_f16x3 bench_views3 bench_views4 bench_train; do [ -s $SRC/$f.json ] && cp $SRC/$f.json profiles/${TAG}_$f.json; done
[ -s $SRC/train_kernel_stats.csv ] && cp $SRC/train_kernel_stats.csv profiles/${TAG}_train_kernel_stats.csv
for f in encoders_timing mvs_stage_timing_256x512 mvs_stage_timing_512x1024 winograd_vs_direct_layers unet2_layer_trace_128x256 frame_timing whole_frame_rank_shares row_tiles_per_rank train_demo views4_mem_pmc costregnet_layer_trace_128x256 costreg_conv_pmc views4_pair_pmc winograd_pmc unet2_conv_pmc cost_volume_pmc cost_volume_in_stage; do
  [ -s $SRC/$f.txt ] && cp $SRC/$f.txt profiles/${TAG}_$f.txt
done
python3 tools/summarize_profiles.py $TAG
ls profiles | grep "^${TAG}_"
