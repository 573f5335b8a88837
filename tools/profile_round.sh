#!/bin/bash
# Collect the round's rocprof evidence on the GPU box from the FINAL binary, in one go:  bash tools/profile_round.sh r03
# outputs under gpurun_out/profiles_<tag>/ ; then, back in the build container:
#   bash tools/install_profiles.sh r03       (copies them to profiles/<tag>_* and regenerates the number blocks of DESIGN.md / README.md)
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. plain bench line as the driver runs it (cpu baseline, bounded eager strip, end-to-end frames, PSNR block)
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 2000 $OUT/bench.json
# 2. kernel trace + stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc > $OUT/bench_trace.log 2>&1
cp $(find $OUT/bench_trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# 3. PMC passes (counters only + kernel trace)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $OUT/pmc$i.log 2>&1
done
python3 - > $OUT/pmc_summary.txt <<PY
import csv, glob, collections
print('# rocprofv3 --pmc passes on: python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc  (per-launch averages)')
for d in sorted(glob.glob('$OUT/pmc*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if 'render_pass' in row['Kernel_Name']:
                agg[row['Counter_Name']][0] += float(row['Counter_Value']); agg[row['Counter_Name']][1] += 1
        for cn, (v, n) in sorted(agg.items()):
            print(f'render_pass_kernel<2,2,false>  {cn:28s} {v/n:.6e}  (launches={n})')
PY
cat $OUT/pmc_summary.txt
# 4. cost volume + regularisers
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cv_trace -- python3 $R/tools/time_cost_volume.py 128 256 > $OUT/cv_128x256.log 2>&1
grep -E "hypotheses|cost volume|CostRegNet|UNet2" $OUT/cv_128x256.log
cp $(find $OUT/cv_trace -name "*kernel_stats.csv" | head -1) $OUT/cv_kernel_stats_128x256.csv
python3 $R/tools/time_cost_volume.py 64 128 2>&1 | grep -E "hypotheses|cost volume|CostRegNet|UNet2" | tee $OUT/cv_64x128.log
# 5. the other bench configurations (one JSON line each)
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fine > $OUT/bench_config4_fine.json 2>/dev/null
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --views 3 > $OUT/bench_views3.json 2>/dev/null
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --views 4 > $OUT/bench_views4.json 2>/dev/null
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision f16x3 > $OUT/bench_f16x3.json 2>/dev/null
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --with-cost-volume > $OUT/bench_config3.json 2>/dev/null
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --with-cost-volume --reg costregnet > $OUT/bench_config3_costregnet.json 2>/dev/null
python3 $R/bench.py --steps 3 --warmup 1 --views 4 --height 1024 --width 2048 > $OUT/bench_config5_1gpu.json 2>/dev/null      # with the PSNR block
# 6. per-frame 2-D encoders and the MVS stage (init net)
python3 $R/tools/time_encoders.py 2>&1 | grep -E "ms|err" > $OUT/encoders_timing.txt
python3 $R/tools/time_mvs.py 256 512 2>&1 | grep -E "ms|err" > $OUT/mvs_stage_timing_256x512.txt
python3 $R/tools/time_mvs.py 512 1024 2>&1 | grep -E "ms|err" > $OUT/mvs_stage_timing_512x1024.txt
# 7. Winograd vs direct 3-D convolution per UNet2 layer shape, and the per-dispatch durations of one UNet2 forward
python3 $R/tools/check_wino.py time 2>&1 | grep "winograd" > $OUT/winograd_vs_direct_layers.txt
bash $R/tools/layer_trace.sh unet2 128 256 2>&1 | grep -E "dur_us|sum_us" > $OUT/unet2_layer_trace_128x256.txt
# 8. whole frames: one GPU (overlap / reuse), per-rank shares of the 8-GPU split, row tiles; training step
python3 $R/tools/time_frame.py 2>&1 | grep -E "renderer\(data\)|max \|" > $OUT/frame_timing.txt
python3 $R/tools/time_frame_shares.py 8 2>&1 | grep -vE "amdgpu.ids" | tail -7 > $OUT/whole_frame_rank_shares.txt
python3 $R/tools/time_tiles.py 8 2>&1 | grep -vE "amdgpu.ids" | tail -8 > $OUT/row_tiles_per_rank.txt
python3 $R/tools/train_demo.py 30 1024 2>&1 | grep -E "step|rays x" > $OUT/train_demo.txt
# 9. memory-side counters of the 4-view kernel and the CostRegNet layer trace
bash $R/tools/pmc_mem.sh v4 --views 4 2>&1 | grep "per-launch" > $OUT/views4_mem_pmc.txt
bash $R/tools/trace_costreg.sh $TAG 128 256 2>&1 | grep -E " us$|sum " > $OUT/costregnet_layer_trace_128x256.txt
{ echo "# SQ counters of CostRegNet's conv kernels at 128x256x64, averages per (kernel, grid) over the launches of tools/time_cost_volume.py (tools/pmc_conv.sh: rocprofv3 --pmc, separate passes)";
  bash $R/tools/pmc_conv.sh $TAG costregnet 128 256 2>&1 | grep "mfma_busy"; } > $OUT/costreg_conv_pmc.txt
# 10. round 4: training step (bench line + kernel stats), SQ counters of the 4-view wave-pair kernel, the Winograd kernels (both
#     generations, per kernel name), UNet2's conv kernels and the cost-volume kernel
python3 $R/bench.py --train --steps 10 --warmup 3 > $OUT/bench_train.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -- python3 $R/bench.py --train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/train_trace.log 2>&1
cp $(find $OUT/train_trace -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
{ echo "# SQ counters of the 4-view render pass (render_pass_pair_kernel), per-launch averages: bash tools/pmc_cmd.sh v4 render_pass bench.py --views 4 ...";
  bash $R/tools/pmc_cmd.sh ${TAG}v4 render_pass $R/bench.py --views 4 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "per-launch"; } > $OUT/views4_pair_pmc.txt
{ echo "# rocprofv3 --kernel-trace --pmc (three passes) on: python3 tools/check_wino.py time (both kernel generations on the 8 UNet2 layer shapes); tools/pmc_wino.sh";
  echo "# mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); SQ_INSTS_VALU counts the MFMAs themselves (subtract 1 per MFMA)";
  bash $R/tools/pmc_wino.sh 2>&1 | grep -vE "amdgpu.ids|^W20"; } > $OUT/winograd_pmc.txt
{ echo "# SQ counters of UNet2's conv kernels at 128x256x64 (tools/pmc_conv.sh: rocprofv3 --pmc, separate passes), averages per (kernel, grid)";
  bash $R/tools/pmc_conv.sh ${TAG}u unet2 128 256 2>&1 | grep "mfma_busy"; } > $OUT/unet2_conv_pmc.txt
{ echo "# memory-side and SQ counters of cost_volume_kernel at 128x256x64, per-launch averages (tools/pmc_cmd.sh ... tools/time_cost_volume.py 128 256 costregnet)";
  bash $R/tools/pmc_cmd.sh ${TAG}cv cost_volume $R/tools/time_cost_volume.py 128 256 costregnet 2>&1 | grep "per-launch"; } > $OUT/cost_volume_pmc.txt
PANOGRF_WINO=32 python3 $R/tools/time_cost_volume.py 128 256 2>&1 | grep -E "UNet2" | sed 's/^/first-generation Winograd kernel (PANOGRF_WINO=32): /' >> $OUT/cv_128x256.log
# 11. round 6: the cost-volume build IN CONTEXT (caches polluted like a frame leaves them; build and build + regulariser per view, the
#     round-5 kernel beside the shipped one, plain vs streamed stores)
{ echo "# tools/cv_stage_nt.py on MI355X ($TAG): per step [1 GB read-modify-write, hypotheses, BUILD, regulariser]; median of 6 steps, two rounds; us per view";
  python3 $R/tools/cv_stage_nt.py 2>&1 | grep -vE "amdgpu.ids"; } > $OUT/cost_volume_in_stage.txt
rm -rf $OUT/pmc[0-9] $OUT/bench_trace $OUT/cv_trace $OUT/train_trace
ls $OUT
