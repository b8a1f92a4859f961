#!/bin/bash
# Collects everything profiles/<round>/ holds, on the GPU box:  bash tools/collect_profiles.sh gpurun_out/<round>
# (run through gpurun from the repo root; copy the directory into profiles/ afterwards)
set -e -o pipefail
OUT=${1:-gpurun_out/profiles}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 50 --warmup 5 --min-time 0.2 --no-cpu-baseline --no-extra-configs"
# 1. kernel trace + stats of the default bench configuration (three streams) and of the single-stream run
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/rp_default -- $B > $ROOT/$OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/rp_single -- $B --streams 1 > $ROOT/$OUT/bench_under_rocprof_single_stream.json 2>/dev/null
cp $ROOT/$OUT/rp_default/*/*_kernel_stats.csv $ROOT/$OUT/kernel_stats_b1_fp32_three_streams.csv
cp $ROOT/$OUT/rp_single/*/*_kernel_stats.csv $ROOT/$OUT/kernel_stats_b1_fp32_single_stream.csv
rm -rf $ROOT/$OUT/rp_default $ROOT/$OUT/rp_single
echo "kernel stats done"
# 1b. kernel stats of the batch-32 configurations (single stream)
for cfg in "b32_fp32:--batch 32 --steps 4" "b32_bf16:--batch 32 --dtype bf16 --steps 4"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/rp_$tag -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  cp $ROOT/$OUT/rp_$tag/*/*_kernel_stats.csv $ROOT/$OUT/kernel_stats_${tag}_single_stream.csv
  rm -rf $ROOT/$OUT/rp_$tag
done
echo "batch-32 kernel stats done"
# 1c. BASELINE config 4 (batch 32, nref 3, scales 0.5 / 1.0 / 1.5 from resident uint8 frames): kernel stats of three steps
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/rp_cfg4 -- python3 $ROOT/tools/cfg4_step.py 3 > $ROOT/$OUT/cfg4_steps.txt 2>/dev/null
cp $ROOT/$OUT/rp_cfg4/*/*_kernel_stats.csv $ROOT/$OUT/kernel_stats_cfg4.csv
rm -rf $ROOT/$OUT/rp_cfg4
echo "config 4 kernel stats done"
# 2. HBM-side traffic counters, one pass each (never together with other trace domains)
for cfg in "b1_fp32:--batch 1 --steps 20" "b32_fp32:--batch 32 --steps 3" "b32_bf16:--batch 32 --dtype bf16 --steps 3"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.05 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.05 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  (cd $ROOT && python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic_$tag.json "$tag 368x656 nref=1, bench.py --streams 1 $fl" > /dev/null)
  rm -rf $ROOT/$OUT/pmc_fetch $ROOT/$OUT/pmc_write
  echo "pmc traffic $tag done"
done
# 2b. matrix-pipe utilisation (its own pass)
for cfg in "b1_fp32:--batch 1 --steps 20" "b32_fp32:--batch 32 --steps 4" "b32_bf16:--batch 32 --dtype bf16 --steps 4"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_mfma -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.05 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  (cd $ROOT && python3 tools/pmc_mfma.py $OUT/pmc_mfma $OUT/mfma_util_$tag.txt "$tag" > /dev/null)
  rm -rf $ROOT/$OUT/pmc_mfma
  echo "mfma util $tag done"
done
# 2c. issue / LDS counters per kernel (two passes; tools/pmc_summary.py): VALU and LDS activity, bank conflicts, wait cycles
for cfg in "b1_fp32:--batch 1 --steps 20" "b32_bf16:--batch 32 --dtype bf16 --steps 3"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_sqa -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.05 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $ROOT/$OUT/pmc_sqb -- python3 $ROOT/bench.py --streams 1 $fl --warmup 1 --preroll 0 --min-time 0.05 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  (cd $ROOT && python3 tools/pmc_summary.py $OUT/pmc_sq_$tag.txt $OUT/pmc_sqa $OUT/pmc_sqb > /dev/null)
  rm -rf $ROOT/$OUT/pmc_sqa $ROOT/$OUT/pmc_sqb
  echo "sq counters $tag done"
done
cd $ROOT
# 3. per-launch tables (HIP events), the depthwise / stem roofline, the bf16 agreement figures and the un-profiled bench line
python3 tools/profile_layers.py --batch 1 > $OUT/launch_table_b1_fp32.txt 2>/dev/null
python3 tools/profile_layers.py --batch 32 > $OUT/launch_table_b32_fp32.txt 2>/dev/null
python3 tools/profile_layers.py --batch 32 --dtype bf16 > $OUT/launch_table_b32_bf16.txt 2>/dev/null
for sz in "368 368 s05" "368 656 s10" "552 984 s15"; do set -- $sz; python3 tools/profile_layers.py --batch 32 --nref 3 --height $1 --width $2 --reps 5 > $OUT/launch_table_cfg4_$3.txt 2>/dev/null; done
python3 tools/dw_roofline.py 32 > $OUT/depthwise_roofline_b32_fp32.txt 2>/dev/null
python3 tools/bf16_agreement.py 4 > $OUT/bf16_agreement.json 2>/dev/null
python3 tools/post_counts.py 32 fp32 > $OUT/post_counts_b32_fp32.txt 2>/dev/null
python3 tools/post_sweep.py 32 fp32 ";" > $OUT/post_chain_b32_fp32.txt 2>/dev/null
python3 tools/post_sweep.py 32 bf16 ";" > $OUT/post_chain_b32_bf16.txt 2>/dev/null
python3 tools/post_sweep.py 1 fp32 ";" > $OUT/post_chain_b1_fp32.txt 2>/dev/null
python3 tools/u8_probe.py > $OUT/u8_boundary_b1_fp32.txt 2>/dev/null
python3 tools/ms_bench.py 32 > $OUT/multiscale_step_b32.txt 2>/dev/null
echo "tables done"
python3 bench.py > $OUT/bench_default.json 2>/dev/null
# 4. multi-rank rehearsal of the N > 1 bench path on this ONE card (real engines, gloo instead of RCCL, 4 ranks: the box allows
#    at most 6 processes on its GPU); every rank checks its replica engines and the ranks' weight digests before timing
LWP_BENCH_DEVICE=0 LWP_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 4 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs > $OUT/bench_rehearsal_4ranks_one_card.json 2>/dev/null || echo "rehearsal failed"
echo done
