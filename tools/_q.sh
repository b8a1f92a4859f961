set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for wl in 1 0; do
  export LWP_STEM_WL=$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/rp_wl$wl -- python3 $ROOT/bench.py --steps 50 --warmup 5 --min-time 0.3 --no-cpu-baseline --no-extra-configs --streams 1 > /dev/null 2>&1
  grep -h "stem_kernel" $ROOT/gpurun_out/rp_wl$wl/*/*_kernel_stats.csv | cut -c1-120
  rm -rf $ROOT/gpurun_out/rp_wl$wl
done
cd $ROOT
python tools/layer_sweep.py 1 fp32 "LWP_STEM_WL=1;LWP_STEM_WL=0;LWP_STEM_WL=1|LWP_STEM_TY=4;LWP_STEM_WL=0|LWP_STEM_TY=4" model.0
