#!/bin/bash
# per-kernel rocprofv3 --stats of one bench workload (run via gpurun): tools/kstats.sh <workload> [steps] -> gpurun_out/kstats_<workload>.csv
R=$GRAFT_REPO_ROOT; wl=$1; steps=${2:-20}; O=$R/gpurun_out/kstats_raw_$wl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 > $R/gpurun_out/kstats_$wl.log 2>&1
cp $(ls $O/*/*kernel_stats.csv | head -1) $R/gpurun_out/kstats_$wl.csv
rm -rf $O
