#!/bin/bash
# Per-launch listing of one train step (kernel trace of a short bench run): bash tools/step_trace.sh [filter] -> gpurun_out/st/trace.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/st; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 > $O/bench.log 2>&1
cd $R
python tools/step_trace.py $(ls $O/raw/*/*kernel_trace.csv | head -1) "${1:-gemm}" > $O/trace.txt 2>&1
rm -rf $O/raw
cat $O/trace.txt | cut -c1-200
