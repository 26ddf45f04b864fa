#!/bin/bash
# HBM-side traffic of the two roofline kernels only (two bounded --pmc passes; run via gpurun): gpurun_out/prof4/{pmc_traffic.txt,wgrad_pmc.json,conv6_fwd_pmc.json}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof4; mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B $SHORT > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B $SHORT > /dev/null 2>&1
cd $R
f=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); w=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python3 tools/pmc_traffic.py $f $w $O > $O/pmc_traffic.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_traffic.txt
