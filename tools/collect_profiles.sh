#!/bin/bash
# Regenerates the round's profile artefacts on the GPU box (run via gpurun from the repo root).
# (every pass runs under its own `timeout`: a profiler pass once sat for 35 minutes on a box and ate the whole call)
# Outputs under gpurun_out/prof/: kernel stats of the default bench run, the two PMC passes for the conv6-forward
# traffic, the bench JSON line.  Copy the summaries into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench_c3_bf16.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python $R/bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 1 > $O/bench_c2.log 2>&1
timeout 300 python $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python $R/bench.py --workload c4 --steps 48 --warmup 24 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 300 python $R/bench.py --workload ref --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_ref.json 2> $O/bench_ref.err
cd $R
f=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); w=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $f $w $O > $O/pmc_traffic.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/c3_bf16_kernel_stats.csv
cp $(ls $O/stats_c2/*/*kernel_stats.csv | head -1) $O/c2_f32_kernel_stats.csv
# keep the merge small: drop the raw traces
rm -rf $O/stats $O/stats_c2 $O/pmc_fetch $O/pmc_write
tail -1 $O/bench_c3_bf16.json | cut -c1-400; cat $O/pmc_traffic.txt; tail -1 $O/bench_c2.log | cut -c1-300
