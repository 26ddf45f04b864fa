R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g8; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
AOCR_SIDE_PLAIN=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_plain -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > $O/pmc_plain.log 2>&1
echo "plain side stream under --pmc rc=$?" > $O/res.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_prio -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > $O/pmc_prio.log 2>&1
echo "priority side stream under --pmc rc=$?" >> $O/res.log
rm -rf $O/pmc_plain $O/pmc_prio
cd $R
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_c2.json 2> $O/bench_c2.err
AOCR_SIDE_PLAIN=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 2 > $O/bench_plain.json 2> $O/bench_plain.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 2 > $O/bench_prio.json 2> $O/bench_prio.err
cat $O/res.log; grep -h -o '"ms_per_step": [0-9.]*' $O/bench_c2.json $O/bench_plain.json $O/bench_prio.json | head -6; tail -3 $O/pmc_prio.log | cut -c1-200
