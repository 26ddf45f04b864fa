R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g7; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests/test_step_gpu.py tests/test_golden_gpu.py tests/test_ops_gpu.py tests/test_abi_harness_gpu.py -x -q -k "not bf16" 2>&1 | tail -8 > $O/t_f32.log
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_c2_lds.json 2> $O/bench_c2_lds.err
AOCR_NO_LDS_F32=1 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_c2_old.json 2> $O/bench_c2_old.err
cd /tmp && export TMPDIR=/tmp
AOCR_NO_SIDE_WGRAD=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $O/t_f32.log
AOCR_NO_SIDE_WGRAD=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 > $O/pmc_write.log 2>&1
echo "write rc=$?" >> $O/t_f32.log
cd $R
f=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); w=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $f $w $O > $O/pmc_traffic.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
tail -3 $O/pmc_fetch.log | cut -c1-300 >> $O/t_f32.log
cat $O/t_f32.log; cat $O/pmc_traffic.txt
