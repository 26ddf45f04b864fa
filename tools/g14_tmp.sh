R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g14; rm -rf $O; mkdir -p $O; cd $R
AOCR_HALO_BREG=1 python -m pytest tests/test_step_gpu.py -x -q -k "halo or c3_full_size" 2>&1 | tail -5 > $O/t.log
for i in 1 2; do
AOCR_HALO_BREG=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 2 > $O/bench_breg$i.json 2> $O/err.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 2 > $O/bench_base$i.json 2>> $O/err.log
done
cat $O/t.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/g14/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); fam=d['families']
    print(f.split('/')[-1], round(d['ms_per_step'],3), {k:round(fam[k]['ms_per_step'],3) for k in ('conv_fwd','conv_dgrad','conv_wgrad')}, round(d['roofline_best']['ms_per_launch'],4))
P
