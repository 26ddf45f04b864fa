#!/bin/bash
# Profile artefacts of a round (run via gpurun from the repo root: `bash tools/round_profiles.sh r06`): everything lands in gpurun_out/prof_<tag>/;
# copy the summaries into profiles/<tag>_*.
# Every pass is bounded by its own `timeout`; the program follows `--` directly (python3 bench.py ...), counters are collected in passes of their own.
set -u
TAG=${1:?usage: round_profiles.sh <round tag, e.g. r06>}
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (this script runs on the GPU box, through gpurun)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 2; }
O="$R/gpurun_out/prof_$TAG"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0"
timeout 600 $B > $O/bench_c3_bf16.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B $SHORT > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B $SHORT > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -- $B $SHORT > /dev/null 2>&1
for w in c2 c3s c4 c5 ref; do
  st=10; wu=3; [ $w = c4 ] && st=48 && wu=24; [ $w = c5 ] && st=4 && wu=2
  timeout 300 $B --workload $w --steps $st --warmup $wu --no-cpu-baseline --no-secondary --sustain-seconds 1 > $O/bench_$w.json 2> $O/bench_$w.err
done
# BASELINE config 5 names no batch: the default line above is 256 strips per GPU (the encoder's groups cover the chip); the smaller batches beside it
for b in 16 64 128; do timeout 300 $B --workload c5 --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --sustain-seconds 0 > $O/bench_c5_b$b.json 2> $O/bench_c5_b$b.err; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- $B --workload c2 --compute f32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3s -- $B --workload c3s --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 > /dev/null 2>&1
cd "$R"
# what the MFMA pipe sustains on resident random fragments, after 5000 warm-up launches (bench.py reads the best "MFMA on resident random fragments" row)
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc gemm4w.hip -o gemm4w 2>/dev/null; timeout 300 ./gemm4w 5000 quick > $O/gemm4w_steady.txt 2>&1)
f=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); w=$(ls $O/pmc_write/*/*counter_collection.csv | head -1); mf=$(ls $O/pmc_mfma/*/*counter_collection.csv | head -1)
python3 tools/pmc_traffic.py $f $w $O > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_hbm_kernels.py $f $w $(ls $O/pmc_fetch/*/*kernel_trace.csv | head -1) > $O/hbm_pmc.txt 2>&1
python3 tools/pmc_mfma_summary.py $mf > $O/mfma_busy.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/c3_bf16_kernel_stats.csv
cp $(ls $O/stats_c3s/*/*kernel_stats.csv | head -1) $O/c3s_bf16_kernel_stats.csv
cp $(ls $O/stats_c2/*/*kernel_stats.csv | head -1) $O/c2_f32_kernel_stats.csv
# one step of C3 / C2 as an ordered launch list, and the clock / MFMA-busy pass of the fp32 kernels
bash tools/ktrace_step.sh c3; cp gpurun_out/ktrace_step_c3.txt $O/c3_step_trace.txt
bash tools/ktrace_step.sh c2 --compute f32; cp gpurun_out/ktrace_step_c2.txt $O/c2_step_trace.txt
bash tools/ktrace_step.sh ref; cp gpurun_out/ktrace_step_ref.txt $O/ref_step_trace.txt
bash tools/ktrace_step.sh c5; cp gpurun_out/ktrace_step_c5.txt $O/c5_step_trace.txt
bash tools/pmc_clock.sh c2 --compute f32; cp gpurun_out/pmc_clock_c2.txt $O/c2_pmc_clock.txt
rm -rf $O/stats $O/stats_c3s $O/stats_c2 $O/pmc_fetch $O/pmc_write $O/pmc_mfma
# cycles per phase of the decoder chain kernels (training forward / backward, beam-5 decode): stamps build of the library
if [ -f $R/torch-attention-ocr_amd/aocr/libaocr_stamps.so ]; then
  AOCR_LIB=$R/torch-attention-ocr_amd/aocr/libaocr_stamps.so timeout 120 python3 tools/ch_stamp.py > $O/chain_stamps.txt 2>&1
  AOCR_LIB=$R/torch-attention-ocr_amd/aocr/libaocr_stamps.so timeout 120 python3 tools/debug/beam_stamp.py 5 2>&1 | grep -v "wid " >> $O/chain_stamps.txt
  for w in c3 c5; do echo "--- encoder cluster kernels, $w shape" >> $O/chain_stamps.txt; AOCR_LIB=$R/torch-attention-ocr_amd/aocr/libaocr_stamps.so timeout 120 python3 tools/debug/enc_stamp.py $w 2>&1 | grep -v amdgpu >> $O/chain_stamps.txt; done
fi
tail -1 $O/bench_c3_bf16.json | cut -c1-300; cat $O/hbm_pmc.txt; head -20 $O/mfma_busy.txt
