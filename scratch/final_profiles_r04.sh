#!/bin/bash
# round-4 measurement batch (one gpurun call): default bench line, other configs, rocprofv3 kernel stats, PMC passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04
mkdir -p $O
SECONDS=0; python bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/r04_bench_final.json; echo "default bench wall: $SECONDS s"
echo "default done: $(python -c "import json;d=json.load(open('$O/r04_bench_final.json'));print(d['ms_per_step'], d['value'], d.get('selfcheck'), d.get('dp_mode_ms_per_step_1gpu'))")"
if [ "$1" != "quick" ]; then
python bench.py --mode infer --dtype fp32 --no-cpu-baseline --no-roofline > $O/c2_fp32.log 2>&1; tail -1 $O/c2_fp32.log > $O/c2_fp32.json
python bench.py --mode infer --dtype bf16 --no-cpu-baseline --no-roofline > $O/c2_bf16.log 2>&1; tail -1 $O/c2_bf16.log > $O/c2_bf16.json
python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-selfcheck > $O/c3_fp32.log 2>&1; tail -1 $O/c3_fp32.log > $O/c3_fp32.json
python bench.py --arch w48 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-selfcheck > $O/c4_w48.log 2>&1; tail -1 $O/c4_w48.log > $O/c4_w48.json
python bench.py --mode dcn --no-cpu-baseline > $O/c5_dcn.log 2>&1; tail -1 $O/c5_dcn.log > $O/c5_dcn.json
echo "configs done"
fi
rm -rf gpurun_out/rp_r04
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_r04 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/rp_r04.log 2>&1
python scratch/rpsum.py gpurun_out/rp_r04 > $O/rp_r04.sum 2>&1
cp gpurun_out/rp_r04/*/*_kernel_stats.csv $O/r04_kernel_stats.csv
find gpurun_out/rp_r04 -name '*_kernel_trace.csv' -delete
echo "rocprof stats done"
bash scratch/pmc_traffic.sh > $O/pmc_traffic.out 2>&1 && cp gpurun_out/traffic.json $O/traffic_r04.json
find gpurun_out/pmc_f gpurun_out/pmc_w -name '*.csv' -size +5M -delete
echo "traffic done"
bash scratch/pmc_mfma.sh > $O/pmc_mfma.out 2>&1 && cp gpurun_out/mfma_busy.json $O/mfma_busy_r04.json
find gpurun_out/pmc_m -name '*.csv' -size +5M -delete
echo "mfma done"
# DCN (config 5): kernel stats + HBM counters of the forward / backward launches (no counter evidence so far)
rm -rf gpurun_out/rp_dcn gpurun_out/pmc_dcn_f gpurun_out/pmc_dcn_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_dcn -- python bench.py --mode dcn --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/rp_dcn.log 2>&1
cp gpurun_out/rp_dcn/*/*_kernel_stats.csv $O/r04_dcn_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_f -- python bench.py --mode dcn --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_dcn_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_w -- python bench.py --mode dcn --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_dcn_w.log 2>&1
python scratch/pmc_dcn.py > $O/r04_dcn_traffic.json 2> $O/pmc_dcn.err
find gpurun_out/pmc_dcn_f gpurun_out/pmc_dcn_w gpurun_out/rp_dcn -name '*.csv' -size +5M -delete
echo "dcn done"
