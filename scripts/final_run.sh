set -e
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.txt 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r02_tests_gpu.txt 2>&1
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench_ops.txt
python bench.py --net lits --vols 8 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun > gpurun_out/r02_lits_bench.json 2> gpurun_out/r02_lits_bench_ops.txt
rm -rf gpurun_out/r02_pmc_final; bash scripts/pmc_conv.sh i8_32 k_conv3d_i8l2e gpurun_out/r02_pmc_final > /dev/null 2>&1
bash scripts/pmc_prox.sh gpurun_out/r02_pmc_prox > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_final_bench_kernel_stats.csv
echo done
