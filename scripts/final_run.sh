# The command sequence that produces the profiles/r03_* files (run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash scripts/final_run.sh'); outputs land in gpurun_out/r03/ and are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/r03_final_bench.json 2> $O/r03_final_bench_ops.txt
python bench.py --config 3 --vols 16 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r03_levels16_bench.json 2> $O/r03_levels16_bench_ops.txt
python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r03_lits_bench.json 2> $O/r03_lits_bench_ops.txt
EFFQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 1 --warmup 1 > $O/r03_dp2_rehearsal_gloo_one_gpu.json 2> $O/r03_dp2_rehearsal.log
python scripts/prof_fp_bracket.py > $O/r03_fp_bracket_vs_passes.txt 2>&1
python scripts/prof_fp_traj.py > $O/r03_fp_traj_vs_older.txt 2>&1
EFFQ_FP_TRAJ_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-subrun --no-conv-subrun 2>&1 | grep "fp_traj" | sed -e "s/'lo'.*'calls'/'calls'/" -e "s/'trace_us'[^]]*\], //" > $O/r03_fp_traj_stats.txt
bash scripts/pmc_prox.sh gpurun_out/r03/r03_pmc_prox > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); cp $f $O/r03_final_bench_kernel_stats.csv
python3 $R/scripts/kstats.py $O/r03_final_bench_kernel_stats.csv 2 > $O/r03_final_kernel_summary.txt
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1); python3 $R/scripts/timeline.py $t 15 40 > $O/r03_timeline.txt 2>&1
echo done
