# The command sequence that produces the profiles/r04_* files (run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash scripts/final_run_r04.sh'); outputs land in gpurun_out/r04/ and are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench_ops.txt
python bench.py --config 3 --vols 16 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r04_levels16_bench.json 2> $O/r04_levels16_bench_ops.txt
python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r04_lits_bench.json 2> $O/r04_lits_bench_ops.txt
EFFQ_DP_FORCE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r04_dp_forced_one_rank.json 2> /dev/null
EFFQ_DP_FORCE=1 EFFQ_DP_GATHER_FIT=0 EFFQ_RCCL_DIRECT=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > $O/r04_dp_forced_one_rank_r3_path.json 2> /dev/null
python scripts/prof_inv.py 865 1729 3457 6913 13825 > $O/r04_inverse.txt 2>&1
EFFQ_GJ_WIDE=0 python scripts/prof_inv.py 6913 13825 > $O/r04_inverse_rank64.txt 2>&1
python scripts/prof_rccl.py > $O/r04_rccl_one_rank.txt 2>&1
python scripts/prof_fp_traj.py > $O/r04_fp_traj_vs_older.txt 2>&1
python scripts/iter_series.py 16 8 > $O/r04_iter_series.txt 2>&1
EFFQ_SIDE=0 python scripts/iter_series.py 16 8 > $O/r04_iter_series_no_side_streams.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subrun --no-conv-subrun > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); cp $f $O/r04_final_bench_kernel_stats.csv
python3 $R/scripts/kstats.py $O/r04_final_bench_kernel_stats.csv 2 > $O/r04_final_kernel_summary.txt
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1); python3 $R/scripts/timeline.py $t 15 40 > $O/r04_timeline.txt 2>&1
python3 $R/scripts/inv_timeline.py $t 0.3 > $O/r04_inverse_timeline_profiled.txt 2>&1
python3 $R/scripts/layer_breakdown.py $t > $O/r04_layer_breakdown_profiled.txt 2>&1
cd $R
bash scripts/pmc_kernel.sh gpurun_out/r04/r04_pmc_gj_big "k_gj_big" scripts/prof_inv.py 13825 > /dev/null 2>&1
bash scripts/pmc_prox.sh gpurun_out/r04/r04_pmc_prox > /dev/null 2>&1
echo done
