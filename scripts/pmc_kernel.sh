#!/bin/bash
# PMC passes over any driver script: counters of one kernel, one pass per counter group (never together with a trace
# domain other than --kernel-trace).  usage (GPU box, repo root): bash scripts/pmc_kernel.sh OUTDIR "KERNEL_SUBSTR" script.py [args]
OUT=$1; KSUB=$2; shift 2
mkdir -p $OUT; rm -f $OUT/summary.txt
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmck_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmck_$i -- python3 $ROOT/$@ > /dev/null 2>&1
  f=$(ls /tmp/pmck_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$KSUB" >> $ROOT/$OUT/summary.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[2]:28s} {k:32s} per-dispatch avg {sum(v)/len(v):.6g}  sum {sum(v):.6g}  (n={len(v)})")
PY
  fi
done
cat $ROOT/$OUT/summary.txt
