#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/r4d_avail.txt 2>&1
grep -c . $O/r4d_avail.txt
for set in "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/r4d_pmc_$tag -o p -- python3 $R/bench.py --workload synth10k_4320p --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/r4d_pmc_$tag.log 2>&1 || { echo "FAILED $set"; tail -3 $O/r4d_pmc_$tag.log; }
  echo "done $set"
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out")
acc=collections.defaultdict(list)
for f in glob.glob(os.path.join(O,"r4d_pmc_*","**","*_counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "qr_render_kernel<false" in r["Kernel_Name"] or "qr_render_kernelILb0" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, sum(acc[k])/len(acc[k]), len(acc[k]))
PY
