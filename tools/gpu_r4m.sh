#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 200 python tools/gpu_synth_probe.py > $O/r4m_probe.txt 2>&1; grep -v amdgpu $O/r4m_probe.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r4m_pmc_$c -o p -- python3 $R/bench.py --workload synth10k_4320p --steps 3 --warmup 2 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/r4m_pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out")
acc=collections.defaultdict(list)
for f in glob.glob(os.path.join(O,"r4m_pmc_*","**","*_counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "qr_render_kernel<false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, sum(acc[k])/len(acc[k]), len(acc[k]))
PY
