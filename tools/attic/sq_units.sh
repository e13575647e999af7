#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqr_units; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
             "SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d "$O/b$i" -o r -- python3 "$R/tools/profile_workload.py" config3_rays 3 > "$O/b$i.log" 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/b*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rays_subtree" not in k: continue
        k = k.split("(")[0].split("<")[0][-40:]
        c = acc[k][r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
for k, v in acc.items():
    print(k)
    for n, c in sorted(v.items()): print(f"    {n:34s} {c[0] / c[1]:16.1f}  (x{c[1]})")
PY
