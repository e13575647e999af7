#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS / cache counters of config 3's ray traversal, one rocprofv3 --pmc pass per counter group (no
# tracing flags), for the binned path (default knobs) and the binary walker (IBVH_TUNING=rays_binned=0).
# usage: tools/sq_rays.sh OUTTAG [extra IBVH_TUNING items]
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqr_$tag
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
(cd "$R" && python3 -c "import bench; print(bench.csrc_sha())") > "$O/csrc_sha.txt" 2>/dev/null  # the sources these counters belong to
for variant in binned walker; do
  if [ $variant = walker ]; then export IBVH_TUNING="rays_binned=0"; else export IBVH_TUNING="$2"; fi
  i=0
  for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
               "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
               "SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $group --output-format csv -d "$O/$variant$i" -o r -- python3 "$R/tools/profile_workload.py" config3_rays 3 > "$O/$variant$i.log" 2>&1
  done
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
for variant in ("binned", "walker"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(sys.argv[1] + f"/{variant}*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rays" not in k: continue
            k = k.split("(")[0].split("<")[0][-40:] + ("<W>" if ", true" in r["Kernel_Name"].split("(")[0] else "")
            c = acc[k][r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
    print("==", variant)
    summary = {}
    for k, v in acc.items():
        print(k)
        summary[k] = {n: round(c[0] / c[1], 1) for n, c in sorted(v.items())}
        for n, c in sorted(v.items()):
            print(f"    {n:34s} {c[0] / c[1]:16.1f}  (x{c[1]})")
    import json
    json.dump(summary, open(sys.argv[1] + f"/summary_{variant}.json", "w"), indent=1)
PY
