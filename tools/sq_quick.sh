#!/bin/bash
# Run ON THE GPU BOX: instruction counters of the LVT count pass (one rocprofv3 --pmc pass, no tracing flags) for a quick look at
# a development build.  usage: tools/sq_quick.sh OUTTAG [IBVH_LIB path]
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_$tag
[ -n "$2" ] && export IBVH_LIB=$R/$2
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$O/a" -o r -- python3 "$R/bench.py" --no-cpu-baseline --no-configs --extra-n 0 --steps 3 --warmup 1 > "$O/a.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lvt_" not in k: continue
        k = k.split("(")[0][-90:]
        c = acc[k][r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
for k, v in acc.items():
    e = {n: c[0] / c[1] for n, c in v.items()}
    w = e.get("SQ_WAVES", 1)
    print(k, {n.replace("SQ_INSTS_", ""): round(x / w, 1) for n, x in e.items() if n != "SQ_WAVES"}, "waves", w)
PY
