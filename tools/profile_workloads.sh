#!/bin/bash
# Run ON THE GPU BOX from the repo root: one rocprofv3 process per workload and per pass (kernel trace; FETCH_SIZE; WRITE_SIZE —
# separate --pmc passes, no tracing flags with them).  usage: tools/profile_workloads.sh TAG [workload...]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_wl_$tag
W=${@:-config2_lvt config2_bfs config3_self config3_rays config3_rays_bfs config4_pair_lvt config4_pair_bfs timestep_1e6 timestep_1e7}
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
(cd "$R" && python3 -c "import bench; print(bench.csrc_sha())") > "$O/csrc_sha.txt" 2>/dev/null  # the sources these counters belong to
for w in $W; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$w/kt" -o r -- python3 "$R/tools/profile_workload.py" $w > "$O/$w.kt.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/$w/fetch" -o r -- python3 "$R/tools/profile_workload.py" $w > "$O/$w.fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/$w/write" -o r -- python3 "$R/tools/profile_workload.py" $w > "$O/$w.write.log" 2>&1
  # (the raw traces are large: keep the summaries only)
  find "$O/$w" -name "*kernel_trace.csv" -delete
  tail -1 "$O/$w.kt.log"
done
