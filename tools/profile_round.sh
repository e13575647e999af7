#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- tools/profile_round.sh [tag]).  Collects, for the bench command,
#   kt*/    rocprofv3 --kernel-trace --stats   (per-kernel durations; must agree with bench.py's HIP-event timings)
#   fetch*/ rocprofv3 --pmc FETCH_SIZE          (separate passes, no tracing flags: MI355X_MICROARCH.md §HBM)
#   write*/ rocprofv3 --pmc WRITE_SIZE
# at n = 1e6 (the bench default) and n = 1e7 (the north-star size) into gpurun_out/prof_<tag>/.
# tools/pmc_traffic.py turns them into the profiles/ files.
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
(cd "$R" && python3 -c "import bench; print(bench.csrc_sha())") > "$O/csrc_sha.txt" 2>/dev/null  # the sources these counters belong to
common="--no-cpu-baseline --no-configs --extra-n 0"
for n in 1000000 10000000; do
  s=$([ $n = 1000000 ] && echo "" || echo "_1e7")
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt$s" -o r -- python3 "$R/bench.py" --n $n $common > "$O/kt$s.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch$s" -o r -- python3 "$R/bench.py" --n $n --steps 5 --warmup 2 $common > "$O/fetch$s.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write$s" -o r -- python3 "$R/bench.py" --n $n --steps 5 --warmup 2 $common > "$O/write$s.log" 2>&1
done
ls -R "$O" | head -40
