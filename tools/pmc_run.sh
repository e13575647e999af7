#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir-under-gpurun_out> "<counters>" [bench args...]
# One rocprofv3 counter pass over a short bench.py run (no tracing flags: gpurun refuses --pmc + traces).
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; ctrs="$2"; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o run -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --extra-n 0 "$@" > "$out/bench.log" 2>&1
