#!/bin/bash
# Run ON THE GPU BOX from the repo root.  Counter passes (one rocprofv3 --pmc pass per group, no tracing flags):
#   sq_a / sq_b / sq_c   SQ issue counters of the bench step's kernels (1e6 leaves)      -> tools/sq_summary.py
#   kt_configs           kernel trace of tools/bench_configs.py (configs 2-4: rays, BFS, pair, skew cases)
#   cfg_fetch / cfg_write / cfg_sq   FETCH_SIZE, WRITE_SIZE and SQ counters of the same script's kernels -> tools/pmc_summary.py
# usage: tools/profile_sq.sh <tag>   -> gpurun_out/prof_sq_<tag>/
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sq_$tag
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
(cd "$R" && python3 -c "import bench; print(bench.csrc_sha())") > "$O/csrc_sha.txt" 2>/dev/null  # the sources these counters belong to
common="--no-cpu-baseline --no-configs --extra-n 0 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$O/sq_a" -o r -- python3 "$R/bench.py" $common > "$O/sq_a.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/sq_b" -o r -- python3 "$R/bench.py" $common > "$O/sq_b.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$O/sq_c" -o r -- python3 "$R/bench.py" $common > "$O/sq_c.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_configs" -o r -- python3 "$R/tools/bench_configs.py" > "$O/kt_configs.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/cfg_fetch" -o r -- python3 "$R/tools/bench_configs.py" > "$O/cfg_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/cfg_write" -o r -- python3 "$R/tools/bench_configs.py" > "$O/cfg_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$O/cfg_sq" -o r -- python3 "$R/tools/bench_configs.py" > "$O/cfg_sq.log" 2>&1
ls "$O"
