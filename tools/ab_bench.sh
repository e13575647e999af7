#!/bin/bash
# Run ON THE GPU BOX: A/B of development builds (tools/build_variant.sh): bench.py headline + per-kernel times per variant.
# usage: tools/ab_bench.sh OUTDIR TAG [TAG...]   (TAG "main" = the shipped in-tree library)
out=$1; shift
mkdir -p $out
for tag in "$@"; do
  if [ "$tag" = main ]; then unset IBVH_LIB; else export IBVH_LIB=$PWD/variants/libibvh_$tag.so; fi
  python bench.py --no-cpu-baseline --no-configs > $out/ab_$tag.json 2> $out/ab_$tag.err
  python - $out/ab_$tag.json $tag <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernels"]; ns = d["north_star_1e7"]
print(f"{sys.argv[2]:8s} step {d['ms_per_step']:.4f} enq {d['ms_per_step_enqueue_only']:.4f} count {k['lvt_queue_kernel_count']['avg_ms']*1e3:.1f} write {k['lvt_queue_kernel_write']['avg_ms']*1e3:.1f} "
      f"sort {d['roofline']['morton_sort_phase']['ms']*1e3:.1f}us | 1e7 step {ns['ms_per_step']:.3f} count {ns['kernels_ms'].get('lvt_queue_kernel_count',0):.3f} sort {ns['morton_sort_phase']['ms']:.3f} contacts {d['config']['contacts_total']} {ns['contacts']}")
PY
done
