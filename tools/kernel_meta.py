"""VGPR / SGPR / scratch / LDS of every kernel in a built object or library, from the code object's metadata notes
(no recompilation).  usage: python tools/kernel_meta.py implicitbvh.jl_amd/csrc/ibvh_msd_finish.o [substring filters...]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
obj, filters = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    co = d + "/dev.co"
    kind = "o" if obj.endswith(".o") else "so"
    # (objects and shared libraries both carry the device code as a clang offload bundle in .hip_fatbin)
    sec = d + "/fatbin"
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + sec, obj, d + "/dummy"])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + sec,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    rows.append((name, get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, v, s, sc, lds), dn in sorted(zip(rows, dem), key=lambda x: x[1]):
    if all(f in dn for f in filters):
        print(f"vgpr {v:4d} sgpr {s:4d} scratch {sc:5d} lds {lds:6d}  {dn[:150]}")
