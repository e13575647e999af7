"""Extract one kernel's ISA from `hipcc -S --cuda-device-only` output.  usage: kernel_asm.py file.s substr [substr...] > out.s"""
import subprocess
import sys
lines = open(sys.argv[1]).read().split("\n")
subs = sys.argv[2:]
labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l]
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in labels), capture_output=True, text=True).stdout.split("\n")
for (i, n), d in zip(labels, names):
    if all(s in d for s in subs):
        j = i
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        sys.stderr.write(d + "\n")
        print("\n".join(lines[i:j]))
        break
