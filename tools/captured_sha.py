"""The hash of the kernel sources a capture directory under gpurun_out/ was taken at: the capture scripts (tools/profile_*.sh,
tools/sq_rays.sh) leave it in csrc_sha.txt ON THE GPU BOX; a directory without the file (older captures) is stamped with the
current sources' hash, as before.  A capture older than the current sources is reported."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha  # noqa: E402


def captured_sha(directory):
    now = csrc_sha()
    p = os.path.join(directory, "csrc_sha.txt")
    if os.path.exists(p):
        got = open(p).read().strip()
        if got and got != now:
            print(f"WARNING: {directory} was captured at csrc {got}, the sources are now {now}: the files are stamped {got} and bench.py will not quote them",
                  file=sys.stderr)
        if got:
            return got
    return now
