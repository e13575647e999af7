"""A short, seeded slice of tests/fuzz_gpu.py (randomised differential testing against the oracle)."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_configurations_match_the_oracle(seed):
    import fuzz_gpu
    assert fuzz_gpu.main(seconds=8.0, seed=seed, verbose=False, light=True) >= 1


def test_large_random_chains_match_the_oracle():
    """A seeded slice of tests/fuzz_gpu_large.py: 1.3e5 .. 5e5 leaves (the sizes at which the MSD sort, its equalised cells and
    rescue workgroups, the shared descent and the binned rays are chosen), `cache=` chains through abrupt changes of the input."""
    import fuzz_gpu_large
    assert fuzz_gpu_large.main(seconds=15.0, seed=5, verbose=False, sizes=(131_072, 200_000, 524_289)) >= 1


def test_64_bit_queue_entries_match_the_oracle():
    """The BBox-node walker's 64-bit queue entries (trees of 29 .. 31 levels, > 134 M leaves) cannot be reached with an
    oracle-sized input; the development knob lvt_wide = 1 (ibvh_set_tuning, applied by the binding from IBVH_TUNING when the
    library is loaded) forces them for every tree.  The knob is process-wide, so the fuzzer runs in a child process
    (started, not exec'ed: this process has initialised the GPU)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, IBVH_TUNING="lvt_wide=1")
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_gpu.py"), "10", "7", "light"], cwd=here, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "fuzz ok" in out.stdout

