"""A short, seeded slice of tests/fuzz_gpu.py (randomised differential testing against the oracle)."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_configurations_match_the_oracle(seed):
    import fuzz_gpu
    assert fuzz_gpu.main(seconds=8.0, seed=seed, verbose=False) >= 1
