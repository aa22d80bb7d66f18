"""A fixed slice of the randomised differential test (tests/fuzz_parity.py): 150 random problems / option
combinations, every PDAS iteration against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_problems_match_the_oracle(gpu):
    import fuzz_parity
    assert fuzz_parity.run(cases=150, seed=20260101, verbose=False) == 0
