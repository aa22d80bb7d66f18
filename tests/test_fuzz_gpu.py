"""A fixed slice of the randomised differential test (tests/fuzz_parity.py): 120 random problems / option
combinations, every PDAS iteration against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_problems_match_the_oracle(gpu):
    import fuzz_parity
    assert fuzz_parity.run(cases=120, seed=20260101, verbose=False) == 0


def test_random_problems_with_chunk_chains_forced(gpu, monkeypatch):
    """The same differential test with every qualifying sequential path run as chunk chains side by side
    (BESSX_KPATH_CHAINS: paths of >= 16 candidates of all four families, weights, warm start, truncated fits): the
    untraced call must walk the path the traced -- single-chain -- call and the oracle walked."""
    import fuzz_parity
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "3")
    assert fuzz_parity.run(cases=90, seed=5, verbose=False) == 0


@pytest.mark.parametrize("hooks_", ["kchunks_merged=1", "panel=lds", "kchunks_merged=1,panel=lds", "kchunks_staged=0",
                                    "light_confirm=0,kchunks_shared_pass=0", "kchunks_pass_groups=2"])
def test_random_problems_with_the_selectable_variants_forced(gpu, monkeypatch, hooks_):
    """The differential test with the selectable variants switched on for every session -- round 5's: the chunk phase as
    merged launches on one stream (LM, covariance form), the fills by the panel kernels of rounds 2-4 (k_cov_panel_lds2 /
    _pair instead of k_cov_panel_dp, the default), round 4's fill rendezvous instead of staged fills; round 6's: every
    PDAS iteration queued whole and a pass over X per chain (rounds 1-5) instead of the light confirming iteration and the
    shared passes, and the shared passes as two alternating groups -- chunk chains forced as above."""
    import fuzz_parity
    monkeypatch.setenv("BESSX_KPATH_CHAINS", "3")
    monkeypatch.setenv("BESSX_TEST_HOOKS", hooks_)
    assert fuzz_parity.run(cases=45, seed=77, verbose=False) == 0
