"""CPU: the kernel's wave-uniform control flow (rc_correct_core.h, lane-serial build) must give
the oracle's results read for read.  This is what lets the search kernel be debugged without a GPU."""
import numpy as np
import pytest

import datasets


@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "k31_mc8", "skew", "nrich", "varlen", "k15", "k32", "pe_var", "edge", "long300", "long600_k31", "max1023", "k11", "polya_k23", "polya_k31", "polya_k15",
                                  "se_151", "pe_151", "pe_160_k15", "tiers_se", "tiers_pe", "tiers_il"])
def test_core_control_flow_matches_oracle(oracle, hostsim, name):
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    got = datasets.run_oracle(oracle, d, fn=lambda p, t, b: hostsim.hostsim_correct_batch(p, t, b, None))
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        assert np.array_equal(w, g), "%s differs on %s" % (what, name)
    assert (want[0] > 0).sum() > 0 or name == "edge"
