"""GPU: a seeded 12-case subset of tools/fuzz_parity.py inside the suite (round 6) -- both models, forward + backward, graphs of
random size and degree (edge counts that are not multiples of the 64-row tiles, missing modalities) against the fp32 CPU oracle,
float64 adjudication where a gradient tensor exceeds the L2 bound (a ReLU unit within rounding of zero), and bitwise repeatability
of the HIP side.  The full sweeps (24 + 30 cases) stay in the tool; their logs are under profiles/."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
def test_seeded_random_graphs_match_the_oracle():
    import fuzz_parity
    lines = []
    bad, worst_out, worst_l2 = fuzz_parity.run_cases(cases=12, seed=11, max_nodes=500, out=lines.append)
    assert len(lines) >= 12
    assert bad == 0, "\n".join(lines)
    assert worst_out < 1e-4, (worst_out, lines)
