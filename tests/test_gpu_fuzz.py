"""A short budget of the randomised C-ABI cross-check (tools/gpu_fuzz.py): random modulus sizes over every kernel configuration
and both small-batch policies, adversarial limb patterns, every primitive compared with Python integers."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_randomised_primitives_against_python_ints(engine):
    import gpu_fuzz

    try:
        rounds = gpu_fuzz.run(25.0, seed=20261003, eng=engine, verbose=False)
    finally:
        engine.set_latency_mode(0)
    assert rounds >= 10
