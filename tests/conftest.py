import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def keys():
    return json.load(open(os.path.join(GOLDEN, "keys.json")))


def oracle_paillier(keys, bits):
    from oracle import sc_oracle as o

    k = keys[f"paillier_{bits}"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    return o.PaillierKey(p * q, p, q)


def oracle_dgk(keys, name):
    from oracle import sc_oracle as o

    k = keys[name]
    p, q = int(k["p"], 16), int(k["q"], 16)
    return o.DGKKey(p * q, int(k["g"], 16), int(k["h"], 16), int(k["u"], 16), k["t"], p, q, int(k["v_p"], 16), int(k["v_q"], 16))


@pytest.fixture(scope="session")
def engine():
    """The HIP engine (GPU tests only).  No fallback: a missing library or GPU is a hard failure."""
    from protocols.secure_comparison_amd.schemes import default_engine

    eng = default_engine()
    # the parity tests use small batches; keep them on each modulus's own kernel configuration.  The small-batch (2G, 9)
    # kernels are exercised explicitly by the tests that take the `latency_engine` fixture.
    eng.set_latency_mode(0)
    return eng


@pytest.fixture()
def latency_engine(engine):
    """The same engine with the small-batch kernel configurations forced on for the duration of one test."""
    engine.set_latency_mode(2)
    yield engine
    engine.set_latency_mode(0)


@pytest.fixture()
def onelane_engine(engine):
    """The same engine with the one-lane (1, 37) kernels forced on for every modulus they fit, whatever the batch size."""
    engine.set_onelane_mode(2)
    yield engine
    engine.set_onelane_mode(1)
