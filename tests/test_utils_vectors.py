"""to_bits / from_bits of the product and of the oracle against vectors produced by the reference's own utils.py
(tests/golden/gen_utils_vectors.py; reference: utils.py:6-38)."""
import json
import os

import pytest

from conftest import GOLDEN
from oracle import sc_oracle as o
from protocols.secure_comparison_amd import utils as product_utils

VEC = json.load(open(os.path.join(GOLDEN, "utils_vectors.json")))


@pytest.mark.parametrize("impl", [o, product_utils], ids=["oracle", "product"])
def test_to_from_bits(impl):
    for c in VEC["to_bits"]:
        bits = impl.to_bits(int(c["value"]), c["bit_length"])
        assert bits == c["bits"] and impl.from_bits(bits) == int(c["roundtrip"])
    for c in VEC["overflow"]:
        assert c["raises"]
        with pytest.raises(AssertionError):
            impl.to_bits(1 << c["bit_length"], c["bit_length"])
    for c in VEC["from_bits_nonbinary"]:
        assert impl.from_bits(c["bits"]) == int(c["value"])
