"""Generate tests/golden/utils_vectors.json by importing the REFERENCE's own utils.py (the only reference module
that loads without its un-vendored dependencies) by file path.  Run in the build container only:
    python tests/golden/gen_utils_vectors.py
"""
import importlib.util
import json
import os
import random

REF = "/root/reference/src/tno/mpc/protocols/secure_comparison/utils.py"
spec = importlib.util.spec_from_file_location("ref_sc_utils", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

rng = random.Random(20261003)
cases = []
for bl in (1, 2, 8, 16, 18, 32, 34, 64, 66, 100):
    vals = [0, 1, (1 << bl) - 1, (1 << bl) // 2] + [rng.randrange(1 << bl) for _ in range(8)]
    for v in vals:
        bits = mod.to_bits(v, bl)
        cases.append({"value": str(v), "bit_length": bl, "bits": bits, "roundtrip": str(mod.from_bits(bits))})
overflow = []
for bl in (1, 16, 32):
    try:
        mod.to_bits(1 << bl, bl)
        overflow.append({"bit_length": bl, "raises": False})
    except AssertionError:
        overflow.append({"bit_length": bl, "raises": True})
odd = [{"bits": [1, 0, 2, 1], "value": str(mod.from_bits([1, 0, 2, 1]))}, {"bits": [], "value": str(mod.from_bits([]))}]
json.dump({"source": REF, "to_bits": cases, "overflow": overflow, "from_bits_nonbinary": odd},
          open(os.path.join(os.path.dirname(__file__), "utils_vectors.json"), "w"), indent=0)
print(len(cases), "cases")
