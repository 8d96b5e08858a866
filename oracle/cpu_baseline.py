"""CPU baseline leg of bench.py (TEST/BENCH INFRASTRUCTURE): times the oracle's full comparison
(oracle.sc_oracle.compare, randomize=True) on a bounded sample over N host processes.

Usage: python oracle/cpu_baseline.py KEYS.json PAILLIER_NAME DGK_NAME COUNT PROCS RBITS [SAMPLE.json]
With SAMPLE.json (written by bench.py::export_sample: rows of the GPU-resident batch -- ciphertext inputs, every random draw --
and the GPU's results for them) the oracle runs those very comparisons and reports how many of its results equal the GPU's
bit for bit ("match_gpu"); without it the inputs are drawn here from a seeded generator (same workload, same op mix).
Prints one JSON object.  Uses gmpy2 when the interpreter has it (the reference's optional fast path,
README.md:49), else Python's built-in pow -- the field "arith" says which.
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import sc_oracle as o  # noqa: E402

_G = {}


def _init(keys_path: str, pname: str, dname: str, rbits: int) -> None:
    k = json.load(open(keys_path))
    pj, dj = k[pname], k[dname]
    p, q = int(pj["p"], 16), int(pj["q"], 16)
    _G["sk"] = o.PaillierKey(p * q, p, q)
    _G["dgk"] = o.DGKKey(int(dj["p"], 16) * int(dj["q"], 16), int(dj["g"], 16), int(dj["h"], 16), int(dj["u"], 16), dj["t"],
                         int(dj["p"], 16), int(dj["q"], 16), int(dj["v_p"], 16), int(dj["v_q"], 16))
    _G["l"] = dj["l"]
    _G["rbits"] = rbits


def _work(seed: int) -> int:
    sk, dgk, l = _G["sk"], _G["dgk"], _G["l"]
    rng = random.Random(seed)
    x, y = rng.randrange(1 << l), rng.randrange(1 << l)
    dr = o.draw(rng, l, sk, dgk, _G["rbits"])
    res = o.compare(sk.enc_raw(x), sk.enc_raw(y), l, sk, dgk, dr, randomize=True)
    return int(sk.dec_raw(res) == int(x <= y))


def _work_row(row) -> tuple[int, int]:
    """One comparison of the exported batch: (result equals the GPU's, result decrypts to 0 or 1)."""
    sk, dgk, l = _G["sk"], _G["dgk"], _G["l"]
    H = lambda s: int(s, 16)  # noqa: E731
    dr = o.Draws(r=H(row["r"]), delta_a=row["delta_a"], rhos=[H(v) for v in row["rhos"]], perm=row["perm"], rho_z=H(row["rho_z"]),
                 r_d=H(row["r_bob"][0]), r_beta=[H(v) for v in row["r_bob"][1:]], r_c=[H(v) for v in row["r_alice"]],
                 rho_zeta1=H(row["rho_zeta_1"]), rho_zeta2=H(row["rho_zeta_2"]), rho_delta_b=H(row["rho_delta_b"]))
    res = o.compare(H(row["x_enc"]), H(row["y_enc"]), l, sk, dgk, dr, randomize=True)
    return int(res == H(row["gpu_result"])), int(sk.dec_raw(res) in (0, 1))


def main() -> None:
    keys_path, pname, dname = sys.argv[1:4]
    count, procs, rbits = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    sample = json.load(open(sys.argv[7])) if len(sys.argv) > 7 else None
    out = {}
    with mp.Pool(procs, initializer=_init, initargs=(keys_path, pname, dname, rbits)) as pool:
        pool.map(_work, range(procs))  # warm-up: imports, key objects
        if sample is None:
            t0 = time.time()
            ok = sum(pool.map(_work, range(1000, 1000 + count), chunksize=1))
            dt = time.time() - t0
        else:
            fields = [k for k in sample if k != "l"]
            count = min(count, len(sample["x_enc"]))
            rows = [{k: (sample[k][i] if sample[k] is not None else None) for k in fields} for i in range(count)]
            t0 = time.time()
            got = pool.map(_work_row, rows, chunksize=1)
            dt = time.time() - t0
            ok = sum(g[1] for g in got)
            out["match_gpu"] = sum(g[0] for g in got)
    out.update({"value": count / dt, "seconds": dt, "count": count, "cores": procs, "correct": ok,
                "arith": "gmpy2" if o._HAVE_GMPY2 else "python-int"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
