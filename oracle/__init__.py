"""CPU oracle package (TEST INFRASTRUCTURE ONLY — see oracle/sc_oracle.py header)."""
