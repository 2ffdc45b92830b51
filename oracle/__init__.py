"""CPU oracle for the scoring path -- test infrastructure, never imported by foodrec_amd (see m2d_oracle.py)."""
