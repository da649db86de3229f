"""CPU oracle of the EMAGE hot path — test infrastructure only (see emage_oracle.py)."""
