"""CPU oracle of the SEA hot path -- test infrastructure only (see sea_oracle.py)."""
