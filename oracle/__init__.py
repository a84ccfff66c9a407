"""CPU oracle (test infrastructure only - see slotvps_oracle.py)."""
