"""Re-export of the deterministic synthetic-workload helpers (they live in the package so that
bench.py, smoke() and tools/ share them)."""
from slotvps_amd.synth import *  # noqa: F401,F403
from slotvps_amd.synth import R50_HEAD_CFG, D  # noqa: F401
