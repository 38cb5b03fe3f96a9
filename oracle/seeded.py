"""Kept for the fixtures' generating scripts and the tests: the recipe itself lives in humannerf_amd/seeded.py
(it is data, and bench.py's GPU legs need it without importing anything of the oracle)."""
from humannerf_amd.seeded import *            # noqa: F401,F403
from humannerf_amd.seeded import default_shapes, seeded_state    # noqa: F401
