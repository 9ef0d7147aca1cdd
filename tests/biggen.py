"""The synthetic workload generator lives in the package (aligngraph2_amd/workload.py: bench.py measures on it); the tests
keep their old name for it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligngraph2_amd.workload import *  # noqa: E402,F401,F403
from aligngraph2_amd.workload import _mapper_starts  # noqa: E402,F401
