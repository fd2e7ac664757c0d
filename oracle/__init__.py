"""CPU oracle for the cvsteer hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (cvsteer_amd/) never does.  See oracle/cvsteer_oracle.h.
"""
from .pyoracle import *  # noqa: F401,F403
