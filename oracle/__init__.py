"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (slicer_amd) never does.
"""
from .oracle import *  # noqa: F401,F403
