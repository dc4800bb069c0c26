"""CPU oracle for the CIPS-3D++ generator-forward hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this package, and only as the checker.
The product path (cips_3dplusplus_amd) never imports it and never falls back
to it: without the HIP library the product raises.

Parity status: PINNED.  tests/golden/make_golden.py imports the reference
(/root/reference, CPU, this container only) and stores its outputs as
fixtures under tests/golden/*.npz; tests/test_oracle_golden.py checks every
function here against them.
"""
from .path import *  # noqa: F401,F403
