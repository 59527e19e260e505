"""CPU oracles -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under checkm_amd/ imports it (tests/test_boundary.py enforces that).
"""
