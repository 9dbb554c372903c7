"""CPU oracle for the linked-cell pair-force hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(ls1-mardyn_amd/) must never do so.
"""
