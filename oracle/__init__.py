"""CPU oracle of the batched step engine -- TEST INFRASTRUCTURE ONLY (see oracle/mjo.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
