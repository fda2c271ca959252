"""CPU restatement of the reference's matching-statistics path (kbo_oracle.c + binding.py).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker,
never by the product (kbo_amd/), which fails loudly without its HIP library.  Pinned against the reference's own golden
vectors (tests/test_oracle_golden.py, tests/golden/).
"""
