"""ORACLE -- CPU restatement of the reference algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under chore_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.
Each function cites the reference file:line it restates.  Pinned by the golden vectors in
tests/golden/ (generated from the reference itself by tests/golden/make_golden.py).
"""
