"""ORACLE -- test infrastructure only, never the product path.

CPU restatements of the reference's hot path, each pinned to golden vectors captured from the imported
reference (tests/golden/make_golden.py, tests/test_oracle_golden.py):

* mvf_numpy.py  closed-form MVF forward/backward in numpy (fp64 capable)
* mvf_ref.c     the same in plain scalar C (built by oracle/Makefile into libmvf_oracle.so; loader mvf_c.py)
* net_torch.py  functional restatement of the whole recognizer step on CPU torch ops

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything here.
"""
