"""CPU oracle for the MuLUT LUT-inference hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker.  ``mulut_amd`` never does.

Two restatements of the reference's algorithm live here, both pinned against fixtures produced by
running the reference itself (tests/golden/gen_golden.py; checked by tests/test_oracle.py):

* ``c_oracle``  -- exact-integer C (oracle/mulut_oracle.c), fast enough for full-size parity.
* ``np_port``   -- NumPy port with the reference's own operation structure (16 corner gathers +
  24 masked simplex cases in float), used as the "reference's CPU path" timing baseline.
"""
