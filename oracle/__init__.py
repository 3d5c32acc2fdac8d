"""CPU oracle for the CODLAD sampling hot path.  TEST INFRASTRUCTURE ONLY.

A restatement, in plain PyTorch-CPU fp32 (float64 numpy for the schedule), of the
reference algorithm between the noise draw and the Cartesian coordinates
(reference test.py:504-582).  Each function cites the reference file:line it follows.

Pinned: every function here is checked against golden vectors produced by importing
and running the reference itself in the build container (tools/gen_golden.py ->
tests/golden/*.npz, tests/test_oracle_vs_golden.py).

Nothing under codlad_amd/ imports this package.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may use it, and only as the checker / the timed CPU
baseline - never as a fallback for the HIP path.
"""
