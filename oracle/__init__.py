"""CPU oracle for the YOLOX-24p hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

A plain torch-CPU / numpy restatement of the reference's algorithm for SURVEY.md section 8 rows a0-a13,
each function citing the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the checker; nothing
under ``exploration-of-potential_amd/`` imports it and the product path fails loudly without its HIP library.

Parity is PINNED: every function here is checked against golden vectors produced by running the reference
itself on this container's CPU (tests/golden/*.npz, generator tests/golden/make_golden.py; the reference
ships no tests or fixtures of its own).

Why torch and not C: the path is fp32 ATen arithmetic (acos/atan2/sin/exp/log, vectorised reductions); the
reference's CPU results are defined by those ATen kernels, so the restatement calls the same primitives in
the same order and is bit-identical on the golden vectors for the index-valued outputs.
"""
