"""CPU oracle for the VFace per-frame DDIM denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU fp32 restatement of the reference
algorithm (Sanoojan/VFace, ``REFace/``); every function cites the reference file:line it follows.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the reported CPU baseline.  Nothing under ``vface_amd/`` imports it: the
product path runs on hand-written HIP kernels and raises when the HIP library is missing.

Parity pin: the oracle is checked against golden vectors produced by importing the reference itself
in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; the reference has
no tests or KATs of its own).  See ``tests/test_oracle_golden.py``.
"""
