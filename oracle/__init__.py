"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the ConvVAE hot path.

Nothing under ``oracle/`` may be imported by the product (``vae-npvc_amd/``).
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

PARITY UNPINNED: the reference (TensorFlow 1.2.1 graph code) cannot be run
here (TensorFlow is not installable, no network) and ships no tests or golden
vectors, so this oracle is a restatement pinned only by (a) two independent
implementations agreeing with each other (float64 NumPy direct definition vs
PyTorch-CPU ops + autograd) and (b) analytic known-answer tests.
"""
