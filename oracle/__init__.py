"""CPU oracle for the rscm hot path -- TEST INFRASTRUCTURE ONLY.

Importable from ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
only.  The product package ``rscm_amd`` must never import this package.
"""
