"""ORACLE package – CPU restatements of the reference's hot-path algorithms.

Test infrastructure only.  Nothing under ``video-diffusion-pipeline-parallel_amd/`` imports
this package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker / reported baseline.
"""
