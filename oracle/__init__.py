"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's GP-MPC hot path.  Nothing
under ``oracle/`` is imported by the product package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it,
and only as the checker / reported baseline.
"""
