"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the rectified-flow sampling hot path.

Nothing under ``rap_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline -- never as the product path.
"""
