"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the SVC hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker / reported CPU
baseline.  The product (``whisper-vits-svc_amd/svcmi``) never imports it and has
no CPU fallback: without the HIP library it raises.

Parity status: PINNED.  The restatement is validated against the reference
modules themselves (imported from /root/reference in the build container by
``oracle/make_golden.py``) and against the golden vectors that script commits
under ``tests/golden/``.  The reference ships no tests or golden vectors of its
own, so executing the reference is the only available pin.
"""
