"""CPU oracle for the Genima hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline.  The product package ``genima_amd`` never
imports it and fails loudly when its HIP library is missing.  See each module's header for what pins it.
"""
