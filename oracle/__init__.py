"""CPU oracle for the DAS3R splat hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product path (``das3r_amd``, ``diff_gaussian_rasterization``, ``simple_knn``) never does and
fails loudly when the HIP library is missing.

PARITY UNPINNED: see the headers of ``raster_oracle.c`` / ``knn_oracle.c``.
"""
