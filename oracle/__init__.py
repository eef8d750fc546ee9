"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference PARSDMM hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker.  The product path (``setintersectionprojection.jl_amd``)
never imports this package and fails loudly when its HIP library is missing.
"""
