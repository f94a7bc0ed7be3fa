"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the defectGAN G+D train step.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline.  The product path
(``de-i2i-gan_amd``) never imports this package and fails loudly when the HIP
library is missing.

Parity status: PINNED.  ``tests/golden/gen_goldens.py`` imports the reference
itself (``/root/reference/defectGAN``) in the build container, checks this
restatement against it and writes the fixtures under ``tests/golden/``; the
``-m "not gpu"`` suite re-checks the restatement against those fixtures.
"""
