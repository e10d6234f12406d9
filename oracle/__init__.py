"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the reference hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import from here, and only as the checker.  The shipped path
(``waveglow_amd``) never imports this package and fails loudly when its HIP
library is missing.
"""
