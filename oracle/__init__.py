"""CPU oracle of the torch-rechub CTR hot path — TEST INFRASTRUCTURE, never imported by the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it.
See ``ctr_oracle.py`` (numpy restatement, reference file:line per function), ``ref_import.py`` (imports the
unmodified reference from /root/reference when it exists) and ``gen_golden.py`` (writes tests/golden/).
"""
