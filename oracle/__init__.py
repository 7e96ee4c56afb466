"""CPU restatement of the reference's ray-marching path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
path (``tinynerf_amd``) never does and fails loudly without ``libtinynerf_hip.so``.  Pinned to the goldens under
``tests/golden/`` (captured from the reference by ``oracle/make_goldens.py``).
"""
