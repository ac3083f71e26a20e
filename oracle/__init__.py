"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the STAGE hot path of jayleicn/TVQAplus, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
Nothing under ``tvqaplus_amd/`` (the product) may import this package.
"""
