"""tvqaplus_amd: the MI355X (gfx950) drop-in for TVQAplus' ``model.stage.STAGE`` hot path.

Layout (only what the path needs):

* ``csrc/``        hand-written HIP kernels + the C-ABI (``include/stage_hip.h``), built into ``libstage_hip.so``
* ``_lib.py``      ctypes binding of that C-ABI (fails loudly if the library is missing)
* ``ops.py``       autograd wrappers around the C-ABI entry points
* ``groups.py``    one autograd node per fused-op group (the K-group entry points ``stage_grp_*``): the model's default path
* ``stage.py``     ``STAGE``: constructor / forward / state_dict compatible with the reference's model/stage.py
* ``att_host.py``  host side of the supervised-attention loss (reference model/stage.py:344-407)
* ``parallel.py``  one-process-per-GPU sharding of the batch and of the 5 answer candidates, flat gradient bucket
* ``prefetch.py``  pinned double-buffered host->HBM batch feed
* ``evaluation.py``span selection, prediction writer and the reference's temporal / grounding metrics
* ``synth.py``     synthetic batches of the BASELINE.json shapes

Nothing here imports ``oracle/`` (test infrastructure).
"""
