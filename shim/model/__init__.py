"""Import shim: put this directory (``<repo>/shim``) in front of ``PYTHONPATH`` and the reference drivers' own line
``from model.stage import STAGE`` (main.py:13, inference.py:7) resolves to the MI355X implementation -- main.py and
inference.py stay byte-for-byte unchanged.  Only ``model.stage`` is provided: it is the only
``model.*`` module the drivers import."""
