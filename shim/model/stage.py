"""``model.stage`` as the reference's drivers import it (main.py:13, inference.py:7): the class is tvqaplus_amd's."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from tvqaplus_amd.stage import STAGE  # noqa: E402,F401

__all__ = ["STAGE"]
