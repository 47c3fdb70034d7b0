"""Re-export of the C-ABI field / phase enums for the SPH package."""
import os
import sys

_pkg_parent = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _pkg_parent not in sys.path:
    sys.path.insert(0, _pkg_parent)

from sph_project_amd._lib import *  # noqa: F401,F403,E402
from sph_project_amd import _lib as lib  # noqa: E402
from sph_project_amd import scene  # noqa: E402
