"""Import shim: the product package lives in the directory ``zero-chain_amd/`` (the name the
project layout prescribes, which is not a valid Python identifier).  ``import zero_chain_amd``
resolves to that directory."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "zero-chain_amd"))
from ._api import *  # noqa: F401,F403,E402
from ._api import __all__  # noqa: E402
