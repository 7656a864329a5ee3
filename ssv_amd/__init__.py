"""Import alias: the package directory is ``self-supervised-vision_amd/`` (not a valid Python
identifier), so ``import ssv_amd`` maps onto it.  All code lives there."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "self-supervised-vision_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
