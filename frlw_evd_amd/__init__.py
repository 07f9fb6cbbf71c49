"""Import alias for the hyphenated package directory ``frlw-evd_amd/``.

``frlw-evd_amd`` is not a valid Python identifier, so this thin package only
redirects its ``__path__`` to that directory: ``import frlw_evd_amd.encoders``
loads ``frlw-evd_amd/encoders.py``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "frlw-evd_amd")
__path__.insert(0, _real)

with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
