"""Import alias: ``import nerfca_amd`` loads the package that lives in ``nerf-ca_amd/``
(a directory name with a hyphen cannot be imported directly)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "nerf-ca_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
