"""Import shim: makes the package directory `generative-recommenders_amd/` (whose name is not a
valid Python identifier) importable as `generative_recommenders_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "generative-recommenders_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
