"""Importable alias of the package directory `yagre-mcmc_amd/` (whose name is
not a Python identifier): `import yagremcmc_amd` resolves every submodule
there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yagre-mcmc_amd"))
_src = _os.path.join(__path__[0], "__init__.py")
with open(_src) as _f:
    exec(compile(_f.read(), _src, "exec"))
del _os, _src, _f
