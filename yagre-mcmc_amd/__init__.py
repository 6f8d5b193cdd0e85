"""yagremcmc on MI355X: ensembles of Metropolis-Hastings chains as HIP kernels.

Import it as `yagremcmc_amd` (the directory name `yagre-mcmc_amd` is not a
Python identifier; the sibling package `yagremcmc_amd/` aliases it).  The
module layout mirrors the reference's (`chain.method.mrw`, `statistics.*`,
`parameter.*`, `model.*`, `test.testSetup`), so an example script switches by
replacing `yagremcmc` with `yagremcmc_amd` in its imports.
"""
from . import _ffi
from ._ffi import YmcError, device_count
from .builders import AEMBuilder, ChainBuilder, MLDABuilder, MRWBuilder, PCNBuilder
from .diagnostics import AcceptanceRateDiagnostics, DummyDiagnostics, FullDiagnostics
from .ensemble import MLDA, AdaptiveErrorModel, MetropolisedRandomWalk, PreconditionedCrankNicolson
from .lowering import UnsupportedPlugin

__all__ = ["ChainBuilder", "MRWBuilder", "MLDABuilder", "PCNBuilder", "AEMBuilder", "AdaptiveErrorModel",
           "MetropolisedRandomWalk", "MLDA",
           "PreconditionedCrankNicolson",
           "AcceptanceRateDiagnostics", "FullDiagnostics", "DummyDiagnostics", "UnsupportedPlugin",
           "YmcError", "device_count"]
