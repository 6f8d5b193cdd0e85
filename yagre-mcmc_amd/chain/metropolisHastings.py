"""yagremcmc.chain.metropolisHastings -> the ensemble template of the samplers."""
from ..ensemble import MetropolisHastingsEnsemble  # noqa: F401
from ..ensemble import MetropolisHastingsEnsemble as MetropolisHastings  # noqa: F401
