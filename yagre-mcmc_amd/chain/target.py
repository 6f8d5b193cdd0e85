"""yagremcmc.chain.target -> the posterior density descriptor."""
from ..plugins import UnnormalisedPosterior  # noqa: F401
