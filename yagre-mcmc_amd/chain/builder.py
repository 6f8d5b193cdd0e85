"""yagremcmc.chain.builder -> the ensemble builder base."""
from ..builders import ChainBuilder  # noqa: F401
