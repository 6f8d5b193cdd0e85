"""yagremcmc.chain.method.pcn -> preconditioned Crank-Nicolson ensemble sampler and its builder."""
from ...builders import PCNBuilder  # noqa: F401
from ...ensemble import PreconditionedCrankNicolson  # noqa: F401
