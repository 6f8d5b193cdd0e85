"""yagremcmc.chain.method.mlda -> two-level delayed-acceptance ensemble sampler and its builder."""
from ...builders import MLDABuilder  # noqa: F401
from ...ensemble import MLDA  # noqa: F401
