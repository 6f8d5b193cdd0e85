"""yagremcmc.chain.method.mrw -> random-walk ensemble sampler and its builder."""
from ...builders import MRWBuilder  # noqa: F401
from ...ensemble import MetropolisedRandomWalk  # noqa: F401
