"""yagremcmc.chain.method.aem -> adaptive-error-model ensemble sampler and its builder."""
from ...builders import AEMBuilder  # noqa: F401
from ...ensemble import AdaptiveErrorModel  # noqa: F401
