from ..ensemble import EnsembleChain  # noqa: F401
from ..interfaces import Chain  # noqa: F401
