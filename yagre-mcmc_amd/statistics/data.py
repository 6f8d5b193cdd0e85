from ..plugins import Data  # noqa: F401
