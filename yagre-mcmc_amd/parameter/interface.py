from ..interfaces import ParameterInterface  # noqa: F401
