from ..interfaces import WelfordAccumulator  # noqa: F401
