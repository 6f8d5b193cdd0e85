from ..interfaces import TransitionData  # noqa: F401
