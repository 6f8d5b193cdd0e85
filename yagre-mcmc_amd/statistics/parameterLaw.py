from ..interfaces import AbsolutelyContinuousParameterLaw  # noqa: F401
