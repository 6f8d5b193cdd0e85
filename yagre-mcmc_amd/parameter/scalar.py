from ..plugins import ScalarParameter  # noqa: F401
