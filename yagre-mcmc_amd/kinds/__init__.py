"""Extension kinds: forward models beyond the reference's own two ("Adding a forward model").
Importing a module here registers its solver class with `lowering.register_solver`."""
from . import logistic  # noqa: F401
