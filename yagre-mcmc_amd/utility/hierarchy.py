from ..plugins import Hierarchy, HierarchyBase, SharedComponent  # noqa: F401
