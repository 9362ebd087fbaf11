"""Same import path as the reference's dafne/utils/sort_corners.py."""
from ..postprocess import sort_quadrilateral  # noqa: F401
