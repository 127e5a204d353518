"""Seeded synthetic scenes for the parity tests: the fixtures live in the top-level `synthetic_scenes` module (bench.py and
smoke() build the same inputs from it without importing the test package)."""
from synthetic_scenes import *  # noqa: F401,F403
from synthetic_scenes import _cell_centres, _make_shelf_scene, _xavier  # noqa: F401
