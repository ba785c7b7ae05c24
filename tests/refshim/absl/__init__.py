"""NOT absl: see tests/refshim/chex/__init__.py."""
