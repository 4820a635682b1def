"""The reference registry imports `AR` from `.ar` (vall_e/vall_e/__init__.py:2); the complete
D3PM model lives in ar_discrete.py, so this module re-exports it."""
from .ar_discrete import AR  # noqa: F401
