from . import config  # noqa: F401  (the reference imports config and train eagerly: lib/lstm/__init__.py:8-9)
from . import train  # noqa: F401
