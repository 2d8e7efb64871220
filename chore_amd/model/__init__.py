# mirrors /root/reference/model/__init__.py:1 (`from model import CHORE`)
from .chore import CHORE  # noqa: F401
