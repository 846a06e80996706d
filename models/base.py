"""`models.base.BaseModel`: the harness base class name of the reference (models/base.py:11)."""
from mdie_amd.host import Model as BaseModel  # noqa: F401
