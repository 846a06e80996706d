"""`models.cbam` -- same import path as /root/reference/models/cbam.py; HIP-backed CBAM."""
from mdie_amd.modules import CBAM  # noqa: F401
