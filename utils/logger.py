"""`utils.logger.ExperimentLogger` (utils/logger.py:33) -> mdie_amd.host.RunLogger."""
from mdie_amd.host import RunLogger as ExperimentLogger  # noqa: F401
