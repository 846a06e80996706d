"""`models.model.Model` (config key model.which_model, config/low_light.json:6-9)."""
from mdie_amd.host import Model  # noqa: F401
