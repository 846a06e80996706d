"""`models.cdan` -- the name the reference's configs resolve (`["models.cdan", "CDAN"]`,
/root/reference/config/low_light.json:12 via /root/reference/utils/parser.py:42-73).
Here it is the MI355X engine's front end; see INTEGRATION.md."""
from mdie_amd.modules import CDAN  # noqa: F401
