"""`from models import hmr` (lib/core/base.py:23): SPIN's constructor name, MI355X arithmetic."""
from poserisk_release_amd.hmr import HMR, hmr  # noqa: F401
