"""`projects.mmdet3d_plugin` — the import target the reference's configs name
(`plugin_dir = 'projects/mmdet3d_plugin/'`, imported by tools/train.py:114-135).  Importing it
registers the MI355X-backed modules under the reference's registry names."""
from occnet_amd.plugin import *  # noqa: F401,F403
