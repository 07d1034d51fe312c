"""Model factories, resolved like the reference does: eval('models.' + cfg.MODEL.NAME + '.get_pose_net')
(/root/reference/lib/models/__init__.py:15-17, tools/fpd_train.py:122)."""
from . import hourglass, pose_hrnet  # noqa: F401
