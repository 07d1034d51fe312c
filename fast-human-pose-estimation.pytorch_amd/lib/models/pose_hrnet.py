"""Placeholder for the reference's `models.pose_hrnet` (/root/reference/lib/models/pose_hrnet.py:495-501).

The MI355X HRNet path (BASELINE.json configs 3-5) is not built yet: it needs stride-2 convolutions with their data
gradients, the post-activation BasicBlock ops and conv tiles for widths that are not powers of two.
Its CPU oracle and golden vectors already exist (oracle/hrnet_ref.py, tests/golden/hrnet_tiny.npz).  There is no CPU
fallback in this package, so the factory fails loudly instead of returning a torch model."""
from ...runtime import FpdError


def get_pose_net(cfg, is_train, **kwargs):
    raise FpdError('models.pose_hrnet: the HRNet FPD path is not implemented on the MI355X back end yet '
                   '(MODEL.NAME hourglass is; see DESIGN.md section 8)')
