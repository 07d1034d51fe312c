from .device_pipeline import DevicePipeline  # noqa: F401
from .synthetic import SyntheticPose  # noqa: F401
