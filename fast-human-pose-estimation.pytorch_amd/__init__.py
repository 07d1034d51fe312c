"""MI355X-native Fast-Pose-Distillation training path (import as `fpd_amd`).

Sub-modules: runtime (ctypes binding of csrc/libfpd_amd.so), graph (host-side op graph),
executor (arenas + plan lowering), lib.models.hourglass / lib.core.{loss,function} (the reference's
module API), dist (data-parallel replicas over RCCL)."""
__version__ = '0.1.0'
