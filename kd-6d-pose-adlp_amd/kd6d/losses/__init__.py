"""Loss-level drop-in surface of the reference's `losses/` package on the kd6d C ABI.

    from kd6d.losses.kd_loss import KDPoseLoss, SamplesLoss        # losses/kd_loss.py:13-161 (+ geomloss.SamplesLoss)
    from kd6d.losses.loss_libs import kd_loss_2d                   # losses/loss_libs.py:1-51

Same constructor / call signatures and return values as the reference; every value and gradient comes from the HIP
kernels (kd6d_ssc_assign, kd6d_focal_*, kd6d_student_points, kd6d_sinkhorn_div_fwd_bwd / _dense_fwd_bwd,
kd6d_loss_backward) behind torch.autograd.Function edges, so they can be dropped into a torch training loop whose
network is NOT the kd6d engine (e.g. the reference's own PoseModule).  PoseModuleKD does not go through this package:
it calls the same kernels on its packed buffers without the NCHW <-> packed-NHWC copies made here."""
from .kd_loss import KDPoseLoss, SamplesLoss  # noqa: F401
from .loss_libs import kd_loss_2d  # noqa: F401
