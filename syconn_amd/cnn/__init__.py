"""Architectures of the dense 3D U-Nets SyConn trains (/root/reference/syconn/cnn/cnn_*.py); inference-side only."""
from .unet_spec import ARCHS, random_state_dict, unet_param_shapes  # noqa: F401
