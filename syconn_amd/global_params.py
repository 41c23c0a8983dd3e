"""Working-directory switch + config singleton, mirroring ``syconn.global_params``
(/root/reference/syconn/global_params.py:19-20: ``wd = None``; ``config = DynConfig()``)."""
from .handler.config import DynConfig

wd = None
config = DynConfig()
