from .basic_unet import BasicUNet, BasicUnet, Basicunet, basicunet  # noqa: F401
from .unetr import UNETR  # noqa: F401
from .unet import UNet, Unet  # noqa: F401
from .dynunet import DynUNet, DynUnet, Dynunet  # noqa: F401
from .segresnet import SegResNet  # noqa: F401
from .swin_unetr import SwinUNETR  # noqa: F401
