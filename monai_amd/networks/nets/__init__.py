from .basic_unet import BasicUNet, BasicUnet, Basicunet, basicunet  # noqa: F401
