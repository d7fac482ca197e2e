from .meta_tensor import MetaTensor  # noqa: F401
from .utils import (  # noqa: F401
    affine_to_spacing,
    compute_importance_map,
    compute_shape_offset,
    dense_patch_slices,
    get_valid_patch_size,
    to_affine_nd,
    window_starts,
    zoom_affine,
)
