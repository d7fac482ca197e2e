from .intensity import GaussianSmooth, GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd  # noqa: F401
from .spatial import Resample, SpatialResample, Spacing, SpacingD, SpacingDict, Spacingd, spatial_resample  # noqa: F401
from .post import Activations, ActivationsD, ActivationsDict, Activationsd, AsDiscrete, AsDiscreteD, AsDiscreteDict, AsDiscreted  # noqa: F401
