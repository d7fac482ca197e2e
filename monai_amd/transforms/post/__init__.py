from .array import Activations, AsDiscrete  # noqa: F401
from .dictionary import ActivationsD, ActivationsDict, Activationsd, AsDiscreteD, AsDiscreteDict, AsDiscreted  # noqa: F401
