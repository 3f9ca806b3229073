from .act import Activation1d  # noqa: F401
from .filter import LowPassFilter1d, kaiser_sinc_filter1d  # noqa: F401
from .resample import DownSample1d, UpSample1d  # noqa: F401
