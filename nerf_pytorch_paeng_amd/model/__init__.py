from .NeRF import NeRF, NeRFModule  # noqa: F401
from .PositionalEncoding import PositionalEncoding, get_positional_encoder  # noqa: F401
