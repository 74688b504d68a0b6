"""MI355X-native batched ``Game.step()`` for the continuous_grid_arctic follow-the-leader env."""
from . import abi  # noqa: F401
from .config import make_config, GameConfig  # noqa: F401
