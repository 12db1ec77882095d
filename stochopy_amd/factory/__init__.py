from .benchmark import *  # noqa: F401,F403
from .benchmark import Objective, __all__  # noqa: F401
