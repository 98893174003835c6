from q1physrl_amd.phys import *  # noqa: F401,F403
from q1physrl_amd.phys import Inputs, PlayerState, apply  # noqa: F401
