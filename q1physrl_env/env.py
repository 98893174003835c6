from q1physrl_amd.env import *  # noqa: F401,F403
from q1physrl_amd.env import (ActionDecoder, Config, INITIAL_YAW_ZERO, Key, Obs, PhysEnv,  # noqa: F401
                              VectorPhysEnv, get_obs_scale, _LazyInfos)
