"""Launch the pure-copy calibration kernel (exactly 85 B read + 85 B written per env, step_kernel's own access
pattern) so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count (MI355X_MICROARCH.md, HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from q1physrl_amd.device import DeviceEnv
from q1physrl_amd.env import Config
for n in (65536, 4194304):
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n})
    d = DeviceEnv(cfg)
    d.calibrate_traffic(20)
    d.close()
