#!/usr/bin/env python3
"""Export the WEIGHTS of the reference's published world-record policy (data file
/root/reference/data/checkpoints/wr/checkpoint, an RLlib 0.8.4 pickle) into tests/golden/wr_policy.npz.
TEST INFRASTRUCTURE, run once in the build container (the reference does not travel to the GPU box).

The pickle references ray classes that are not installed; a restricted Unpickler maps every non-NumPy class to an inert
stub, which is enough because the policy state is a plain dict of NumPy arrays
('default_policy/fc_1/kernel' (6,256) ... 'default_policy/value_out/bias' (1,)) and the observation filter is NoFilter.
The fixture holds data only: 12 float32 arrays (137 995 numbers) + the env_config the run used (params.json)."""
import io
import json
import os
import pickle
import sys

import numpy as np

sys.dont_write_bytecode = True
SRC = "/root/reference/data/checkpoints/wr"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wr_policy.npz")


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, s):
        self.state = s


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("numpy", "builtins", "collections", "_codecs"):
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


def main():
    top = _Unpickler(open(os.path.join(SRC, "checkpoint"), "rb")).load()
    worker = _Unpickler(io.BytesIO(top["worker"])).load()
    assert type(worker["filters"]["default_policy"]).__name__ == "NoFilter"
    weights = worker["state"]["default_policy"]
    out = {k.replace("default_policy/", "").replace("/", "."): np.asarray(v, dtype=np.float32) for k, v in weights.items()}
    assert sum(v.size for v in out.values()) == 137995
    params = json.load(open(os.path.join(SRC, "params.json")))
    out["env_config_json"] = np.array(json.dumps(params["env_config"]))
    meta = _Unpickler(open(os.path.join(SRC, "checkpoint.tune_metadata"), "rb")).load()
    out["tune_metadata_json"] = np.array(json.dumps({k: v for k, v in meta.items() if isinstance(v, (int, float, str))}))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes;", {k: v.shape for k, v in out.items() if v.ndim})
    print(out["tune_metadata_json"])


if __name__ == "__main__":
    main()
