"""Soak of the tick server's hand-rolled store protocol under the ASSERTION build of the library (libq1env_check.so = the same sources
compiled with -DQ1_CHECK, `python -m q1physrl_amd.build --check`): every inline-assembly 16-byte sc1 granule-pair store is read back
on the device and compared with the registers it was issued from (q1server.hpp, granule_pair_store); q1env_debug_counters reports how
many stores were checked and how many differed.  Runs the two-stream form (granules on EVERY tick) and the LDS pair (granules on the
last tick of a launch) and also compares the results with the per-tick kernels, as tools/soak_pair.py does.

    python tools/soak_check.py [--envs 4096 65536] [--launches 4] [--ticks 360]

The check library is selected through Q1ENV_LIB_PATH before q1physrl_amd is imported; the product path never loads it.
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHECK_SO = os.path.join(ROOT, "q1physrl_amd", "libq1env_check.so")


def run(envs=(4096, 65536), launches=4, ticks=360, verbose=True):
    """Returns {"stores": .., "mismatches": .., "env_steps": ..}; raises if the library is not the assertion build."""
    import numpy as np, torch
    from q1physrl_amd.env import Config
    from q1physrl_amd.tensor_env import TensorVectorEnv
    total = {"stores": 0, "mismatches": 0, "env_steps": 0}
    for n in envs:
        cfgd = dict(Config.get_default().__dict__, num_envs=n, time_limit=2.0, zero_start_prob=0.3)
        for two in (True, False):
            a = TensorVectorEnv(Config(**cfgd), device=0, seed=3); b = TensorVectorEnv(Config(**cfgd), device=0, seed=3)
            built, _, _ = a._dev.debug_counters(clear=True)
            if not built:
                raise RuntimeError("soak_check: the loaded library was not built with -DQ1_CHECK (set Q1ENV_LIB_PATH to libq1env_check.so)")
            a.reset(); b.reset()
            g = torch.Generator(device="cuda").manual_seed(8)
            keys = torch.randint(0, 16, (ticks, n), dtype=torch.uint8, device="cuda", generator=g)
            mouse = ((torch.rand((ticks, n), device="cuda", generator=g) * 2 - 1) * 10.0).contiguous()
            t0 = time.time()
            for l in range(launches):
                for t in range(ticks):
                    obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
                res = a.serve_ticks(keys, mouse, two_streams=two)
                assert not res["status"].any(), (n, two, l, res["status"])
                assert torch.equal(res["obs"], obs_b) and torch.equal(res["obs_from_granules"], obs_b) and torch.equal(res["reward"], rew_b) \
                    and torch.equal(res["done"], done_b), (n, two, l)
            sa, sb = a.get_state(), b.get_state()
            for k in sa:
                assert np.array_equal(sa[k], sb[k]), (n, two, k)
            _, stores, bad = a._dev.debug_counters(clear=True)
            want = n * 4 * launches * (ticks if two else 1)          # four pairs per env per tick that leaves as granules
            assert stores == want, (n, two, stores, want)
            total["stores"] += stores; total["mismatches"] += bad; total["env_steps"] += n * launches * ticks
            if verbose:
                print(f"soak_check: {n} envs, {'two streams (granules every tick)' if two else 'LDS pair (granules on the last tick)'}: "
                      f"{launches} x {ticks} ticks, {stores} granule-pair stores read back, {bad} mismatches, results identical to the per-tick "
                      f"kernels ({time.time() - t0:.1f} s)", flush=True)
            a.close(); b.close()
    return total


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[4096, 65536])
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--ticks", type=int, default=360)
    args = ap.parse_args()
    if os.environ.get("Q1ENV_LIB_PATH") != CHECK_SO:          # re-exec with the assertion build selected before the binding loads
        if not os.path.exists(CHECK_SO):
            from q1physrl_amd import build
            build.build_lib(check=True)
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, Q1ENV_LIB_PATH=CHECK_SO))
    tot = run(args.envs, args.launches, args.ticks)
    print(f"soak_check total: {tot['env_steps'] / 1e9:.3f} G env-steps, {tot['stores']} stores checked, {tot['mismatches']} mismatches")
    sys.exit(1 if tot["mismatches"] else 0)
