#!/usr/bin/env python3
"""End-to-end check of the env through learning: PPO on the GPU-resident sampler (one process per GPU; with
torch.distributed.run the policy gradient is all-reduced, the envs are sharded).  Prints one line per iteration with the
reference's headline training metric, zero_start_total_reward_mean (train.py:54-57; README: ~5700 after 150 M steps).

    python tools/train_ppo.py --iters 150 --envs 8192 --horizon 128
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# The learner's float32 GEMMs are tall-skinny (weight gradients reduce over a 32 768-row minibatch); hipBLASLt's default
# heuristic picks poor tiles for them.  PyTorch's TunableOp times the candidate rocBLAS / hipBLASLt solutions once per shape
# (during the learner's warm-up steps) - measured 0.67 -> 0.47 s per iteration.  Q1_TUNABLEOP=0 turns it off.
if os.environ.get("Q1_TUNABLEOP", "1") != "0":
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/q1physrl_tunableop_%d.csv")
import torch
import torch.distributed as dist
from q1physrl_amd import policy as P, ppo, sharding
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--envs", type=int, default=8192, help="total envs over all ranks")
ap.add_argument("--horizon", type=int, default=128)
ap.add_argument("--lr", type=float, default=3e-4)
ap.add_argument("--epochs", type=int, default=4)
ap.add_argument("--minibatch", type=int, default=65536)
ap.add_argument("--entropy", type=float, default=0.003)
ap.add_argument("--kl-target", type=float, default=0.01)
ap.add_argument("--zero-start-prob", type=float, default=0.1)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--time-limit", type=float, default=0.0, help="Config.time_limit in seconds (0 = the Config's own: 10 s = 720-tick episodes)")
ap.add_argument("--out", default="")
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--fused-policy", action="store_true", help="sample with the fused MFMA forward kernel (bf16 hidden layer)")
ap.add_argument("--resident", action="store_true", help="sample every horizon as ONE dispatch (q1env_sample_resident) + one batched value forward; needs --fused-policy")
ap.add_argument("--learner-bf16", action="store_true", help="learner GEMMs under torch.autocast(bfloat16) (float32 master weights, loss and Adam)")
ap.add_argument("--no-fused-adam", action="store_true", help="plain torch.optim.Adam instead of the multi-tensor fused one (0.48 instead of 0.41 s per iteration)")
ap.add_argument("--fused-loss", action="store_true", help="PPO loss + gradient from the q1env_ppo_loss_grad kernel")
ap.add_argument("--native", action="store_true", help="native learner step: gather, both MLPs forward + backward, loss gradient and weight gradients as the library's gfx950 kernels (q1env_learner_step); torch runs only Adam")
ap.add_argument("--native-splits", type=int, default=32, help="workgroups per network of the split-K weight-gradient kernel")
ap.add_argument("--save", default="", help="write the final policy weights (npz, RLlib fcnet naming) here")
ap.add_argument("--discrete-yaw-steps", type=int, default=-1, help="Config.discrete_yaw_steps: the mouse becomes Discrete(2S+1) (a Categorical policy head)")
ap.add_argument("--refcfg", action="store_true",
                help="the REFERENCE's training configuration (VERDICT r3 item 5): data/params.yml trainer_config (lr 5e-6, train_batch_size 50 000, "
                     "kl_target 0.0036, entropy 0.01, gamma 0.99, lambda 0.95, vf_clip 100) and env_config (params.yml:16-33), RLlib 0.8.4 PPO defaults "
                     "for the rest (sgd_minibatch_size 128, num_sgd_iter 30, clip 0.3, kl_coeff 0.2), 4 workers x 100 envs = 400 envs x 125 ticks "
                     "per iteration; overrides --envs / --horizon / --lr / --epochs / --minibatch / --entropy / --kl-target / --zero-start-prob")
ap.add_argument("--no-persistent", action="store_true", help="drive q1env_learner_sgd_step per minibatch instead of ONE q1env_learner_sgd_epochs dispatch per update (A/B)")
ap.add_argument("--learner-fp32", action="store_true", help="the persistent learner in float32 arithmetic (q1env_learner_sgd_epochs_f32: float32 operands, no loss scale, "
                                                           "nothing saturates - RLlib's own arithmetic; 27 instead of 11.8 us per step): the control of the float16 learner "
                                                           "(STATE.md: same seed spread over 8 / 9 seeds)")
ap.add_argument("--dynamic-loss-scale", action="store_true", help="choose the native learner's float16 loss scales per update from the previous update's largest gradient element instead of the static (256, 1): no saturation, but measured to cost the large-minibatch configuration its result (PPOLearner docstring)")
ap.add_argument("--checkpoint-dir", default="", help="trainer checkpoints (policy weights, optimizer state incl. the native Adam moments + step count, adaptive KL "
                                                      "coefficient, iteration, best metric): every --checkpoint-every iterations and whenever "
                                                      "zero_start_total_reward_mean exceeds its previous best - the reference's schedule (q1physrl/train.py:110-133)")
ap.add_argument("--checkpoint-every", type=int, default=100)
ap.add_argument("--restore", default="", help="resume from a checkpoint written by --checkpoint-dir (the reference's params['checkpoint_fname'], train.py:110-111)")
ap.add_argument("--step-mode", default="auto", choices=["auto", "four_launch", "fused", "fused_dw1"],
                help="kernel sequence of q1env_learner_sgd_step (include/q1env.h q1env_learner_set_step_mode): auto = the fused forward + backward kernel with per-tile "
                     "dW1 / dW3 products from 2 048 samples on; four_launch = round 4's; fused = the fused kernel, bit-identical to four_launch")
ap.add_argument("--log-every", type=int, default=5)
ap.add_argument("--out-stride", type=int, default=1, help="--out keeps every Nth iteration's row (+ every row with an evaluation or a checkpoint, + the last): a 2 989-iteration log is 1.4 MB at stride 1")
ap.add_argument("--eval-every", type=int, default=0, help="every N iterations: 256 zero-start episodes of 720 ticks on a separate env, stochastic (the "
                                                         "training metric's policy) and deterministic; 0 = only at the end")
args = ap.parse_args()
if args.refcfg:
    args.envs, args.horizon, args.lr, args.epochs, args.minibatch = 400, 125, 5e-6, 30, 128
    args.entropy, args.kl_target, args.zero_start_prob = 0.01, 0.0036, 0.01
    args.native_splits = min(args.native_splits, 4)            # a 128-sample minibatch is 4 tiles: more split-K workgroups would only add partial sums

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("nccl" if torch.cuda.device_count() >= world else "gloo")
torch.manual_seed(args.seed)                                   # identical initial weights on every rank
start, count = sharding.shard_range(args.envs, rank, world)
import bench
if args.refcfg:
    cfg = Config(num_envs=count, **bench.PARAMS_YML)           # data/params.yml:16-33
else:
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": count, "zero_start_prob": args.zero_start_prob,
                    "discrete_yaw_steps": args.discrete_yaw_steps, **({"time_limit": args.time_limit} if args.time_limit > 0 else {})})
# a resumed run does not replay the first run's resets and action noise: its env / sampler seed is offset by the iteration it resumes at
ck = torch.load(args.restore, map_location="cuda", weights_only=False) if args.restore else None
env = TensorVectorEnv(cfg, device=local, seed=args.seed + 1 + (1000003 * (int(ck["iter"]) + 1) if ck is not None else 0), env_index_base=start)
pol = P.Q1Policy(discrete_yaw_steps=args.discrete_yaw_steps).cuda()
fused = P.FusedPolicyForward(pol, env) if args.fused_policy else None
smp = GpuSampler(env, fused if fused is not None else pol, horizon=args.horizon, use_graph=not args.no_graph, resident=args.resident)
env._dev.learner_set_step_mode(args.step_mode)
lrn = ppo.PPOLearner(pol, float(cfg.action_range), lr=args.lr, num_sgd_iter=args.epochs, minibatch_size=args.minibatch,
                     entropy_coeff=args.entropy, kl_target=args.kl_target, seed=args.seed + rank, use_graph=not args.no_graph, fused_loss=args.fused_loss, env=env,
                     discrete_yaw_steps=args.discrete_yaw_steps, autocast_dtype=torch.bfloat16 if args.learner_bf16 else None, fused_adam=not args.no_fused_adam,
                     native=args.native, native_splits=args.native_splits, persistent=False if args.no_persistent else None,
                     dynamic_loss_scale=args.dynamic_loss_scale, precision="f32" if args.learner_fp32 else "f16")
log = []
start_iter, best_metric, best_file = 0, float("-inf"), None


def save_checkpoint(it, tag):
    """One file per checkpoint (torch.save): the policy, the learner (optimizer moments + step count, adaptive KL coefficient, float16 loss scales,
    the minibatch-permutation generator and the permutations already drawn ahead), torch's generator, the iteration and the best metric: a
    resumed run continues the LEARNER exactly.  The environment is not checkpointed: a resumed run's sampler starts from a fresh reset with
    a seed offset by the iteration it resumes at (env state is not part of the reference's RLlib checkpoints either, train.py:110-133), so
    trajectories after a resume differ from the uninterrupted run's.  Rank 0 writes; each rank's learner generator differs by its rank (seed +
    rank): the file holds rank 0's, the other ranks re-derive theirs at resume (`lrn.seed`)."""
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    fn = os.path.join(args.checkpoint_dir, f"checkpoint_{tag}.pt")
    torch.save({"iter": it, "policy": pol.state_dict(), "learner": lrn.state_dict(), "best_metric": best_metric, "best_file": best_file,
                "sampler_stats": smp.stats, "torch_rng": torch.get_rng_state(), "torch_cuda_rng": torch.cuda.get_rng_state(), "rank": rank,
                "args": vars(args)}, fn + ".tmp")
    os.replace(fn + ".tmp", fn)
    return fn


if args.restore:
    pol.load_state_dict(ck["policy"])
    lsd = dict(ck["learner"])
    if ck.get("learner_gen") is not None and lsd.get("gen") is None:       # (checkpoints written before round 6)
        lsd["gen"] = ck["learner_gen"]
    if rank != int(ck.get("rank", 0)):                                      # another rank's permutation stream: re-derived from this rank's seed, not collapsed onto rank 0's
        lsd["gen"], lsd["perms_next"] = None, None
    lrn.load_state_dict(lsd)
    if ck.get("torch_rng") is not None:
        torch.set_rng_state(ck["torch_rng"].cpu())
    if ck.get("torch_cuda_rng") is not None:
        torch.cuda.set_rng_state(ck["torch_cuda_rng"].cpu())
    start_iter, best_metric, best_file = int(ck["iter"]) + 1, float(ck["best_metric"]), ck.get("best_file")
    if fused is not None:
        fused.refresh()
    if rank == 0:
        print(json.dumps({"restored": args.restore, "resume_at_iter": start_iter, "best_metric": best_metric}), flush=True)


def zero_start_eval(n_eval=256):
    """Mean y distance (sum of the 720 rewards) of n_eval zero-start episodes under the current policy on a fresh env: stochastic
    (the policy the reference's zero_start_total_reward_mean averages over, train.py:54-57) and deterministic (what mkdemo plays back)."""
    base = bench.PARAMS_YML if args.refcfg else {**Config.get_default().__dict__, "discrete_yaw_steps": args.discrete_yaw_steps}
    ec = Config(**{**{k: v for k, v in base.items() if k != "num_envs"}, "num_envs": n_eval, "zero_start_prob": 1.0})
    out = {}
    for det in (False, True):
        ee = TensorVectorEnv(ec, device=local, seed=4242 + len(log))
        tr_ = GpuSampler(ee, pol, horizon=720).collect(deterministic=det)
        out["eval_det" if det else "eval_stochastic"] = float(tr_["reward"].double().sum(0).mean())
        ee.close()
    return out


t0 = time.time()
prev = smp.stats
for it in range(start_iter, args.iters):
    ts = time.time()
    traj = smp.collect()
    adv, vtarg = smp.advantages(traj, lrn.gamma, lrn.lam)
    torch.cuda.synchronize(); t_sample = time.time() - ts
    st = lrn.update(traj, adv, vtarg)
    if fused is not None:
        fused.refresh()
    torch.cuda.synchronize(); t_iter = time.time() - ts
    cur = smp.stats
    dz = cur["zero_start_episodes"] - prev["zero_start_episodes"]
    zmean = (cur["zero_start_return_sum"] - prev["zero_start_return_sum"]) / dz if dz else float("nan")
    de = cur["episodes"] - prev["episodes"]
    emean = (cur["return_sum"] - prev["return_sum"]) / de if de else float("nan")
    prev = cur
    row = {"iter": it, "steps": (it + 1) * args.envs * args.horizon, "zero_start_total_reward_mean": zmean, "episode_reward_mean": emean,
           "kl": st["kl"], "entropy": st["entropy"], "vf_loss": st["vf_loss"], "kl_coeff": st["kl_coeff"],
           "sample_s": t_sample, "iter_s": t_iter, "wall_s": time.time() - t0}
    row.update({k: st[k] for k in ("grad_saturated_pi", "grad_saturated_vf", "grad_max_abs_pi", "grad_max_abs_vf", "pi_upscale", "value_downscale") if k in st})
    if args.eval_every and (it % args.eval_every == 0 or it == args.iters - 1):
        row.update(zero_start_eval())
    if args.checkpoint_dir and rank == 0:
        improved = zmean == zmean and zmean > best_metric                 # (NaN: no zero-start episode finished in this iteration)
        if improved:
            best_metric = zmean
            best_file = save_checkpoint(it, "best")
            row["checkpoint_best"] = best_file
        if it % args.checkpoint_every == 0 or it == args.iters - 1:
            row["checkpoint"] = save_checkpoint(it, f"{it:06d}")
    log.append(row)
    if rank == 0 and (it % args.log_every == 0 or it == args.iters - 1 or "eval_det" in row):
        print(json.dumps(row), flush=True)

# deterministic evaluation: zero-start, 720 ticks, argmax keys / squashed mean (what mkdemo would play back)
ecfg = Config(**{**Config.get_default().__dict__, "num_envs": 64, "zero_start_prob": 1.0, "discrete_yaw_steps": args.discrete_yaw_steps})
eenv = TensorVectorEnv(ecfg, device=local, seed=123)
es = GpuSampler(eenv, pol, horizon=720)
tr = es.collect(deterministic=True)
dist_y = tr["reward"].double().sum(0)
if rank == 0:
    final = {"eval_zero_start_distance_mean": float(dist_y.mean()), "eval_min": float(dist_y.min()), "eval_max": float(dist_y.max()),
             "total_steps": args.iters * args.envs * args.horizon, "wall_s": time.time() - t0}
    print(json.dumps(final), flush=True)
    if args.out:
        kept = [r for r in log if r["iter"] % max(1, args.out_stride) == 0 or r["iter"] == args.iters - 1 or "eval_det" in r or "checkpoint" in r]
        json.dump({"args": vars(args), "log": kept, "final": final}, open(args.out, "w"))
    if args.save:
        import numpy as np
        names = [(pol.pi[0], "fc_1"), (pol.pi[2], "fc_2"), (pol.pi[4], "fc_out"), (pol.vf[0], "fc_value_1"), (pol.vf[2], "fc_value_2"), (pol.vf[4], "value_out")]
        np.savez_compressed(args.save, **{f"{n}.kernel": l.weight.detach().cpu().numpy().T for l, n in names},
                            **{f"{n}.bias": l.bias.detach().cpu().numpy() for l, n in names})
