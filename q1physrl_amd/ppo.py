"""PPO learner for the GPU-resident sampler ("next" row 3 of SURVEY.md section 8f): the counterpart of the RLlib
PPOTrainer the reference configures in q1physrl/train.py:60-64 with data/params.yml:4-13 (gamma 0.99, lambda 0.95,
kl_target 0.0036, entropy_coeff 0.01, vf_clip_param 100; RLlib 0.8.4 PPO defaults: clip_param 0.3, kl_coeff 0.2
adaptive, vf_loss_coeff 1.0, advantages standardised per train batch).

Loss (RLlib ppo_tf_policy.PPOLoss restated for torch):
    ratio      = exp(logp_new - logp_old)
    surrogate  = min(adv * ratio, adv * clip(ratio, 1 - clip, 1 + clip))
    vf_loss    = max((v - vtarg)^2, (v_old + clip(v - v_old, +-vf_clip) - vtarg)^2)
    total      = mean(-surrogate + kl_coeff * KL(old || new) + vf_loss_coeff * vf_loss - entropy_coeff * entropy)
Multi-GPU: every rank samples its own env shard; the only collective of the whole system is the gradient
all-reduce of the 138 k-parameter policy (one 552 KB flat bucket per SGD step; RCCL when the backend is nccl).
"""
import os
import torch
import torch.distributed as dist

from .policy import Q1PhysActionDist


def ppo_loss(policy, batch, action_range, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff, num_keys=4,
             discrete_yaw_steps=-1, allow_yaw=True, autocast_dtype=None):
    """autocast_dtype: reduced-precision operands for the two MLPs' matrix products only; the distribution / loss arithmetic below
    always runs in float32 (the same split as the fused-loss path of PPOLearner._sgd_step)."""
    if autocast_dtype is not None and batch["obs"].is_cuda:
        with torch.autocast("cuda", dtype=autocast_dtype):
            logits, value = policy(batch["obs"])
        logits, value = logits.float(), value.float()
    else:
        logits, value = policy(batch["obs"])
    new = Q1PhysActionDist(logits, action_range, num_keys, discrete_yaw_steps, allow_yaw)
    old = Q1PhysActionDist(batch["old_logits"], action_range, num_keys, discrete_yaw_steps, allow_yaw)
    logp = new.logp(batch["keys"], batch["mouse"])
    ratio = torch.exp(logp - batch["logp"])
    adv = batch["adv"]
    surrogate = torch.minimum(adv * ratio, adv * torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param))
    v_clipped = batch["value"] + torch.clamp(value - batch["value"], -vf_clip_param, vf_clip_param)
    vf_loss = torch.maximum((value - batch["vtarg"]) ** 2, (v_clipped - batch["vtarg"]) ** 2)
    kl = old.kl(new)
    entropy = new.entropy()
    total = torch.mean(-surrogate + kl_coeff * kl + vf_loss_coeff * vf_loss - entropy_coeff * entropy)
    stats = {"total_loss": total.detach(), "policy_loss": -surrogate.mean().detach(), "vf_loss": vf_loss.mean().detach(),
             "kl": kl.mean().detach(), "entropy": entropy.mean().detach()}
    return total, stats


STAT_KEYS = ("entropy", "kl", "policy_loss", "total_loss", "vf_loss")      # sorted: the order of the stats vector


def allreduce_grads_(params, world):
    """Average gradients over ranks with ONE collective on a flat bucket."""
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class NativeStep:
    """One PPO SGD step's forward + loss gradient + backward + weight gradients as the library's own gfx950 kernels
    (q1env_learner_step: float16 matrix-core operands, float32 accumulation; include/q1env.h) on the float32 master weights of a
    Q1Policy: after step() every parameter's .grad holds d mean-loss / d parameter (the tensors are allocated once and overwritten - do
    not zero_grad(set_to_none=True)); the caller runs the optimizer and then images() to rebuild the kernels' float16 weight images.
    The minibatch is rows idx of the FULL trajectory arrays (gathered inside the kernels, nothing is copied)."""

    def __init__(self, policy, env, minibatch, splits=32, adam_state=None):
        from . import _lib
        self.policy, self.env, self.mb, self.splits = policy, env, int(minibatch), int(splits)
        self._lib = _lib
        dev = next(policy.parameters()).device
        for p in policy.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self.pi = self._net(policy.pi)
        self.vf = self._net(policy.vf)
        nbytes = env._dev.learner_workspace_bytes(self.mb, policy.pi[4].out_features, self.splits)
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.partials = torch.zeros(((self.mb + 255) // 256, len(STAT_KEYS)), dtype=torch.float32, device=dev)
        # the optimizer's moments and step count: owned by the caller when given (PPOLearner keeps ONE across re-builds of this object -
        # a different minibatch size must not zero Adam's history; its size depends on the network shape only)
        nstate = env._dev.learner_adam_state_bytes(policy.pi[4].out_features)
        if adam_state is None:
            adam_state = torch.zeros((nstate,), dtype=torch.uint8, device=dev)
        assert adam_state.numel() == nstate and adam_state.dtype == torch.uint8 and adam_state.device == dev
        self.adam_state = adam_state
        self.stats_acc = self.adam_state[16:36].view(torch.float32)      # running sums of the step statistics (q1env_learner_adam)
        self.cursor = self.adam_state[72:80].view(torch.int64)           # minibatch cursor: += mb per adam() (q1env_learner_batch.idx_cursor_dev)
        # saturation report of the backward pass (q1env_learner_batch.saturation_dev): [count, max |element| bits] x (policy, value)
        self.saturation = torch.zeros((4,), dtype=torch.int32, device=dev)
        self._pws = None                                         # the persistent learner's exchange buffers (epochs()), made on first use
        self.images()

    def _net(self, seq):
        l1, l2, l3 = seq[0], seq[2], seq[4]
        assert l1.in_features == 6 and l1.out_features == 256 and l2.in_features == 256 and l2.out_features == 256 and l3.in_features == 256
        for l in (l1, l2, l3):
            assert l.weight.is_contiguous() and l.weight.dtype == torch.float32 and l.weight.grad.is_contiguous()
        return self._lib.Q1LearnerNet(l1.weight.data_ptr(), l1.bias.data_ptr(), l2.weight.data_ptr(), l2.bias.data_ptr(), l3.weight.data_ptr(),
                                      l3.bias.data_ptr(), l1.weight.grad.data_ptr(), l1.bias.grad.data_ptr(), l2.weight.grad.data_ptr(),
                                      l2.bias.grad.data_ptr(), l3.weight.grad.data_ptr(), l3.bias.grad.data_ptr(), l3.out_features)

    def images(self):
        self.env._dev.learner_images_dev(self.pi, self.vf, self.ws.data_ptr(), self.mb, self.splits)

    def forward(self, obs, idx=None):
        """logits (B, out) / value (B,) of rows idx of obs (float32 (total, 6)); bit-identical to FusedPolicyForward on those rows."""
        out = self.policy.pi[4].out_features
        logits = torch.empty((self.mb, out), dtype=torch.float32, device=obs.device)
        value = torch.empty((self.mb,), dtype=torch.float32, device=obs.device)
        self.env._dev.learner_forward_dev(self.pi, self.vf, self.ws.data_ptr(), self.mb, self.splits, obs.data_ptr(),
                                          idx.data_ptr() if idx is not None else 0, logits.data_ptr(), value.data_ptr())
        return logits, value

    def backward(self, obs, idx, dlogits, dvalue, grad_scale):
        """.grad of all parameters <- d(sum_i <dlogits_i, logits_i> + dvalue_i value_i) / grad_scale after a forward() on the same rows;
        dlogits (B, out) / dvalue (B,) float32, already multiplied by grad_scale."""
        self.env._dev.learner_backward_dev(self.pi, self.vf, self.ws.data_ptr(), self.mb, self.splits, obs.data_ptr(),
                                           idx.data_ptr() if idx is not None else 0, dlogits.data_ptr(), dvalue.data_ptr(), grad_scale)

    def adam(self, lr, betas=(0.9, 0.999), eps=1e-8):
        """After step(..., skip_reduce=True): gradient reduction + torch.optim.Adam's update of the masters + weight images, ONE kernel
        (q1env_learner_adam; moments and step count in self.adam_state)."""
        self.env._dev.learner_adam_dev(self.pi, self.vf, self.ws.data_ptr(), self.mb, self.splits, float(self.mb), lr, betas[0], betas[1], eps,
                                       self.adam_state.data_ptr(), self.partials.data_ptr())

    PERSISTENT_MB = 128                                        # q1env_learner_sgd_epochs is built for RLlib's sgd_minibatch_size

    def persistent_ok(self):
        """The persistent learner serves the reference's own shape: 128-sample minibatches, 4 keys + continuous mouse (10 policy outputs)."""
        return self.mb == self.PERSISTENT_MB and self.policy.pi[4].out_features == 10 and self.env._dev.num_keys == 4

    def epochs(self, full, perms, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, klc_dev, adam, steps=None, refresh_images=True, f32=False):
        """ALL the SGD steps of an update as ONE dispatch (q1env_learner_sgd_epochs, csrc/q1learner_persist.hpp): perms int64
        (epochs, total) - one permutation of the train batch per epoch; every epoch runs total // 128 steps on consecutive windows of its
        permutation (what update()'s loop feeds q1env_learner_sgd_step one call at a time).  adam = (lr, betas, eps).  Masters, moments,
        step count, .grad (the last step's) and the running statistics [0, 1, 2, 4] of self.stats_acc are updated; [3] (total loss) is
        left to the caller.  Returns the number of steps.  f32: the float32-arithmetic kernel (q1env_learner_sgd_epochs_f32: no float16 operand
        rounding, no loss scale, no saturation - RLlib's own arithmetic; about twice the time per step)."""
        L = self._lib
        assert self.persistent_ok() and perms.dtype == torch.int64 and perms.is_contiguous() and perms.dim() == 2
        total = perms.shape[1]
        spe = total // self.mb
        n = perms.shape[0] * spe if steps is None else int(steps)
        if not 0 < n <= perms.shape[0] * spe:
            raise ValueError(f"epochs(): {n} steps asked for, the permutations hold {perms.shape[0]} epochs x {spe} minibatches")
        rows = full["adv"].shape[0]
        need = self.env._dev.learner_persistent_bytes(rows)
        if self._pws is None or self._pws.numel() < need:
            self._pws = torch.zeros((need,), dtype=torch.uint8, device=perms.device)
        ol = full["old_logits"]
        b = L.Q1LearnerBatch(self.mb, perms.data_ptr(), None, full["obs"].data_ptr(), ol.data_ptr(), ol.shape[1],
                             full["keys_packed"].data_ptr(), full["mouse"].data_ptr(), full["logp"].data_ptr(), full["adv"].data_ptr(),
                             full["value"].data_ptr(), full["vtarg"].data_ptr(), clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff,
                             klc_dev.data_ptr(), None, 0, self.saturation.data_ptr())
        lr, betas, eps = adam
        self.env._dev.learner_sgd_epochs_dev(self.pi, self.vf, self._pws.data_ptr(), b, rows, perms.numel(), n, spe, total, lr, betas[0], betas[1], eps,
                                             self.adam_state.data_ptr(), f32=f32)
        if refresh_images:
            self.images()                                      # the four-launch path's float16 weight images follow the new masters
        return n

    def persistent_status(self):
        """[0] != 0: a group barrier of the last epochs() launch timed out (1 + its index in the step), [1] = the step; [2], [3]: the exchange mode
        the policy / value group ran in (0 = agent scope, otherwise 1 + the XCD its eight workgroups shared).  Synchronises."""
        return self.env._dev.learner_persistent_status(self._pws.data_ptr()) if self._pws is not None else [0, 0, 0, 0]

    def step(self, full, idx, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, klc_dev, skip_reduce=False, use_cursor=False, adam=None):
        """full: dict of the whole trajectory batch (obs (total,6), old_logits (total,W), keys_packed, mouse, logp, adv, value, vtarg);
        idx int64 (B,) or None.  Returns the statistics vector (STAT_KEYS order, means over the minibatch).
        skip_reduce: the parameter gradients stay as split-K partial sums for adam() (.grad is then written by adam()).
        adam = (lr, betas, eps): the optimizer step too, in the same call (same result as step(skip_reduce=True) + adam(lr, betas, eps))."""
        L = self._lib
        ol = full["old_logits"]
        # use_cursor: idx is a whole epoch's permutation and the minibatch is idx[cursor : cursor + mb] with the device-resident cursor
        # that adam() advances - nothing to copy per minibatch, and the step replays from a captured graph as it is
        assert ol.is_contiguous() and full["obs"].is_contiguous() and (idx is None or idx.dtype == torch.int64)
        assert idx is None or (idx.numel() >= self.mb if use_cursor else idx.numel() == self.mb)
        b = L.Q1LearnerBatch(self.mb, idx.data_ptr() if idx is not None else None, self.cursor.data_ptr() if (use_cursor and idx is not None) else None,
                             full["obs"].data_ptr(), ol.data_ptr(), ol.shape[1],
                             full["keys_packed"].data_ptr(), full["mouse"].data_ptr(), full["logp"].data_ptr(), full["adv"].data_ptr(),
                             full["value"].data_ptr(), full["vtarg"].data_ptr(), clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff,
                             klc_dev.data_ptr(), self.partials.data_ptr(), int(bool(skip_reduce)), self.saturation.data_ptr())
        if adam is not None:                         # the whole SGD step as one call (q1env_learner_sgd_step): step(skip_reduce) + adam() in four launches
            lr, betas, eps = adam
            self.env._dev.learner_sgd_step_dev(self.pi, self.vf, self.ws.data_ptr(), self.splits, b, lr, betas[0], betas[1], eps, self.adam_state.data_ptr())
            return None
        self.env._dev.learner_step_dev(self.pi, self.vf, self.ws.data_ptr(), self.splits, b)
        if skip_reduce:
            return None                              # adam() folds the statistics into self.stats_acc on the device
        return self.partials.sum(dim=0) / self.mb


class PPOLearner:
    """SGD epochs over trajectory batches.

    use_graph:   one SGD step (forward, loss, backward, Adam) captured once into a hipGraph and replayed per minibatch (single
                 process only).
    fused_loss:  the loss and its gradient w.r.t. (logits, value) come from ONE HIP kernel (q1env_ppo_loss_grad, closed-form
                 derivatives) instead of ~100 elementwise torch launches and their autograd twins; torch autograd only runs the
                 two MLPs.  Needs `env` (a TensorVectorEnv: its handle supplies num_keys / action_range and the stream).
    native:      the WHOLE step up to the gradients - minibatch gather, both MLPs forward and backward, the loss gradient, the weight
                 gradients - is the library's own gfx950 kernels (NativeStep / q1env_learner_step: float16 matrix-core operands,
                 float32 accumulation and master weights); torch only runs the (fused) Adam.  Needs `env`; implies the fused loss.
    """

    def __init__(self, policy, action_range, lr=5e-6, gamma=0.99, lam=0.95, clip_param=0.3, vf_clip_param=100.0,
                 vf_loss_coeff=1.0, entropy_coeff=0.01, kl_coeff=0.2, kl_target=0.0036, num_sgd_iter=30,
                 minibatch_size=128, num_keys=4, seed=0, use_graph=False, fused_loss=False, env=None, discrete_yaw_steps=-1,
                 allow_yaw=True, autocast_dtype=None, fused_adam=False, native=False, native_splits=32, native_adam=True, persistent=None,
                 dynamic_loss_scale=False, precision="f16"):
        self.policy = policy
        self.action_range = float(action_range)
        self.gamma, self.lam = gamma, lam
        self.clip_param, self.vf_clip_param = clip_param, vf_clip_param
        self.vf_loss_coeff, self.entropy_coeff = vf_loss_coeff, entropy_coeff
        self.kl_coeff, self.kl_target = kl_coeff, kl_target
        self.num_sgd_iter, self.minibatch_size, self.num_keys = num_sgd_iter, minibatch_size, num_keys
        self.discrete_yaw_steps, self.allow_yaw = discrete_yaw_steps, allow_yaw
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.use_graph = bool(use_graph) and self.world == 1
        self.fused_loss, self.env = bool(fused_loss), env
        self.native, self.native_splits, self.native_adam = bool(native), int(native_splits), bool(native_adam)
        # persistent: the whole update (num_sgd_iter epochs of 128-sample minibatches) as ONE dispatch (NativeStep.epochs); None = whenever
        # the shape allows it (native, own Adam, single process, minibatch 128, the reference's action structure)
        self.persistent = persistent
        # precision of the PERSISTENT learner's arithmetic (the reference's shape: minibatch 128; the four-launch path of larger minibatches is float16
        # only).  "f16" (default) = float16 matrix operands, float32 accumulation / masters / optimizer, 11.8 us per step, per-sample gradients saturating
        # at 65 504 / loss scale; "f32" = q1env_learner_sgd_epochs_f32: float32 everywhere, no loss scale, nothing saturates - RLlib's own arithmetic (TF
        # PPO is float32 end to end, grad_clip None), 27 us per step.  Round 6 measured both under the reference's configuration over 8 / 9 seeds
        # (STATE.md "fp32 control"): 3 of 8 float16 runs and 3 of 9 float32 runs end below 5 500, medians 5 660 / 5 708 - the seed spread belongs to the
        # configuration, not to the arithmetic -, so the fast kernel stays the default and the float32 one is the explicit choice for the reference's numerics.
        if precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.precision = precision
        # dynamic_loss_scale (native learner; OFF by default - measured, see below): the float16 loss scales of the next
        # update are chosen from THIS update's largest per-sample gradient element (NativeStep.saturation), as exact powers of two, so that
        # the largest element sits a factor LOSS_SCALE_HEADROOM below float16's largest finite value: (almost) nothing saturates (RLlib:
        # grad_clip = None).  Measured on MI355X (round 5): under the reference's configuration (minibatch 128, lr 5e-6) the run is unchanged
        # (5 702 against 5 697 zero-start reward; 176 saturated elements in the run against 1e8), but with large minibatches (32 768, lr 3e-5)
        # it COSTS the result - 5 290 / 5 240 against 5 652 for two seeds (profiles/r5_train_ppo_largebatch_*.json): the largest element is a
        # handful of outlier samples (probability ratios that explode), a scale that fits them (down to 2^-6) pushes the bulk of the
        # gradients into float16's subnormal range, and it is the BULK that carries the learning signal - the same plateau round 3 saw with a
        # scale of 1.  Saturating the outliers at the static scale (a per-sample clip of a few samples in 10^4) is the better trade; it stays
        # counted and reported.
        self.dynamic_loss_scale = bool(dynamic_loss_scale) and self.native
        self.pi_upscale, self.value_downscale = 256.0, 1.0      # the library's defaults (csrc/q1env_learner.hip)
        if (self.fused_loss or self.native) and env is None:
            raise ValueError("fused_loss=True / native=True need env= (the TensorVectorEnv whose handle runs the kernels)")
        self._native = None
        self._adam_state = None                 # native own-Adam moments + step count: allocated once, survives NativeStep re-builds
        self._graph_hparams = None              # (lr, betas, eps) baked into the captured graph's kernel arguments
        self._full = None
        self._idx = None
        self._perm = None                       # native + own Adam: the epoch's permutation, read through the device-resident cursor
        self._perms = None                      # persistent learner: all epochs' permutations of one update
        self._perms_next, self._perms_ready, self._perm_stream = None, None, None      # ... and the next update's, drawn ahead
        self.prefetch_perms = os.environ.get("Q1_PPO_PREFETCH_PERMS", "1") != "0"
        # autocast_dtype (e.g. torch.bfloat16): the two MLPs' matrix products run on reduced-precision operands with float32
        # accumulation (master weights, loss, its gradient and Adam stay float32); fused_adam: one multi-tensor Adam launch
        self.autocast_dtype = autocast_dtype
        self.opt = self._make_adam(policy, lr, self.use_graph, fused_adam)
        self._graph = None
        self._klc = None                        # device scalar: the KL coefficient as the captured graph / the kernel reads it
        self._work = None
        self.gen = None
        self.seed = seed

    @staticmethod
    def _make_adam(policy, lr, capturable, fused_adam):
        """Adam, multi-tensor fused when asked for and supported: older torch versions reject fused=True together with
        capturable=True (or on CPU parameters) at construction - fall back to the plain implementation instead of failing."""
        if fused_adam:
            try:
                return torch.optim.Adam(policy.parameters(), lr=lr, capturable=capturable, fused=True)
            except (RuntimeError, ValueError, TypeError) as ex:
                import warnings
                warnings.warn(f"PPOLearner: fused Adam unavailable here ({ex}); using the unfused optimizer", RuntimeWarning, stacklevel=3)
        return torch.optim.Adam(policy.parameters(), lr=lr, capturable=capturable)

    def _flatten(self, traj, adv, vtarg, old_logits):
        t, n = traj["reward"].shape
        keys = ((traj["keys"].reshape(-1, 1).long() >> torch.arange(self.num_keys, device=adv.device)) & 1)
        return {"obs": traj["obs"][:t].reshape(t * n, 6), "keys": keys, "keys_packed": traj["keys"].reshape(-1),
                "mouse": traj["mouse"].reshape(-1, 1), "logp": traj["logp"].reshape(-1), "value": traj["value"][:t].reshape(-1),
                "adv": adv.reshape(-1), "vtarg": vtarg.reshape(-1), "old_logits": old_logits}

    def _sgd_step(self, mb):
        """One minibatch: forward, loss, backward, (gradient all-reduce,) Adam.  Returns the stats vector (STAT_KEYS order).
        native: `mb` is ignored - the minibatch is rows self._idx (or, with the library's own Adam, the window of self._perm at the
        device-resident cursor) of the persistent full-batch arrays self._full."""
        if self.native:
            own_adam = self.world == 1 and self.native_adam       # no all-reduce between gradients and optimizer: one fused kernel
            # own optimizer: the minibatch is a window of the epoch's permutation (self._perm) at the cursor adam() advances
            if own_adam:                              # statistics accumulate in self._native.stats_acc
                g = self.opt.param_groups[0]
                self._native.step(self._full, self._perm, self.clip_param, self.vf_clip_param, self.vf_loss_coeff, self.entropy_coeff, self._klc,
                                  skip_reduce=True, use_cursor=True, adam=(g["lr"], g["betas"], g["eps"]))
                return None
            stats = self._native.step(self._full, self._idx, self.clip_param, self.vf_clip_param, self.vf_loss_coeff, self.entropy_coeff, self._klc)
            if self.world > 1:
                allreduce_grads_([p for p in self.policy.parameters()], self.world)
            self.opt.step()
            self._native.images()
            return stats
        if self.fused_loss:
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    logits, value = self.policy(mb["obs"])
                logits, value = logits.float().contiguous(), value.float().contiguous()
            else:
                logits, value = self.policy(mb["obs"])
            bsz, width = logits.shape
            if self._work is None or self._work[0].shape != logits.shape:
                dev = logits.device
                self._work = (torch.empty_like(logits), torch.empty_like(value),
                              torch.zeros(((bsz + 255) // 256, len(STAT_KEYS)), dtype=torch.float32, device=dev))
            dlogits, dvalue, partials = self._work
            assert logits.is_contiguous() and value.is_contiguous() and mb["old_logits"].is_contiguous()
            self.env._dev.ppo_loss_grad_dev(bsz, logits.data_ptr(), mb["old_logits"].data_ptr(), width, mb["keys_packed"].data_ptr(),
                                            mb["mouse"].data_ptr(), mb["logp"].data_ptr(), mb["adv"].data_ptr(), value.data_ptr(),
                                            mb["value"].data_ptr(), mb["vtarg"].data_ptr(), self.clip_param, self.vf_clip_param,
                                            self.vf_loss_coeff, self.entropy_coeff, self._klc.data_ptr(), dlogits.data_ptr(),
                                            dvalue.data_ptr(), partials.data_ptr())
            self.opt.zero_grad(set_to_none=True)
            torch.autograd.backward([logits, value], [dlogits, dvalue])
            stats = partials.sum(dim=0) / bsz
        else:
            loss, st = ppo_loss(self.policy, mb, self.action_range, self.clip_param, self.vf_clip_param, self.vf_loss_coeff,
                                self.entropy_coeff, self._klc, self.num_keys, self.discrete_yaw_steps, self.allow_yaw,
                                autocast_dtype=self.autocast_dtype)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            stats = torch.stack([st[k].float() for k in STAT_KEYS])
        if self.world > 1:
            allreduce_grads_([p for p in self.policy.parameters()], self.world)
        self.opt.step()
        return stats

    def _capture(self, b, mb, dev):
        """Capture one SGD step on static minibatch buffers into a hipGraph (warm-up on a side stream first; the warm-up and
        capture steps trained on the first minibatch, so parameters and Adam moments are rolled back afterwards)."""
        own_adam = self.native and self.world == 1 and self.native_adam
        if self.native:
            self._mb = None
            self._idx.copy_(torch.arange(mb, device=dev))
            if own_adam:
                self._perm.copy_(torch.arange(self._perm.numel(), device=dev))
        else:
            self._mb = {k: torch.empty((mb,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev) for k, v in b.items()}
            for k, v in b.items():
                self._mb[k].copy_(v[:mb])
        self._acc = torch.zeros((len(STAT_KEYS),), dtype=torch.float32, device=dev)
        snapshot = [p.detach().clone() for p in self.policy.parameters()]
        # Adam's moments / step counts as they are NOW (empty before the very first step, accumulated on a re-capture after
        # the minibatch size changed): the warm-up and capture steps below must not leave a trace in them either
        opt_snapshot = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                        for p, st in self.opt.state.items()}

        adam_snapshot = self._native.adam_state.clone() if self.native else None      # the native optimizer's moments and step count

        def one_step():
            st = self._sgd_step(self._mb)
            if st is not None:
                self._acc += st

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if self.env is not None:
                self.env.use_current_stream()
            for _ in range(3):
                if own_adam:
                    self._native.cursor.zero_()        # every warm-up step on the first minibatch (adam() advances the cursor)
                one_step()
            if own_adam:
                self._native.cursor.zero_()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if self.env is not None:
                self.env.use_current_stream()          # the kernel launch must land on the capturing stream
            one_step()
        if self.env is not None:
            self.env.use_current_stream()              # back on the caller's stream
        with torch.no_grad():
            for p, q in zip(self.policy.parameters(), snapshot):
                p.copy_(q)
            for p, st_ in self.opt.state.items():
                saved = opt_snapshot.get(id(p))
                for k_, v_ in st_.items():
                    if torch.is_tensor(v_):          # in place: the captured graph holds these addresses
                        if saved is not None and k_ in saved:
                            v_.copy_(saved[k_])
                        else:
                            v_.zero_()
        if self.native:
            self._native.adam_state.copy_(adam_snapshot)
            self._native.saturation.zero_()          #  the discarded warm-up / capture steps leave no trace in the report either
            self._native.images()                    # the float16 images follow the restored masters
        self._graph = g

    LOSS_SCALE_HEADROOM = 8.0
    PI_UPSCALE_RANGE = (2.0 ** -6, 2.0 ** 12)
    VALUE_DOWNSCALE_RANGE = (2.0 ** -8, 2.0 ** 10)

    def _next_loss_scales(self, max_abs_pi, max_abs_vf):
        """max_abs_*: the largest |element| the backward pass converted to float16 in the update that just ended, AS IT TRAVELLED (i.e.
        times pi_upscale / divided by value_downscale).  Returns the (pi_upscale, value_downscale) of the next update."""
        import math
        target = 65504.0 / self.LOSS_SCALE_HEADROOM
        up, down = self.pi_upscale, self.value_downscale
        if max_abs_pi > 0.0 and math.isfinite(max_abs_pi):
            raw = max_abs_pi / self.pi_upscale                                  # the largest unscaled element
            up = 2.0 ** math.floor(math.log2(target / raw))
            up = min(max(up, self.PI_UPSCALE_RANGE[0]), self.PI_UPSCALE_RANGE[1])
        if max_abs_vf > 0.0 and math.isfinite(max_abs_vf):
            raw = max_abs_vf * self.value_downscale
            down = 2.0 ** math.ceil(math.log2(raw / target))
            down = min(max(down, self.VALUE_DOWNSCALE_RANGE[0]), self.VALUE_DOWNSCALE_RANGE[1])
        return up, down

    def _hparams(self):
        g = self.opt.param_groups[0]
        return (float(g["lr"]), tuple(float(x) for x in g["betas"]), float(g["eps"]))

    def _graph_key(self):
        """everything a captured graph of the native step bakes into kernel arguments"""
        return (self._hparams(), self.pi_upscale, self.value_downscale)

    def state_dict(self):
        """Everything a resume needs: torch Adam's state (the non-native / multi-rank paths), the native optimizer's moments and step
        count (q1env_learner_adam's state block; self.opt is never stepped on that path, so its state_dict() is empty there), and the
        adaptive KL coefficient."""
        if self._perms_ready is not None:
            self._perms_ready.synchronize()
        return {"opt": self.opt.state_dict(), "kl_coeff": float(self.kl_coeff), "pi_upscale": self.pi_upscale, "value_downscale": self.value_downscale,
                "native_adam": None if self._adam_state is None else self._adam_state.detach().cpu().clone(),
                # the minibatch permutations: the generator (already advanced past the NEXT update's permutations when those were drawn ahead) and
                # the prefetched permutations themselves - a resumed run continues with exactly the permutations the uninterrupted one uses
                "gen": None if self.gen is None else self.gen.get_state().cpu().clone(),
                "perms_next": None if (self._perms_next is None or self._perms_ready is None) else self._perms_next.detach().cpu().clone()}

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd["opt"])
        self.kl_coeff = float(sd.get("kl_coeff", self.kl_coeff))
        self.pi_upscale, self.value_downscale = float(sd.get("pi_upscale", self.pi_upscale)), float(sd.get("value_downscale", self.value_downscale))
        na = sd.get("native_adam")
        if na is not None:
            dev = next(self.policy.parameters()).device
            if self._adam_state is None:
                self._adam_state = na.to(dev).clone()
            else:
                self._adam_state.copy_(na.to(dev))       # in place: a captured graph / NativeStep hold this address
        dev = next(self.policy.parameters()).device
        if sd.get("gen") is not None:
            self.gen = torch.Generator(device=dev).manual_seed(0)
            self.gen.set_state(sd["gen"].cpu())
        if sd.get("perms_next") is not None and dev.type == "cuda":
            self._perms_next = sd["perms_next"].to(dev).contiguous()
            self._perms_ready = torch.cuda.Event()
            self._perms_ready.record(torch.cuda.current_stream(dev))

    def update(self, traj, adv, vtarg):
        """SGD epochs over one trajectory batch; adv/vtarg from q1env_gae.  Returns averaged stats (python floats)."""
        dev = adv.device
        t, n = traj["reward"].shape
        if "logits" in traj:                    # the behaviour policy's own outputs (exact even when sampling ran in bf16)
            old_logits = traj["logits"].reshape(t * n, -1)
        else:
            with torch.no_grad():
                old_logits, _ = self.policy(traj["obs"][:t].reshape(t * n, 6))
        b = self._flatten(traj, adv, vtarg, old_logits)
        a = b["adv"]
        mean, sq = a.mean(), (a * a).mean()
        if self.world > 1:                      # standardise over the GLOBAL batch
            ms = torch.stack([mean, sq])
            dist.all_reduce(ms, op=dist.ReduceOp.SUM)
            mean, sq = ms[0] / self.world, ms[1] / self.world
        b["adv"] = (a - mean) / torch.sqrt(torch.clamp(sq - mean * mean, min=1e-8))
        total = t * n
        mb = min(self.minibatch_size, total)
        if self.gen is None:
            self.gen = torch.Generator(device=dev).manual_seed(self.seed)
        if self._klc is None:
            self._klc = torch.zeros((), dtype=torch.float32, device=dev)
        self._klc.fill_(float(self.kl_coeff))
        if self.native:
            # the kernels gather the minibatch themselves: the whole batch sits in PERSISTENT arrays (a captured graph holds their
            # addresses), refreshed once per update; the minibatch is the static index vector self._idx
            keys_ = ("obs", "old_logits", "keys_packed", "mouse", "logp", "adv", "value", "vtarg")
            if self._full is None or self._full["adv"].shape[0] != total or self._idx.shape[0] != mb:
                self._full = {k: torch.empty_like(b[k].reshape(total, -1) if k in ("obs", "old_logits") else b[k].reshape(-1)).contiguous() for k in keys_}
                self._idx = torch.zeros((mb,), dtype=torch.int64, device=dev)
                self._perm = torch.zeros((total,), dtype=torch.int64, device=dev)
                self._native = NativeStep(self.policy, self.env, mb, self.native_splits, adam_state=self._adam_state)
                self._adam_state = self._native.adam_state      # first build allocates it; later builds (other mb / total) reuse it
                self._graph = None
            for k in keys_:
                self._full[k].copy_(b[k].reshape(self._full[k].shape))
        own_adam = self.native and self.world == 1 and self.native_adam
        use_persistent = bool(own_adam and self.persistent is not False and self._native.persistent_ok() and total >= mb)
        if (self.persistent is True or self.precision == "f32") and not use_persistent:
            raise ValueError("PPOLearner(persistent=True / precision='f32') needs native=True, native_adam=True, one process, minibatch_size 128 and the reference's action structure")
        # lr / betas / eps are kernel ARGUMENTS of the native Adam, baked into a captured graph: a changed param_group re-captures
        if self.native:
            self.env._dev.learner_set_loss_scale(self.pi_upscale, self.value_downscale)
        if self.use_graph and self._graph is not None and self.native and self._graph_hparams != self._graph_key():
            self._graph = None
        if self.use_graph and not use_persistent and (self._graph is None or (not self.native and self._mb["adv"].shape[0] != mb)):
            self._capture(b, mb, dev)
            self._graph_hparams = self._graph_key()
        acc, steps = torch.zeros((len(STAT_KEYS),), dtype=torch.float32, device=dev), 0
        if self.use_graph and not use_persistent:
            self._acc.zero_()
        if self._native is not None:
            self._native.stats_acc.zero_()
        if use_persistent:
            # the same permutations, in the same generator order, as the per-step loop below draws - then one launch for all of them
            # (drawn one update AHEAD on a side stream - 30 device sorts, ~3 ms - while the learner's 16 workgroups run; same generator, same order)
            shape = (self.num_sgd_iter, total)
            if self._perms_next is not None and self._perms_next.shape == shape and self._perms_ready is not None:
                torch.cuda.current_stream(dev).wait_event(self._perms_ready)
                self._perms, self._perms_next = self._perms_next, self._perms
            else:
                if self._perms is None or self._perms.shape != shape:
                    self._perms = torch.empty(shape, dtype=torch.int64, device=dev)
                for e in range(self.num_sgd_iter):
                    self._perms[e].copy_(torch.randperm(total, device=dev, generator=self.gen))
            self.env.use_current_stream()
            free_evt = torch.cuda.Event()
            free_evt.record(torch.cuda.current_stream(dev))     # (the buffer about to be refilled was read by the PREVIOUS update's kernel)
            steps = self._native.epochs(self._full, self._perms, self.clip_param, self.vf_clip_param, self.vf_loss_coeff, self.entropy_coeff,
                                        self._klc, self._hparams(), f32=self.precision == "f32")
            if self.prefetch_perms:
                if self._perm_stream is None:
                    self._perm_stream = torch.cuda.Stream(device=dev)
                if self._perms_next is None or self._perms_next.shape != shape:
                    self._perms_next = torch.empty(shape, dtype=torch.int64, device=dev)
                self._perm_stream.wait_event(free_evt)
                with torch.cuda.stream(self._perm_stream):
                    for e in range(self.num_sgd_iter):
                        self._perms_next[e].copy_(torch.randperm(total, device=dev, generator=self.gen))
                    self._perms_ready = torch.cuda.Event()
                    self._perms_ready.record(self._perm_stream)
        for _ in range(0 if use_persistent else self.num_sgd_iter):
            perm = torch.randperm(total, device=dev, generator=self.gen)
            if own_adam:                            # the whole epoch's order once; each step reads its window at the device-resident cursor
                self._perm.copy_(perm)
                self._native.cursor.zero_()
            for s in range(0, total - mb + 1, mb):
                idx = perm[s:s + mb]
                if self.native:
                    if not own_adam:
                        self._idx.copy_(idx)
                    if self.use_graph:
                        self._graph.replay()
                    else:
                        st = self._sgd_step(None)
                        if st is not None:
                            acc = acc + st
                elif self.use_graph:
                    for k, v in b.items():
                        torch.index_select(v, 0, idx, out=self._mb[k])
                    self._graph.replay()
                else:
                    acc = acc + self._sgd_step({k: v[idx] for k, v in b.items()})
                steps += 1
        if self.use_graph and not use_persistent:
            acc = self._acc.clone()
        if self.native and self.world == 1 and self.native_adam:
            acc = self._native.stats_acc.clone()
            if use_persistent:                        # total = policy + kl_coeff kl + vf_coeff vf - entropy_coeff entropy (all means: linear)
                i = {k: j for j, k in enumerate(STAT_KEYS)}
                acc[i["total_loss"]] = (acc[i["policy_loss"]] + float(self.kl_coeff) * acc[i["kl"]] + self.vf_loss_coeff * acc[i["vf_loss"]]
                                        - self.entropy_coeff * acc[i["entropy"]])
                st = self._native.persistent_status()
                if st[0]:
                    raise RuntimeError(f"persistent learner: group barrier {st[0] - 1} timed out in step {st[1]} (the 16 workgroups were not co-resident?)")
        acc = acc / steps
        if self.world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            acc /= self.world
        out = dict(zip(STAT_KEYS, acc.tolist()))
        # adaptive KL coefficient (RLlib KLCoeffMixin.update_kl)
        if out["kl"] > 2.0 * self.kl_target:
            self.kl_coeff *= 1.5
        elif out["kl"] < 0.5 * self.kl_target:
            self.kl_coeff *= 0.5
        out["kl_coeff"] = self.kl_coeff
        out["sgd_steps"] = steps
        if self._native is not None:
            # float16 gradient operands: how often the backward pass had to clamp one at 65504 during this update, and the largest
            # magnitude it met (scaled as it travels: x minibatch x loss scale) - a silent per-sample clip otherwise
            sat = self._native.saturation.cpu()
            out["grad_saturated_pi"], out["grad_saturated_vf"] = int(sat[0]), int(sat[2])
            out["grad_max_abs_pi"] = float(sat[1:2].view(torch.float32)[0])
            out["grad_max_abs_vf"] = float(sat[3:4].view(torch.float32)[0])
            self._native.saturation.zero_()
            out["pi_upscale"], out["value_downscale"] = self.pi_upscale, self.value_downscale      # the scales THIS update ran with
            if self.dynamic_loss_scale:
                self.pi_upscale, self.value_downscale = self._next_loss_scales(out["grad_max_abs_pi"], out["grad_max_abs_vf"])
        return out
