"""Policy-side counterpart of the env fast path ("next" row 1 of SURVEY.md section 8f).

Torch restatement of the reference's TF action distribution (q1physrl/action_dist.py):
  GaussianSquashedGaussian   action_dist.py:141-196 (base class 46-138): a diagonal Gaussian squashed through
                             a Normal CDF into (low, high), with closed-form kl / entropy
  Q1PhysActionDist           action_dist.py:199-243: a Discrete(2) categorical per key + the squashed Gaussian for the mouse
and the network shape of the published WR checkpoint (data/checkpoints/wr: policy 6->256->256->10, separate value
net 6->256->256->1, tanh).  torch here is the learner-side library (GEMMs via hipBLASLt);
the per-tick sampling itself has a fused HIP kernel (q1env_policy_sample, used by q1physrl_amd.sampler).

RLlib 0.8.4 constants (ray/rllib/utils): SMALL_NUMBER = 1e-6, MIN_LOG_NN_OUTPUT = -20, MAX_LOG_NN_OUTPUT = 2.
"""
import math

import torch
from torch import nn

SMALL_NUMBER = 1e-6
TANH_PRESCALE = 2.8853900817779268        # 2 log2(e): folded into the W2 block of the fused kernel's weight image (q1policy.hpp)
MIN_LOG_NN_OUTPUT = -20.0
MAX_LOG_NN_OUTPUT = 2.0
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _normal_logpdf(x, mean, log_std):
    z = (x - mean) * torch.exp(-log_std)
    return -0.5 * z * z - log_std - _HALF_LOG_2PI


class GaussianSquashedGaussian:
    """Normal(mean, std) pushed through NormalCDF(. / _SCALE) and an affine map onto (low, high)."""
    _SCALE = 0.5 * 1.8137            # action_dist.py:151: Var(N(0, 2*_SCALE)) = Var(Logistic(0, 1))

    def __init__(self, inputs: torch.Tensor, low=-1.0, high=1.0):
        mean, log_std = torch.chunk(inputs, 2, dim=-1)                       # action_dist.py:65
        self.log_std = torch.clamp(log_std, MIN_LOG_NN_OUTPUT, MAX_LOG_NN_OUTPUT)   # :69-70
        self.mean = torch.clamp(mean, -3.0, 3.0)                              # :72
        self.std = torch.exp(self.log_std)
        assert low < high
        self.low, self.high = float(low), float(high)

    # -- squash / unsquash (action_dist.py:186-196)
    def _squash(self, raw):
        v = torch.special.ndtr(raw / self._SCALE)
        return torch.clamp(v, SMALL_NUMBER, 1.0 - SMALL_NUMBER) * (self.high - self.low) + self.low

    def _unsquash(self, values):
        return self._SCALE * torch.special.ndtri((values - self.low) / (self.high - self.low))

    def _log_squash_grad(self, raw):                                          # :180-184
        return _normal_logpdf(raw, torch.zeros_like(raw), torch.full_like(raw, math.log(self._SCALE))) + math.log(self.high - self.low)

    def sample(self, generator=None):                                         # :98-101
        eps = torch.randn(self.mean.shape, dtype=self.mean.dtype, device=self.mean.device, generator=generator)
        return self._squash(self.mean + self.std * eps)

    def deterministic_sample(self):                                           # :84-89
        return self._squash(self.mean)

    def logp(self, x):                                                        # :91-96
        raw = self._unsquash(x)
        return torch.sum(_normal_logpdf(raw, self.mean, self.log_std) - self._log_squash_grad(raw), dim=1)

    def kl(self, other):                                                      # :153-165
        return torch.sum(other.log_std - self.log_std
                         + (self.std ** 2 + (self.mean - other.mean) ** 2) / (2.0 * other.std ** 2) - 0.5, dim=1)

    def entropy(self):                                                        # :167-178
        return torch.sum(math.log(self.high - self.low)
                         - (math.log(self._SCALE) - self.log_std
                            + (self.std ** 2 + self.mean ** 2) / (2.0 * self._SCALE ** 2) - 0.5), dim=1)


class Categorical2:
    """RLlib's TF Categorical on a Discrete(M) space: logits (B, M) - M = 2 for a key, M = 2S+1 for a discrete mouse
    (Config.discrete_yaw_steps = S, env.py:216-219; the reference obtains it from ModelCatalog.get_action_dist,
    action_dist.py:221-222)."""

    def __init__(self, logits):
        self.logits = logits
        self.logprobs = torch.log_softmax(logits, dim=-1)

    def sample(self, generator=None):
        u = torch.rand(self.logits.shape[0], device=self.logits.device, generator=generator)
        if self.logits.shape[1] == 2:
            return (u < torch.exp(self.logprobs[:, 1])).long()
        cdf = torch.cumsum(torch.exp(self.logprobs), dim=1)                   # inverse CDF, like the sampling kernel
        return torch.clamp((cdf <= u[:, None]).sum(dim=1), max=self.logits.shape[1] - 1)

    def deterministic_sample(self):
        return torch.argmax(self.logits, dim=-1)

    def logp(self, a):
        return torch.gather(self.logprobs, 1, a.long().view(-1, 1)).squeeze(1)

    def entropy(self):
        return -torch.sum(torch.exp(self.logprobs) * self.logprobs, dim=1)

    def kl(self, other):
        return torch.sum(torch.exp(self.logprobs) * (self.logprobs - other.logprobs), dim=1)


CategoricalN = Categorical2


def policy_row_width(num_keys=4, discrete_yaw_steps=-1, allow_yaw=True):
    """Q1PhysActionDist.required_model_output_shape (action_dist.py:236-241): two logits per key, then (mean, log_std) of the
    continuous mouse, or the 2S+1 logits of a discrete mouse, or nothing without a mouse dimension."""
    if not allow_yaw:
        return 2 * num_keys
    return 2 * num_keys + (2 if discrete_yaw_steps == -1 else 2 * discrete_yaw_steps + 1)


class Q1PhysActionDist:
    """Tuple distribution over (key_0 .. key_{K-1}, mouse) (action_dist.py:199-243).

    inputs: (B, 2K + 2) = K x (logit0, logit1), then (mean, log_std); with discrete_yaw_steps = S > 0 the mouse child is a
    Categorical over 2S+1 steps and the row is (B, 2K + 2S+1); without a mouse dimension (allow_yaw=False) it is (B, 2K).
    Actions are (keys (B,K) int64, mouse (B,1) float - the step index for a discrete mouse)."""

    def __init__(self, inputs, action_range, num_keys=4, discrete_yaw_steps=-1, allow_yaw=True):
        self.num_keys = num_keys
        self.discrete = allow_yaw and discrete_yaw_steps != -1
        self.keys = [Categorical2(inputs[:, 2 * k:2 * k + 2]) for k in range(num_keys)]
        if not allow_yaw:
            self.mouse = None
        elif self.discrete:
            self.mouse = CategoricalN(inputs[:, 2 * num_keys:2 * num_keys + 2 * discrete_yaw_steps + 1])
        else:
            self.mouse = GaussianSquashedGaussian(inputs[:, 2 * num_keys:2 * num_keys + 2], low=-float(action_range), high=float(action_range))

    @staticmethod
    def required_model_output_shape(num_keys=4, discrete_yaw_steps=-1, allow_yaw=True):   # :236-241
        return policy_row_width(num_keys, discrete_yaw_steps, allow_yaw)

    def _mouse_value(self, m):
        return m.view(-1, 1).to(self.keys[0].logits.dtype) if self.discrete else m

    def sample(self, generator=None):
        keys = torch.stack([c.sample(generator) for c in self.keys], dim=1)
        if self.mouse is None:
            return keys, torch.zeros((keys.shape[0], 1), dtype=self.keys[0].logits.dtype, device=keys.device)
        return keys, self._mouse_value(self.mouse.sample(generator))

    def deterministic_sample(self):
        keys = torch.stack([c.deterministic_sample() for c in self.keys], dim=1)
        if self.mouse is None:
            return keys, torch.zeros((keys.shape[0], 1), dtype=self.keys[0].logits.dtype, device=keys.device)
        return keys, self._mouse_value(self.mouse.deterministic_sample())

    def logp(self, keys, mouse):
        if self.mouse is None:
            lp = 0.0
        elif self.discrete:
            lp = self.mouse.logp(mouse.reshape(-1))
        else:
            lp = self.mouse.logp(mouse)
        for k, c in enumerate(self.keys):
            lp = lp + c.logp(keys[:, k])
        return lp

    def entropy(self):
        return sum(c.entropy() for c in self.keys) + (self.mouse.entropy() if self.mouse is not None else 0.0)

    def kl(self, other):
        return sum(a.kl(b) for a, b in zip(self.keys, other.keys)) + (self.mouse.kl(other.mouse) if self.mouse is not None else 0.0)


def pack_keys(keys: torch.Tensor) -> torch.Tensor:
    """(B, K) 0/1 -> uint8 bitmask (bit k = Key k), the packed action layout of q1env_step."""
    w = (1 << torch.arange(keys.shape[1], device=keys.device, dtype=torch.int64))
    return (keys.long() * w).sum(dim=1).to(torch.uint8)


class Q1Policy(nn.Module):
    """RLlib fcnet as used by the reference run (params.yml + PPO defaults): tanh MLP 6 -> 256 -> 256 -> 2K+2 and a
    separate value MLP 6 -> 256 -> 256 -> 1 (the WR checkpoint has 70 154 + 67 841 = 137 995 parameters)."""

    def __init__(self, num_keys=4, hidden=256, obs_dim=6, discrete_yaw_steps=-1, allow_yaw=True):
        super().__init__()
        self.num_keys, self.discrete_yaw_steps, self.allow_yaw = num_keys, discrete_yaw_steps, allow_yaw
        out = Q1PhysActionDist.required_model_output_shape(num_keys, discrete_yaw_steps, allow_yaw)
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, out))
        self.vf = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, 1))
        with torch.no_grad():                      # RLlib: normc_initializer(0.01) on the logits layer -> near-uniform start
            self.pi[-1].weight.mul_(0.01)
            self.pi[-1].bias.zero_()

    def forward(self, obs):
        return self.pi(obs), self.vf(obs).squeeze(-1)

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())


def load_rllib_fcnet_weights(policy: Q1Policy, weights) -> Q1Policy:
    """Load an RLlib 0.8.4 TF fcnet state (as exported by oracle/export_wr_weights.py from the reference's
    data/checkpoints/wr: fc_1, fc_2, fc_out, fc_value_1, fc_value_2, value_out; TF kernels are (in, out)) into a Q1Policy."""
    import numpy as np
    pairs = [(policy.pi[0], "fc_1"), (policy.pi[2], "fc_2"), (policy.pi[4], "fc_out"),
             (policy.vf[0], "fc_value_1"), (policy.vf[2], "fc_value_2"), (policy.vf[4], "value_out")]
    with torch.no_grad():
        for layer, name in pairs:
            w = torch.from_numpy(np.ascontiguousarray(np.asarray(weights[name + ".kernel"], dtype=np.float32).T))
            b = torch.from_numpy(np.asarray(weights[name + ".bias"], dtype=np.float32))
            assert layer.weight.shape == w.shape and layer.bias.shape == b.shape, (name, layer.weight.shape, w.shape)
            layer.weight.copy_(w)
            layer.bias.copy_(b)
    return policy


class FusedPolicyForward:
    """Inference-side twin of a Q1Policy for the sampler loop: both networks evaluated by the fused gfx950 kernel
    (q1env_policy_forward: all three layers on the matrix cores with float16 operands and
    float32 accumulation; biases and tanh float32).  The float32 torch modules stay the learner's master copy; call refresh() after an optimiser step.
    Callable like the module: fused(obs) -> (logits (N, row width <= 32) float32, value (N,) float32)."""

    def __init__(self, policy: Q1Policy, env):
        self.policy, self.env = policy, env
        n = env.num_envs
        dev = env.device
        self.logits = torch.empty((n, policy.pi[-1].out_features), dtype=torch.float32, device=dev)
        self.value = torch.empty((n, 1), dtype=torch.float32, device=dev)
        self._w = {}
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Copy the master weights into the kernel's persistent device buffers IN PLACE: a captured hipGraph of the
        sampler holds these addresses, so they must never be re-allocated.  W2 / W3 go into the float16 LDS image the kernel
        copies verbatim: 288 rows x 264 (256 weights + 8 pad), columns permuted (0,2,1,3 groups of four within each 16), the
        W2 rows multiplied by 2 log2(e) before the float16 rounding (tanh's exp2 argument is then the accumulator itself)."""
        for name, net in (("pi", self.policy.pi), ("vf", self.policy.vf)):
            l1, l2, l3 = net[0], net[2], net[4]
            if l3.out_features > 32:
                raise ValueError(f"FusedPolicyForward: {l3.out_features} outputs do not fit the kernel's one 32-row output tile")
            # float16 operands: fine for weights of O(1..10) (the WR checkpoint's largest is 7.4); refuse silently wrong results
            wmax = max(float(l.weight.abs().max()) for l in (l1, l2, l3)) * TANH_PRESCALE
            if not wmax < 6.0e4:
                raise ValueError(f"FusedPolicyForward: |weight| * 2log2(e) = {wmax:.3g} does not fit float16; use the float32 modules")
            dev = l1.weight.device
            if name not in self._w:
                self._w[name] = (torch.empty_like(l1.weight, dtype=torch.float32), torch.empty_like(l1.bias, dtype=torch.float32),
                                 torch.zeros((288, 264), dtype=torch.float16, device=dev),
                                 torch.empty_like(l2.bias, dtype=torch.float32), torch.empty_like(l3.bias, dtype=torch.float32))
                g = torch.arange(256, device=dev)
                grp = (g >> 2) & 3
                self._perm = (g & ~0xC) | ((((grp & 1) << 1) | (grp >> 1)) << 2)      # dest column -> source column
            w1, b1, img, b2, b3 = self._w[name]
            w1.copy_(l1.weight); b1.copy_(l1.bias); b2.copy_(l2.bias); b3.copy_(l3.bias)
            img[:256, :256].copy_(l2.weight[:, self._perm] * TANH_PRESCALE)     # the kernel's exp2 argument is the accumulator itself
            img[256:256 + l3.out_features, :256].copy_(l3.weight[:, self._perm])

    def _mlp(self, name, out):
        from . import _lib
        w1, b1, img, b2, b3 = self._w[name]
        return _lib.Q1Mlp(w1.data_ptr(), b1.data_ptr(), img.data_ptr(), b2.data_ptr(), b3.data_ptr(), out.data_ptr(), out.shape[1])

    def forward_into(self, obs, logits_out, value_out):
        """Both networks in one launch, written straight into caller-owned buffers (the sampler's trajectory rows):
        logits_out (N, 10) float32, value_out (N,) float32, both contiguous."""
        n = self.env.num_envs
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape == (n, 6)
        assert logits_out.dtype == torch.float32 and logits_out.is_contiguous() and logits_out.shape == self.logits.shape
        assert value_out.dtype == torch.float32 and value_out.is_contiguous() and value_out.numel() == n
        self.env._dev.policy_value_forward_dev(obs.data_ptr(), self._mlp("pi", logits_out), self._mlp("vf", value_out.view(n, 1)))

    def __call__(self, obs, separate_launches=False):
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape == (self.env.num_envs, 6)
        d = self.env._dev
        if separate_launches:                  # one q1env_policy_forward per network (same bits; kept for tests / measurement)
            for name, out in (("pi", self.logits), ("vf", self.value)):
                w1, b1, img, b2, b3 = self._w[name]
                d.policy_forward_dev(obs.data_ptr(), w1.data_ptr(), b1.data_ptr(), img.data_ptr(), b2.data_ptr(), b3.data_ptr(),
                                     out.shape[1], out.data_ptr())
        else:                                  # both networks in one launch, half of the CUs each
            d.policy_value_forward_dev(obs.data_ptr(), self._mlp("pi", self.logits), self._mlp("vf", self.value))
        return self.logits, self.value[:, 0]

    def parameters(self):
        return self.policy.parameters()
