"""Single-episode evaluation and move-command export ("next" row 4 of SURVEY.md section 8f).

`eval_sim` is the data half of the reference's q1physrl/analyse.py:197-240: it drives a one-env VectorPhysEnv with a
trainer-like object AND, in parallel, a stand-alone ActionDecoder fed with the observed z velocity and the env's time -
which is how the reference derives the (yaw, smove, fmove, jump) move commands that mkdemo.py:47-55 sends to the real
game.  It deliberately goes through the reference-compatible NumPy surface (VectorPhysEnv / ActionDecoder / phys.apply),
so it doubles as an end-to-end exercise of the drop-in boundary.  Plotting (analyse.py:120-148, matplotlib/cv2) is out of scope.
"""
import dataclasses

import numpy as np

from . import env, phys


@dataclasses.dataclass
class EvalSimResult:
    """Field-for-field the reference's EvalSimResult (analyse.py:71-82)."""
    time_delta: float
    player_state: phys.PlayerState
    action: np.ndarray
    obs: np.ndarray
    reward: np.ndarray
    yaw: np.ndarray
    smove: np.ndarray
    fmove: np.ndarray
    jump: np.ndarray

    @property
    def move_angle(self):                                                     # analyse.py:84-86
        return 180. * np.arctan2(self.player_state.vel[:, 1], self.player_state.vel[:, 0]) / np.pi

    @property
    def wish_angle(self):                                                     # analyse.py:88-90
        return self.yaw - (180. * np.arctan2(self.smove, self.fmove) / np.pi)

    @property
    def hypothetical_delta_speeds(self):
        """(360, frames): speed gain of each frame had the wish direction been move_angle + d, d = -180..179
        (analyse.py:92-118); 360 stateless phys.apply launches."""
        rows = []
        move_angle = self.move_angle
        speed_before = np.linalg.norm(self.player_state.vel[:, :2], axis=1)
        for rel in np.arange(-180, 180):
            inputs = phys.Inputs(yaw=move_angle + rel, pitch=np.zeros_like(move_angle), roll=np.zeros_like(move_angle),
                                 fmove=np.full_like(move_angle, 800.), smove=np.zeros_like(move_angle), button2=self.jump,
                                 time_delta=np.full_like(move_angle, 0.014))
            nxt = phys.apply(inputs, self.player_state)
            rows.append(np.linalg.norm(nxt.vel[:, :2], axis=1) - speed_before)
        return np.stack(rows)

    def move_commands(self):
        """What mkdemo._apply_action (mkdemo.py:47-55) would send per frame: yaw in radians, forward, side, buttons."""
        return {"yaw_rad": self.yaw * np.pi / 180, "forward": self.fmove.astype(np.int64), "side": self.smove.astype(np.int64),
                "buttons": np.where(self.jump, 2, 0)}


def eval_sim(trainer, env_config: env.Config, *, device: int = 0) -> EvalSimResult:
    """Run one episode (analyse.py:197-240).  `trainer.compute_action(obs)` returns an RLlib-style action
    (sequence of K key components + the mouse component, each a scalar or a length-1 array)."""
    e = env.VectorPhysEnv(dataclasses.asdict(env_config), device=device)
    o, = e.vector_reset()
    action_decoder = env.ActionDecoder(env_config, device=device)
    action_decoder.vector_reset(e._yaw)
    obs, reward, actions, player_states, yaws, smoves, fmoves, jumps = [], [], [], [], [], [], [], []
    done = False
    while not done:
        a = trainer.compute_action(o)
        (yaw,), (smove,), (fmove,), (jump,) = action_decoder.map([a], o[None, env.Obs.Z_VEL], e._time_remaining)
        player_states.append(e.player_state)
        obs.append(o)
        actions.append(np.array([np.ravel(x)[0] for x in a], dtype=np.float64))
        yaws.append(yaw); smoves.append(smove); fmoves.append(fmove); jumps.append(jump)
        (o,), (r,), (done,), _ = e.vector_step([a])
        reward.append(r)
    e.close()
    return EvalSimResult(time_delta=env_config.time_delta, player_state=phys.PlayerState.concatenate(player_states),
                         action=np.stack(actions), obs=np.stack(obs), reward=np.stack(reward), yaw=np.stack(yaws),
                         smove=np.stack(smoves), fmove=np.stack(fmoves), jump=np.stack(jumps))


class TorchTrainer:
    """Adapter giving a Q1Policy the one method eval_sim / mkdemo need from an RLlib trainer (`compute_action`)."""

    def __init__(self, policy, action_range, deterministic=True, generator=None):
        self.policy, self.action_range, self.deterministic, self.generator = policy, float(action_range), deterministic, generator

    def compute_action(self, obs):
        import torch
        from .policy import Q1PhysActionDist
        p = next(self.policy.parameters())
        with torch.no_grad():
            logits, _ = self.policy(torch.as_tensor(np.asarray(obs, dtype=np.float32)[None, :], device=p.device))
            dist = Q1PhysActionDist(logits.float().cpu(), self.action_range)
            keys, mouse = dist.deterministic_sample() if self.deterministic else dist.sample(self.generator)
        return [int(k) for k in keys[0]] + [np.array([float(mouse[0, 0])], dtype=np.float32)]
