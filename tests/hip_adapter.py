"""Gives BatchedMobiEnv the OracleEnv call surface (numpy in / numpy out) so replay.py drives both."""
import numpy as np


class HipEnvAdapter:
    def __init__(self, env):
        self.env = env

    def _np(self):
        import torch

        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in self.env.out.items()}

    def init(self, **kw):
        self.env.init(**kw)

    def warmup(self, theta_u=None, group_u=None):
        self.env.warmup(1, theta_u=theta_u, group_u=group_u)

    def reset(self, mask=None, **inj):
        self.env.reset(mask=mask, **inj)
        return self._np()

    def step(self, actions, **inj):
        import torch

        self.env.step(torch.as_tensor(np.asarray(actions, np.int64)), **inj)
        return self._np()

    @property
    def s(self):
        return self.env.state_fields()

    def reset_trace(self, ue_xy, mask=None, fading=None):
        self.env.reset_trace(ue_xy, mask=mask, fading=fading)
        return self._np()

    def step_trace(self, actions, ue_xy, fading=None):
        import torch

        self.env.step_trace(torch.as_tensor(np.asarray(actions, np.int64)), ue_xy, fading=fading)
        return self._np()
