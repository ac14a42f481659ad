"""Multi-hypothesis tracking (BASELINE.json configs[4]: "track_one() with 64-hypothesis refine per frame").

The reference's track_one refines exactly one pose (src/estimater.py:263) and never scores; the 64-hypothesis mode is this
build's extension on top of the same two predictors: the previous pose plus n-1 fixed, seeded perturbations of it are
refined together, scored by ScoreNet, and the best one becomes the new pose (FoundationPose.track_multi).  The
perturbation set is a constant of (n, seed, sigmas), so the CPU oracle and the HIP path start from identical hypotheses.
"""
import functools

import numpy as np
import torch


@functools.lru_cache(maxsize=8)
def perturbation_set(n, trans_sigma=0.01, rot_sigma_deg=5.0, seed=0):
  """(n,4,4) float32: identity first, then n-1 small rigid motions [dR | dt] with dt ~ N(0, trans_sigma) per axis and
  dR = exp(hat(w)), w ~ N(0, rot_sigma) per axis."""
  rs = np.random.RandomState(seed)
  out = np.tile(np.eye(4), (n, 1, 1))
  dt = rs.randn(n, 3) * trans_sigma
  w = rs.randn(n, 3) * np.deg2rad(rot_sigma_deg)
  for i in range(1, n):
    th = np.linalg.norm(w[i])
    Kx = np.array([[0, -w[i, 2], w[i, 1]], [w[i, 2], 0, -w[i, 0]], [-w[i, 1], w[i, 0], 0]])
    out[i, :3, :3] = np.eye(3) + (np.sin(th) / max(th, 1e-12)) * Kx + ((1 - np.cos(th)) / max(th * th, 1e-12)) * (Kx @ Kx)
    out[i, :3, 3] = dt[i]
  return out.astype(np.float32)


@functools.lru_cache(maxsize=8)
def _device_set(n, trans_sigma, rot_sigma_deg, seed, device):
  """The perturbation set on `device`, uploaded once (no host copy per frame: a frame can be captured in a hipGraph)."""
  return torch.as_tensor(perturbation_set(n, trans_sigma, rot_sigma_deg, seed), device=device)


def tracking_hypotheses(pose, n, trans_sigma=0.01, rot_sigma_deg=5.0, seed=0):
  """pose (4,4) tensor -> (n,4,4) hypotheses on its device: R_i = dR_i R, t_i = t + dt_i (the egocentric update form of
  the refiner, src/Utils.py:848-855); hypothesis 0 is `pose` itself."""
  P = _device_set(int(n), float(trans_sigma), float(rot_sigma_deg), int(seed), str(pose.device))
  pose = pose.reshape(4, 4).to(torch.float)
  hyp = torch.eye(4, dtype=torch.float, device=pose.device).repeat(n, 1, 1)
  hyp[:, :3, :3] = P[:, :3, :3] @ pose[:3, :3]
  hyp[:, :3, 3] = pose[:3, 3] + P[:, :3, 3]
  return hyp
