"""Predictor configuration: the keys the reference reads from weights/<run>/config.yml through
OmegaConf (learning/training/predict_pose_refine.py:97-131, predict_score.py:120-143)."""
import os

import torch
import yaml


class Cfg(dict):
  """dict with attribute access (the subset of OmegaConf's DictConfig behaviour the hot path uses)."""
  def __getattr__(self, k):
    try:
      return self[k]
    except KeyError:
      raise AttributeError(k)

  def __setattr__(self, k, v):
    self[k] = v


# what the released config.yml files were generated from (learning/training/training_config.py:20-91)
# plus the upstream normalisers - used for synthetic-weight runs.
REFINE_DEFAULT = dict(input_resize=[160, 160], c_in=6, use_BN=True, normalize_xyz=True, use_normal=False, crop_ratio=1.2,
                      trans_rep='tracknet', rot_rep='axis_angle', trans_normalizer=[0.02, 0.02, 0.05], rot_normalizer=0.3490659)
SCORE_DEFAULT = dict(input_resize=[160, 160], c_in=6, use_BN=True, normalize_xyz=True, use_normal=False, crop_ratio=1.1)


def weights_root_default():
  env = os.environ.get('FOUNDATIONPOSE_WEIGHTS')
  if env:
    return env
  return os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'weights')


def load_run_dir(run_name, weights_root=None):
  """weights/<run_name>/model_best.pth (+ ['model'] unwrap) and config.yml, loaded with loaders that
  execute nothing from the files (torch.load(weights_only=True), yaml.safe_load)."""
  root = weights_root or weights_root_default()
  ckpt_dir = os.path.join(root, run_name, 'model_best.pth')
  cfg_path = os.path.join(root, run_name, 'config.yml')
  if not os.path.exists(ckpt_dir):
    raise FileNotFoundError(f'{ckpt_dir} not found: pass state_dict=/cfg= or set FOUNDATIONPOSE_WEIGHTS to the directory '
                            f'holding the upstream weight folders')
  with open(cfg_path) as f:
    cfg = yaml.safe_load(f) or {}
  ckpt = torch.load(ckpt_dir, map_location='cpu', weights_only=True)
  if 'model' in ckpt:
    ckpt = ckpt['model']
  cfg['ckpt_dir'] = ckpt_dir
  return ckpt, cfg
