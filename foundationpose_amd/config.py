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


def check_network_cfg(cfg, state_dict, stem_key, ref_file):
  """The configurations the HIP networks are built for, checked BEFORE anything touches the device, each refusal naming the
  reference lines whose behaviour would be needed (learning/training/<ref_file>).  Everything else the reference's config
  offers (use_BN, normalize_xyz, use_normal, crop_ratio, trans_rep, rot_rep, zfar) is implemented."""
  size = tuple(int(x) for x in cfg['input_resize'])
  if size != (160, 160):
    raise NotImplementedError(
      f"input_resize={list(size)}: the HIP trunk is laid out for 160x160 crops (80/40/20-pixel feature maps, 400 tokens = "
      f"pos_embed max_len, network_modules.py:115-137).  The reference renders at input_resize ({ref_file}:38-49, output_size=cfg['input_resize']) "
      f"and only warps side A again when the sizes differ ({ref_file}:64-71); no released model uses another size")
  # use_normal=True is accepted: the refiner then carries normalAs / normalBs in its BatchPoseData (predict_pose_refine.crop_normals), the
  # scorer ignores the flag as the reference's does; the networks read 6 channels either way (the c_in check below)
  stem = state_dict.get(stem_key) if hasattr(state_dict, 'get') else None
  if stem is not None:
    c_sd = int(stem.shape[1])
    if int(cfg['c_in']) != c_sd:
      raise ValueError(f"config c_in={cfg['c_in']} but '{stem_key}' has {c_sd} input channels: the reference builds the model with "
                       f"c_in=cfg['c_in'] and load_state_dict would raise on the size mismatch ({ref_file}: model = ...(c_in=self.cfg['c_in']); "
                       f"note the back-compat default c_in=4 when the key is absent)")
    if c_sd != 6:
      raise ValueError(f"c_in={c_sd}: predict() feeds 6 channels, cat([rgb, xyz_map], 1) ({ref_file}: A = torch.cat([rgbAs, xyz_mapAs], dim=1)), "
                       f"so a stem with {c_sd} input channels raises in the reference's first convolution; only c_in=6 checkpoints can run")
