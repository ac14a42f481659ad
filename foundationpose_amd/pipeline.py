"""The caller-side protocol the reference's readme documents (readme.md:122-185) but does not ship
(SURVEY.md D1): `Processor.process(data) -> data`, optional `visualize(data)`,
`Pipeline(name).add_processor(p)`, `pipeline.run(PipelineData()) -> data` with `data.pose_6d`,
`FoundationPoseEstimator(mesh_file, intrinsics_file)` and `PoseTransformer` (src/transform.py:4-67).
"""
import logging
import math

import numpy as np

METERS_TO_INCHES = 39.3701     # src/transform.py:47


class PipelineData:
  """Attribute bag passed from processor to processor (`rgb, depth, mask, K, pose, pose_6d, errors`)."""

  def __init__(self, **kw):
    self.rgb = self.depth = self.mask = self.K = None
    self.pose = None          # 4x4 ob_in_cam of the original (un-centred) mesh
    self.pose_6d = None       # (x, y, z, roll, pitch, yaw)
    self.errors = []
    self.__dict__.update(kw)


class Processor:
  def process(self, data):
    raise NotImplementedError

  def visualize(self, data):
    return None


class Pipeline:
  def __init__(self, name='Pipeline', stop_on_error=False, visualize=False):
    self.name, self.stop_on_error, self.visualize = name, stop_on_error, visualize
    self.processors = []

  def add_processor(self, processor):
    if not callable(getattr(processor, 'process', None)):
      raise TypeError('a processor needs a process(data) method')
    self.processors.append(processor)
    return self

  def run(self, data=None):
    """Runs the processors in order.  A failing processor is recorded in `data.errors`; with
    `stop_on_error` (readme.md:120) the run ends there, otherwise the next processor still runs."""
    data = PipelineData() if data is None else data
    for proc in self.processors:
      try:
        out = proc.process(data)
        data = data if out is None else out
        if self.visualize and hasattr(proc, 'visualize'):
          proc.visualize(data)
      except Exception as exc:          # noqa: BLE001 - the protocol isolates processors from each other
        logging.error(f'[{self.name}] {type(proc).__name__} failed: {exc}')
        data.errors.append((type(proc).__name__, exc))
        if self.stop_on_error:
          break
    return data


def rotation_to_roll_pitch_yaw(R):
  """ZYX Euler angles (radians) with the reference's gimbal-lock convention (src/transform.py:52-67):
  |R[2,0]| > 0.9999 -> yaw = 0, pitch = +-pi/2, roll from atan2(R[0,1], R[1,1])."""
  s = R[2, 0]
  if abs(s) > 0.9999:
    roll = math.atan2(R[0, 1], R[1, 1])
    return (roll, math.pi / 2, 0) if s < 0 else (-roll, -math.pi / 2, 0)
  pitch = -math.asin(s)
  c = math.cos(pitch)
  return math.atan2(R[2, 1] / c, R[2, 2] / c), pitch, math.atan2(R[1, 0] / c, R[0, 0] / c)


class PoseTransformer(Processor):
  """4x4 pose -> (x, y, z, roll, pitch, yaw); inches / degrees by default (src/transform.py:4-50)."""

  def __init__(self, to_inches=True, to_degrees=True):
    self.to_inches, self.to_degrees = to_inches, to_degrees

  def transform_pose(self, center_pose):
    pose = np.asarray(center_pose)
    xyz = [float(v) * (METERS_TO_INCHES if self.to_inches else 1.0) for v in pose[:3, 3]]
    rpy = [math.degrees(a) if self.to_degrees else a for a in rotation_to_roll_pitch_yaw(pose[:3, :3])]
    return (*xyz, *rpy)

  def process(self, data):
    data.pose_6d = self.transform_pose(data.pose)
    return data


class FoundationPoseEstimator(Processor):
  """Wraps foundationpose_amd.estimater.FoundationPose as a pipeline stage (what main.py:34-79 does by hand):
  first frame -> register(), later frames -> track_one()."""

  def __init__(self, mesh_file=None, intrinsics_file=None, mesh=None, K=None, est_refine_iter=5, track_refine_iter=2,
               scorer=None, refiner=None, debug=0):
    from .mesh_io import load_intrinsics, load_obj
    self.mesh = load_obj(mesh_file) if mesh is None else mesh
    self.K = np.asarray(K, dtype=np.float64) if K is not None else (load_intrinsics(intrinsics_file) if intrinsics_file else None)
    self.est_refine_iter, self.track_refine_iter = est_refine_iter, track_refine_iter
    self._kw = dict(scorer=scorer, refiner=refiner, debug=debug)
    self.est = None

  def process(self, data):
    from .estimater import FoundationPose
    K = self.K if data.K is None else np.asarray(data.K, dtype=np.float64)
    if self.est is None:
      self.est = FoundationPose(model_pts=self.mesh.vertices, model_normals=self.mesh.vertex_normals, mesh=self.mesh, **self._kw)
    if self.est.pose_last is None or data.mask is not None:
      data.pose = self.est.register(K=K, rgb=data.rgb, depth=data.depth, ob_mask=np.asarray(data.mask).astype(bool),
                                    iteration=self.est_refine_iter)
    else:
      data.pose = self.est.track_one(rgb=data.rgb, depth=data.depth, K=K, iteration=self.track_refine_iter)
    return data
