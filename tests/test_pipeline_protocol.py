"""CPU: the readme's Processor / Pipeline protocol, PoseTransformer against the reference's own
src/transform.py outputs (golden), OBJ round trip."""
import numpy as np
import pytest

from foundationpose_amd import synthetic as S
from foundationpose_amd.mesh_io import load_intrinsics, load_obj, save_obj
from foundationpose_amd.pipeline import Pipeline, PipelineData, PoseTransformer, Processor


def test_pose_transformer_matches_reference(golden):
  for M, a, b in zip(golden['pt_mats'], golden['pt_inch_deg'], golden['pt_m_rad']):
    np.testing.assert_allclose(PoseTransformer().transform_pose(M), a, rtol=0, atol=1e-12)
    np.testing.assert_allclose(PoseTransformer(to_inches=False, to_degrees=False).transform_pose(M), b, rtol=0, atol=1e-12)


class _Add(Processor):
  def __init__(self, k): self.k = k
  def process(self, data):
    data.trace = getattr(data, 'trace', []) + [self.k]
    return data


class _Boom(Processor):
  def process(self, data):
    raise ValueError('boom')


def test_pipeline_order_and_error_policy():
  out = Pipeline('p').add_processor(_Add(1)).add_processor(_Boom()).add_processor(_Add(2)).run(PipelineData())
  assert out.trace == [1, 2] and [n for n, _ in out.errors] == ['_Boom']
  out = Pipeline('p', stop_on_error=True).add_processor(_Add(1)).add_processor(_Boom()).add_processor(_Add(2)).run()
  assert out.trace == [1] and len(out.errors) == 1
  with pytest.raises(TypeError):
    Pipeline().add_processor(object())
  d = PipelineData(pose=np.eye(4))
  assert Pipeline().add_processor(PoseTransformer()).run(d).pose_6d == (0.0, 0.0, 0.0, 0.0, -0.0, 0.0)


def test_obj_roundtrip_and_intrinsics(tmp_path):
  mesh = S.make_mustard_mesh(seed=3, n_theta=12, n_z=8)
  p = tmp_path / 'm.obj'
  save_obj(mesh, str(p))
  back = load_obj(str(p))
  np.testing.assert_allclose(back.vertices, mesh.vertices, rtol=1e-8)
  np.testing.assert_array_equal(back.faces, mesh.faces)
  assert np.abs(back.visual.vertex_colors[:, :3].astype(int) - mesh.visual.vertex_colors[:, :3].astype(int)).max() <= 1
  np.testing.assert_allclose(back.vertex_normals, mesh.vertex_normals, atol=1e-6)
  (tmp_path / 'q.obj').write_text('v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\nf -4/1 -3/2 -2/3\n')
  quad = load_obj(str(tmp_path / 'q.obj'), texture_image=np.zeros((2, 2, 3), np.uint8))
  assert quad.faces.shape == (3, 3) and quad.visual.uv.shape == (4, 2)
  (tmp_path / 'k.txt').write_text('1066.778 0 312.9869\n0 1067.487 241.3109\n0 0 1\n')
  np.testing.assert_allclose(load_intrinsics(str(tmp_path / 'k.txt')), S.YCB_K)
  (tmp_path / 'bad.obj').write_text('# nothing\n')
  with pytest.raises(ValueError):
    load_obj(str(tmp_path / 'bad.obj'))


def test_batch_pose_data_matches_reference(golden):
  """BatchPoseData.select_by_indices against the reference's own class (tests/golden/gen_golden.py:168-171), plus the
  record's own rules: unknown fields refused, the fused net tensor keeps side A over side B."""
  import torch
  from foundationpose_amd.pose_dataset import BatchPoseData, planar_views
  bp = BatchPoseData(rgbAs=torch.arange(24.).reshape(4, 6), poseA=torch.arange(8.).reshape(4, 2))
  sel = bp.select_by_indices(torch.tensor([2, 0]))
  np.testing.assert_array_equal(sel.rgbAs.numpy(), golden['bpd_rgbAs'])
  np.testing.assert_array_equal(sel.poseA.numpy(), golden['bpd_poseA'])
  assert sel.rgbBs is None and sel.Ks is None and len(sel) == 2 and len(bp) == 4
  assert bp.pin_memory() is bp
  with pytest.raises(TypeError):
    BatchPoseData(rgbs=torch.zeros(1))
  net = torch.arange(6 * 2 * 2 * 8, dtype=torch.float16).reshape(6, 2, 2, 8)      # 3 hypotheses
  bp = BatchPoseData(poseA=torch.arange(3.).reshape(3, 1), net_input=net)
  sel = bp.select_by_indices([2, 1])
  assert len(sel) == 2
  np.testing.assert_array_equal(sel.net_input.numpy(), net[[2, 1, 5, 4]].numpy())
  rgbA, xyzA, rgbB, xyzB = planar_views(net)
  assert rgbA.shape == (3, 3, 2, 2) and rgbA.dtype == torch.float32
  np.testing.assert_array_equal(xyzB[1, 2].numpy(), net[4, :, :, 5].float().numpy())


def test_shared_translation_detection_is_host_only():
  """PoseRefinePredictor._shares_translation (the hint behind FP_REFINE_SHARED_TRANSLATION): host arrays - what the reference hands to
  predict(), src/estimater.py:215 - are compared exactly; an explicit True / False is the caller's word; a single hypothesis has nothing to
  share.  (Device tensors are never inspected: that would synchronise; tests/test_gpu_pipeline.py covers them.)"""
  import numpy as np
  import torch
  from foundationpose_amd.predict_pose_refine import PoseRefinePredictor as P
  poses = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1))
  poses[:, :3, 3] = [0.1, -0.2, 0.7]
  assert P._shares_translation(poses, None) and P._shares_translation(torch.from_numpy(poses), None)
  moved = poses.copy()
  moved[3, 0, 3] = np.nextafter(np.float32(0.1), np.float32(1))          # one ulp is not "the same translation"
  assert not P._shares_translation(moved, None)
  assert not P._shares_translation(poses[:1], None)
  assert P._shares_translation(moved, True) and not P._shares_translation(poses, False)
