"""CPU oracle for the FoundationPose render-and-compare hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``foundationpose_amd/`` may import this
package: it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check (and time) the HIP path against a
plain CPU restatement of the reference algorithm.

Every function cites the reference file:line (relative to the upstream repo
SavaRobotics/FoundationPose) whose arithmetic it restates.

Pinning status (see DESIGN.md "Oracle"):
  * networks (RefineNet / ScoreNetMultiPair), projection matrix, depth2xyzmap,
    pose algebra, guess_translation: PINNED against the reference's own Python
    modules imported in the build container (tests/golden/gen_golden.py ->
    tests/golden/*.npz).
  * rasteriser (nvdiffrast), warp_perspective (kornia 0.7.2), so3_exp_map
    (pytorch3d), icosphere order (trimesh), Warp depth kernels: the third-party
    packages are absent from the reference checkout and from this image ->
    restated from their published behaviour; PARITY UNPINNED for those pieces.
"""
