"""CPU oracle for the UmeTrack hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``umetrack_amd/`` imports
it, and the product path has no CPU fallback: without the HIP library it
raises.

Pinning status (details in DESIGN.md section "Oracle"):

* model graph (backbone, FTL/fusion, temporal, skeleton encoder, regressors,
  Procrustes decode, wrist recovery): pinned against golden vectors produced by
  importing the reference's ``lib.models`` in the build container
  (tests/golden/make_goldens.py -> tests/golden/*.npz).
* warp coordinate maps, crop cameras, float bilinear resample, metrics: pinned
  the same way (``lib.common.camera/crop``, ``lib.batched_dataset``,
  ``lib.common.metric_utils`` / ``load_eval``).
* ``cv2.remap`` fixed-point sampling and ``pytorch3d.transforms.so3_exp_map``:
  third-party arithmetic absent from /root/reference and from this image ->
  **parity unpinned** at those two boundaries; the published algorithms are
  restated and checked by properties only.
"""
