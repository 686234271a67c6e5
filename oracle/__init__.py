"""CPU oracle for the PoseRisk per-frame hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``poserisk_release_amd/`` may import,
call, link or execute anything in this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
there only as the checker / reported baseline, never as the thing shipped.

Each module restates, in numpy / torch-CPU, one piece of the reference's
algorithm and cites the reference file:line it follows (paths are relative to
the reference checkout, ``hygenie1228/PoseRisk_RELEASE``).

Parity status:
  * smpl_ref, coord_ref (Euler part), reba_ref, rula_ref are
    PINNED by golden vectors produced by running the reference's own Python in
    the build container (tests/golden/make_golden.py, fixtures committed);
    aggregate_ref (5 lines of numpy, base.py:263-271) by hand-computed known answers.
  * hmr_ref (SPIN ``models/hmr.py`` + ``utils/geometry.py``), crop_ref (OpenCV
    ``warpAffine``/``getAffineTransform`` fixed-point bilinear warp) and rodrigues_cv
    (OpenCV ``cv2.Rodrigues``) restate third-party code that is NOT in the
    reference tree and is unpinned upstream (``script/install_conda.sh:24``
    clones SPIN's default branch; ``requirements.txt:10`` leaves opencv-python
    unpinned).  The reference holds no test or fixture at those call sites, so
    for these three modules: PARITY UNPINNED.
"""
