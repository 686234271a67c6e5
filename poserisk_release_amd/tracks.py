"""Tracker hand-off: what `DataProcessing.__call__` does with multi_person_tracker's output
before the crops are cut (lib/core/base.py:47-74, lib/utils/funcs_utils.py:55-64).  Pure host logic on small
arrays; MPT itself (YOLOv3 + SORT) stays outside this package.

tracking_results: {person_id: {'bbox': f32[n,4] (cx,cy,w,h), 'frames': int[n]}}  (MPT output_format='dict').
"""
import numpy as np

MIN_FRAME_RATIO = 0.33   # lib/core/config.py:33
MIN_FRAME_CAP = 1000     # base.py:55


def filter_tracks(tracking_results, n_frames, min_frame_ratio=MIN_FRAME_RATIO):
    """base.py:53-69: keep tracks seen in >= min(ratio*n_frames, 1000) frames; if none qualifies keep all.
    Returns a list in the dict's iteration order (what `select_target_id` indexes)."""
    need = n_frames * min_frame_ratio
    if need > MIN_FRAME_CAP:
        need = MIN_FRAME_CAP
    kept = [t for t in tracking_results.values() if np.asarray(t['frames']).shape[0] >= need]
    return kept if kept else list(tracking_results.values())


def select_target_id(results):
    """funcs_utils.py:55-64: index of the track with the largest mean bbox area (first one on ties)."""
    areas = [(np.asarray(r['bbox'])[:, 2] * np.asarray(r['bbox'])[:, 3]).mean() for r in results]
    return int(np.argmax(np.array(areas)))


def target_track(tracking_results, n_frames, min_frame_ratio=MIN_FRAME_RATIO):
    """-> (bbox f32[n,4], frames int[n]) of the person the reference scores (base.py:71-73)."""
    kept = filter_tracks(tracking_results, n_frames, min_frame_ratio)
    t = kept[select_target_id(kept)]
    return np.asarray(t['bbox'], np.float32), np.asarray(t['frames'])
