"""Text / CSV / OBJ outputs of `main/run.py`, byte-compatible with the reference's writers.
Pure host I/O on the path's results.

  * `<title>_result.txt`            lib/core/base.py:161-165, 178-182
  * `<TITLE>_score_log.csv`, `<TITLE>_eval_pose_log.csv`   base.py:351-397
  * `pose_log.csv`                  base.py:329-349
  * `<TITLE>_score.png`             base.py:254-262 (the plot inside post_processing)
  * `<TITLE>_video.mp4`             base.py:284-327 (+ vis_utils.py:278-294 visualize_box); drawn with OpenCV, so
                                    only where `cv2` is importable
  * `pose_to_str`                   lib/utils/vis_utils.py:9-16
  * `smpl_model.obj`                vis_utils.py:238-245
  * `joint_3d.png`                  vis_utils.py:181-235 (vis_3d_pose, joint set 'smpl'), base.py:273-282
"""
import csv
import os.path as osp

import numpy as np


def pose_to_str(poses):
    """f64[N,24,3] -> N lists of '(x, y, z)' strings with 3 decimals (vis_utils.py:9-16)."""
    return [[f"({p[0]:.3f}, {p[1]:.3f}, {p[2]:.3f})" for p in frame] for frame in poses]


def write_result_txt(output_path, title, final_score, action_level, action_name):
    """`reba_result.txt` / `rula_result.txt`.  The reference's f-string has a line continuation inside the
    literal, so the REBA file carries 20 spaces before the MAX line and one trailing space; RULA has the
    20 spaces but no trailing space (base.py:162-163 vs 179-180)."""
    avg, s50, s10, smax, smode = final_score
    tail = " " if title.upper() == "REBA" else ""
    data = (f"AVG Score: {avg} \n%50 Score: {s50} \n%10 Score: {s10} " + " " * 20 +
            f"\nMAX Score: {smax} \nMODE Score: {smode} \nAction level: {action_level} \nAction: {action_name}{tail}")
    with open(osp.join(output_path, f"{title.lower()}_result.txt"), "w") as f:
        f.write(data)
    return data


def _frame_rows(timestamp):
    """base.py iterates range(timestamp[0], timestamp[-1]) and fills only the frames in timestamp[1]."""
    first, frames, last = timestamp
    frames = np.asarray(frames)
    for i in range(first, last):
        hit = np.where(frames == i)[0]
        yield i, (int(hit[0]) if hit.size else None)


def save_score_csv(output_path, title, timestamp, scores, joint_names, logs, pose_logs):
    """`<title>_score_log.csv` and `<title>_eval_pose_log.csv` (base.py:351-397)."""
    with open(osp.join(output_path, title + "_score_log.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Frame", "Final_score", "Joint Score"] + list(joint_names))
        for i, idx in _frame_rows(timestamp):
            row = [i]
            if idx is not None:
                row += [str(scores[idx]), ""] + [str(logs[idx][j]) for j in range(len(joint_names))]
            wr.writerow(row)
    names = list(pose_logs[0].keys())
    with open(osp.join(output_path, title + "_eval_pose_log.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Frame", ""] + names)
        for i, idx in _frame_rows(timestamp):
            row = [i]
            if idx is not None:
                row += [""] + [str(pose_logs[idx][n]) for n in names]
            wr.writerow(row)


def save_pose_log_csv(output_path, timestamp, pose_str, debug_joints, joints_name_upper):
    """`pose_log.csv` (base.py:329-349)."""
    with open(osp.join(output_path, "pose_log.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Frame", "Joint Pose"] + list(debug_joints))
        for i, idx in _frame_rows(timestamp):
            row = [i]
            if idx is not None:
                row += [""] + [str(pose_str[idx][joints_name_upper.index(j.upper())]) for j in debug_joints]
            wr.writerow(row)


def save_obj(v, f=None, file_name=""):
    """Wavefront OBJ of a mesh: 'v x y z' per vertex, 'f a/a b/b c/c' per face, 1-based (vis_utils.py:238-245)."""
    with open(file_name, "w") as fh:
        for p in v:
            fh.write("v " + str(p[0]) + " " + str(p[1]) + " " + str(p[2]) + "\n")
        for t in (f if f is not None else []):
            fh.write("f " + str(t[0] + 1) + "/" + str(t[0] + 1) + " " + str(t[1] + 1) + "/" + str(t[1] + 1) + " "
                     + str(t[2] + 1) + "/" + str(t[2] + 1) + "\n")


def save_score_plot(output_path, title, timestamp, scores):
    """`<title>_score.png`: score over frame index, x range = the whole video (base.py:254-262).  Drawn on a
    private Agg figure of matplotlib's default size, so the caller's pyplot state is left alone (the reference
    draws on the global figure and clears it)."""
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    fig = Figure()
    FigureCanvasAgg(fig)
    ax = fig.add_subplot(111)
    ax.set_title(title + ' Score')
    ax.set_xlim([timestamp[0], timestamp[2]])
    ax.set_xlabel('frames')
    ax.set_ylabel('score')
    ax.plot(np.asarray(timestamp[1]), np.asarray(scores))
    path = osp.join(output_path, title + '_score.png')
    fig.savefig(path)
    return path


def draw_track_box(cv2, img, box):
    """vis_utils.py:278-294: the (cx, cy, w, h) box as four green lines of thickness 2 on a copy of the frame."""
    img = img.copy()
    x_min, y_min = int(box[0]) - int(box[2]) // 2, int(box[1]) - int(box[3]) // 2
    x_max, y_max = int(box[0]) + int(box[2]) // 2, int(box[1]) + int(box[3]) // 2
    corners = ((x_min, y_min), (x_min, y_max), (x_max, y_min), (x_max, y_max))
    for a, b in ((0, 1), (0, 2), (1, 3), (2, 3)):
        img = cv2.line(img, corners[a], corners[b], (0, 255, 0), 2)
    return img


def write_annotated_video(output_path, title, frames_bgr, bboxes, timestamp, fps, scores, joint_names, logs, cv2=None):
    """`<title>_video.mp4` (base.py:284-327): every decoded frame resized to width 720 beside a 280-pixel panel with the
    frame number, and -- on frames of the target track -- the track's box, the score and the per-part log of frame
    `idx // 2 * 2` (the reference shows every other frame's numbers, Q21), else "Not detected target".
    frames_bgr: sequence of uint8[H,W,3] in OpenCV's channel order (what cv2.imread returns).  Needs OpenCV for the
    drawing and the mp4 container; returns the path, or None when `cv2` is not importable."""
    if cv2 is None:
        try:
            import cv2
        except ImportError:
            return None
    if not all(hasattr(cv2, n) for n in ("VideoWriter", "putText", "line", "resize")):
        return None
    height, width = frames_bgr[0].shape[:2]
    resize_w = 720
    resize_h = int(height * resize_w / width)
    canvas_w, canvas_h = resize_w + 280, resize_h
    path = osp.join(output_path, title + '_video.mp4')
    writer = cv2.VideoWriter(path, 0x7634706d, fps, (canvas_w, canvas_h))
    font, white = cv2.FONT_HERSHEY_SIMPLEX, (255, 255, 255)
    track_frames = np.asarray(timestamp[1])
    x0 = resize_w + 15
    for i, img in enumerate(frames_bgr):
        canvas = np.zeros((canvas_h, canvas_w, 3))
        cv2.putText(canvas, "frame: " + str(i), (x0, canvas_h - 14), font, 0.5, white, 1, cv2.LINE_AA)
        hit = np.where(track_frames == i)[0]
        if hit.size:
            idx = int(hit[0]) // 2 * 2
            img = draw_track_box(cv2, img, bboxes[idx])
            cv2.putText(canvas, title + " Score: " + str(scores[idx]), (x0, 35), font, 0.7, (0, 255, 0), 1, cv2.LINE_AA)
            cv2.putText(canvas, "- Score per Joints ", (x0, 122), font, 0.6, white, 1, cv2.LINE_AA)
            for j, joint in enumerate(joint_names):
                cv2.putText(canvas, joint + ": " + str(logs[idx][j]), (x0, 153 + 24 * j), font, 0.5, white, 1, cv2.LINE_AA)
        else:
            cv2.putText(canvas, "Not detected target", (x0, canvas_h - 65), font, 0.6, white, 1, cv2.LINE_AA)
        canvas[:resize_h, :resize_w, :] = cv2.resize(img, (resize_w, resize_h), interpolation=cv2.INTER_AREA)
        writer.write(np.uint8(canvas))
    writer.release()
    return path


def save_joint_3d_plot(joint_cam, skeleton, file_path, frame=0):
    """`joint_3d.png` of the --debug_frame branch (base.py:273-282 -> vis_utils.py:181-235 with joint set 'smpl'):
    the 24 root-relative joints (mm) as a 3-D skeleton, x / z / -y axes, right-side joints green, 5 x 3.75 inch figure,
    limits +-800 mm made equal per axis.  Drawn on a private Agg figure."""
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    r_joints = (2, 5, 8, 11, 14, 17, 19, 21, 23)
    kps = np.asarray(joint_cam)
    fig = Figure()
    FigureCanvasAgg(fig)
    ax = fig.add_subplot(111, projection='3d')
    fig.set_size_inches(5, 3.75)
    for i1, i2 in skeleton:
        ax.plot(np.array([kps[i1, 0], kps[i2, 0]]), np.array([kps[i1, 2], kps[i2, 2]]), -np.array([kps[i1, 1], kps[i2, 1]]),
                c='r', linewidth=1)
        for i in (i1, i2):
            ax.scatter(kps[i, 0], kps[i, 2], -kps[i, 1], c='g' if i in r_joints else 'b', marker='o')
    ax.set_xlabel('X axis')
    ax.set_ylabel('Z axis')
    ax.set_zlabel('Y axis')
    ax.set_xlim3d(-800, 800)
    ax.set_ylim3d(-800, 800)
    ax.set_zlim3d(-800, 800)
    ax.set_title(f'3D Skeleton - frame: {frame}')
    extents = np.array([getattr(ax, 'get_{}lim'.format(dim))() for dim in 'xyz'])      # axisEqual3D
    radius = max(abs(extents[:, 1] - extents[:, 0])) / 2
    for ctr, dim in zip(np.mean(extents, axis=1), 'xyz'):
        getattr(ax, 'set_{}lim'.format(dim))(ctr - radius, ctr + radius)
    fig.savefig(file_path)
    return file_path
