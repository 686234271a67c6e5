"""ctypes binding of libposerisk_hip.so (the C ABI declared in include/poserisk_hip.h).

The product path has no CPU fallback: if the library is missing or fails to load, importing
any compute entry point raises.  `python -m poserisk_release_amd.build` (or
`__graft_entry__.build()`) produces the library in-tree.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The shipped library.  POSERISK_LIB_PATH selects another build for A/B and ablation runs (scripts/ab_libs.sh builds them
# beside the tree with `python -m poserisk_release_amd.build --out ...`): nothing ever overwrites the shipped file, and
# pr_build_info() -- printed by bench.py as `library` -- says which build a record came from.
LIB_PATH = os.environ.get("POSERISK_LIB_PATH") or os.path.join(HERE, "libposerisk_hip.so")

ABI_VERSION = 10


class PoseRiskHipError(RuntimeError):
    pass


class RebaInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("legs_bilateral", "sitting", "load_force", "arm_supported_l",
                                          "arm_supported_r", "coupling", "activity")]


class RulaInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("arm_supported_l", "arm_supported_r", "a_muscle_l", "a_muscle_r",
                                          "a_load_l", "a_load_r", "legs_bilateral", "b_muscle", "b_load")]


class FramesOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rotmat", "betas", "cam", "axis_angle", "euler_deg", "joint_cam",
                                           "verts", "reba", "rula", "status")]


_P = C.c_void_p
_I = C.c_int
# name -> (restype, argtypes); every symbol include/poserisk_hip.h declares
SIGNATURES = {
    "pr_last_error": (C.c_char_p, []),
    "pr_abi_version": (_I, []),
    "pr_build_info": (C.c_char_p, []),
    "pr_declare_stream": (_I, [_P, _I]),
    "pr_hmr_weight_floats": (C.c_size_t, []),
    "pr_hmr_create": (_I, [_I, _P, C.c_size_t, _I, _I, _I, C.POINTER(_P)]),
    "pr_hmr_destroy": (_I, [_P]),
    "pr_hmr_forward": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "pr_hmr_set_streams": (_I, [_P, _I]),
    "pr_hmr_set_concurrency": (_I, [_P, _I]),
    "pr_hmr_profile_enable": (_I, [_P, _I]),
    "pr_hmr_profile_read": (_I, [_P, _P, _P, _P, _P, _I]),
    "pr_hmr_num_conv_layers": (_I, []),
    "pr_hmr_conv_form": (_I, [_P]),
    "pr_hmr_plan_counts": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "pr_conv_num_tile_cfgs": (_I, []),
    "pr_conv2d_nhwc": (_I, [_I, _P, _P, _P, _P, _P] + [_I] * 14 + [_P, _P]),
    "pr_conv1x1_dual_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P] + [_I] * 12 + [_P]),
    "pr_conv3x3_conv1x1_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P, _P] + [_I] * 7 + [_P]),
    "pr_bottleneck_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "pr_bottleneck128_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "pr_bottleneck256_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "pr_stem_pool_nhwc": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "pr_stem_pool_f32_nhwc": (_I, [_I, _P, _P, _P, _P, _I, _I, _P, _P]),
    "pr_crop_frames": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, C.c_float, _P, _P, _P]),
    "pr_rot6d_to_rotmat": (_I, [_P, _I, _P, _P]),
    "pr_pose_to_euler": (_I, [_P, _I, _P, _P, _P, _P]),
    "pr_axis_angle_to_euler": (_I, [_P, _I, _P, _P, _P]),
    "pr_smpl_create": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_P)]),
    "pr_smpl_destroy": (_I, [_P]),
    "pr_smpl_forward": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "pr_smpl_joint_cam": (_I, [_P, _P, _I, _P, _P, _P]),
    "pr_reba": (_I, [_P, _I, C.POINTER(RebaInfo), _P, _P]),
    "pr_rula": (_I, [_P, _I, C.POINTER(RulaInfo), _P, _P]),
    "pr_frames_forward": (_I, [_P, _P, _P, _I, C.POINTER(RebaInfo), C.POINTER(RulaInfo),
                               C.POINTER(FramesOut), _P]),
}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises PoseRiskHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PoseRiskHipError(
            f"{LIB_PATH} is missing: build it with `python -m poserisk_release_amd.build` "
            "(there is no CPU fallback for the hot path)")
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64): it has to be in the process BEFORE this library resolves the
    # same soname, or the two copies each try to own the device and the second one sees none ("no HIP device visible" from
    # pr_hmr_create when the library was loaded first, e.g. by __graft_entry__.build() followed by smoke() in one process).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise PoseRiskHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pr_abi_version() != ABI_VERSION:
        raise PoseRiskHipError(f"ABI version mismatch: library {lib.pr_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def declare_stream(device=None):
    """Tell the library which stream this thread enqueues on (torch's current one) before a call that allocates -- create,
    set_streams, destroy -- so that it can refuse instead of invalidating a hipGraph capture in progress
    (include/poserisk_hip.h, pr_declare_stream)."""
    import torch
    lib = load()
    if torch.cuda.is_available():
        lib.pr_declare_stream(torch.cuda.current_stream(device).cuda_stream, 1)
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().pr_last_error().decode("utf-8", "replace")
        raise PoseRiskHipError(f"{what} failed (status {status}): {msg}")


def reba_info_struct(info):
    """add_info["REBA"] dict (example/additional_information.json:2-11) -> RebaInfo."""
    return RebaInfo(int(info["Legs_bilateral_weight_bearing/walking"]), int(info["Sitting"]),
                    int(info["Load/Force Score"]), int(info["Arm_supported_leaning_L"]),
                    int(info["Arm_supported_leaning_R"]), int(info["Coupling"]), int(info["Activity_Score"]))


def rula_info_struct(info):
    """add_info["RULA"] dict (example/additional_information.json:13-24) -> RulaInfo."""
    return RulaInfo(int(info["Arm_supported_leaning_L"]), int(info["Arm_supported_leaning_R"]),
                    int(info["A_Muscle_use_L"]), int(info["A_Muscle_use_R"]), int(info["A_Load/Force_L"]),
                    int(info["A_Load/Force_R"]), int(info["Legs_bilateral_weight_bearing"]),
                    int(info["B_Muscle_use"]), int(info["B_Load/Force"]))
