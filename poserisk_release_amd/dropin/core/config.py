"""Mirror of lib/core/config.py: the global `cfg` and `update_config` that main/run.py:8 and
lib/core/base.py:19 import by the bare name `core.config`.

    from core.config import cfg, update_config           main/run.py:8
    cfg.DATASET.batch_size / workers / min_frame_ratio / bbox_scale / default_information     config.py:30-35
    cfg.SPIN.SMPL_MEAN_PARAMS / checkpoint / SMPL_MODEL_DIR / FOCAL_LENGTH / IMG_RES           config.py:44-50

Same knob names, same defaults, attribute and item access like the reference's `easydict` (which is not a
dependency here).  The reference computes its paths from the location of its own config.py
(`root_dir = lib/core/../../`); this file lives in another repository, so the root of the PoseRisk checkout is
found as: $POSERISK_ROOT, else the directory above the reference's `lib/` that main/__init_path.py:16-17 put on
sys.path, else the current directory (the reference is run from its root: lib/utils/smpl.py:9 is CWD-relative).
`cfg.DATASET.hip_batch_size` (not in the reference) is the frames per `pr_frames_forward` call of the MI355X path;
`cfg.DATASET.batch_size` (8) stays what the reference hands to the tracker.  The other knobs of the MI355X path, all
absent from the reference and all optional:
    cfg.SPIN.precision        'fp32' (default: the reference's arithmetic) | 'bf16' (encoder on the bf16 MFMAs, fp32
                              accumulate; regressor and SMPL stay fp32: BASELINE configs[2])
    cfg.SPIN.conv_form        'default' | 'direct' | 'winograd5' ... (fp32 encoder; poserisk_release_amd.hmr.CONV_FORMS)
    cfg.DATASET.hip_lanes     whole batches in flight on separate HIP streams (2)
    cfg.DATASET.hip_world_size  0 = whatever the launcher says ($WORLD_SIZE of torch.distributed.run, else 1); N > 1 insists
                              on N ranks, one per GPU, frames sharded and gathered once over RCCL
main/run.py has its `--cfg` option commented out (run.py:20-24), so a YAML of overrides named by $POSERISK_CFG is applied
when this module is imported -- `POSERISK_CFG=bf16.yaml python main/run.py ...` with the one line `SPIN: {precision: bf16}`.
"""
import os
import os.path as osp
import sys


class _Cfg(dict):
    """dict with attribute access (the part of easydict the reference uses); nested dicts convert on assignment."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, _Cfg):
            value = _Cfg(value)
        super().__setitem__(key, value)

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


edict = _Cfg


def _reference_root():
    env = os.environ.get('POSERISK_ROOT')
    if env:
        return osp.abspath(env)
    here = osp.dirname(osp.abspath(__file__))
    for p in sys.path:
        if not p:
            continue
        cand = osp.join(p, 'core', 'base.py')
        if osp.isfile(cand) and osp.abspath(osp.join(p, 'core')) != here:
            return osp.abspath(osp.join(p, '..'))       # p is the reference's lib/
    return osp.abspath(os.getcwd())


def _defaults(root):
    """The reference's knobs and default values (config.py:17-60), as one table."""
    core_dir = osp.join(root, 'lib', 'core')
    spin_dir = osp.join(root, 'lib', 'SPIN')
    spin_data = osp.join(spin_dir, 'data')
    return {
        'root_dir': root, 'cur_dir': core_dir, 'data_dir': osp.join(root, 'data'),
        'smpl_dir': osp.join(root, 'smplpytorch'),
        'DATASET': {'workers': 16, 'batch_size': 8, 'min_frame_ratio': 0.33, 'bbox_scale': 1.2,
                    'default_information': osp.join(core_dir, 'default_information.json'),
                    'hip_batch_size': 64, 'hip_lanes': 2, 'hip_world_size': 0},
        'MODEL': {'input_shape': (224, 224)},
        'SPIN': {'spin_dir': spin_dir, 'SMPL_MEAN_PARAMS': osp.join(spin_data, 'smpl_mean_params.npz'),
                 'checkpoint': osp.join(spin_data, 'model_checkpoint.pt'),
                 'SMPL_MODEL_DIR': osp.join(spin_data, 'smpl'), 'FOCAL_LENGTH': 5000, 'IMG_RES': 224,
                 'precision': 'fp32', 'conv_form': 'default'},
        'AUG': {'flip': False, 'rotate_factor': 0},
        'TEST': {},
    }


cfg = edict(_defaults(_reference_root()))


def update_config(config_file):
    """YAML overrides with the reference's rules (config.py:63-85): a top-level or nested key that `cfg` does
    not already hold raises ValueError; nested sections are updated key by key, scalars replaced."""
    import yaml
    with open(config_file) as f:
        overrides = yaml.safe_load(f) or {}
    for section, value in overrides.items():
        if section not in cfg:
            raise ValueError("{} not exist in config.py".format(section))
        if not isinstance(value, dict):
            cfg[section] = value
            continue
        unknown = [k for k in value if k not in cfg[section]]
        if unknown:
            raise ValueError("{}.{} not exist in config.py".format(section, unknown[0]))
        for k, v in value.items():
            cfg[section][k] = v


if os.environ.get('POSERISK_CFG'):
    update_config(os.environ['POSERISK_CFG'])
