"""Drop-in modules with the reference's bare import names (main/__init_path.py:16-32 puts `lib/`,
`lib/utils/`, `lib/SPIN/`, `lib/smplpytorch/` on sys.path; base.py then does `from models import hmr`,
`from smpl import SMPL`, `from coord_utils import ...`, `from reba import REBA`, `from rula import RULA`).

`poserisk_release_amd.dropin.install()` puts this directory first on sys.path so those imports resolve
to the MI355X implementation."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    repo = os.path.dirname(os.path.dirname(HERE))
    for p in (repo, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
