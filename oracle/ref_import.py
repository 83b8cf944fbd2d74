"""Import the *real* reference model in the build container (never on the GPU box).

TEST INFRASTRUCTURE ONLY.  Used by `oracle/gen_golden.py` and by the optional
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent).
Recipe: SURVEY.md §8(c).  Stubs for the three absent third-party packages live in
`oracle/stubs/` and contain no reference code.
"""
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stubs')


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'model', 'BDE2VID'))


def _prepare():
    import torch
    sys.dont_write_bytecode = True
    for p in (REFERENCE_ROOT, _STUBS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, _STUBS)
    # V5.py:12 imports model.losses.losses, which drags in sklearn/LPIPS/torchvision.
    if 'model.losses.losses' not in sys.modules:
        from mmengine.registry import MODELS
        pkg = types.ModuleType('model.losses')
        pkg.__path__ = []
        mod = types.ModuleType('model.losses.losses')
        mod.LOSSES = MODELS
        sys.modules['model.losses'] = pkg
        sys.modules['model.losses.losses'] = mod
    # hard-coded .cuda() on the hot path (V5.py:153,166): identity on a CPU-only box
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self


def import_model_classes():
    """Returns (BDE2VID, module DTransformer, module submodules) of the reference."""
    _prepare()
    from model.BDE2VID.bde2vid import BDE2VID
    import model.BDE2VID.DTransformer as DT
    import model.BDE2VID.submodules as SM
    return BDE2VID, DT, SM


def import_event_utils():
    _prepare()
    for name in ('h5py', 'cv2'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    import events_contrast_maximization.utils.event_utils as EU
    return EU


def import_h5_dataset():
    """The reference's data_loader.h5_dataset with the absent third-party modules it imports but does not need for the
    index logic (h5py, cv2, skimage) replaced by empty modules."""
    _prepare()

    def stub(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m
    for name in ('h5py', 'cv2', 'skimage', 'skimage.io'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                stub(name)
    if not hasattr(sys.modules['skimage'], 'io'):
        sys.modules['skimage'].io = sys.modules['skimage.io']
    import data_loader.h5_dataset as D
    return D


def import_croper():
    _prepare()
    from utils_func.inference_utils import Croper
    return Croper


def build_reference_model(cfg, seed=4, cpu_cache_length=100):
    """Instantiate the reference BDE2VID with formula weights (eval mode, CPU)."""
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.append(repo)
    from bde2vid_amd.weights import formula_state_dict
    BDE2VID, _, _ = import_model_classes()
    model = BDE2VID(generator=cfg.to_reference_kwargs(), cpu_cache_length=cpu_cache_length)
    sd = formula_state_dict(cfg, seed)
    own = model.state_dict()
    float_keys = [k for k, v in own.items() if v.is_floating_point()]
    missing = sorted(set(float_keys) - set(sd))
    extra = sorted(set(sd) - set(float_keys))
    assert not missing and not extra, (missing, extra)
    for k in float_keys:
        assert tuple(own[k].shape) == tuple(sd[k].shape), (k, own[k].shape, sd[k].shape)
    model.load_state_dict(sd, strict=False)
    model.eval()
    return model
