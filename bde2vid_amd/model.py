"""Host-side mirror of the reference model interface for the hot path.

`BDE2VID` has the surface `eval_models_seq.py` touches:
constructor `(generator=dict(...), cpu_cache_length=100)` (bde2vid.py:14), `load_state_dict`
(eval_models_seq.py:86), `eval()`, `to(device)` (:116-117), `reset_states()` (:169) and
`model(inputs)` == `forward(inputs, mode='tensor')` (bde2vid.py:30-50).  Everything that
computes runs in libbde2vid.so through the C ABI; PyTorch only owns the device memory
and the stream.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .config import GeneratorConfig
from .weights import PREFIX, infer_config, state_dict_spec


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class BDE2VID:
    """Drop-in for `model/BDE2VID/bde2vid.py:BDE2VID` (inference, mode='tensor')."""

    def __init__(self, generator=None, cpu_cache_length: int = 100, init_cfg=None):
        if isinstance(generator, GeneratorConfig):
            self.cfg = generator
        elif generator is None:
            self.cfg = None                     # inferred from the state dict on load
        else:
            self.cfg = GeneratorConfig.from_dict(generator)
        if self.cfg is not None:
            self.cfg.validate()
        # The reference off-loads feature maps to the host when T > cpu_cache_length (V5.py:102);
        # with 288 GB of HBM nothing is off-loaded here, the argument is accepted for compatibility.
        self.cpu_cache_length = cpu_cache_length
        self.training = False
        self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        self._h = None
        self._loaded = False
        self._ext_streams = {}              # pipelined mode: torch views of the library's internal streams
        self._inflight = []                 # pipelined mode: inputs / outputs of calls not yet waited for (see forward)

    # ---- reference surface ---------------------------------------------------------------
    @property
    def num_encoders(self):
        return self.cfg.num_encoders

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('bde2vid_amd is an inference path (mode="tensor") only')
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('bde2vid_amd runs on an MI355X only; there is no CPU path')
        if self._loaded and self.device is not None and device.index not in (None, self.device.index):
            raise RuntimeError('move the model before load_state_dict(): packed weights live on one device')
        if device.index is not None:
            self.device = device
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', device if device is not None else torch.cuda.current_device()))

    def reset_states(self):
        """State never survives a forward call (bde2vid.py:31); nothing to clear."""
        return None

    def state_dict_keys(self) -> List[str]:
        return [k for k, _ in state_dict_spec(self.cfg)]

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        if self.cfg is None:
            self.cfg = infer_config(state_dict)
        spec = state_dict_spec(self.cfg)
        missing = [k for k, _ in spec if k not in state_dict]
        used = {k for k, _ in spec}
        unexpected = [k for k in state_dict
                      if k not in used and not k.endswith(('relative_position_index', 'num_batches_tracked'))]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing keys {missing[:5]}..., unexpected {unexpected[:5]}...')
        if missing:
            raise RuntimeError(f'load_state_dict: missing keys {missing[:8]}')
        L = _lib.lib()
        self._create()
        for k, shape in spec:
            if '.fusion_layers.' in k:          # dead on the forward path (V5.py:54-57)
                continue
            t = state_dict[k].detach().to('cpu', torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f'load_state_dict: {k} has shape {tuple(t.shape)}, expected {tuple(shape)}')
            sh = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(L.bde_load_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()), sh, t.dim()))
        with torch.cuda.device(self.device):
            _lib.check(L.bde_finalize_weights(self._h))
        self._loaded = True
        return self

    def alloc_packed(self):
        """Allocate the packed device image without weights (receiver side of the broadcast)."""
        self._create()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bde_alloc_packed(self._h))
        self._loaded = True
        return self

    def packed_view(self) -> torch.Tensor:
        """The packed weights as a 1-D fp32 CUDA tensor aliasing the library's buffer."""
        L = _lib.lib()
        n = int(L.bde_packed_numel(self._h))
        ptr = int(L.bde_packed_ptr(self._h))
        if n == 0 or ptr == 0:
            raise RuntimeError('weights are not finalized')

        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 3}
        self._packed_holder = h
        return torch.as_tensor(h, device=self.device)

    def __call__(self, inputs, mode='tensor', **kwargs):
        return self.forward(inputs, mode=mode, **kwargs)

    def forward(self, inputs: Sequence[dict], mode: str = 'tensor', **kwargs) -> List[torch.Tensor]:
        if mode != 'tensor':
            raise NotImplementedError(f"mode={mode!r}: only the inference mode 'tensor' is built")
        if not self._loaded:
            raise RuntimeError('load_state_dict() first')
        T = len(inputs)
        if T < 1:
            raise ValueError('empty input sequence')
        evs = []
        for d in inputs:
            e = d['events']
            if not e.is_cuda or e.dtype != torch.float32:
                raise TypeError("inputs[t]['events'] must be float32 CUDA tensors")
            evs.append(e.contiguous())
        B, nb, H, W = evs[0].shape
        if nb != self.cfg.num_bins or any(tuple(e.shape) != (B, nb, H, W) for e in evs):
            raise ValueError('all frames must have shape [B, num_bins, Hp, Wp]')
        dev = evs[0].device
        out = torch.empty((T, B, 1, H, W), dtype=torch.float32, device=dev)
        ev_ptrs = (C.c_void_p * T)(*[e.data_ptr() for e in evs])
        im_ptrs = (C.c_void_p * T)(*[out[t].data_ptr() for t in range(T)])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().bde_forward(self._h, ev_ptrs, T, B, H, W, im_ptrs,
                                              C.c_void_p(_stream_ptr(dev))))
        if self.get_info('pipeline') > 1:
            # Pipelined mode: the library reads the inputs and writes the outputs on an internal stream after this
            # call has returned.  Tell the caching allocator, so that memory of a tensor the caller drops early
            # (the .contiguous() temporaries above, an output that is never read) is not re-issued before that
            # stream is done with it.
            ptr = self.get_info('last_stream')
            es = self._ext_streams.get(ptr)
            if es is None:
                es = self._ext_streams[ptr] = torch.cuda.ExternalStream(ptr, device=dev)
            for t in evs:
                t.record_stream(es)
            out.record_stream(es)
            # ... and keep the output buffer alive until wait(): a forward whose operands leave the range of the two-term
            # format is recomputed there ("sb_auto", csrc/split.h) into the same buffer, long after the first run is done with it
            self._inflight.append(out)
            if len(self._inflight) > 8:
                del self._inflight[0]
        return [out[t] for t in range(T)]

    def wait(self):
        """Pipelined mode only (set_tuning('pipeline', 2)): order the outputs of all forward calls issued so
        far into the current stream.  Call before reading them (a device synchronize also suffices)."""
        try:
            _lib.check(_lib.lib().bde_wait_outputs(self._h, C.c_void_p(_stream_ptr(self.device))))
        finally:
            self._inflight.clear()
        return self

    def get_info(self, key: str) -> int:
        v = C.c_int64()
        _lib.check(_lib.lib().bde_get_info(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_tuning(self, key: str, value: int):
        _lib.check(_lib.lib().bde_set_tuning(self._h, key.encode(), int(value)))
        return self

    def get_intermediate(self, name: str, shape) -> torch.Tensor:
        t = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().bde_get_intermediate(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.numel(),
                                                   C.c_void_p(_stream_ptr(self.device))))
        return t

    # ---- internals -------------------------------------------------------------------------
    def _create(self):
        if self.device is None:
            raise RuntimeError('no MI355X visible (torch.cuda.is_available() is False)')
        if self._h is not None:
            _lib.lib().bde_destroy(self._h)
            self._h = None
        c = _lib.make_config(self.cfg)
        h = C.c_void_p()
        _lib.check(_lib.lib().bde_create(C.byref(c), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                _lib.lib().bde_destroy(self._h)
                self._h = None
        except Exception:
            pass


def build_model(cfg: GeneratorConfig, state_dict=None, device=None) -> BDE2VID:
    m = BDE2VID(generator=cfg)
    if device is not None:
        m.to(device)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m
