"""Checkpoint loading without mmengine.

The reference stores the model hyper-parameters as mmengine *config source text* inside the
checkpoint: `{'state_dict': ..., 'meta': {'cfg': "<python source>"}}` and rebuilds the model with
`Config.fromstring(cfg, '.py').model` -> `MODELS.build` (`eval_models_seq.py:52-60,86`).  Here the
source is parsed with `ast` (never executed): assignments of literals, `dict(...)`/`list(...)`/
`tuple(...)` calls, references to earlier variables and simple arithmetic are understood, which is
what mmengine config files consist of.  Without a `meta.cfg` the hyper-parameters are inferred from
the tensor shapes (`weights.infer_config`).
"""
import ast
import operator
from typing import Any, Dict, Optional

import torch

from .config import GeneratorConfig
from .weights import infer_config

_BINOPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
           ast.FloorDiv: operator.floordiv, ast.Pow: operator.pow, ast.Mod: operator.mod}


def _eval(node: ast.AST, env: Dict[str, Any]):
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.Name):
        if node.id in env:
            return env[node.id]
        if node.id in ('True', 'False', 'None'):
            return {'True': True, 'False': False, 'None': None}[node.id]
        raise ValueError(f'config refers to unknown name {node.id!r}')
    if isinstance(node, (ast.List, ast.Tuple)):
        vals = [_eval(e, env) for e in node.elts]
        return vals if isinstance(node, ast.List) else tuple(vals)
    if isinstance(node, ast.Dict):
        return {_eval(k, env): _eval(v, env) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _eval(node.operand, env)
        return -v if isinstance(node.op, ast.USub) else +v
    if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
        return _BINOPS[type(node.op)](_eval(node.left, env), _eval(node.right, env))
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in ('dict', 'list', 'tuple'):
        if node.func.id == 'dict':
            out = {}
            for a in node.args:
                out.update(_eval(a, env))
            for kw in node.keywords:
                if kw.arg is None:
                    out.update(_eval(kw.value, env))
                else:
                    out[kw.arg] = _eval(kw.value, env)
            return out
        seq = [_eval(a, env) for a in node.args]
        seq = list(seq[0]) if seq else []
        return seq if node.func.id == 'list' else tuple(seq)
    if isinstance(node, ast.Subscript):
        base = _eval(node.value, env)
        idx = node.slice
        if isinstance(idx, ast.Slice):
            lo = _eval(idx.lower, env) if idx.lower else None
            hi = _eval(idx.upper, env) if idx.upper else None
            st = _eval(idx.step, env) if idx.step else None
            return base[slice(lo, hi, st)]
        return base[_eval(idx, env)]
    raise ValueError(f'unsupported construct in checkpoint config: {ast.dump(node)[:80]}')


def parse_config_source(src: str) -> Dict[str, Any]:
    """Top-level assignments of an mmengine python config, evaluated without executing anything."""
    env: Dict[str, Any] = {}
    for stmt in ast.parse(src).body:
        if isinstance(stmt, ast.Assign) and len(stmt.targets) == 1 and isinstance(stmt.targets[0], ast.Name):
            try:
                env[stmt.targets[0].id] = _eval(stmt.value, env)
            except ValueError:
                if stmt.targets[0].id == 'model':
                    raise
        # imports, custom_imports etc. are irrelevant to the model dict and are skipped
    return env


def generator_config_from_cfg(cfg_source: str) -> GeneratorConfig:
    env = parse_config_source(cfg_source)
    if 'model' not in env:
        raise ValueError("checkpoint config has no top-level 'model = dict(...)'")
    model = env['model']
    mtype = str(model.get('type', ''))
    if not mtype.startswith('BDE2VID'):
        raise ValueError(f'checkpoint holds a {mtype!r} model; this build implements BDE2VID only')
    gen = dict(model['generator'])
    gtype = gen.get('type', 'BDE2VIDCrossscalePropogationV5')
    if gtype != 'BDE2VIDCrossscalePropogationV5':
        raise ValueError(f'generator type {gtype!r} is not built (only BDE2VIDCrossscalePropogationV5)')
    cfg = GeneratorConfig.from_dict(gen)
    cfg.validate()
    return cfg


def read_checkpoint(checkpointfile: str, trust_checkpoint: bool = False):
    """torch.load restricted to tensors and plain containers (`weights_only=True`): nothing in the file is executed.
    A file that needs the full unpickler (custom classes, e.g. an mmengine object saved into `meta`) is refused
    unless the caller states that it trusts the file -- unpickling can run arbitrary code."""
    if trust_checkpoint:
        return torch.load(checkpointfile, map_location='cpu', weights_only=False)
    try:
        return torch.load(checkpointfile, map_location='cpu', weights_only=True)
    except FileNotFoundError:
        raise
    except Exception as e:
        raise RuntimeError(
            f'{checkpointfile}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}). '
            'If the file comes from a source you trust, pass trust_checkpoint=True to use the full unpickler.') from e


def load_model(checkpointfile: str, device='cuda', cpu_cache_length: Optional[int] = None,
               trust_checkpoint: bool = False):
    """Counterpart of `load_model` (`eval_models_seq.py:41-96`) for BDE2VID checkpoints."""
    from .model import BDE2VID
    ckpt = read_checkpoint(checkpointfile, trust_checkpoint)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        sd = ckpt['state_dict']
        meta = ckpt.get('meta') or {}
        if 'cfg' in meta:
            cfg = generator_config_from_cfg(meta['cfg'])
            ccl = parse_config_source(meta['cfg']).get('model', {}).get('cpu_cache_length', 100)
        else:
            cfg, ccl = infer_config(sd), 100
    else:                                   # bare state dict (eval_models_seq.py:87-95 style)
        sd, cfg, ccl = ckpt, infer_config(ckpt), 100
    model = BDE2VID(generator=cfg, cpu_cache_length=cpu_cache_length if cpu_cache_length is not None else ccl)
    model.to(device)
    model.load_state_dict(sd)
    return model.eval()
