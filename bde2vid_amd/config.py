"""Generator hyper-parameters of the BDE2VID reconstructor.

Mirrors the constructor arguments of the reference generator
(`model/BDE2VID/bde2vid_cross_scale_propogation_V5.py:19-23`).  The shipped
checkpoint (and with it the shipped values) is absent from the reference mount,
so everything here is shape-generic; `canonical()` is SURVEY.md §0 "config A".
"""
from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple


@dataclass
class GeneratorConfig:
    num_bins: int = 5
    basechannels: int = 32
    num_encoders: int = 3
    ks: int = 5
    num_res_blocks: int = 2
    norm: Optional[str] = None
    recurrent_block_type: str = 'convlstm'
    useRC: bool = True
    skip_type: str = 'sum'
    activation: str = 'Sigmoid'          # reference: dict(type='Sigmoid'), V5.py:25-26
    num_output_channels: int = 1
    act_net: str = 'default'             # -> ReLU, V5.py:246
    buffer_index: Tuple[int, ...] = (-1, 0, 1)
    q_idx: int = 1
    window_size: Tuple[int, int] = (7, 7)
    nwindow_size: Optional[Tuple[int, int]] = None
    depths: Tuple[int, ...] = (4, 0, 6)
    num_heads: int = 16
    act_attn: str = 'default'            # -> GELU, DTransformer.py:345
    mlp_ratio: float = 4.0

    # ---- derived -----------------------------------------------------
    @property
    def frame_num(self) -> int:
        return len(self.buffer_index)

    def enc_in(self, l: int) -> int:
        return self.basechannels * 2 ** l

    def enc_out(self, l: int) -> int:
        return self.basechannels * 2 ** (l + 1)

    @property
    def norm_kind(self) -> int:
        """0 = no norm layer, 1 = BatchNorm2d, 2 = InstanceNorm2d(track_running_stats=True) (submodules.py:96-103); any other
        string builds a ConvLayer without a norm layer, exactly as None does."""
        return {'BN': 1, 'IN': 2}.get(self.norm, 0)

    @property
    def bottleneck(self) -> bool:
        """depths[-1] == 0: the last level runs num_res_blocks ResidualBlockNoBN on buffer slot 0 instead of attention
        (V5.py:77-80, 262-282)."""
        return self.depths[-1] == 0

    def validate(self) -> None:
        """Reject what the reference itself cannot run or what the kernels are not built for: everything else the
        constructor of the reference accepts (V5.py:19-98) is built -- ConvLSTM / ConvGRU / plain encoders, skip sum / concat,
        norm None / BN / IN (eval mode), attention or the residual-block bottleneck on the last level."""
        bad = []
        if self.norm not in (None, 'none', 'BN', 'IN'):
            bad.append(f'norm={self.norm!r} (None, "BN" or "IN")')
        if self.recurrent_block_type not in ('convlstm', 'convgru'):
            bad.append(f'recurrent_block_type={self.recurrent_block_type!r} (the reference asserts convlstm | convgru)')
        if self.skip_type not in ('sum', 'concat'):
            bad.append(f'skip_type={self.skip_type!r} ("no_skip" hands a list to the decoder in the reference and raises there)')
        if self.activation not in ('Sigmoid', 'Identity'):
            bad.append(f'activation={self.activation!r}')
        if self.act_net not in ('default', 'ReLU'):
            bad.append(f'act_net={self.act_net!r}')
        if self.act_attn not in ('default', 'GELU'):
            bad.append(f'act_attn={self.act_attn!r}')
        if self.nwindow_size is not None:
            bad.append('nwindow_size is not None (reduction_conv path)')
        if self.num_output_channels != 1:
            bad.append(f'num_output_channels={self.num_output_channels}')
        if len(self.depths) != self.num_encoders:
            bad.append('len(depths) != num_encoders')
        if self.bottleneck and self.num_res_blocks < 0:
            bad.append(f'num_res_blocks={self.num_res_blocks}')
        if tuple(self.window_size) != (7, 7):
            bad.append(f'window_size={self.window_size} (kernels are built for 7x7)')
        if self.ks not in (3, 5):
            bad.append(f'ks={self.ks}')
        if not (0 <= self.q_idx < self.frame_num):
            bad.append('q_idx outside buffer_index')
        for l, d in enumerate(self.depths):
            if d > 0 and self.enc_out(l) % self.num_heads:
                bad.append(f'level {l}: C={self.enc_out(l)} not divisible by heads')
        if bad:
            raise ValueError('unsupported BDE2VID generator config: ' + '; '.join(bad))

    def to_reference_kwargs(self) -> dict:
        """kwargs for the reference constructor (used only by oracle/gen_golden.py)."""
        return dict(type='BDE2VIDCrossscalePropogationV5',
                    num_bins=self.num_bins, basechannels=self.basechannels,
                    num_encoders=self.num_encoders, ks=self.ks,
                    num_res_blocks=self.num_res_blocks, norm=self.norm,
                    recurrent_block_type=self.recurrent_block_type, useRC=self.useRC,
                    skip_type=self.skip_type, activation=dict(type=self.activation),
                    num_output_channels=self.num_output_channels, act_net=self.act_net,
                    buffer_index=list(self.buffer_index), q_idx=self.q_idx,
                    window_size=tuple(self.window_size), nwindow_size=self.nwindow_size,
                    depths=list(self.depths), num_heads=self.num_heads,
                    act_attn=self.act_attn, losses=[])

    def to_dict(self) -> dict:
        return asdict(self)

    @staticmethod
    def from_dict(d: dict) -> 'GeneratorConfig':
        d = dict(d)
        d.pop('type', None)
        act = d.get('activation')
        if isinstance(act, dict):
            d['activation'] = act.get('type', 'Sigmoid')
        elif act is None:
            d['activation'] = 'Sigmoid'
        for k in ('losses', 'loss_inds', 'init_cfg', 'drop_path_rate', 'use_checkpoint'):
            d.pop(k, None)
        for k in ('buffer_index', 'window_size', 'depths'):
            if k in d and d[k] is not None:
                d[k] = tuple(int(v) for v in d[k])
        if d.get('nwindow_size') is not None:
            d['nwindow_size'] = tuple(d['nwindow_size'])
        return GeneratorConfig(**d)


def canonical() -> GeneratorConfig:
    """SURVEY.md §0 assumed canonical config "A" (20,870,433 parameters)."""
    return GeneratorConfig()
