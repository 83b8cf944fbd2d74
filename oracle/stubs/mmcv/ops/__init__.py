import torch.nn as nn


class DeformConv2d(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError('mmcv stub: DeformConv2d is not on the BDE2VID path')


class DeformConv2dPack(DeformConv2d):
    pass
