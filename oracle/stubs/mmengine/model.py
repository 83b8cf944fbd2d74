import torch.nn as nn


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class BaseModel(BaseModule):
    def __init__(self, data_preprocessor=None, init_cfg=None):
        super().__init__(init_cfg)
