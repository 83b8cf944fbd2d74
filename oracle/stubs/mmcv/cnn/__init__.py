def constant_init(module, val, bias=0):
    pass


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    pass
