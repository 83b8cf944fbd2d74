def deform_conv2d(*a, **k):
    raise NotImplementedError('mmcv stub: deform_conv2d is not on the BDE2VID path')
