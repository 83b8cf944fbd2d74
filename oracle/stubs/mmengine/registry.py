"""Name -> class registry with the two calls the reference uses:
`@REG.register_module()` and `REG.build(dict(type=..., **kwargs))`."""


class Registry:
    def __init__(self, name, *args, **kwargs):
        self.name = name
        self._classes = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._classes[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._classes.get(key)

    def build(self, cfg, *args, **kwargs):
        cfg = dict(cfg)
        typ = cfg.pop('type')
        cls = self._classes[typ] if isinstance(typ, str) else typ
        return cls(**cfg)


MODELS = Registry('model')
METRICS = Registry('metric')
