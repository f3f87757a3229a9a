import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_internal_dict", {}))
        cfg.update(kwargs)
        self._internal_dict = FrozenDict(cfg)

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        sig = inspect.signature(cls.__init__)
        args = {k: v for k, v in dict(config).items() if k in sig.parameters}
        args.update(kwargs)
        return cls(**args)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.items())[1:]
        cfg = {name: p.default for name, p in params if p.default is not inspect.Parameter.empty}
        for (name, _), a in zip(params, args):
            cfg[name] = a
        cfg.update(kwargs)
        ignore = getattr(self, "ignore_for_config", [])
        self.register_to_config(**{k: v for k, v in cfg.items() if k not in ignore and not k.startswith("_")})
        init(self, *args, **kwargs)

    return inner
