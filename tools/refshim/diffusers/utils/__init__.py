import logging as _pylog


def deprecate(*args, **kwargs):
    return None


def is_scipy_available():
    return True


def is_ftfy_available():
    return False


def is_torch_xla_available():
    return False


def replace_example_docstring(doc):
    def deco(fn):
        return fn

    return deco


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _pylog.getLogger(name)


logging = _Logging()


class BaseOutput:
    """diffusers.utils.BaseOutput: a dataclass base whose fields can also be read by index / key."""

    def __getitem__(self, k):
        import dataclasses
        vals = [getattr(self, f.name) for f in dataclasses.fields(self)]
        if isinstance(k, str):
            return getattr(self, k)
        return [v for v in vals if v is not None][k]

    def to_tuple(self):
        import dataclasses
        return tuple(getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) is not None)
