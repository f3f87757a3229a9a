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
