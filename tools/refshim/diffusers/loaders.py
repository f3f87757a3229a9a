class WanLoraLoaderMixin:
    pass


class FromOriginalModelMixin:
    pass
