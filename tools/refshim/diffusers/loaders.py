class WanLoraLoaderMixin:
    pass
