class AutoencoderKLWan:  # marker only
    pass


class WanTransformer3DModel:  # marker only
    pass
