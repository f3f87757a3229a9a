class PipelineCallback:
    pass


class MultiPipelineCallbacks:
    pass
