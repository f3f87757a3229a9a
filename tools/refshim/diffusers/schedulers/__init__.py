class FlowMatchEulerDiscreteScheduler:  # marker only
    pass
