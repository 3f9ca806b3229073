class ExecutorType:  # noqa: D101
    pass


class PrecomputedFeatures:  # noqa: D101
    def __init__(self, *a, **k):
        pass


def _get_executor(*a, **k):
    raise NotImplementedError
