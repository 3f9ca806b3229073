"""Stand-in for ruamel.yaml (imported by utils/util.py:21-24, unused on the vocoder path)."""


class YAML:
    def __init__(self, *a, **k):
        pass

    def load(self, *a, **k):
        raise NotImplementedError

    def dump(self, *a, **k):
        raise NotImplementedError
