"""Stand-in for lhotse (imported by modules/general/input_strategies.py:14-21, unused here)."""


class CutSet:  # noqa: D101
    pass
