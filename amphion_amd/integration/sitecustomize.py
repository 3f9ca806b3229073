# Put this directory first on PYTHONPATH to run the unmodified Amphion CLI on the MI355X kernels.
try:
    import os
    import sys

    _repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if _repo not in sys.path:
        sys.path.append(_repo)
    from amphion_amd.integration import install

    install()
except Exception as _e:  # never break interpreter start-up
    import sys

    print(f"[amphion_amd] integration hook not installed: {_e}", file=sys.stderr)
