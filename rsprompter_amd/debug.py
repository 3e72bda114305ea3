"""Parity-test hooks.  The predict path can keep references to intermediate tensors (`_last_*` attributes of the
detectors and heads: RoI features, low-resolution mask logits, attention masks, ...) for the tests that compare them with
the oracle.  They pin hundreds of MB across steps, so they are OFF unless asked for:

    import rsprompter_amd.debug as dbg;  dbg.KEEP_TRACES = True      (tests/conftest.py, bench.py's parity canary)
    or RSP_KEEP_TRACES=1 in the environment."""
import os

KEEP_TRACES = os.environ.get('RSP_KEEP_TRACES', '0') == '1'


def keep(value):
    """value (or the result of calling it) when traces are on, else None"""
    if not KEEP_TRACES:
        return None
    return value() if callable(value) else value
