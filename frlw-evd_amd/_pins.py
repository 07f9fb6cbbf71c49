"""Lifetime of module-level device caches under HIP-graph replay.

A captured graph (``Trainer(graph=True)``) bakes in the raw addresses of whatever cached scratch buffers, weight-operand
caches and launch tables its kernels used.  The caches are grown or rebuilt on demand by eager calls (a larger batch, another
resolution): without care the old tensor would be freed while the graph still launches kernels on it.  So: while at least
one graph is live (``pin()`` .. ``unpin()``), a cache that replaces a buffer hands the old one to ``retire()``, which keeps it
alive until the last graph is gone.
"""
_PINS = 0
_RETIRED = []


def pin():
    global _PINS
    _PINS += 1


def unpin():
    global _PINS
    _PINS = max(0, _PINS - 1)
    if _PINS == 0:
        _RETIRED.clear()


def retire(obj):
    """Call with the buffer (or tuple of buffers) a cache is about to drop."""
    if _PINS and obj is not None:
        _RETIRED.append(obj)


def live():
    return _PINS, len(_RETIRED)
