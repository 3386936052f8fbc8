"""`maybe_execute_in_stream`: run a piece of the cache / scoring chain on the side ("store") stream.

Calling convention of `compactor_vllm/utils/helpers.py:6-28` (positional / keyword arguments are forwarded, the
keyword-only `STORE_STREAM=None` means "run inline").  What it guarantees here:
  * the side stream first waits for everything already enqueued on the caller's stream (inputs are ready);
  * tensors passed in are marked as in use on the side stream and tensor results as in use on the caller's stream
    (caching-allocator lifetime - NOT an execution dependency);
  * a consumer on another stream must still `wait_stream(STORE_STREAM)` before reading a result; the places that need
    it do so explicitly (the reference has one such edge missing, hazard H1 of SURVEY 3.1 - see compression/compactor.py).
"""
from collections.abc import Callable
from itertools import chain

import torch


def _tensors_in(values):
    return [x for x in values if isinstance(x, torch.Tensor)]


def maybe_execute_in_stream(fn: Callable, *args, STORE_STREAM: torch.cuda.Stream = None, **kwargs):
    if STORE_STREAM is None:
        return fn(*args, **kwargs)
    caller = torch.cuda.current_stream()
    STORE_STREAM.wait_stream(caller)
    with torch.cuda.stream(STORE_STREAM):
        result = fn(*args, **kwargs)
    inputs = _tensors_in(chain(args, kwargs.values()))
    bound_to = getattr(fn, "__self__", None)  # e.g. tensor.index_copy_
    if isinstance(bound_to, torch.Tensor):
        inputs.append(bound_to)
    for t in inputs:
        t.record_stream(STORE_STREAM)
    for t in _tensors_in(result if isinstance(result, tuple) else (result,)):
        t.record_stream(caller)
    return result
