"""Stream helper of the boundary (`compactor_vllm/utils/helpers.py:6-28`), with real dependency edges."""
from collections.abc import Callable

import torch


def maybe_execute_in_stream(fn: Callable, *args, STORE_STREAM: torch.cuda.Stream = None, **kwargs):
    """Run `fn` on STORE_STREAM (after everything already enqueued on the current stream), or inline.

    Same calling convention as the reference.  The reference orders store-after-main with
    `wait_stream(default_stream)` and then only calls `record_stream` (allocator lifetime) on the outputs;
    consumers on the main stream must add their own `wait_stream(STORE_STREAM)` before reading the result
    (the reference forgets this for Compactor's pre-RoPE scores, hazard H1 — see compactor.py here)."""
    if STORE_STREAM is None:
        return fn(*args, **kwargs)
    cur = torch.cuda.current_stream()
    tensors = [a for a in args if isinstance(a, torch.Tensor)]
    tensors += [v for v in kwargs.values() if isinstance(v, torch.Tensor)]
    obj = getattr(fn, "__self__", None)
    if isinstance(obj, torch.Tensor):
        tensors.append(obj)
    STORE_STREAM.wait_stream(cur)
    with torch.cuda.stream(STORE_STREAM):
        output = fn(*args, **kwargs)
    for t in tensors:
        t.record_stream(STORE_STREAM)
    outs = output if isinstance(output, tuple) else (output,)
    for o in outs:
        if isinstance(o, torch.Tensor):
            o.record_stream(cur)
    return output
