"""Engine loop around the hot path: scheduler, KV-cache manager, model runner, `LLM`."""
