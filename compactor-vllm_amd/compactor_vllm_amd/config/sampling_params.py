"""Per-request sampling knobs - same two attributes and defaults as `compactor_vllm/config/sampling_params.py:4-11`."""
import dataclasses


@dataclasses.dataclass
class SamplingParams:
    """temperature 0 selects greedy decoding (what `bench.py` uses); the engine-side sampler is out of scope here."""

    temperature: float = 1.0
    max_new_tokens: int = 256

    def __post_init__(self):
        self._validate()

    def _validate(self):
        if not self.temperature >= 0:  # also rejects NaN
            raise ValueError("Temperature cannot be negative")

    @property
    def greedy(self) -> bool:
        return self.temperature == 0
