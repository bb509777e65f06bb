from .harness import evaluate_sharded, local_batches, shard_indices  # noqa: F401
from .stereo import EpeAccumulator, calc_error, remove_padding  # noqa: F401
