from .frame_shard import frames_of_rank, gather_fitted, init_distributed  # noqa: F401
