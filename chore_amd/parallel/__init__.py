from .frame_shard import frames_of_rank, gather_fitted, init_distributed  # noqa: F401
from .grad_arena import FlatGradReducer, backward_in_segments, chore_segments  # noqa: F401
from .graph_train import GraphedTrainStep, drain_collectives  # noqa: F401
