"""Drop-in for the reference's ``pointops`` package (libs/pointops/functions/__init__.py:1-14),
backed by hand-written gfx950 kernels behind the C ABI of include/pdfops.h."""
from ._ops import (
    knn_query,
    ball_query,
    random_ball_query,
    farthest_point_sampling,
    grouping,
    grouping2,
    interpolation,
    interpolation2,
    subtraction,
    aggregation,
    attention_relation_step,
    attention_fusion_step,
)
from ._compose import (
    query_and_group,
    knn_query_and_group,
    ball_query_and_group,
    batch2offset,
    offset2batch,
)

__all__ = [
    "knn_query", "ball_query", "random_ball_query", "farthest_point_sampling", "grouping", "grouping2",
    "interpolation", "interpolation2", "subtraction", "aggregation", "attention_relation_step",
    "attention_fusion_step", "query_and_group", "knn_query_and_group", "ball_query_and_group",
    "batch2offset", "offset2batch",
]
