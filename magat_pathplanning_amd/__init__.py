"""MI355X-native MAGAT batched graph-attention forward (drop-in for the reference's
DecentralPlannerGATNet / GraphFilterBatchAttentional).  See DESIGN.md and INTEGRATION.md."""
from .graphml import GraphFilterBatch, GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin  # noqa: F401
from .planner import DecentralPlannerGATNet, DecentralPlannerNet  # noqa: F401

__all__ = ["DecentralPlannerGATNet", "DecentralPlannerNet", "GraphFilterBatchAttentional", "GraphFilterBatchAttentional_Origin", "GraphFilterBatch"]
