"""Host-side helpers of the MI355X-native ggml block-quant path (formats, synthetic
inputs, C-ABI loader, row-sharded multi-GPU op)."""
from .formats import GGMLType, BLOCK, WEIGHT_TYPES, block_elems, block_bytes, row_bytes  # noqa: F401
