"""sea_attention_amd -- SEA's sparse-attention hot path, MI355X-native.

    from sea_attention_amd.perlin_attention import PerlinSelfAttention, PerlinAttention, ...   # module API
    from sea_attention_amd.perlin_attention import ops                                          # operator API

The HIP kernels live in csrc/ and are reached through the C ABI of libsea_hip.so (include/sea_hip.h).
"""
from . import _build, _lib, utils
from .utils import get_bench, seed
from . import perlin_attention
from .perlin_attention import (PerlinAttention, PerlinAttentionConfig, PerlinAttentionOutput, PerlinSelfAttention,
                               get_default_config, register_default_config)

__all__ = ["perlin_attention", "PerlinAttention", "PerlinAttentionConfig", "PerlinAttentionOutput",
           "PerlinSelfAttention", "get_default_config", "register_default_config", "get_bench", "seed"]
