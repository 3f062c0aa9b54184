"""Same import surface as the reference package (src/models/perlin_attention/__init__.py:1-3)."""
from .config import PerlinAttentionConfig, get_default_config, register_default_config
from .self_attention import PerlinSelfAttention
from .attention import PerlinAttention, PerlinAttentionOutput
from . import modules, ops
