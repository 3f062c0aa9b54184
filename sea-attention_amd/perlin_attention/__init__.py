"""Drop-in package for `src.models.perlin_attention`: the names its importers use (perlin_opt.py:21,178,
perlin_trainer.py:26,137,155,244, benchmark_bert.py:17) plus the `modules` / `ops` sub-packages."""
from . import config as _config
from . import attention as _attention
from . import self_attention as _self_attention
from . import modules, ops

PerlinAttentionConfig = _config.PerlinAttentionConfig
register_default_config = _config.register_default_config
get_default_config = _config.get_default_config
PerlinAttention = _attention.PerlinAttention
PerlinAttentionOutput = _attention.PerlinAttentionOutput
PerlinSelfAttention = _self_attention.PerlinSelfAttention

__all__ = ["PerlinAttentionConfig", "register_default_config", "get_default_config", "PerlinAttention",
           "PerlinAttentionOutput", "PerlinSelfAttention", "modules", "ops"]
