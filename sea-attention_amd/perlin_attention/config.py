"""`PerlinAttentionConfig` and the process-wide default it is constructed from.

The field set (names, types, defaults) is the drop-in contract with the reference's trainer and model code
(src/models/perlin_attention/config.py:12-61): `src/trainer/perlin_trainer.py:137-155` fills it from CLI flags and
registers it as the default, the attention modules pick the registered default up at construction
(attention.py:142).  Each field below notes which step of this build consumes it.
"""
import dataclasses
import json


@dataclasses.dataclass
class PerlinAttentionConfig:
    # ---- baselines of the reference that share the config object (not used by the SEA hot path) -------------------
    reformer_n_hashs: int = 8                      # reformer baseline only
    # ---- step B: Performer estimator ---------------------------------------------------------------------------
    performer_nb_factor: int = 1                   # nb_features = d * ln(d) / factor
    # ---- step H: grouped top-k -----------------------------------------------------------------------------------
    k: int = 7                                     # keys kept per (query, head) on average
    k_flatten: bool = True                         # pool the heads of a row before selecting
    k_flatten_dim: str = 'causal_batch'            # the only pooling the causal model uses
    random_lookup: bool = False                    # (reference ablation; off)
    random_lookup_count: int = 3
    # ---- steps D-G: attention predictor ------------------------------------------------------------------------------
    attention_predictor_method: str = 'mlp'        # 'mlp' | 'comp' (codebook predictor, attention.py:293-312, 649-661)
    attention_predictor_length: int = 128          # T_M, width of the compressed attention map
    attention_predictor_backend: str = 'performer'
    attention_predictor_comp_book_size: int = 8    # 'comp' predictor only
    attention_predictor_comp_patch_size: int = 16
    attention_predictor_comp_patch_count: int = 16
    attention_predictor_enc_per_layer: bool = False
    # ---- training-time plumbing ----------------------------------------------------------------------------------
    layerwise: bool = False                        # detach the layer input while distilling layer by layer
    lora_r: int = 32
    lora_enabled: bool = False
    lora_in_approx_enabled: bool = False
    # ---- steps J-L: sparse attention and output ----------------------------------------------------------------------
    partial_attention_scaler: bool = True          # multiply the sparse probabilities by sigmoid(scale_0)
    out_add_performer_context: bool = False
    v_eye_length: int = 128
    out_norm: bool = False
    causal: bool = False                           # perlin_opt forces True
    use_cache: bool = False                        # kv-cache decoding (attention_state.py)
    compile: bool = False
    context_output_method: str = 'mix'             # lerp with the cumulative-average value by sigmoid(scale_1)
    k_oversample: float = 1.0                      # multiplies k in the per-row keep count

    def to_json(self) -> dict:
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

    def check_validity(self) -> None:
        causal_pooling = self.causal and self.k_flatten
        assert (not causal_pooling) or self.k_flatten_dim in ('causal_batch',), self.k_flatten_dim

    def __repr__(self) -> str:
        return "PerlinAttentionConfig(" + json.dumps(self.to_json()) + ")"


class _Registry:
    """Holder of the process-wide default (the reference keeps a module global, config.py:53-61)."""
    current = PerlinAttentionConfig()


def register_default_config(config: PerlinAttentionConfig) -> None:
    _Registry.current = config


def get_default_config() -> PerlinAttentionConfig:
    return _Registry.current


def __getattr__(name):                             # `config.DEFAULT_CONFIG` stays readable for code that peeks at it
    if name == "DEFAULT_CONFIG":
        return _Registry.current
    raise AttributeError(name)
