"""Operator package, same seven exports as the reference (src/models/perlin_attention/ops/__init__.py:1-7)."""
from .resize_dense import resize_from_m_to_t
from .flat_csr import resize_from_m_to_t_csr
from .flat_csr import flat_csr_elmul
from .flat_csr import flat_csr_masked_bmm
from .flat_csr import flat_csr_sdbmm
from .flat_csr import flat_csr_softmax
from .flat_csr import flat_csr_to_dense
# fused MI355X entry points (not in the reference)
from .flat_csr import (FlatCSR, fused_interp_supported, attention_few_rows, topk_to_csr, topk_mask, sparse_attention, sparse_attention_bytes, attention_plan, plan_blocks, make_plan,
                       sparse_attention_autograd,
                       keep_table_causal, keep_table_kernel_test, z_capacity, csr_from_selection)
from .predictor import (split_layernorm, predictor_tail, cumavg, performer_value, performer_supported, performer_avg_supported,
                        performer_step, performer_plan, performer_chunk_rows,
                        split_layernorm_c8, causal_conv_c8, causal_conv_c8_z, conv_z_supported, conv_c8_f32_supported, predictor_tail_z, pack_conv_weight, to_c8, from_c8,
                        predictor_mlp, predictor_mlp_supported, predictor_tail_select,
                        predictor_tail_select_supported, clear_prep_cache, prep_generation, pinned_prep, LazyTensor, realize,
                        decode_stage, c8_window_shift, decode_cnn_tail_select, decode_cnn_supported, decode_cnn_emits)
