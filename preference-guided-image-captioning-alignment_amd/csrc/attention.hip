// Fused multi-head self-attention for head_dim 64, forward and backward, on MFMA 16x16x32 bf16: the C-ABI entry points.
// The kernels are the key-tiled ones of attention_tiled.hip for every sequence length (GPT-2 captions S = 128 / 256,
// CLIP ViT-B/32 T = 50, ViT-L/14 T = 257); at S <= 128 they run one tile per (batch, head) and measured faster than the
// round-1 single-tile kernels they replace (forward 87 vs 94 us, backward 238 vs 286 us at 256 sequences x 16 heads,
// dropout 0.1), mainly because the probabilities never cross LDS.
#include "common.h"

using namespace pgca;

namespace pgca {
int attention_fwd_tiled(const void* qkv, const int32_t* key_mask, int B, int S, int heads, int causal, void* out,
                        float* lse, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale, const int32_t* cu,
                        void* stream);
int attention_bwd_tiled(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_mask,
                        int B, int S, int heads, int causal, void* dqkv, uint32_t drop_seed, uint32_t drop_threshold,
                        float drop_scale, const int32_t* cu, void* stream);
}  // namespace pgca

extern "C" int pgca_attention_fwd(const void* qkv, const int32_t* key_mask, int32_t B, int32_t S, int32_t heads,
                                  int32_t causal, void* out, float* lse, uint32_t drop_seed, uint32_t drop_threshold,
                                  float drop_scale, const int32_t* cu_seqlens, void* stream) {
  if (!qkv || !out || B <= 0 || S <= 0 || heads <= 0 || B > 65535) {
    set_error("pgca_attention_fwd: bad arguments (B=%d S=%d heads=%d)", B, S, heads);
    return PGCA_ERR_INVALID;
  }
  return attention_fwd_tiled(qkv, key_mask, B, S, heads, causal, out, lse, drop_seed, drop_threshold, drop_scale,
                             cu_seqlens, stream);
}

extern "C" int pgca_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                  const int32_t* key_mask, int32_t B, int32_t S, int32_t heads, int32_t causal,
                                  void* dqkv, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale,
                                  const int32_t* cu_seqlens, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || S <= 0 || S > PGCA_ATTN_MAX_S || heads <= 0 || B > 65535) {
    set_error("pgca_attention_bwd: bad arguments (B=%d S=%d heads=%d; S must be <= %d)", B, S, heads, PGCA_ATTN_MAX_S);
    return PGCA_ERR_INVALID;
  }
  return attention_bwd_tiled(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, drop_seed, drop_threshold,
                             drop_scale, cu_seqlens, stream);
}
