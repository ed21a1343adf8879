from .layers import Dropout, DropPath, Embedding, LayerNorm, Linear, OfaEmbedding, OfaLayerNorm, OfaLinear, init_bert_params
from .multihead_attention import MultiheadAttention
from .transformer_layer import TransformerDecoderLayer, TransformerEncoderLayer

__all__ = ["Dropout", "DropPath", "Embedding", "LayerNorm", "Linear", "MultiheadAttention", "TransformerDecoderLayer",
           "TransformerEncoderLayer", "init_bert_params", "OfaEmbedding", "OfaLayerNorm", "OfaLinear"]
