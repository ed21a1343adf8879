from .ofa import GeneralistModel, GeneralistModelConfig, OFAEncoderDecoderExecutor, OFAExecutor
from .transformer import TransformerDecoder, TransformerEncoder

__all__ = ["GeneralistModel", "GeneralistModelConfig", "OFAEncoderDecoderExecutor", "OFAExecutor", "TransformerDecoder",
           "TransformerEncoder"]
