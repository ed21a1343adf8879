"""ofasys_amd: MI355X-native (gfx950) implementation of the OFASys unified encoder-decoder hot path."""
__version__ = "0.1.0"
