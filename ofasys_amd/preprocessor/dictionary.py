"""Symbol <-> index mapping (reference: preprocessor/dictionary.py:21-140), the part the model needs:
specials bos=0, pad=1, eos=2, unk=3, `add_symbol`, `index`, `len`, and <bin>_k box tokens
(preprocessor/default/box.py:37-124)."""


class Dictionary:
    def __init__(self, *, bos="<s>", pad="<pad>", eos="</s>", unk="<unk>", extra_special_symbols=None):
        self.bos_word, self.unk_word, self.pad_word, self.eos_word = bos, unk, pad, eos
        self.symbols, self.count, self.indices = [], [], {}
        self.bos_index = self.add_symbol(bos)
        self.pad_index = self.add_symbol(pad)
        self.eos_index = self.add_symbol(eos)
        self.unk_index = self.add_symbol(unk)
        for s in extra_special_symbols or []:
            self.add_symbol(s)
        self.nspecial = len(self.symbols)

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, idx):
        return self.symbols[idx] if idx < len(self.symbols) else self.unk_word

    def __contains__(self, sym):
        return sym in self.indices

    def __eq__(self, other):
        return self.indices == other.indices

    def index(self, sym):
        return self.indices.get(sym, self.unk_index)

    def add_symbol(self, word, n=1, overwrite=False, check=True):
        """Index of `word`, appended at the end when it is new (or `overwrite`: a second entry shadows the first in the lookup, the
        reference's semantics for duplicate lines of a dictionary file); a known word only has its count raised by n."""
        known = self.indices.get(word)
        if known is not None and not overwrite:
            self.count[known] += n
            return known
        self.symbols.append(word)
        self.count.append(n)
        self.indices[word] = len(self.symbols) - 1
        return self.indices[word]

    def get_start_end_idx(self, prefix: str):
        """[start, end) spanned by the symbols carrying `prefix` -- first and last occurrence, whatever lies between them; (-1, -1)
        when there is none (the contract of preprocessor/dictionary.py:66-74: the `<bin>_` / `<code>_` ranges of the vocabulary)."""
        hits = [i for i, token in enumerate(self.symbols) if token.startswith(prefix)]
        return (hits[0], hits[-1] + 1) if hits else (-1, -1)

    def encode_line(self, line, add_if_not_exist=True, append_eos=True, reverse_order=False):
        """Whitespace-split symbols -> int32 ids (preprocessor/dictionary.py:322-347)."""
        import torch
        words = line.strip().split()
        if reverse_order:
            words = list(reversed(words))
        ids = [self.add_symbol(w) if add_if_not_exist else self.index(w) for w in words]
        if append_eos:
            ids.append(self.eos_index)
        return torch.tensor(ids, dtype=torch.int32)

    def encode(self, words, add_if_not_exist=True, append_eos=True, reverse_order=False):
        """Dictionary-relative text ids -> global ids: word w maps to the symbol '<text>_w' (preprocessor/dictionary.py:349-374)."""
        import torch
        words = [str(w) for w in (reversed(list(words)) if reverse_order else words)]
        ids = [self.add_symbol(w) if add_if_not_exist else self.index("<text>_" + w) for w in words]
        if append_eos:
            ids.append(self.eos_index)
        return torch.tensor(ids, dtype=torch.int32)

    def bos(self):
        return self.bos_index

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    # ---- box <-> <bin>_k tokens: preprocessor/default/box.py:42-45, 101-110, 119-124 (integer, bit-exact)
    def add_bins(self, num_bins):
        self.bin_start = len(self.symbols)
        for i in range(num_bins):
            self.add_symbol(f"<bin>_{i}")
        self.num_bins = num_bins
        return self.bin_start

    def box_to_tokens(self, coords, max_image_size):
        import numpy as np   # float32 arithmetic + round-half-to-even, as the reference's tensor expression
        return [self.bin_start + int(np.round(np.float32(c) / np.float32(max_image_size) * np.float32(self.num_bins - 1)))
                for c in coords]

    def tokens_to_box(self, tokens, max_image_size):
        return [(t - self.bin_start) / (self.num_bins - 1) * max_image_size for t in tokens]
