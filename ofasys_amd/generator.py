"""Step-wise incremental decoding driver: the model side of the reference's generation loop
(generator/sequence_generator.py:258-306 -- encoder once, then one decoder call per step with an `incremental_state`,
beams reordered in between) with the launch-bound step captured into hipGraphs.

One decoding step of OFA-base is ~130 small kernels; launched eagerly from Python it takes 2.65 ms whether 32 or 160 rows
are decoded (host-bound).  `StepDecoder` keeps every buffer a step touches at a fixed address -- target tokens, encoder
output, the KV caches at their final capacity -- and records ONE hipGraph PER STEP LENGTH (the prefix length and the cache
length are kernel arguments, so each length is its own graph; a sequence of T steps replays T graphs).  Graphs are keyed by
(rows, source length, step) and reused by every later batch of the same shape.  The beam-search policy itself (scoring,
length penalty, finalisation) stays out of scope; `greedy` below is the minimal loop used by tests and benchmarks.
"""
from typing import Dict, List, Optional

import torch

from .preprocessor import ModalityType, Slot


class StepDecoder:
    def __init__(self, model, max_len: int, use_graph: bool = True, warmup_sequences: int = 1):
        self.model, self.max_len, self.use_graph = model, int(max_len), use_graph
        self.warmup_sequences = warmup_sequences
        self._shape = None
        self._graphs: Dict[int, tuple] = {}
        self._pool = None
        self._seq = 0

    # ------------------------------------------------------------------ sequence state
    def begin(self, src_slots: List[Slot], beam_order: Optional[torch.Tensor] = None):
        """Encode the sources (eagerly), optionally expand to beams (`beam_order`: row index per output row), and point the
        static buffers at the new batch."""
        m = self.model
        m.eval()                                          # decoding is an inference-mode activity: the model STAYS in eval mode (call
        #                                                   model.train() before the next train step, as with the reference's generator)
        for mod in m.decoder.modules():                   # packed decode projections follow the current parameters
            if getattr(mod, "_decode_pack_cache", None) is not None:
                mod._decode_pack()
        with torch.no_grad():
            enc = m.encoder(src_slots)
            if beam_order is not None:
                enc = m.encoder.reorder_encoder_out(enc, beam_order)
        out = enc["encoder_out"][0]
        # captured step graphs bake in parameter ADDRESSES: a TrainStep built afterwards (its arenas re-point p.data) or a
        # model.to(...) must drop them, so the storage of one parameter is part of the cache key
        fingerprint = next(m.decoder.parameters()).data_ptr()
        shape = (out.shape[1], out.shape[0], out.dtype, fingerprint)
        if shape != self._shape:                          # another batch shape: new buffers, new graphs
            self._shape, self._graphs, self._pool, self._seq = shape, {}, None, 0
            self.enc = {k: [t.clone() if torch.is_tensor(t) else t for t in v] if isinstance(v, list) else v for k, v in enc.items()}
            self.tokens = torch.zeros(shape[0], self.max_len, dtype=torch.long, device=out.device)
            self.inc = {"__capacity__": self.max_len, "__static__": True}
        else:
            with torch.no_grad():
                for k, v in enc.items():
                    if isinstance(v, list):
                        for dst, srct in zip(self.enc[k], v):
                            if torch.is_tensor(dst):
                                dst.copy_(srct)
            for mod in m.decoder.modules():
                if hasattr(mod, "reset_incremental_state"):
                    mod.reset_incremental_state(self.inc)
            self._seq += 1
        self.t = 0
        return self

    def step(self, next_tokens: torch.Tensor) -> torch.Tensor:
        """Append one token per row and return the logits of that position: [rows, V]."""
        t = self.t
        if t >= self.max_len:
            raise ValueError(f"StepDecoder: max_len={self.max_len} exceeded")
        self.tokens[:, t].copy_(next_tokens.reshape(-1))
        graphed = self.use_graph and self._seq >= self.warmup_sequences
        if graphed and t in self._graphs:
            g, logits = self._graphs[t]
            self._set_lengths(t + 1)
            g.replay()
        elif graphed:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.no_grad(), torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                logits = self._run(t)
            if self._pool is None:
                self._pool = g.pool()
            self._graphs[t] = (g, logits)
            g.replay()                                    # capture only records: run the step once
        else:
            with torch.no_grad():
                logits = self._run(t)
        self.t = t + 1
        return logits

    def _run(self, t):
        out, _ = self.model.decoder([Slot(ModalityType.TEXT, False, self.tokens[:, :t + 1])], encoder_out=self.enc,
                                    incremental_state=self.inc)
        return out[:, -1]

    def _set_lengths(self, n):
        """The host-side cache lengths (only read when a step is recorded or run eagerly) follow the replayed steps."""
        for c in self.inc.values():
            if isinstance(c, dict) and "len" in c and not c.get("static"):
                c["len"] = n
        return n

    def reorder(self, new_order: torch.Tensor):
        """Beam reorder between steps: caches, encoder output and the token prefix follow `new_order` (in place)."""
        m = self.model
        if new_order.numel() != self._shape[0]:
            raise ValueError("StepDecoder.reorder keeps the row count (the buffers of the captured steps are fixed): "
                             f"got {new_order.numel()} indices for {self._shape[0]} rows")
        m.decoder.reorder_incremental_state_scripting(self.inc, new_order)
        enc = m.encoder.reorder_encoder_out(self.enc, new_order)
        for k, v in enc.items():
            if isinstance(v, list):
                for dst, srct in zip(self.enc[k], v):
                    if torch.is_tensor(dst):
                        dst.copy_(srct)
        self.tokens.copy_(self.tokens.index_select(0, new_order))

    # ------------------------------------------------------------------ minimal loop
    def greedy(self, src_slots: List[Slot], bos: int, steps: int, beam_order: Optional[torch.Tensor] = None):
        """Forced-length greedy decoding: returns tokens [rows, steps + 1] (bos first) and the per-step logits."""
        self.begin(src_slots, beam_order)
        rows = self._shape[0]
        nxt = torch.full((rows,), bos, dtype=torch.long, device=self.tokens.device)
        logits_all = []
        for _ in range(steps):
            logits = self.step(nxt)
            logits_all.append(logits.float().clone())
            nxt = logits.argmax(-1)
        toks = torch.cat([self.tokens[:, :steps], nxt.view(-1, 1)], 1)
        return toks, torch.stack(logits_all)
