"""TEST-ONLY stack backend: lets the host-side logic of `Transformer` (pipeline partitioning, rank-filtered
weight loading, send/recv/broadcast order, cache bookkeeping, generate()) run on CPU under gloo by
delegating the layer stack to the oracle.  Lives under tests/ on purpose: the product ships only
HipStackBackend and has no CPU path."""
import torch
import torch.nn.functional as F

import mistral_oracle as mo


class OracleStackBackend:
    HIST_BRANCH_DECODE = 2  # mi_branch MI_BRANCH_DECODE

    def __init__(self, fail_from_step=None):
        self.steps = 0  # decode-branch calls so far: the workspace's step counter of the HIP backend (status word 5)
        # sabotage (tests): from decode step `fail_from_step` on this stage behaves like an engine launch that failed its
        # residency gate - it writes NOTHING (no activations, no ring row, no sample, no counter) and raises status 0x700
        self.fail_from_step = fail_from_step
        self.status = 0

    def invalidate(self):
        pass

    # -- the three hooks GreedySession needs from a backend (HipStackBackend: the workspace's control words)
    def session_status(self):
        return {"steps": self.steps, "status": self.status, "arrivals": 0, "engine_launches": 0}

    def session_rewind(self, steps):
        self.steps = steps

    def prepare_session(self, model, B, cache):
        pass

    def session_disable_engine(self):
        self.status, self.fail_from_step = 0, None

    def run_stack(self, model, h, input_ids, meta, cache, logits, greedy=None):
        if meta.branch == self.HIST_BRANCH_DECODE and (self.status != 0 or (self.fail_from_step is not None and self.steps >= self.fail_from_step)):
            self.status = 0x700   # (sticky: every later "launch" on this workspace leaves at once)
            return
        a = model.args
        oargs = mo.OracleArgs(dim=a.dim, n_layers=a.n_layers, head_dim=a.head_dim, hidden_dim=a.hidden_dim,
                              n_heads=a.n_heads, n_kv_heads=a.n_kv_heads, norm_eps=a.norm_eps, vocab_size=a.vocab_size,
                              rope_theta=a.rope_theta, num_experts=a.moe.num_experts if a.moe else 0,
                              num_experts_per_tok=a.moe.num_experts_per_tok if a.moe else 0,
                              sliding_window=a.sliding_window)
        om = mo.OracleModel(oargs, dict(model.state_dict()), model.pipeline_rank, model.num_pipeline_ranks)
        ocache = None
        if cache is not None:
            ocache = getattr(cache, "_oracle", None)
            if ocache is None or cache._seen is None or all(p == 0 for p in cache._seen):
                ocache = mo.OracleCache(model.n_local_layers, cache.max_batch_size, cache.max_seq_len, cache.n_kv_heads,
                                        cache.head_dim, a.sliding_window, dtype=model.dtype)
                cache._oracle = ocache
            ocache.seen = list(cache._seen)
        seqlens = meta.seqlens if cache is not None else None
        if cache is None:
            # NOCACHE: meta.seqlens is [T]; positions restart per original sequence -> recover from tok_pos
            pos = meta.tok_pos.tolist()
            seqlens, run = [], 0
            for i, p in enumerate(pos):
                if p == 0 and i:
                    seqlens.append(run)
                    run = 0
                run += 1
            seqlens.append(run)
        ids = input_ids if input_ids is not None else torch.zeros(h.shape[0], dtype=torch.long)
        # input_ids present: the stack embeds them; absent: h is the input (later pipeline rank / multimodal embeddings)
        out = om.forward_partial(ids, seqlens, ocache, h_in=None if input_ids is not None else h.clone())
        if logits is not None:
            logits.copy_(F.linear(out, om.w["output.weight"]).float())
        h.copy_(out)
        if meta.branch == self.HIST_BRANCH_DECODE:
            self.steps += 1
        if greedy is not None:  # mi_batch_t.greedy_token & co. (ABI v4): argmax + log-softmax behind the LM head, history ring
            assert logits is not None and greedy.temperature == 0
            tok = torch.argmax(logits, dim=-1)
            lp = torch.log_softmax(logits, dim=-1).gather(1, tok[:, None])[:, 0]
            greedy.tok.copy_(tok)
            greedy.lp.copy_(lp)
            row = (self.steps - 1) % greedy.hist_tok.shape[0]
            greedy.hist_tok[row].copy_(tok)
            greedy.hist_lp[row].copy_(lp)
