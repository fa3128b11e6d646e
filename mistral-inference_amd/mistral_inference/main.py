"""`mistral-chat` / `mistral-demo` entry points (reference main.py:102-283).

Distributed contract unchanged (reference main.py:36-38,110-118): launched by torchrun with one process
per GPU; the world size is the number of pipeline stages; rank r drives device r.  `init_process_group`
gets the explicit backend "nccl" -- on ROCm that is RCCL, which runs the stage-to-stage send/recv and the
logits broadcast over xGMI.  Tokenisation is delegated to `mistral_common` exactly as in the reference and
is imported lazily (it is CPU string processing outside the hot path; benchmarks feed token ids).
"""
import json
import logging
import os
from pathlib import Path
from typing import List, Optional, Type, Union

import torch

from .generate import generate  # noqa: F401  (the reference's tests import generate from main)
from .transformer import Transformer


def is_torchrun() -> bool:
    return all(v in os.environ for v in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"))


def init_pipeline() -> int:
    """Process-group setup of reference main.py:110-118; returns the number of pipeline ranks."""
    if not is_torchrun():
        return 1
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    torch.distributed.init_process_group(backend=backend)
    if torch.cuda.is_available():
        torch.cuda.set_device(torch.distributed.get_rank() % max(1, torch.cuda.device_count()))
    return torch.distributed.get_world_size()


def share_prompt_length(n_tokens: int) -> int:
    """Rank 0's prompt length on every pipeline rank (reference main.py:161-170: only the LENGTH is shared, the other
    ranks feed that many placeholder ids).  RCCL ("nccl") moves device memory only, so the one-element tensor lives
    on the current device for the broadcast and its RECEIVED value is what is returned."""
    buf = torch.tensor([n_tokens], dtype=torch.int)
    if torch.distributed.get_backend() == "nccl":
        buf = buf.cuda()
    torch.distributed.broadcast(buf, src=0)
    return int(buf.item())


def _should_print() -> bool:
    return not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0


def _require(mod: str):
    import importlib
    try:
        return importlib.import_module(mod)
    except ImportError as e:  # pragma: no cover
        raise RuntimeError(f"the CLI needs `{mod}` (pip install mistral_common fire); "
                           "library use (Transformer / generate) does not") from e


def load_tokenizer(model_path: Path):
    MistralTokenizer = _require("mistral_common.tokens.tokenizers.mistral").MistralTokenizer
    files = [f for f in os.listdir(model_path) if f.startswith("tekken.json") or f.startswith("tokenizer.model")]
    assert len(files) > 0, f"No tokenizer in {model_path}"
    assert len(files) == 1, f"Multiple tokenizers {', '.join(files)} found in `model_path`"
    tok = MistralTokenizer.from_file(str(model_path / files[0]))
    logging.info("Loaded tokenizer of type %s", tok.instruct_tokenizer.__class__)
    return tok


def get_model_cls(model_path: str) -> Type[Transformer]:
    with open(Path(model_path) / "params.json", "r") as f:
        kind = json.load(f).get("model_type", "transformer")
    if kind != "transformer":
        raise NotImplementedError(f"model_type {kind!r}: only the transformer family is on the hot path")
    return Transformer


def interactive(model_path: str, max_tokens: int = 35, temperature: float = 0.7, num_pipeline_ranks: int = 1,
                instruct: bool = False, lora_path: Optional[str] = None) -> None:
    num_pipeline_ranks = init_pipeline() if is_torchrun() else num_pipeline_ranks
    should_print = _should_print()
    mistral_tokenizer = load_tokenizer(Path(model_path))
    tokenizer = mistral_tokenizer.instruct_tokenizer.tokenizer
    model = get_model_cls(model_path).from_folder(Path(model_path), max_batch_size=3,
                                                  num_pipeline_ranks=num_pipeline_ranks, dtype=torch.bfloat16)
    if lora_path is not None:  # reference main.py:131-132
        model.load_lora(Path(lora_path))
    messages: List = []
    while True:
        tokens: List[int] = []
        if should_print:
            user_input = input("Prompt: ")
            if instruct:
                req = _require("mistral_common.protocol.instruct.request")
                msg = _require("mistral_common.protocol.instruct.messages")
                messages += [msg.UserMessage(content=user_input)]
                tokens = mistral_tokenizer.encode_chat_completion(req.ChatCompletionRequest(messages=messages)).tokens
            else:
                tokens = tokenizer.encode(user_input, bos=True, eos=False)
        if is_torchrun():
            # only the prompt LENGTH is shared; other ranks feed dummy ids (reference main.py:161-170)
            n = share_prompt_length(len(tokens))
            if not should_print:
                tokens = n * [0]
        generated, _ = generate([tokens], model, max_tokens=max_tokens, temperature=temperature,
                                eos_id=tokenizer.eos_id)
        answer = tokenizer.decode(generated[0])
        if should_print:
            print(answer)
            print("=====================")
        if instruct:
            msg = _require("mistral_common.protocol.instruct.messages")
            messages += [msg.AssistantMessage(content=answer)]


def demo(model_path: str, max_tokens: int = 35, temperature: float = 0, lora_path: Optional[str] = None) -> None:
    num_pipeline_ranks = init_pipeline()
    should_print = _should_print()
    model = get_model_cls(model_path).from_folder(Path(model_path), max_batch_size=3,
                                                  num_pipeline_ranks=num_pipeline_ranks, dtype=torch.bfloat16)
    if lora_path is not None:  # reference main.py:224-225
        model.load_lora(Path(lora_path))
    tokenizer = load_tokenizer(Path(model_path)).instruct_tokenizer.tokenizer
    prompts = ["This is a test", "This is another great test", "This is a third test, mistral AI is very good at testing. "]
    encoded = [tokenizer.encode(p, bos=True, eos=False) for p in prompts]
    generated, logprobs = generate(encoded, model, max_tokens=max_tokens, temperature=temperature,
                                   eos_id=tokenizer.eos_id)
    if should_print:
        for p, g, lp in zip(prompts, generated, logprobs):
            print(p + tokenizer.decode(g))
            logging.debug("logprobs: %s", lp)
            print("=====================")


def mistral_chat() -> None:
    _require("fire").Fire(interactive)


def mistral_demo() -> None:
    _require("fire").Fire(demo)


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    _require("fire").Fire({"interactive": interactive, "demo": demo})
