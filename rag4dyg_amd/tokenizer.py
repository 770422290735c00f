"""WordLevel tokenizer + model construction with the reference's token-id and file layout.

Mirrors ``utils/tokenizer.py:10-68`` (and its near-duplicates ``utils/tokenizer_generator.py``,
``main_SimpleDyG.py:53-123``): vocab from ``vocabs/<ds>/<t>/vocab.json`` (``"id": id``), Whitespace
pre-tokenizer, no normaliser / post-processor, no UNK (an out-of-vocabulary word raises), then the special
tokens in the reference's order:  <|endoftext|>=V0 (bos == eos), <|history|>, <|endofhistory|>, <|pre|>,
<|endofpre|>, <|time0..t|>, [PAD], and (retriever only, ``utils/tokenizer.py:48``) [MASK].
Saved as ``tokenizers/<ds>/<t>/{tokenizer.json,special_tokens_map.json,tokenizer_config.json}`` in the HF
tokenizers JSON schema, so either side can read the other's files.  Host logic only -- no device code.
"""
import json
import os
import re

import numpy as np
import torch

_WORD = re.compile(r"\w+|[^\w\s]+")          # tokenizers.pre_tokenizers.Whitespace


class WordLevelTokenizer:
    def __init__(self, vocab, added_tokens=()):
        self.vocab = dict(vocab)                         # base WordLevel vocab (token -> id)
        self.added = []                                  # [(content, id)] in insertion order
        self.bos_token = self.eos_token = self.pad_token = self.mask_token = None
        self.additional_special_tokens = []
        self.truncation_side = "left"                    # utils/tokenizer.py:43
        self.max_len = 1024
        self._rebuild()
        for content in added_tokens:
            self._add(content)

    # ------------------------------------------------------------------ construction
    def _rebuild(self):
        self._added_map = dict(self.added)
        self._id2tok = {i: t for t, i in self.vocab.items()}
        self._id2tok.update({i: t for t, i in self.added})
        if self.added:
            alt = "|".join(re.escape(t) for t, _ in sorted(self.added, key=lambda x: -len(x[0])))
            self._split = re.compile("(" + alt + ")")
        else:
            self._split = None

    def _add(self, content):
        if content in self._added_map or content in self.vocab:
            return 0
        self.added.append((content, len(self.vocab) + len(self.added)))
        self._rebuild()
        return 1

    def add_special_tokens(self, mapping):
        """Subset of PreTrainedTokenizer.add_special_tokens used at ``utils/tokenizer.py:44-48``."""
        n = 0
        for key, val in mapping.items():
            if key == "additional_special_tokens":
                for t in val:
                    n += self._add(t)
                self.additional_special_tokens = list(val)
            else:
                n += self._add(val)
                setattr(self, key, val)
        return n

    @property
    def vocab_size(self):                                # size of the base vocabulary (without added tokens)
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab) + len(self.added)

    def convert_tokens_to_ids(self, tok):
        if isinstance(tok, (list, tuple)):
            return [self.convert_tokens_to_ids(t) for t in tok]
        if tok in self._added_map:
            return self._added_map[tok]
        return self.vocab[tok]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, (list, tuple)):
            return [self._id2tok[int(i)] for i in ids]
        return self._id2tok[int(ids)]

    for _name in ("bos", "eos", "pad", "mask"):
        locals()[_name + "_token_id"] = property(
            lambda self, _n=_name: None if getattr(self, _n + "_token") is None
            else self.convert_tokens_to_ids(getattr(self, _n + "_token")))
    del _name

    # ------------------------------------------------------------------ encoding
    def encode(self, text, max_length=None, truncation=True):
        ids = []
        parts = self._split.split(text) if self._split else [text]
        for part in parts:
            if part in self._added_map:
                ids.append(self._added_map[part])
                continue
            for w in _WORD.findall(part):
                try:
                    ids.append(self.vocab[w])
                except KeyError:
                    raise ValueError(f"WordLevel error: token {w!r} is not in the vocabulary and there is no UNK token")
        if truncation and max_length is not None and max_length > 0 and len(ids) > max_length:
            ids = ids[-max_length:] if self.truncation_side == "left" else ids[:max_length]
        return ids

    def __call__(self, lines, add_special_tokens=True, max_length=None, truncation=True, **_):
        """``tokenizer(lines, ...)["input_ids"]`` (batch) -- ``dataloader/retriever.py:23``."""
        if isinstance(lines, str):
            return {"input_ids": self.encode(lines, max_length, bool(truncation))}
        return {"input_ids": [self.encode(line, max_length, bool(truncation)) for line in lines]}

    batch_encode_plus = __call__                         # dataloader/retriever.py:23,53,90

    def decode(self, ids, skip_special_tokens=False):
        toks = [self._id2tok[int(i)] for i in ids]
        if skip_special_tokens:
            toks = [t for t in toks if t not in self._added_map]
        return " ".join(toks)

    # ------------------------------------------------------------------ files
    def to_json(self):
        added = [{"id": i, "content": t, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False,
                  "special": True} for t, i in self.added]
        return {"version": "1.0", "truncation": None, "padding": None, "added_tokens": added, "normalizer": None,
                "pre_tokenizer": {"type": "Whitespace"}, "post_processor": None, "decoder": None,
                "model": {"type": "WordLevel", "vocab": self.vocab, "unk_token": "<unk>"}}

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "tokenizer.json"), "w") as f:
            json.dump(self.to_json(), f, indent=2)
        smap = {k: getattr(self, k) for k in ("bos_token", "eos_token", "pad_token", "mask_token") if getattr(self, k)}
        if self.additional_special_tokens:
            smap["additional_special_tokens"] = self.additional_special_tokens
        with open(os.path.join(path, "special_tokens_map.json"), "w") as f:
            json.dump(smap, f, indent=2)
        with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
            json.dump(dict(smap, tokenizer_class="PreTrainedTokenizerFast", truncation_side=self.truncation_side,
                           model_max_length=self.max_len), f, indent=2)
        return (os.path.join(path, "tokenizer.json"),)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "tokenizer.json")) as f:
            j = json.load(f)
        tok = cls(j["model"]["vocab"])
        for a in sorted(j.get("added_tokens", []), key=lambda a: a["id"]):
            tok.added.append((a["content"], a["id"]))
        tok._rebuild()
        smf = os.path.join(path, "special_tokens_map.json")
        if os.path.exists(smf):
            with open(smf) as f:
                for k, v in json.load(f).items():
                    v = [x["content"] if isinstance(x, dict) else x for x in v] if isinstance(v, list) else \
                        (v["content"] if isinstance(v, dict) else v)
                    setattr(tok, k, v)
        return tok


def build_tokenizer(dataset, timestamp, with_mask=True, root="."):
    """Tokenizer of ``utils/tokenizer.py:27-49`` (``with_mask=False``: ``main_SimpleDyG.py:70-95`` / generator)."""
    spl_tokens = ['<|history|>', '<|endofhistory|>', '<|pre|>', '<|endofpre|>'] + \
                 ['<|time' + str(i) + '|>' for i in range(int(timestamp) + 1)]
    with open(os.path.join(root, "vocabs", dataset, str(timestamp), "vocab.json")) as f:
        vocab = json.load(f)
    tok = WordLevelTokenizer(vocab)
    tok.add_special_tokens({'bos_token': '<|endoftext|>'})
    tok.add_special_tokens({'eos_token': '<|endoftext|>'})
    tok.add_special_tokens({'additional_special_tokens': spl_tokens})
    tok.add_special_tokens({'pad_token': '[PAD]'})
    if with_mask:
        tok.add_special_tokens({'mask_token': '[MASK]'})
    return tok, spl_tokens


def get_model_tokenizer(args, MODEL_CLASSES):
    """Drop-in for ``utils/tokenizer.get_model_tokenizer`` (``utils/tokenizer.py:10-68``).

    Deviation: ``--model_name_or_path gpt2`` / ``--config_name gpt2`` name a hub model the reference downloads
    (:13-16); there is no network here, so any non-directory name falls back to the default ``GPT2Config()``,
    whose values equal the hub ``gpt2`` config for every field the path reads (n_ctx = n_positions = 1024,
    eps 1e-5) before the architecture flags overwrite n_head / n_layer / n_embd.
    """
    config_class, model_class, _tok_class = MODEL_CLASSES[args.model_type]
    name = getattr(args, "config_name", None) or getattr(args, "model_name_or_path", None)
    if name and os.path.isdir(name):
        config = config_class.from_pretrained(name)
    else:
        config = config_class()
    config.n_head = args.n_head
    config.n_layer = args.n_layer
    config.n_embd = args.n_embed
    config.eta = getattr(args, "eta", 0.0)
    config.gamma = getattr(args, "gamma", 0.0)
    config.beta = getattr(args, "beta", 0.0)
    tok, spl = build_tokenizer(args.dataset, args.timestamp, with_mask=getattr(args, "with_mask_token", True))
    args.spl_tokens = spl
    tok.save_pretrained(os.path.join('./tokenizers/', args.dataset, str(args.timestamp)))
    print('vocab size: ', tok.vocab_size)
    config.max_token_id = tok.vocab_size
    # evaluation only: every parameter comes from a checkpoint (strict load_state_dict) -- no random draws (gpt2.skip_random_init)
    import contextlib
    from .gpt2 import skip_random_init
    eval_only = bool(getattr(args, "do_eval", False)) and not bool(getattr(args, "do_train", False)) and args.dataset != 'hepth'
    with (skip_random_init() if eval_only else contextlib.nullcontext()):
        model = model_class(config=config)
        model.to(args.device)
        model.resize_token_embeddings(len(tok))
    if args.dataset == 'hepth':                          # node-feature injection, utils/tokenizer.py:56-66
        feats = np.load(args.node_feat_file)[:tok.vocab_size]
        if feats.shape[1] < args.n_embed:
            feats = np.concatenate((feats, np.zeros((feats.shape[0], args.n_embed - feats.shape[1]))), axis=1)
        wte = model.transformer.wte.weight
        sp_feat = wte.data[tok.vocab_size:]
        weights = torch.cat([torch.as_tensor(feats, dtype=torch.float32, device=wte.device), sp_feat])
        model.transformer.wte = torch.nn.Embedding.from_pretrained(embeddings=weights, freeze=False)
        # NOT re-tied, exactly like the reference (utils/tokenizer.py:65, utils/tokenizer_generator.py:116): lm_head.weight
        # stays the resized random-init Parameter, trained separately from the injected node features.
    return model, tok, model_class, args
