"""Line datasets and the sequential eval loader of the retriever (``dataloader/retriever.py``), host side.

Batching semantics are parity-critical: file order, blank lines dropped, left-truncation to
the last ``block_size`` tokens, batches of ``per_gpu_eval_batch_size * max(1, n_gpu)`` right-padded with
[PAD] to the batch max, ``drop_last=False``.
"""
import os

import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset, SequentialSampler


def read_nonblank_lines(file_path):
    assert os.path.isfile(file_path)
    with open(file_path, encoding="utf-8") as f:
        return [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]


class LineByLineTextDataset(Dataset):
    """Full line per example -- ``dataloader/retriever.py:12-35`` (queries: val/test files are history-only)."""

    def __init__(self, tokenizer, args, file_path, block_size=512):
        lines = read_nonblank_lines(file_path)
        self.examples = tokenizer.batch_encode_plus(lines, add_special_tokens=True, max_length=block_size,
                                                    truncation='longest_first')["input_ids"]

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        return torch.tensor(self.examples[i], dtype=torch.long)


class LineByLineTextDatasetHistory(LineByLineTextDataset):
    """Text before ``<|pre|>`` only -- ``dataloader/retriever.py:37-65`` (pool side)."""

    def __init__(self, tokenizer, args, file_path, block_size=512):
        lines = [line.split('<|pre|>')[0].strip() for line in read_nonblank_lines(file_path)]
        print('line0: ', lines[0])
        self.examples = tokenizer.batch_encode_plus(lines, add_special_tokens=True, max_length=block_size,
                                                    truncation='longest_first')["input_ids"]


def load_and_cache_examples(args, tokenizer, evaluate=False, test=False, pair=True):
    """``dataloader/retriever.py:113-125``: the (anchor, positive, negative) triples of ``--train_pair_data_file`` for the
    training step (``rag4dyg_amd.training``, forward half), line datasets otherwise."""
    if evaluate:
        file_path = args.eval_data_file
    elif test:
        file_path = args.test_data_file
    elif pair:
        from .training import PairSequenceDataset
        return PairSequenceDataset(tokenizer, args, file_path=args.train_pair_data_file, block_size=args.block_size)
    else:
        file_path = args.train_data_file
    return LineByLineTextDataset(tokenizer, args, file_path=file_path, block_size=args.block_size)


def get_dataloader(dataset, tokenizer, args, split='eval'):
    """``dataloader/retriever.py:128-168``: sequential right-padded eval batches; shuffled six-tensor training batches
    (anchor / positive / negative sequences and their pool indices, all padded with the [PAD] id like upstream)."""
    pad = {} if tokenizer.pad_token is None else {"padding_value": tokenizer.pad_token_id}

    def collate(examples):
        if split == 'train':
            return tuple(pad_sequence([ex[j] for ex in examples], batch_first=True, **pad) for j in range(6))
        return pad_sequence(examples, batch_first=True, **pad)

    if split == 'train':
        from torch.utils.data import RandomSampler
        args.train_batch_size = args.per_gpu_train_batch_size * max(1, args.n_gpu)
        if getattr(args, "data_parallel_world", 1) <= 1:
            sampler = RandomSampler(dataset)
        else:                                                       # one process per GPU: each rank its share of the triples
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dataset, num_replicas=int(args.data_parallel_world),
                                         rank=int(getattr(args, "data_parallel_rank", 0)))
        loader = DataLoader(dataset, sampler=sampler, batch_size=args.train_batch_size, collate_fn=collate, drop_last=False)
        return loader, args
    args.eval_batch_size = args.per_gpu_eval_batch_size * max(1, args.n_gpu)
    loader = DataLoader(dataset, sampler=SequentialSampler(dataset), batch_size=args.eval_batch_size,
                        collate_fn=collate, drop_last=False)
    return loader, args
