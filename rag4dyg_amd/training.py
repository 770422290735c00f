"""Retriever TRAINING step, forward side (staged): the data, the augmentation, the five encoder
forwards on the gfx950 kernels and the two contrastive losses of one ``train_epoch`` iteration.

Mirrors ``dataloader/retriever.py:68-111`` (``PairSequenceDataset``), ``models/modeling_rag.py:774-840`` (``_aug``) and
``train/train_retriever.py:40-98,177-196`` (``CLtime_loss``, ``mask_correlated_samples``, ``info_nce``, the step).  The
BACKWARD pass (dgrad / wgrad of the GEMMs, LayerNorm, GELU and attention backward) and the optimizer are not built yet:
``main_retriever.py --do_train`` still raises.  What is here runs the forward half exactly as the reference computes it in
``model.eval()`` terms -- dropout is the identity (the reference trains with p = 0.1 drawn from its device RNG, which no
other device reproduces) -- so that the loss values can be checked against the reference before any backward kernel exists.
The five batches of a step (anchor, positive, hard negative, two augmented views) go through ONE fused launch sequence
(``r4d_gpt2_encode_groups_f32``); the losses themselves are [B, 3B] / [2B, 2B] similarity tables: torch ops on the device.
"""
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from .dataloader import read_nonblank_lines


class PairSequenceDataset(torch.utils.data.Dataset):
    """``dataloader/retriever.py:68-111``: (anchor, positive, negative) index triples of ``train_index.retrieval`` resolved
    against the history part of the training lines."""

    def __init__(self, tokenizer, args, file_path, block_size=512):
        assert os.path.isfile(file_path)
        train_lines = [line.split('<|pre|>')[0].strip() for line in read_nonblank_lines(args.train_data_file)]
        triples = [list(map(int, line.split())) for line in read_nonblank_lines(file_path)]
        examples = tokenizer(train_lines, add_special_tokens=True, max_length=block_size)["input_ids"]
        self.anchor, self.positive, self.negative = [], [], []
        self.anchor_idx, self.positive_idx, self.negative_idx = [], [], []
        for a, p, n in triples:
            self.anchor.append(examples[a]); self.positive.append(examples[p]); self.negative.append(examples[n])
            self.anchor_idx.append([a]); self.positive_idx.append([p]); self.negative_idx.append([n])

    def __len__(self):
        return len(self.anchor)

    def __getitem__(self, i):
        t = lambda x: torch.tensor(x, dtype=torch.long)
        return (t(self.anchor[i]), t(self.positive[i]), t(self.negative[i]), t(self.anchor_idx[i]), t(self.positive_idx[i]),
                t(self.negative_idx[i]))


def aug(batch_seqs, eta, gamma, mask_token):
    """``_aug`` (``models/modeling_rag.py:774-840``): view 1 = crop (``eta``), view 2 = mask (``gamma``), python ``random``
    as upstream.  Kept as is: a row's length is its number of non-zero ids; the crop lands at the end of an all-zero row;
    the mask view overwrites the row it was taken from (``seq[:]`` of an ndarray is a view)."""
    seqs = batch_seqs.tolist()
    lengths = batch_seqs.count_nonzero(dim=1).tolist()
    view1, view2 = [], []
    for seq, length in zip(seqs, lengths):
        seq = np.asarray(seq.copy(), dtype=np.int64)
        if length > 1:
            num_left = math.floor(length * eta)                              # item_crop
            crop_begin = random.randint(4, length - num_left)
            c = np.zeros_like(seq)
            if crop_begin != 0:
                c[-num_left:] = seq[-(crop_begin + num_left):-crop_begin]
            else:
                c[-num_left:] = seq[-(crop_begin + num_left):]
            view1.append(c.tolist() if num_left > 0 else seq.tolist())
            num_mask = math.floor(length * gamma)                            # item_mask
            mask_index = [-i - 1 for i in random.sample(range(length), k=num_mask)]
            seq[mask_index] = mask_token
            view2.append(seq.tolist())
        else:
            view1.append(seq.tolist()); view2.append(seq.tolist())
    dev = batch_seqs.device
    return torch.tensor(view1, dtype=torch.long, device=dev), torch.tensor(view2, dtype=torch.long, device=dev)


def CLtime_loss(args, anchors, positives, hard_negatives, anchors_time, positives_time, negatives_time):
    """``train/train_retriever.py:40-72``."""
    B = anchors.size(0)
    allv = torch.cat([anchors, positives, hard_negatives], dim=0)
    sim = F.cosine_similarity(allv.unsqueeze(1), allv.unsqueeze(0), dim=2)
    dev = anchors.device
    d_pos = torch.exp(-args.lambda_decay * torch.abs(anchors_time.unsqueeze(1) - positives_time).squeeze()).to(dev)
    d_neg = torch.exp(-args.lambda_decay * torch.abs(anchors_time.unsqueeze(1) - anchors_time).squeeze())
    d_neg.fill_diagonal_(0)
    d_neg = d_neg.to(dev)
    d_hard = torch.exp(-args.lambda_decay * torch.abs(anchors_time.unsqueeze(1) - negatives_time).squeeze()).to(dev)
    logits = torch.cat([sim[:B, B:2 * B] * d_pos, sim[:B, :B] * d_neg, sim[:B, 2 * B:] * d_hard], dim=1) / args.temperature
    return F.cross_entropy(logits, torch.arange(B, device=dev))


def mask_correlated_samples(batch_size):
    """``train/train_retriever.py:74-82``."""
    N = 2 * batch_size
    mask = torch.ones((N, N), dtype=bool)
    mask = mask.fill_diagonal_(0)
    for i in range(batch_size):
        mask[i, batch_size + i] = 0
        mask[batch_size + i, i] = 0
    return mask


def info_nce(args, z_i, z_j, temp, batch_size, mask):
    """``train/train_retriever.py:84-98`` (the mask is rebuilt for a last, smaller batch, :92-93)."""
    N = 2 * batch_size
    z = torch.cat((z_i, z_j), dim=0)
    sim = torch.mm(z, z.T) / temp
    positive = torch.cat((torch.diag(sim, batch_size), torch.diag(sim, -batch_size)), dim=0).reshape(N, 1)
    if mask is None or batch_size != args.per_gpu_train_batch_size:
        mask = mask_correlated_samples(batch_size)
    negative = sim[mask.to(sim.device)].reshape(N, -1)
    labels = torch.zeros(N, device=sim.device).long()
    return F.cross_entropy(torch.cat((positive, negative), dim=1), labels)


@torch.no_grad()
def training_step_forward(args, model, batch, all_query_time, mask_nce=None):
    """Forward half of one ``train_epoch`` iteration (``train/train_retriever.py:164-196``): ``batch`` = (anchor, positive,
    negative, anchor_idx, positive_idx, negative_idx) as the ``PairSequenceDataset`` loader yields them.  Returns
    dict(cl_loss, aug_loss, loss, embeddings [5, B, d], aug1, aug2)."""
    anchor_seq, pos_seq, neg_seq, anchor_idx, pos_idx, neg_idx = batch
    dev = args.device
    anchor_seq, pos_seq, neg_seq = anchor_seq.to(dev), pos_seq.to(dev), neg_seq.to(dev)
    aug1, aug2 = aug(anchor_seq, model.config.eta, model.config.gamma, model.config.vocab_size - 1)
    B = anchor_seq.size(0)
    emb = model.encode_groups_meanpool([anchor_seq, pos_seq, neg_seq, aug1, aug2]).view(5, B, -1)   # one fused launch sequence
    t = all_query_time
    cl = CLtime_loss(args, emb[0], emb[1], emb[2], t[anchor_idx], t[pos_idx], t[neg_idx])
    au = args.alpha * info_nce(args, emb[3], emb[4], args.temperature, B, mask_nce)
    return dict(cl_loss=cl, aug_loss=au, loss=cl + au, embeddings=emb, aug1=aug1, aug2=aug2)
