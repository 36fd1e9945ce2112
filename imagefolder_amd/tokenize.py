"""Bulk tokenisation / index emission: images -> code indices -> the two on-disk formats the generators read.

Mirrors the job of reference scripts/pretokenization.py:150-259 (per-rank loop: augment, tokenise, collect
{"class_id", "tokens"} records, dump `pretokenized_{rank}.json`, rank 0 merges them into `pretokenized.jsonl`) with the
XQ-GAN tokenizer on MI355X:

  * the code search runs through VQModel.img_to_idx — encoder + quant_conv + the fused HIP nearest-code kernels
    (f_to_idxBl_or_fhat(to_fhat=False)); the N x V distance matrix never exists, so inference batches of hundreds of images
    fit next to the model;
  * the device -> host copy of each batch's tokens goes to a pinned double buffer on a side stream and is only waited for
    one batch later: the encoder of batch t+1 runs while the tokens of batch t travel and are serialised;
  * formats: RAR  — one JSON object per line {"class_id": int, "tokens": [int, ...]}, read by PretoeknizedDataSetJSONL
                    (data/webdataset_reader.py:253-267) as torch.tensor(data["tokens"]);
             VAR / LlamaGen — `{code_dir}/{i}.npy` int codes of shape (1, n_aug, L) and `{label_dir}/{i}.npy` labels of shape (1,),
                    read by CustomDataset (dataset/imagenet.py:8-50: features[:, aug_idx] picks one augmentation).
  * token layout per image: for every product branch p (xqgan_model.py:126-134) the scales of its ladder in order
    (1 scale for VectorQuantizer, SN for VectorQuantizer2: the list f_to_idxBl_or_fhat returns), flattened and offset by
    p * codebook_size so that the P codebooks share one vocabulary of P * V ids (configs/VP2-16384.yaml: vocab 32768).

NB upstream's script calls `tokenizer.encode(...)`, which for the XQ-GAN VQModel returns the pre-quantisation latent map
(xqgan_model.py:241-255), not indices — it was written for the TiTok tokenizer; the indices come from img_to_idx here.
"""
import glob
import json
import os
from typing import Iterable, List, Optional, Tuple

import numpy as np
import torch


def augment_flip(samples: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """pretokenization.py:227-228: every image and its horizontal mirror"""
    return torch.cat([samples, torch.flip(samples, dims=[-1])]), torch.cat([target, target])


def augment_ten_crop(samples: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """pretokenization.py:224-225: (B, 10, 3, H, W) crops -> (10 B, 3, H, W), labels repeated"""
    return samples.flatten(0, 1), target.unsqueeze(1).repeat(1, samples.shape[1]).flatten(0, 1)


@torch.no_grad()
def tokens_from_images(model, imgs: torch.Tensor, amp_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(B, 3, H, W) in [-1, 1] -> int64 (B, L): L = P * sum_s pn_s^2 token ids in [0, P * V)."""
    with torch.autocast(device_type=imgs.device.type, dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
        branches = model.img_to_idx(imgs)          # list over product branches of lists over scales
    B = imgs.shape[0]
    V = model.quantizes[0].vocab_size if getattr(model, "product_quant", 1) > 1 else model.quantize.vocab_size
    out = []
    for p, scales in enumerate(branches):
        for idx in scales:
            out.append(idx.reshape(B, -1).to(torch.int64) + p * V)
    return torch.cat(out, dim=1)


class BulkTokenizer:
    """Streams (images, labels) batches through the tokenizer and collects the records of this rank."""

    def __init__(self, model, amp_dtype: Optional[torch.dtype] = None, augment: str = "flip"):
        if augment not in ("flip", "ten_crop", "none"):
            raise ValueError(f"augment must be flip / ten_crop / none, got {augment}")
        self.model = model.eval().requires_grad_(False)      # pretokenization.py:206-207
        self.amp_dtype = amp_dtype
        self.augment = augment
        self.class_ids: List[np.ndarray] = []
        self.tokens: List[np.ndarray] = []
        self.batch_sizes: List[int] = []          # SOURCE images per batch (before augmentation): the grouping key of write_code_npy
        self._pending = None
        self._side = None
        self._pinned = [None, None]
        self._flip = 0

    # -- device -> host pipeline ---------------------------------------------------------------------------------
    def _drain(self):
        if self._pending is not None:
            ev, host, target = self._pending
            if ev is not None:
                ev.synchronize()
            self.tokens.append(host.numpy().copy())
            self.class_ids.append(target)
            self._pending = None

    def _submit(self, tok: torch.Tensor, target: np.ndarray):
        if tok.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=tok.device)
            buf = self._pinned[self._flip]
            if buf is None or buf.shape != tok.shape:
                buf = torch.empty(tok.shape, dtype=tok.dtype, pin_memory=True)
                self._pinned[self._flip] = buf
            self._side.wait_stream(torch.cuda.current_stream(tok.device))
            with torch.cuda.stream(self._side):
                buf.copy_(tok, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
            tok.record_stream(self._side)
            prev = self._pending
            self._pending = (ev, buf, target)
            self._flip ^= 1
            if prev is not None:                      # the previous batch had a whole encoder pass to arrive
                pev, phost, ptarget = prev
                pev.synchronize()
                self.tokens.append(phost.numpy().copy())
                self.class_ids.append(ptarget)
        else:
            self.tokens.append(tok.numpy().copy())
            self.class_ids.append(target)

    # -- main loop (pretokenization.py:219-239) -----------------------------------------------------------------------
    def run(self, batches: Iterable, device=None):
        for samples, target in batches:
            if device is not None:
                samples = samples.to(device, non_blocking=True)
            target = torch.as_tensor(target)
            self.batch_sizes.append(int(target.shape[0]))
            if self.augment == "ten_crop":
                samples, target = augment_ten_crop(samples, target)
            elif self.augment == "flip":
                samples, target = augment_flip(samples, target)
            tok = tokens_from_images(self.model, samples, self.amp_dtype)
            self._submit(tok, target.cpu().numpy().astype(np.int64))
        self._drain()
        return self

    @property
    def records(self):
        cls = np.concatenate(self.class_ids) if self.class_ids else np.zeros(0, np.int64)
        tok = np.concatenate(self.tokens) if self.tokens else np.zeros((0, 0), np.int64)
        return cls, tok

    # -- RAR format (pretokenization.py:236-255) -------------------------------------------------------------------
    def write_rank_json(self, cached_path: str, rank: int = 0) -> str:
        os.makedirs(cached_path, exist_ok=True)
        cls, tok = self.records
        path = os.path.join(cached_path, f"pretokenized_{rank}.json")
        with open(path, "w") as f:
            json.dump([{"class_id": int(c), "tokens": t.tolist()} for c, t in zip(cls, tok)], f)
        return path

    # -- VAR / LlamaGen format (dataset/imagenet.py:8-50) ------------------------------------------------------------
    @property
    def n_aug(self) -> int:
        """augmented views per source image in `records`"""
        if self.augment == "flip":
            return 2
        if self.augment == "none":
            return 1
        cls, _ = self.records
        return len(cls) // max(1, sum(self.batch_sizes))      # ten_crop: whatever the crop count of the input was

    def write_code_npy(self, code_dir: str, label_dir: str, n_aug: Optional[int] = None, first_index: int = 0) -> int:
        """one `{i}.npy` pair per SOURCE image: codes (1, n_aug, L) = its augmented views, label (1,).
        The record stream is ordered per batch as the augmentation emitted it — flip: [batch, flipped batch]
        (pretokenization.py:227-228), so the two views of an image are B records apart and are brought together here
        (regroup_flip over the recorded batch sizes); ten_crop: the crops of an image are already adjacent.  `n_aug`, if given,
        must be the augmentation's own count (2 / crops / 1)."""
        os.makedirs(code_dir, exist_ok=True)
        os.makedirs(label_dir, exist_ok=True)
        cls, tok = self.records
        own = self.n_aug
        if n_aug is not None and n_aug != own:
            raise ValueError(f"n_aug={n_aug}, but augment='{self.augment}' produced {own} views per image")
        n_aug = own
        n = sum(self.batch_sizes)
        if len(cls) != n * n_aug:
            raise ValueError(f"{len(cls)} records for {n} source images x {n_aug} views: records were edited behind run()")
        if self.augment == "flip":
            cls, tok = regroup_flip(cls, tok, self.batch_sizes)
        for i in range(n):
            views = cls[i * n_aug:(i + 1) * n_aug]
            if not (views == views[0]).all():
                raise ValueError(f"source image {i}: its {n_aug} views carry different labels {views.tolist()}")
            np.save(os.path.join(code_dir, f"{first_index + i}.npy"), tok[i * n_aug:(i + 1) * n_aug].reshape(1, n_aug, -1))
            np.save(os.path.join(label_dir, f"{first_index + i}.npy"), views[:1].reshape(1))
        return n


def regroup_flip(cls: np.ndarray, tok: np.ndarray, batch_sizes: List[int]):
    """augment_flip emits [batch, flipped batch] per batch; returns the records re-ordered so that the two views of every
    source image are adjacent (the grouping write_code_npy / CustomDataset expect)."""
    out_c, out_t, off = [], [], 0
    for b in batch_sizes:
        c, t = cls[off:off + 2 * b], tok[off:off + 2 * b]
        order = np.stack([np.arange(b), np.arange(b) + b], axis=1).reshape(-1)
        out_c.append(c[order]); out_t.append(t[order])
        off += 2 * b
    return np.concatenate(out_c), np.concatenate(out_t)


def convert_json_to_jsonl(input_pattern: str, output_file: str) -> int:
    """pretokenization.py:139-146: merge the per-rank dumps into one JSON-lines file (rank 0, after a barrier)"""
    n = 0
    with open(output_file, "w") as out:
        for filename in sorted(glob.glob(input_pattern)):
            with open(filename, "r") as f:
                for item in json.load(f):
                    json.dump(item, out)
                    out.write("\n")
                    n += 1
    return n


def read_jsonl_record(path: str, idx: int):
    """what PretoeknizedDataSetJSONL.__getitem__ does with a line (data/webdataset_reader.py:263-267)"""
    import linecache
    data = json.loads(linecache.getline(path, idx + 1).strip())
    return torch.tensor(data["class_id"]), torch.tensor(data["tokens"])


def pretokenize(model, batches: Iterable, cached_path: str, device=None, amp_dtype=None, augment="flip") -> str:
    """The whole job of scripts/pretokenization.py main() for one process / all ranks of a torch.distributed job."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    bt = BulkTokenizer(model, amp_dtype=amp_dtype, augment=augment).run(batches, device=device)
    bt.write_rank_json(cached_path, rank)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    out = os.path.join(cached_path, "pretokenized.jsonl")
    if rank == 0:
        convert_json_to_jsonl(os.path.join(cached_path, "pretokenized_*.json"), out)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    return out
