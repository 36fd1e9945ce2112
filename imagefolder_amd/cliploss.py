"""Contrastive (CLIP-style) loss used as the semantic regulariser of the tokenizer.
Mirror of the part of reference tokenizer/tokenizer_image/cliploss.py the hot path runs (ClipLoss :65-134 with
gather_with_grad=True, local_loss=False, no horovod: xqgan_model.py:181-196).  Plain tensor ops: (B x C) features."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributed as dist


class ClipLoss(nn.Module):
    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1, use_horovod=False):
        super().__init__()
        assert not use_horovod and not local_loss, "only the configuration used by VQModel is mirrored"
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels
        self.rank = rank
        self.world_size = world_size
        self.prev_num_logits = 0
        self.labels = {}

    def get_ground_truth(self, device, num_logits):
        if self.prev_num_logits != num_logits or device not in self.labels:
            labels = torch.arange(num_logits, device=device, dtype=torch.long)
            if self.cache_labels:
                self.labels[device] = labels
                self.prev_num_logits = num_logits
        else:
            labels = self.labels[device]
        return labels

    def get_logits(self, image_features, text_features, logit_scale):
        if self.world_size > 1:
            if self.gather_with_grad:
                import torch.distributed.nn
                all_i = torch.cat(torch.distributed.nn.all_gather(image_features), dim=0)
                all_t = torch.cat(torch.distributed.nn.all_gather(text_features), dim=0)
            else:
                gi = [torch.zeros_like(image_features) for _ in range(self.world_size)]
                gt = [torch.zeros_like(text_features) for _ in range(self.world_size)]
                dist.all_gather(gi, image_features)
                dist.all_gather(gt, text_features)
                gi[self.rank], gt[self.rank] = image_features, text_features
                all_i, all_t = torch.cat(gi, dim=0), torch.cat(gt, dim=0)
            logits_per_image = logit_scale * all_i @ all_t.T
            logits_per_text = logits_per_image.T
        else:
            logits_per_image = logit_scale * image_features @ text_features.T
            logits_per_text = logit_scale * text_features @ image_features.T
        return logits_per_image, logits_per_text

    def forward(self, image_features, text_features, logit_scale, output_dict=False):
        lpi, lpt = self.get_logits(image_features, text_features, logit_scale)
        labels = self.get_ground_truth(image_features.device, lpi.shape[0])
        total = (F.cross_entropy(lpi, labels) + F.cross_entropy(lpt, labels)) / 2
        return {"contrastive_loss": total} if output_dict else total
