"""GroupedBatchSampler (wetectron/data/samplers/grouped_batch_sampler.py:13-266).

Two behaviours of the reference live here:
  * aspect-ratio grouping (:62-121): batches only hold images of one group (wide / tall), inside a group images keep
    the sampler's order, and the batches are ordered by the sampler position of their first element;
  * SOLVER.CLASS_BATCH (:123-205, the contrastive model's sampler): every batch is a PAIR of images that share a
    class -- walking the sampled order, image i draws one of its classes (`np.random.choice`) and takes the first later
    image that has it, unless both carry the same multi-class label set; a taken image leaves the pool.
The image-level class sets come from `dataset.get_groundtruth(i).get_field("labels")` (the reference re-opens the VOC
splits for that, :42-60)."""
import itertools

import numpy as np
import torch
from torch.utils.data.sampler import BatchSampler, Sampler


def group_batches(sampled_ids, group_ids, batch_size, drop_uneven=False):
    """:62-121 on plain lists: `sampled_ids` in sampler order, `group_ids[i]` the group of dataset item i."""
    sampled = list(sampled_ids)
    position = {v: k for k, v in enumerate(sampled)}           # later duplicates win, like the reference's dict
    group_ids = np.asarray(group_ids)
    order = np.full(len(group_ids), -1, np.int64)
    order[np.asarray(sampled, np.int64)] = np.arange(len(sampled))
    batches = []
    for g in np.unique(group_ids):
        rel = np.sort(order[(group_ids == g) & (order >= 0)])
        members = [sampled[k] for k in rel.tolist()]
        batches += [members[k:k + batch_size] for k in range(0, len(members), batch_size)]
    firsts = torch.as_tensor([position[b[0]] for b in batches])
    batches = [batches[k] for k in firsts.sort(0)[1].tolist()] if batches else []
    if drop_uneven:
        batches = [b for b in batches if len(b) == batch_size]
    return batches


def class_pair_batches(sampled_ids, class_labels):
    """:181-190: pairs of images sharing a drawn class; consumes one `np.random.choice` per visited image."""
    inds = list(sampled_ids)
    batch = []
    for i, ind1 in enumerate(inds):              # `inds` shrinks while it is walked -- the reference's semantics
        rand_class1 = np.random.choice(class_labels[ind1])
        for ind2 in inds[i + 1:]:
            if rand_class1 in class_labels[ind2] and (class_labels[ind1] != class_labels[ind2]
                                                      or len(class_labels[ind1]) == 1):
                batch.append([ind1, ind2])
                inds.remove(ind2)
                break
    return batch


class GroupedBatchSampler(BatchSampler):
    def __init__(self, sampler, group_ids, batch_size, b_size=None, dataset=None, class_batch=False, data_args=None,
                 drop_uneven=False):
        if not isinstance(sampler, Sampler):
            raise ValueError("sampler should be an instance of torch.utils.data.Sampler, but got sampler={}".format(sampler))
        self.sampler = sampler
        self.group_ids = torch.as_tensor(group_ids)
        assert self.group_ids.dim() == 1
        self.batch_size, self.b_size, self.drop_uneven = batch_size, b_size, drop_uneven
        self.dataset, self.class_batch = dataset, class_batch
        self._can_reuse_batches = False
        self._class_labels = None

    def class_labels(self):
        if self._class_labels is None:
            def labels_of(d):       # VOC: a BoxList with a `labels` field; COCO: the class tensor itself (coco.py:191-196)
                gt = self.dataset.get_groundtruth(d)
                return (gt.get_field("labels") if hasattr(gt, "get_field") else gt).tolist()
            self._class_labels = [list(set(labels_of(d))) for d in range(len(self.dataset))]
        return self._class_labels

    def _prepare_batches(self):
        sampled_ids = list(self.sampler)
        if self.class_batch:
            return class_pair_batches(sampled_ids, self.class_labels())
        return group_batches(sampled_ids, self.group_ids.tolist(), self.batch_size, self.drop_uneven)

    def __iter__(self):
        if self._can_reuse_batches:
            batches = self._batches
            self._can_reuse_batches = False
        else:
            batches = self._prepare_batches()
        self._batches = batches
        return iter(batches)

    def __len__(self):
        if not hasattr(self, "_batches"):
            self._batches = self._prepare_batches()
            self._can_reuse_batches = True
        return len(self._batches)
