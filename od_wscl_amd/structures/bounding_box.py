"""BoxList -- the container the reference passes for proposals and targets
(wetectron/structures/bounding_box.py:13-260): `.bbox` (n,4) fp32, `.size` = (W,H), `.mode`, fields, len,
indexing, `.to`, `.area` (+1 convention), xyxy<->xywh conversion, and the geometry the data boundary applies on
the host next to the pixel plan (resize :95-131, transpose :133-169, crop :171-197, clip_to_image :218-229)."""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be (n,4), got %s" % (tuple(bbox.shape),))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode = bbox, image_size, mode
        self.extra_fields = {}

    def add_field(self, field, data):
        self.extra_fields[field] = data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def convert(self, mode):
        if mode == self.mode:
            return self
        b = self.bbox
        if mode == "xyxy":      # from xywh
            out = torch.stack((b[:, 0], b[:, 1], b[:, 0] + (b[:, 2] - 1).clamp(min=0),
                               b[:, 1] + (b[:, 3] - 1).clamp(min=0)), dim=1)
        else:
            out = torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0] + 1, b[:, 3] - b[:, 1] + 1), dim=1)
        r = BoxList(out, self.size, mode)
        r.extra_fields = dict(self.extra_fields)
        return r

    def _with_fields(self, bbox, mode, size=None, op=None):
        r = BoxList(bbox, self.size if size is None else size, mode)
        for k, v in self.extra_fields.items():
            r.add_field(k, v if isinstance(v, torch.Tensor) or op is None else op(v))
        return r

    def resize(self, size, *args, **kwargs):
        """Boxes of the image resized to `size` = (width, height): plain scaling, one ratio when both agree
        (bounding_box.py:95-131)."""
        ratios = tuple(float(s) / float(s_orig) for s, s_orig in zip(size, self.size))
        op = lambda v: v.resize(size, *args, **kwargs)
        if ratios[0] == ratios[1]:
            return self._with_fields(self.bbox * ratios[0], self.mode, size, op)
        b = self.convert("xyxy").bbox
        rw, rh = ratios
        scaled = torch.stack((b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh), dim=1)
        return self._with_fields(scaled, "xyxy", size, op).convert(self.mode)

    def transpose(self, method):
        """Horizontal flip keeps the +1 pixel convention (x' = W - x - 1), the vertical one does not
        (y' = H - y) -- bounding_box.py:147-158."""
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        b = self.convert("xyxy").bbox
        if method == FLIP_LEFT_RIGHT:
            out = torch.stack((w - b[:, 2] - 1, b[:, 1], w - b[:, 0] - 1, b[:, 3]), dim=1)
        else:
            out = torch.stack((b[:, 0], h - b[:, 3], b[:, 2], h - b[:, 1]), dim=1)
        return self._with_fields(out, "xyxy", None, lambda v: v.transpose(method)).convert(self.mode)

    def crop(self, box):
        b = self.convert("xyxy").bbox
        w, h = box[2] - box[0], box[3] - box[1]
        out = torch.stack(((b[:, 0] - box[0]).clamp(min=0, max=w), (b[:, 1] - box[1]).clamp(min=0, max=h),
                           (b[:, 2] - box[0]).clamp(min=0, max=w), (b[:, 3] - box[1]).clamp(min=0, max=h)), dim=1)
        return self._with_fields(out, "xyxy", (w, h), lambda v: v.crop(box)).convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        """In place, like the reference (bounding_box.py:218-229); empty = not strictly positive extent."""
        self.bbox[:, 0].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 1].clamp_(min=0, max=self.size[1] - 1)
        self.bbox[:, 2].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 3].clamp_(min=0, max=self.size[1] - 1)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def copy_with_fields(self, fields, skip_missing=False):
        r = BoxList(self.bbox, self.size, self.mode)
        for f in fields if isinstance(fields, (list, tuple)) else [fields]:
            if self.has_field(f):
                r.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return r

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def to(self, device):
        r = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            r.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return r

    def __getitem__(self, item):
        r = BoxList(self.bbox[item].reshape(-1, 4), self.size, self.mode)
        for k, v in self.extra_fields.items():
            r.add_field(k, v[item])
        return r

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return "BoxList(num_boxes=%d, image_width=%s, image_height=%s, mode=%s)" % (
            len(self), self.size[0], self.size[1], self.mode)
