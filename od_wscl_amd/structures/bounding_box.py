"""BoxList -- the container the reference passes for proposals and targets
(wetectron/structures/bounding_box.py:13-58,209-233).  Only what the hot path
consumes: `.bbox` (n,4) fp32, `.size` = (W,H), `.mode`, fields, len, indexing,
`.to`, `.area` (+1 convention), xyxy<->xywh conversion."""
import torch


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
