import torch


def nms(boxes, scores, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    b = boxes.detach().cpu().float()
    s = scores.detach().cpu().float()
    order = torch.sort(s, descending=True, stable=True)[1].tolist()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = ((x2 - x1) * (y2 - y1))
    n = len(order)
    dead = [False] * b.shape[0]
    keep = []
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        for c in range(a + 1, n):
            j = order[c]
            if dead[j]:
                continue
            w = (torch.minimum(x2[i], x2[j]) - torch.maximum(x1[i], x1[j])).clamp(min=0)
            h = (torch.minimum(y2[i], y2[j]) - torch.maximum(y1[i], y1[j])).clamp(min=0)
            inter = w * h
            iou = inter / (area[i] + area[j] - inter)
            if iou > iou_threshold:
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64, device=boxes.device)
