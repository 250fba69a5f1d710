"""A miniature VOC devkit written from arrays (shared by tests/golden/make_golden.py and the data-boundary tests).
Images are stored losslessly (PNG bytes under the devkit's .jpg names -- PIL picks the decoder from the content), so
both sides decode exactly the pixels held in the fixture."""
import os

import numpy as np
from PIL import Image

CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
           "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


def make_case(seed, shapes):
    """Deterministic content: pixels (smooth gradients + noise), 1-based VOC objects, raw int16 proposals that
    exercise de-duplication, clipping and the small-box filter."""
    rng = np.random.RandomState(seed)
    images, objects, proposals = [], [], []
    for k, (h, w) in enumerate(shapes):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7) % 256], axis=2)
        noise = rng.randint(-40, 41, size=(h, w, 3))
        images.append(np.clip(base + noise, 0, 255).astype(np.uint8))
        objs = []
        for j in range(1 + (k % 3)):
            x1, y1 = rng.randint(1, w // 2), rng.randint(1, h // 2)
            x2, y2 = rng.randint(x1 + 4, w + 1), rng.randint(y1 + 4, h + 1)
            objs.append((CLASSES[(3 * k + 5 * j) % 20], int(j == 1), x1, y1, x2, y2))
        objects.append(objs)
        n = 40 + 7 * k
        x1 = rng.randint(-6, w - 8, size=n)
        y1 = rng.randint(-6, h - 8, size=n)
        bw = rng.randint(2, w, size=n)
        bh = rng.randint(2, h, size=n)
        b = np.stack([x1, y1, x1 + bw, y1 + bh], axis=1)
        b[5] = b[2]
        b[11] = b[2]
        b[17] = b[9]
        proposals.append(b.astype(np.int16))
    return images, objects, proposals


def write_devkit(root, split, ids, images, objects):
    for d in ("JPEGImages", "Annotations", os.path.join("ImageSets", "Main")):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, "ImageSets", "Main", split + ".txt"), "w") as f:
        f.write("".join(i + "\n" for i in ids))
    for i, img, objs in zip(ids, images, objects):
        with open(os.path.join(root, "JPEGImages", i + ".jpg"), "wb") as f:
            Image.fromarray(np.asarray(img, np.uint8), "RGB").save(f, format="PNG")
        h, w = img.shape[:2]
        xml = ["<annotation><size><width>%d</width><height>%d</height><depth>3</depth></size>" % (w, h)]
        for name, difficult, x1, y1, x2, y2 in objs:
            xml.append("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin>"
                       "<xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (name, difficult, x1, y1, x2, y2))
        xml.append("</annotation>")
        with open(os.path.join(root, "Annotations", i + ".xml"), "w") as f:
            f.write("".join(xml))


def make_pair_case(seed, n):
    """n tiny images (wide and tall) whose objects come from a 5-class pool, so that many images share a class:
    input of the aspect-grouping / class-pair sampler tests."""
    rng = np.random.RandomState(seed)
    images, objects = [], []
    pool = ("bird", "cat", "dog", "horse", "sheep")
    for k in range(n):
        h, w = (16, 24) if rng.randint(2) else (24, 16)
        images.append(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8))
        names = sorted(set(pool[c] for c in rng.randint(0, 5, size=1 + rng.randint(2))))
        objects.append([(nm, 0, 2, 2, 10 + j, 12 + j) for j, nm in enumerate(names)])
    return images, objects


def make_detections(seed, images, objects):
    """Per image (boxes fp32 (k,4), scores (k), labels int64 (k)): jittered copies of the annotated objects (hits and
    near misses, duplicates), plus random false positives; some images get none."""
    rng = np.random.RandomState(seed + 1000)
    cls_index = {c: i + 1 for i, c in enumerate(CLASSES)}
    out = []
    for k, (img, objs) in enumerate(zip(images, objects)):
        h, w = img.shape[:2]
        boxes, scores, labels = [], [], []
        if k % 7 != 3:
            for name, difficult, x1, y1, x2, y2 in objs:
                for rep in range(1 + rng.randint(3)):
                    j = rng.randint(-3, 4, size=4) if rng.rand() < 0.7 else rng.randint(-9, 10, size=4)
                    boxes.append([x1 - 1 + j[0], y1 - 1 + j[1], x2 - 1 + j[2], y2 - 1 + j[3]])
                    scores.append(rng.rand())
                    labels.append(cls_index[name] if rng.rand() < 0.85 else 1 + rng.randint(20))
            for _ in range(rng.randint(3)):
                x1, y1 = rng.randint(0, w - 6), rng.randint(0, h - 6)
                boxes.append([x1, y1, x1 + rng.randint(3, 6), y1 + rng.randint(3, 6)])
                scores.append(rng.rand() * 0.6)
                labels.append(1 + rng.randint(20))
        out.append((np.asarray(boxes, np.float32).reshape(-1, 4), np.asarray(scores, np.float32),
                    np.asarray(labels, np.int64)))
    return out


def make_coco_case(seed, n):
    """A miniature COCO instances file: non-contiguous category ids, crowd objects, an image without annotations, one
    whose only box is degenerate, boxes sticking out of the image; raw int16 proposals per image id."""
    rng = np.random.RandomState(seed)
    cat_ids = [1, 2, 4, 7, 9, 16, 18, 44, 62, 90]
    cats = [{"id": c, "name": "cat%d" % c, "supercategory": "s"} for c in cat_ids]
    images, pixels, anns, proposals = [], {}, [], {}
    aid = 1
    for k in range(n):
        img_id = 100 + 7 * (n - k)              # descending in file order: the dataset sorts
        h, w = (40, 56) if k % 2 else (52, 36)
        images.append({"id": img_id, "file_name": "im%06d.jpg" % img_id, "height": h, "width": w})
        pixels[img_id] = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        if k == 2:
            pass                                                    # no annotation at all
        elif k == 4:
            anns.append({"id": aid, "image_id": img_id, "category_id": 4, "bbox": [5.0, 6.0, 1.0, 20.0], "iscrowd": 0, "area": 20.0})
            aid += 1                                                # only a degenerate box
        else:
            for j in range(1 + rng.randint(3)):
                x, y = rng.randint(-4, w - 10), rng.randint(-4, h - 10)
                bw, bh = rng.randint(4, w), rng.randint(4, h)
                anns.append({"id": aid, "image_id": img_id, "category_id": int(cat_ids[rng.randint(len(cat_ids))]),
                             "bbox": [float(x), float(y), float(bw), float(bh)], "iscrowd": int(j == 2), "area": float(bw * bh)})
                aid += 1
        m = 30
        x1, y1 = rng.randint(-3, w - 4, size=m), rng.randint(-3, h - 4, size=m)
        b = np.stack([x1, y1, x1 + rng.randint(0, w, size=m), y1 + rng.randint(0, h, size=m)], axis=1)
        b[3] = b[1]
        proposals[img_id] = b.astype(np.int16)
    return {"images": images, "annotations": anns, "categories": cats}, pixels, proposals


def write_coco(root, data, pixels, proposals):
    import json
    import pickle
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    for im in data["images"]:
        with open(os.path.join(root, "img", im["file_name"]), "wb") as f:
            Image.fromarray(pixels[im["id"]], "RGB").save(f, format="PNG")
    ann = os.path.join(root, "instances.json")
    with open(ann, "w") as f:
        json.dump(data, f)
    pkl = os.path.join(root, "props.pkl")
    ids = sorted(proposals.keys())
    with open(pkl, "wb") as f:
        pickle.dump(dict(boxes=[proposals[i] for i in ids], scores=[np.ones(len(proposals[i]), np.float32) for i in ids],
                         indexes=ids), f, pickle.HIGHEST_PROTOCOL)
    return ann, os.path.join(root, "img"), pkl
