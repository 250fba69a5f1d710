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
