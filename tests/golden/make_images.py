"""Writes the 8 small image files of tests/golden/images/ (4 PNG + 4 JPEG, colour and gray, odd sizes) that
tests/test_loader_values.py decodes.  Deterministic content: gradients + seeded noise + a few hard edges, so that a
wrong interpolation rule (antialiasing, half-pixel centres, align_corners) changes many pixels.
    python tests/golden/make_images.py"""
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'images')
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(2026)
specs = [('a.png', 37, 29, 3), ('b.png', 128, 96, 3), ('c.png', 60, 60, 1), ('d.png', 250, 250, 3),
         ('e.jpg', 144, 122, 3), ('f.jpg', 112, 112, 3), ('g.jpg', 71, 203, 1), ('h.jpg', 300, 180, 3)]
for name, h, w, ch in specs:
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(255 * xx / max(w - 1, 1)), (255 * yy / max(h - 1, 1)), (255 * ((xx // 7 + yy // 5) % 2))], -1)[..., :ch]
    img = np.clip(base + rng.normal(0, 25, (h, w, ch)), 0, 255).astype(np.uint8)
    im = Image.fromarray(img if ch == 3 else img[..., 0], 'RGB' if ch == 3 else 'L')
    if name.endswith('.jpg'):
        im.save(os.path.join(HERE, name), quality=92)
    else:
        im.save(os.path.join(HERE, name))
with open(os.path.join(HERE, 'list.txt'), 'w') as f:
    for i, (name, _, _, _) in enumerate(specs):
        f.write('%s %d\n' % (name, i % 4))
print('wrote', len(specs), 'images to', HERE)
