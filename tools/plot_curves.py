#!/usr/bin/env python3
"""Loss curves from a cli/gan.py --log-jsonl file -> PNG (PIL only; matplotlib is not in the image).

    python tools/plot_curves.py gpurun_out/gan_learn/split/log.jsonl profiles/gan_learning/split_curves.png
Upper panel: d_loss (blue) and g_loss (red), log10 axis.  Lower panel: the discriminator's mean logits on the real
half (green), the fake half of the D phase (blue) and the fake batch of the G phase (red): the last two coincide
when no batch-composition information leaks through BatchNorm.
"""
import json
import math
import sys

from PIL import Image, ImageDraw


def main(src, dst, title=""):
    rows = [json.loads(l) for l in open(src)]
    loss = [(r["iteration"], r["d_loss"], r["g_loss"]) for r in rows if r.get("event") == "loss"]
    ev = [r for r in rows if r.get("event") == "eval"]
    W, H, L, R, T = 900, 560, 60, 20, 24
    im = Image.new("RGB", (W, H), "white")
    d = ImageDraw.Draw(im)
    d.text((L, 4), title or src, fill="black")
    n = max(r[0] for r in loss) or 1

    def panel(y0, y1, lo, hi, series, ticks, fmt):
        d.rectangle([L, y0, W - R, y1], outline="black")
        for t in ticks:
            y = y1 - (t - lo) / (hi - lo) * (y1 - y0)
            d.line([L, y, W - R, y], fill=(225, 225, 225))
            d.text((4, y - 6), fmt(t), fill="black")
        for k in range(0, 11):
            x = L + k / 10 * (W - R - L)
            d.text((x - 10, y1 + 3), str(int(n * k / 10)), fill="black")
        for pts, col in series:
            xy = [(L + i / n * (W - R - L), y1 - (min(max(v, lo), hi) - lo) / (hi - lo) * (y1 - y0)) for i, v in pts]
            if len(xy) > 1:
                d.line(xy, fill=col, width=1)

    lg = lambda v: math.log10(max(v, 1e-6))
    panel(T, 300, -6, 1.5, [([(i, lg(a)) for i, a, _ in loss], (30, 60, 200)), ([(i, lg(b)) for i, _, b in loss], (200, 40, 40))],
          range(-6, 2), lambda t: f"1e{t}")
    d.text((L + 8, T + 4), "d_loss", fill=(30, 60, 200))
    d.text((L + 60, T + 4), "g_loss", fill=(200, 40, 40))
    if ev:
        lo = min(min(r["logit_real"], r["logit_fake"], r["logit_gphase"]) for r in ev) - 1
        hi = max(max(r["logit_real"], r["logit_fake"], r["logit_gphase"]) for r in ev) + 1
        step = max(1, int((hi - lo) / 6))
        panel(330, H - 24, lo, hi,
              [([(r["iteration"], r["logit_real"]) for r in ev], (30, 150, 60)),
               ([(r["iteration"], r["logit_fake"]) for r in ev], (30, 60, 200)),
               ([(r["iteration"], r["logit_gphase"]) for r in ev], (200, 40, 40))],
              range(int(lo), int(hi) + 1, step), lambda t: str(t))
        d.text((L + 8, 334), "logit real", fill=(30, 150, 60))
        d.text((L + 90, 334), "fake (D phase)", fill=(30, 60, 200))
        d.text((L + 200, 334), "fake (G phase)", fill=(200, 40, 40))
    im.save(dst)


if __name__ == "__main__":
    main(*sys.argv[1:])
