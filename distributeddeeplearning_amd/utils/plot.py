"""Model graph rendering: `plot_model(spec, path, show_shapes=True)`.

The reference calls keras.utils.plot_model (pydot + graphviz) for every exported model
(/root/reference/DDL.py:158, DDL_Normal4.py:175-177, GAN_distributed.py:130-131).  Neither pydot
nor graphviz exists here, so this renders the same boxes-and-arrows layout directly: a DOT text
file (always, next to the image, loadable by graphviz elsewhere) and a PNG/JPEG drawn with PIL.
Node label format follows Keras: `name: ClassName | input: (None, ...) | output: (None, ...)`.
"""
from __future__ import annotations

from pathlib import Path

from ..models.ir import ModelSpec, _KIND_TO_CLASS


def _rows(spec: ModelSpec, show_shapes: bool):
    shapes = spec.infer_shapes()
    rows = []
    prev = None
    for l in spec.layers:
        out = (None,) + tuple(shapes[l.name])
        inp = out if prev is None else (None,) + tuple(shapes[prev.name])
        rows.append((f"{l.name}: {_KIND_TO_CLASS[l.kind]}", str(inp), str(out)))
        prev = l
    return rows


def to_dot(spec: ModelSpec, show_shapes: bool = True) -> str:
    lines = ["digraph G {", "  node [shape=record];"]
    rows = _rows(spec, show_shapes)
    for i, (title, inp, out) in enumerate(rows):
        label = f"{title}|{{input:|output:}}|{{{inp}|{out}}}" if show_shapes else title
        lines.append(f'  n{i} [label="{label}"];')
    for i in range(1, len(rows)):
        lines.append(f"  n{i - 1} -> n{i};")
    lines.append("}")
    return "\n".join(lines)


def plot_model(spec: ModelSpec, to_file, show_shapes: bool = True):
    to_file = Path(to_file)
    to_file.parent.mkdir(parents=True, exist_ok=True)
    to_file.with_suffix(".dot").write_text(to_dot(spec, show_shapes))
    try:
        from PIL import Image, ImageDraw, ImageFont
    except ImportError:          # DOT file is still written
        return to_file.with_suffix(".dot")
    font = ImageFont.load_default()
    rows = _rows(spec, show_shapes)
    pad, lh, gap = 6, 14, 28

    def tw(s):
        return int(font.getlength(s)) if hasattr(font, "getlength") else 6 * len(s)

    boxes = []
    for title, inp, out in rows:
        w_title = tw(title) + 2 * pad
        w_io = (tw("output:") + 2 * pad) if show_shapes else 0
        w_shape = (max(tw(inp), tw(out)) + 2 * pad) if show_shapes else 0
        boxes.append((w_title, w_io, w_shape))
    width = max(sum(b) for b in boxes) + 40
    bh = 2 * lh + 2 * pad
    height = len(rows) * (bh + gap) + 20
    img = Image.new("RGB", (width, height), "white")
    dr = ImageDraw.Draw(img)
    y = 10
    for i, ((title, inp, out), (w1, w2, w3)) in enumerate(zip(rows, boxes)):
        x0 = (width - (w1 + w2 + w3)) // 2
        dr.rectangle([x0, y, x0 + w1 + w2 + w3, y + bh], outline="black")
        dr.text((x0 + pad, y + pad + lh // 2), title, fill="black", font=font)
        if show_shapes:
            x1, x2 = x0 + w1, x0 + w1 + w2
            dr.line([x1, y, x1, y + bh], fill="black")
            dr.line([x2, y, x2, y + bh], fill="black")
            dr.line([x1, y + bh // 2, x0 + w1 + w2 + w3, y + bh // 2], fill="black")
            dr.text((x1 + pad, y + pad - 2), "input:", fill="black", font=font)
            dr.text((x1 + pad, y + bh // 2 + pad - 2), "output:", fill="black", font=font)
            dr.text((x2 + pad, y + pad - 2), inp, fill="black", font=font)
            dr.text((x2 + pad, y + bh // 2 + pad - 2), out, fill="black", font=font)
        if i + 1 < len(rows):
            cx = width // 2
            dr.line([cx, y + bh, cx, y + bh + gap], fill="black")
            dr.polygon([(cx - 4, y + bh + gap - 7), (cx + 4, y + bh + gap - 7), (cx, y + bh + gap)], fill="black")
        y += bh + gap
    img.save(to_file)
    return to_file
