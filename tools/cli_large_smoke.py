"""End-to-end CLI check at the PRODUCTION architecture (synthetic weights): compress.py -> search.py -> decompress.py
on a handful of generated PNGs, incl. a non-multiple-of-256 size.  usage: python tools/cli_large_smoke.py [workdir]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

import sgic_amd  # noqa
from sgic_amd import compress, decompress, search
from sgic_amd.data import synth_images

work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sgic_cli_large"
src, out = os.path.join(work, "imgs"), os.path.join(work, "out")
os.makedirs(src, exist_ok=True)
sizes = [(256, 256)] * 4 + [(300, 500), (512, 512)]
for i, (h, w) in enumerate(sizes):
    x = synth_images(1, 256 * ((h + 255) // 256), 256 * ((w + 255) // 256), 500 + i)[0, :, :h, :w]
    Image.fromarray(((x * 0.5 + 0.5) * 255).round().byte().permute(1, 2, 0).numpy()).save(os.path.join(src, f"im{i}.png"))
t0 = time.time()
assert compress.main(["--dataset_dir", src, "--save_dir", out, "--batch_size", "4"]) == 0
t1 = time.time()
files = sorted(os.listdir(os.path.join(out, "bitstreams")))
print("compress:", files, [os.path.getsize(os.path.join(out, "bitstreams", f)) for f in files], f"{t1 - t0:.1f}s")
q, _ = search.decode_clip_from_c2df(os.path.join(out, "bitstreams", "im4.c2df"))
vecs, paths = search.load_index(os.path.join(out, "faiss"))
hits = search.do_search(q[None], vecs, paths, 3)
print("query-c2df im4 ->", json.dumps(hits))
assert os.path.basename(hits[0][0]) == "im4.c2df"
assert decompress.main(["--dataset_dir", os.path.join(out, "bitstreams"), "--save_dir", out]) == 0
for i, (h, w) in enumerate(sizes):
    im = Image.open(os.path.join(out, "results", f"im{i}.png"))
    assert im.size == (w, h), (i, im.size)
print("decompress: ok", f"{time.time() - t1:.1f}s")
