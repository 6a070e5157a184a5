"""CLI-level throughput of the compress driver (files -> .c2df): N synthetic JPEG files on local disk ->
`sgic_amd.compress.main` with the production architecture -> the driver's own JSON record (images/s including the header
pass, JPEG decode, H2D, GPU work, container and .npy writes, index assembly excluded from the rate's clock).
usage: python tools/cli_throughput.py [N=960] [size=256] [batch=32]"""
import io
import json
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

import sgic_amd  # noqa
from sgic_amd import compress
from sgic_amd.data import synth_images

N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
with tempfile.TemporaryDirectory() as tmp:
    src, out = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    os.makedirs(src)
    base = ((synth_images(64, S, S, 3) * 0.5 + 0.5) * 255).round().byte().permute(0, 2, 3, 1).numpy()
    t0 = time.perf_counter()
    for i in range(N):
        Image.fromarray(np.roll(base[i % 64], i // 64, axis=0)).save(os.path.join(src, f"im{i:05d}.jpg"), quality=90)
    print(f"wrote {N} JPEGs {S}x{S} in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    for rep in range(2):          # the second pass runs with a warm tile cache and page cache: the steady-state CLI rate
        buf = io.StringIO()
        t0 = time.perf_counter()
        with redirect_stdout(buf):
            rc = compress.main(["--dataset_dir", src, "--save_dir", out + str(rep), "--batch_size", str(B)])
        wall = time.perf_counter() - t0
        rec = json.loads(buf.getvalue().strip().splitlines()[-1])
        rec.update(rc=rc, pass_=rep, wall_incl_model_build_s=round(wall, 2), size=S)
        print(json.dumps(rec), flush=True)
