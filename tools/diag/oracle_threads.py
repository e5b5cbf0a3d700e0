"""How the CPU oracle's wall time on this box depends on torch's thread count (the GPU suite's long tests are oracle-bound)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle.dbnet import dbnet_forward  # noqa: E402
from oracle.parseq import PRESETS, make_cfg, parseq_forward  # noqa: E402
from oracle.rtdetr import rtdetr_forward  # noqa: E402
from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_line_batch  # noqa: E402
from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict  # noqa: E402

sd_d, sd_p, sd_r = dbnet_state_dict(1234), parseq_state_dict(1235, eos_bias=5.5), rtdetr_state_dict(1242, num_classes=6)
ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
xd = torch.randn(1, 3, 1600, 1184)
xp = synthetic_line_batch(3, 64, 320)
xr = torch.rand(2, 3, 640, 640)
print("default threads", torch.get_num_threads(), flush=True)
for n in (256, 128, 64, 32, 16):
    torch.set_num_threads(n)
    row = [n]
    for fn in (lambda: dbnet_forward(sd_d, xd), lambda: parseq_forward(sd_p, ocfg, xp), lambda: rtdetr_forward(sd_r, xr)):
        fn()
        t = time.perf_counter()
        fn()
        row.append(round(time.perf_counter() - t, 2))
    print("threads, dbnet 1600x1184, parseq 64x320, rtdetr 2x640^2 [s]:", row, flush=True)
