"""PARSeq with bf16-split operands in the ViT encoder ONLY (decoder, memory projection and vocabulary head exact fp32):
how far do the logits move, against the fp32 kernels and against the CPU oracle?  Wave-sized grouped forward of the --lite
recogniser (753 lines, 30 mini-batches) and the open-beta geometry (D = 512, 3 mini-batches); prints one JSON document."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

dev = torch.device("cuda:0")


def run(net, xs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    torch.cuda.synchronize()
    return logits.cpu(), list(out_lens), list(steps), time.perf_counter() - t


def compare(a, b, xs):
    row, worst, same, n = 0, 0.0, 0, 0
    for x, n_out in zip(xs, a[1]):
        p, q = a[0][row : row + x.shape[0], :n_out], b[0][row : row + x.shape[0], :n_out]
        worst = max(worst, float((p - q).abs().max()))
        same += int((p.argmax(-1) == q.argmax(-1)).sum())
        n += p.shape[0] * p.shape[1]
        row += x.shape[0]
    return {"max_d_logit": worst, "argmax_equal_frac": same / n, "ar_steps_equal": a[2] == b[2]}


def main():
    from oracle.parseq import parseq_forward
    from test_parseq_gpu import _groups, _net
    from yomitoku_amd.utils.synth import parseq_state_dict

    out = {}
    rng = np.random.default_rng(8)
    shapes = []
    while sum(b for b, _ in shapes) < 640 or len(shapes) < 30:
        w = int(rng.choice([72, 96, 128, 160, 200, 240, 320, 400, 560, 800]))
        shapes.append((int(min(32, max(1, 8000 // w - int(rng.integers(0, 4))))), w))
    cases = {"parseq-tiny-dynw-v4 (753 lines, 30 mini-batches)": (parseq_state_dict(1235, eos_bias=5.5), "parseq-tiny-dynw-v4", {}, _groups(300, shapes)),
             "parseq open-beta geometry, depth 12 (9 lines... 3 mini-batches)": (parseq_state_dict(1236, patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, eos_bias=6.5),
                                                                                "parseq", {}, _groups(31, [(24, 160), (8, 640), (32, 96)]))}
    for name, (sd, preset, over, xs) in cases.items():
        ocfg, net = _net(dev, sd, preset, **over)
        res = {}
        for label, split, enc in (("fp32", 0, -1), ("encoder_2_planes", 0, 2), ("all_2_planes", 2, -1), ("encoder_3_planes", 0, 3),
                                  ("encoder_f16x2", 0, 16), ("all_f16x2", 16, -1)):
            net.set_conv_split(split)
            net.set_param("conv_split_encoder", enc)
            run(net, xs)  # shapes seen once
            res[label] = run(net, xs)
        r = {"lines": int(res["fp32"][0].shape[0]), "forward_ms": {k: round(v[3] * 1e3, 2) for k, v in res.items()}}
        for k in ("encoder_2_planes", "all_2_planes", "encoder_3_planes", "encoder_f16x2", "all_f16x2"):
            r[k + "_vs_fp32_kernel"] = compare(res[k], res["fp32"], xs)
        # two small groups against the CPU oracle (fp32 PyTorch)
        small = sorted(range(len(xs)), key=lambda g: xs[g].shape[0] * xs[g].shape[3])[:2]
        for k in ("fp32", "encoder_2_planes", "all_f16x2"):
            worst = 0.0
            for g in small:
                row = sum(x.shape[0] for x in xs[:g])
                ref = parseq_forward(sd, ocfg, xs[g])
                got = res[k][0][row : row + xs[g].shape[0], : ref.shape[1]]
                assert torch.equal(got.argmax(-1), ref.argmax(-1)), (name, k, g)
                worst = max(worst, float((got - ref).abs().max()))
            r[k + "_vs_oracle_max_d_logit"] = worst
        out[name] = r
        net.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
