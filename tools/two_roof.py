"""Which layers hold the convolution path away from its roofs: the per-launch lines of ymk_debug_option("prof_dump", 1)
grouped by layer shape and priced against the binding roof of each.

    YMK_DEBUG_OPTIONS=prof_dump=1 python bench.py --roofline-only --no-cpu-baseline 2> dump.txt > line.json
    python tools/two_roof.py dump.txt out.md [passes]

A dump line: `[ymk-prof] i M=.. Cin=.. Cout=.. k=KxK s=. d=. res=. tile=BMxBN ksplit=. grid=.  T us  R TFLOP/s  B MB`
(yomitoku_amd/csrc/ymk_conv.hip prof_end).  `ksplit` >= 160 marks the fp16-plane kernels (MFMA roof 2500 / 3 TFLOP/s-
equivalent), 20 / 30 the bf16 forms, anything else the exact fp32 MFMA (157.3).  HBM roof: 8 TB/s over the algorithmic bytes
(input view, weights, output, residual - each once).  `passes`: how many repetitions of the same pass the dump holds
(bench.py repeats the serial pass three times): only the first `passes` spans of the dump are read (a span ends where the
launch index starts again from 0; bench.py's detector-alone leg follows the analyzer's passes) and per-pass figures are the
totals divided by it.  The table is sorted by the time a shape spends ABOVE its bound - the order in which fixing shapes pays."""
import re
import sys
from collections import defaultdict

FP32_MFMA, F16_MFMA, HBM = 157.3e12, 2500e12, 8e12
LINE = re.compile(r"\[ymk-prof\]\s+(\d+)\s+(M=\s*\d+ Cin=\s*\d+ Cout=\s*\d+ k=\dx\d s=\d d=\d res=\d) tile=(\d+x\d+) ksplit=(\d+) grid=\d+\s+([\d.]+) us\s+([\d.]+) TFLOP/s\s+([\d.]+) MB")


def main(src, dst, passes=3):
    groups = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0])
    import gzip

    span, last = 0, -1
    with (gzip.open(src, "rt", errors="replace") if src.endswith(".gz") else open(src, errors="replace")) as f:
        for line in f:
            m = LINE.search(line)
            if not m:
                continue
            idx = int(m.group(1))
            if idx <= last:
                span += 1
            last = idx
            if span >= passes:
                break
            shape, tile, ksplit, us, tf, mb = m.group(2), m.group(3), int(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))
            g = groups[(re.sub(r"\s+", " ", shape).replace("= ", "="), tile, ksplit)]
            g[0] += 1
            g[1] += us
            g[2] += tf * 1e12 * us * 1e-6  # FLOPs
            g[3] += mb * 1e6
            g[4] = 3 if ksplit >= 160 or ksplit == 20 else (6 if ksplit == 30 else 0)
    rows = []
    for (shape, tile, ksplit), (n, us, flop, nbytes, products) in groups.items():
        peak = FP32_MFMA if products == 0 else F16_MFMA / products
        t_m, t_h = flop / peak * 1e6, nbytes / HBM * 1e6
        bound = max(t_m, t_h)
        rows.append((us - bound, shape, tile, "f16" if ksplit >= 160 else ("bf16" if ksplit in (20, 30) else "f32"), n, us, flop, nbytes,
                     "hbm" if t_h > t_m else "mfma", bound))
    rows.sort(reverse=True)
    tot_us, tot_bound = sum(r[5] for r in rows), sum(r[9] for r in rows)
    with open(dst, "w") as f:
        f.write(f"{sum(r[4] for r in rows) // passes} launches per pass, {tot_us / passes / 1e3:.2f} ms measured, {tot_bound / passes / 1e3:.2f} ms at the binding roofs "
                f"(= {tot_bound / tot_us:.3f}); sorted by time above the bound\n\n")
        f.write("| layer shape | tile | kernel | launches / pass | ms / pass | TFLOP/s | TB/s (algorithmic) | binding roof | of that roof | ms above it / pass | cumulative share of the excess |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        excess_total = sum(max(0.0, r[0]) for r in rows)
        run = 0.0
        for excess, shape, tile, kern, n, us, flop, nbytes, roof, bound in rows:
            if us < 2e-3 * tot_us:
                continue
            run += max(0.0, excess)
            f.write(f"| `{shape}` | {tile} | {kern} | {n // passes} | {us / passes / 1e3:.3f} | {flop / us / 1e6:.1f} | {nbytes / us / 1e6:.2f} | {roof} | "
                    f"{bound / us:.2f} | {excess / passes / 1e3:.3f} | {run / excess_total:.2f} |\n")
    print(f"{dst}: {len(rows)} shapes, {tot_bound / tot_us:.3f} of the two-roof bound")


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
