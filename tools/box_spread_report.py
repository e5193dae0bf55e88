#!/usr/bin/env python3
"""profiles/r05_box_spread.txt from the rows tools/box_spread.sh left under gpurun_out/box_spread/ (one fresh box each).

usage: python3 tools/box_spread_report.py [dir] > profiles/r05_box_spread.txt
"""
import glob
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/box_spread"
rows = [json.load(open(f)) for f in sorted(glob.glob(d + "/*.json"))]
print("The bench's timed step (reference-shaped loop, 64 x 256 frames) on fresh pool boxes, one gpurun call each (tools/box_spread.sh;")
print("this table: tools/box_spread_report.py); `domains` = HBM domains the placement probe saw among its chunks; `pair/one` = a mover on")
print("the placed pair over the same mover inside one domain (~0.9: the pair lies apart; ~1.0: the box behaves as ONE domain whatever the")
print("probe's small timing differences said); `mover_frac` = roofline.box_mover_frac (the box's own ceiling for this read / write mix on")
print("the scale of 8 TB/s); `frac` = roofline.frac.")
print()
print("box domains  pair/one  mover ms  mover_frac  kernel ms  frac    k/mover   decoder ms  frames/s   step ms min/median/max")
for i, r in enumerate(rows, 1):
    s = r["step_ms"]
    print(f"{i:<3} {r['domains_seen']:<8} {r['pair_over_one_domain']:<9.3f} {r['mover_ms']:<9.3f} {r['box_mover_frac']:<11.4f} "
          f"{r['kernel_ms']:<10.3f} {r['frac']:<7.4f} {r['kernel_over_mover']:<9.3f} {r['decoder_ms']:<11.3f} {r['value']:<10d} "
          f"{s['min']:.2f} / {s['median']:.2f} / {s['max']:.2f}")
one = [r for r in rows if r["pair_over_one_domain"] >= 0.985]
two = [r for r in rows if r["pair_over_one_domain"] < 0.985]
rng = lambda xs, k, f="{:.3f}": (f + "-" + f).format(min(x[k] for x in xs), max(x[k] for x in xs))
print()
print(f"{len(rows)} boxes; {len(one)} of them behave as one domain (pair/one >= 0.985): mover_frac {rng(one, 'box_mover_frac') if one else '-'} there against "
      f"{rng(two, 'box_mover_frac')} on the others;")
print(f"kernel / mover {rng(rows, 'kernel_over_mover')} on every box: the kernel follows its box's ceiling; frames/s "
      f"{min(r['value'] for r in rows) / 1e6:.3f}-{max(r['value'] for r in rows) / 1e6:.3f} M "
      f"({min(r['value'] for r in two) / 1e6:.3f}-{max(r['value'] for r in two) / 1e6:.3f} M on the boxes whose pair lies apart).")
