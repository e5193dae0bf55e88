#!/usr/bin/env python3
"""profiles/r05_box_spread.txt from the rows tools/box_spread.sh left under gpurun_out/box_spread/ (one fresh box each).

usage: python3 tools/box_spread_report.py [dir] > profiles/r05_box_spread.txt
"""
import glob
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/box_spread"
rows = [json.load(open(f)) for f in sorted(glob.glob(d + "/*.json"))]


def placed_ratio(r):
    """mover on the placed pair over the mover inside one domain, of the FIRST placement (a re-check reports it inside its record)"""
    rc = r.get("recheck") or {}
    if "placed_pair_was" in rc:
        return rc["placed_pair_was"]["mover_on_pair_over_mover_in_one_domain"]
    return r["pair_over_one_domain"]


print("The bench's timed step (reference-shaped loop, 64 x 256 frames) on fresh pool boxes, one gpurun call each (tools/box_spread.sh;")
print("this table: tools/box_spread_report.py); `domains` = HBM domains the placement probe saw among its chunks; `pair/one` = a mover on")
print("the placed pair over the same mover inside one domain (~0.9: the pair lies apart; ~1.0: the virtual-memory API handed out")
print("chunks of ONE domain only -- the box \"behaves as one domain\"); `mover_frac` = roofline.box_mover_frac (the ceiling for this")
print("read / write mix on the buffers the run used, on the scale of 8 TB/s); `frac` = roofline.frac.  Rows 1-14: before bench.py")
print("held the placed pair against plain allocations; rows 15-51: only when pair/one >= 0.985; from row 52 on: always -- the mover")
print("is timed on the placed pair and on two plain hipMallocs and the faster pair is kept (`re-check`).")
print()
print("box domains  pair/one  mover ms  mover_frac  kernel ms  frac    k/mover   decoder ms  frames/s   step ms min/median/max   re-check")
for i, r in enumerate(rows, 1):
    s = r["step_ms"]
    rc = r.get("recheck")
    note = ""
    if rc:
        note = "placed %.3f ms, plain %.3f ms: kept %s" % (rc["mover_ms_on_the_placed_pair"], rc["mover_ms_on_two_plain_allocations"], rc["kept"])
    dom = r["domains_seen"] if r["domains_seen"] is not None else (rc or {}).get("placed_pair_was", {}).get("domains_seen")
    print(f"{i:<3} {str(dom):<8} {placed_ratio(r):<9.3f} {r['mover_ms']:<9.3f} {r['box_mover_frac']:<11.4f} "
          f"{r['kernel_ms']:<10.3f} {r['frac']:<7.4f} {r['kernel_over_mover']:<9.3f} {r['decoder_ms']:<11.3f} {r['value']:<10d} "
          f"{s['min']:.2f} / {s['median']:.2f} / {s['max']:.2f}     {note}")
one = [r for r in rows if placed_ratio(r) >= 0.985]
two = [r for r in rows if placed_ratio(r) < 0.985]
kept_plain = [r for r in one if (r.get("recheck") or {}).get("kept") == "plain"]
no_better = [r for r in one if (r.get("recheck") or {}).get("kept") == "placed"]
stuck = [r for r in one if not r.get("recheck")]
cmp_two = [r for r in two if r.get("recheck")]
rng = lambda xs, k, f="{:.3f}": (f + "-" + f).format(min(x[k] for x in xs), max(x[k] for x in xs))
val = lambda xs: "%.3f-%.3f M" % (min(r["value"] for r in xs) / 1e6, max(r["value"] for r in xs) / 1e6)
print()
print(f"{len(rows)} runs (a pool machine can come up more than once: rows 70-74 are one machine by every number).  On {len(one)} of them the placed pair was no")
print(f"better than one domain (pair/one >= 0.985): {len(stuck)} before bench.py compared (mover_frac {rng(stuck, 'box_mover_frac')}, {val(stuck)} frames/s);")
if kept_plain:
    print(f"{len(kept_plain)} where two plain allocations were faster and were kept (mover_frac {rng(kept_plain, 'box_mover_frac')}, {val(kept_plain)}: the virtual-memory API's chunks came")
    print(f"from one domain, plain allocations did not);")
if no_better:
    print(f"{len(no_better)} where plain allocations were just as slow (mover_frac {rng(no_better, 'box_mover_frac')}, {val(no_better)}: whatever is allocated on such a machine behaves as one domain).")
print(f"The other {len(two)}: mover_frac {rng(two, 'box_mover_frac')}, {val(two)}; of the {len(cmp_two)} compared, the placed pair was kept on "
      f"{sum(1 for r in cmp_two if r['recheck']['kept'] == 'placed')}.  kernel / mover {rng(rows, 'kernel_over_mover')} on every run: the kernel follows its buffers' ceiling.")
